#!/bin/bash
# Round-4 call R: the whole GPU suite, smoke(), then the driver's N = 1 bench command on the final library
set -u
OUT=gpurun_out/r04r
mkdir -p $OUT; rm -f $OUT/*
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
tail -n 6 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
( time python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4
echo "bench rc=$?" >> $OUT/rc.txt
tail -n 5 $OUT/bench.err
cat $OUT/rc.txt; tail -2 $OUT/smoke.log
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04r/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['frac_basis'][:40])
print('gt', j['config'].get('ground_truth'))
print('host', j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'])
print('matched', {k: j['config']['matched_recall_point'].get(k) for k in ('window','recall_at_10','gpu_qps','kernel_ms','frac','gpu_over_cpu','gpu_over_full_host_if_linear')})
for k,v in j['config'].get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('workload','graph','bound')})
print('sharded', j.get('sharded_search'))
print('cpu', {k: j['cpu_baseline'].get(k) for k in ('value','cores','gpu_over_cpu','gpu_over_full_host_if_linear')})
PY
