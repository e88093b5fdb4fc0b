#!/bin/bash
# Round-4 final check: the whole GPU suite, smoke(), then the driver's N = 1 bench command on the final library
set -u
OUT=gpurun_out/r04final
mkdir -p $OUT; rm -f $OUT/*
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
grep -E "passed|failed" $OUT/tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
( time python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4
echo "bench rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -1 $OUT/smoke.log
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04final/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])
print('flat', {k:v for k,v in j['config']['other_configs']['c3'].get('flat_search',{}).items() if k!='what'})
print('host', j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'])
print('c4', {k:v for k,v in j['config']['other_configs']['c4'].items() if k in ('qps','recall_at_10','window_for_recall_0.9','oracle_sample','error')})
print('sharded', j.get('sharded_search',{}).get('value'), 'gt', j['config']['ground_truth'].get('queries_recomputed_exhaustively'), j['config']['ground_truth'].get('id_sets_equal_to_float64_brute_force'))
PY
