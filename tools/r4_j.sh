#!/bin/bash
# Round-4 call J: the tests that changed since call I, then the driver's N = 1 bench command with every new point
set -u
OUT=gpurun_out/r04j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_iterator.py tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q --timeout 600 -k "filtered or sharded or beyond or contraction or two_gpus or config5" > $OUT/tests.log 2>&1
tail -n 6 $OUT/tests.log
( time python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4
tail -n 12 $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04j/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'], j['roofline']['frac_basis'][:40])
print('host', j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'])
print('matched', {k: j['config']['matched_recall_point'].get(k) for k in ('window','recall_at_10','gpu_qps','kernel_ms','frac','gpu_over_cpu','gpu_over_full_host_if_linear')})
for k,v in j['config'].get('other_configs',{}).items(): print(k, {a:b for a,b in v.items() if a not in ('workload','graph','bound')})
print('sharded', j.get('sharded_search'))
print('cpu', {k: j['cpu_baseline'].get(k) for k in ('value','cores','gpu_over_cpu','gpu_over_full_host_if_linear')})
print('scaling', j['cpu_baseline'].get('thread_scaling'))
PY
