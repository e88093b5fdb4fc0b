#!/bin/bash
# Round-3 closing call: the driver's bench command, smoke, the whole GPU suite.
set -u
OUT=gpurun_out/r03_final
mkdir -p $OUT
python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_final/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'], j['roofline']['frac'], j['roofline']['kernel_src_sha'], j['cpu_baseline']['gpu_over_cpu'])
PY
python __graft_entry__.py smoke 2>&1 | tail -n 1 | cut -c1-200
timeout 900 python -m pytest tests -m gpu -q --timeout 400 > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -n 3
