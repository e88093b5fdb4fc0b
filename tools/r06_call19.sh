#!/bin/bash
# round 6, call 19: the driver's bench command once more (bench.py's rule-sized-records point rebuilt its records; library and tests as in call 18)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06u; mkdir -p $OUT; rm -f $OUT/*
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4; echo "bench rc=$?"
python - <<'PY'
import json
b = json.loads(open("gpurun_out/r06u/bench_1gpu.json").read().strip().splitlines()[-1])
print(b["value"], b["roofline"]["frac"], b["roofline"]["kernel_ms"], b["config"]["host_pointer"]["over_device_resident"])
print(json.dumps(b["config"]["other_configs"]["c2_records_rule"])[:900])
PY
