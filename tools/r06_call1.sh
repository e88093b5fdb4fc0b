#!/bin/bash
# round 6, call 1: the tests round 6 touched so far + the driver's own bench command without the extras
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06a; mkdir -p $OUT; rm -f $OUT/*
timeout 1700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_iterator.py -m gpu -x -q -k "float16_scalar or scalar_loop or range_and_filtered or scalar_quantized" > $OUT/tests_a.log 2>&1; echo "tests_a rc=$?" > $OUT/rc.txt
tail -3 $OUT/tests_a.log
timeout 1700 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "config4_large" > $OUT/tests_c4.log 2>&1; echo "tests_c4 rc=$?" >> $OUT/rc.txt
tail -3 $OUT/tests_c4.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06a/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "prewarm", j.get("prewarm_steps"), "kernel_ms", j["roofline"]["kernel_ms"])
print("steps", j["config"]["step_ms_each"])
print("matched", j["config"]["matched_recall_point"])
print("host", j["config"].get("host_pointer"))
PY
cat $OUT/rc.txt
