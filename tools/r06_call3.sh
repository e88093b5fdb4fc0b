#!/bin/bash
# round 6, call 3: integer-path records (all of tests/test_gpu_bound_codes.py), C4(b) against the reference, bench with the C4(b) point at 2 M
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06c; mkdir -p $OUT; rm -f $OUT/*
timeout 1700 python -m pytest tests/test_gpu_bound_codes.py -m gpu -q > $OUT/tests_codes.log 2>&1; echo "tests_codes rc=$?" > $OUT/rc.txt
tail -5 $OUT/tests_codes.log
timeout 1700 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "config4b" > $OUT/tests_c4b.log 2>&1; echo "tests_c4b rc=$?" >> $OUT/rc.txt
tail -5 $OUT/tests_c4b.log
timeout 1500 python bench.py --steps 20 --warmup 5 --c4-n 2000000 --c5-n -1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06c/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "prewarm", j.get("prewarm_steps"))
c4 = j["config"]["other_configs"]["c4"]
print("c4", {k: v for k, v in c4.items() if k not in ("raw_int8", "workload")})
print("c4b", c4.get("raw_int8"))
PY
cat $OUT/rc.txt; tail -5 $OUT/bench.err
