#!/bin/bash
# round 6, call 8: the mean direction in the similarity basis (offset sets), the tests that exercise similarity records, the uint8 regression point
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06h; mkdir -p $OUT; rm -f $OUT/*
timeout 900 python tools/offset_ab.py > $OUT/offset_meandir.jsonl 2> $OUT/offset.err; echo "offset rc=$?" > $OUT/rc.txt
cut -c1-330 $OUT/offset_meandir.jsonl
timeout 2400 python -m pytest tests/test_gpu_bound_codes.py tests/test_gpu_configs.py -m gpu -q -x -k "not config2_full and not config4_large" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -4 $OUT/tests.log
timeout 1500 python bench.py --steps 20 --warmup 5 --c4-n 0 --c5-n -1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06h/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"])
print("u8", j["config"]["other_configs"].get("regression_uint8_r128"))
print("c3", {k: v for k, v in j["config"]["other_configs"]["c3"].items() if k in ("qps", "kernel_ms", "recall_at_10", "bound_code_rejects_per_query", "distance_evals_per_query")})
PY
cat $OUT/rc.txt
