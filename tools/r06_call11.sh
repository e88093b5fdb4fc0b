#!/bin/bash
# round 6, call 11: the mean direction as a float32 coordinate of the similarity records -- offset sets, then every similarity / records test
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06k; mkdir -p $OUT; rm -f $OUT/*
timeout 900 python tools/offset_ab.py > $OUT/offset_u0.jsonl 2> $OUT/offset.err; echo "offset rc=$?" > $OUT/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r06k/offset_u0.jsonl"):
    j = json.loads(l); print(j["set"][:44], j["mode"], j["ms"], j["recall"], round(j["rejects"]), round(j["evals"]), j["ids_equal_plain"], j["kernel"][-24:])
PY
timeout 2700 python -m pytest tests/test_gpu_bound_codes.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -5 $OUT/tests.log
cat $OUT/rc.txt
