#!/bin/bash
# round 6, call 15: 28-byte records -- parity tests, record size against spectrum with the new size, the headline with what the rule now picks
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06q; mkdir -p $OUT; rm -f $OUT/*
timeout 1500 python -m pytest tests/test_gpu_bound_codes.py -m gpu -q -x -k "28_bytes or small_world or fuzz or fall_back or d128 or any_dimension" > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
tail -4 $OUT/tests.log
timeout 900 python tools/spectrum_ab.py --windows 128,23 > $OUT/spectrum_ab.jsonl 2> $OUT/spectrum.err; echo "spectrum rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
for l in open("gpurun_out/r06q/spectrum_ab.jsonl"):
    j = json.loads(l); print(j["set"], j["window"], j["mode"], j["record"], j["ms"], round(j["rejects"]), round(j["evals"]), j["ids_equal_plain"], j["kernel"][-22:])
PY
cat $OUT/rc.txt
