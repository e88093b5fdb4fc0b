#!/bin/bash
# round 6, call 4: timeline at the matched window, record size against spectrum, touched tests, bench with the spectrum points
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06d; mkdir -p $OUT; rm -f $OUT/*
SVSX_LIB=variants/diag3/libsvsx.so timeout 600 python tools/diag_prologue.py --nq 10000 --windows 23,128 > $OUT/diag_prologue.jsonl 2> $OUT/diag.err; echo "diag rc=$?" > $OUT/rc.txt
cut -c1-1200 $OUT/diag_prologue.jsonl
timeout 900 python tools/spectrum_ab.py > $OUT/spectrum_ab.jsonl 2> $OUT/spectrum.err; echo "spectrum rc=$?" >> $OUT/rc.txt
cat $OUT/spectrum_ab.jsonl | cut -c1-330
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bound_codes.py -m gpu -q -k "sharded or fall_back or integer_rows or multi" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -4 $OUT/tests.log
timeout 1500 python bench.py --steps 20 --warmup 5 --c4-n 0 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06d/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "prewarm", j.get("prewarm_steps"))
for p in j["config"]["other_configs"]["c2_spectrum"]: print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()})
print("sharded", j.get("sharded_search")); print(j["config"].get("sharded_c5", {}).get("phases_ms"))
PY
cat $OUT/rc.txt; tail -3 $OUT/bench.err
