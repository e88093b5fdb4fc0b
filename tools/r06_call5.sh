#!/bin/bash
# round 6, call 5: timeline at the matched window; occupancy floor 6 for the runtime-d 64- / 128-byte record kernels (A/B); eight-lane integer rows
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06e; mkdir -p $OUT; rm -f $OUT/*
SVSX_LIB=variants/diag3/libsvsx.so timeout 600 python tools/diag_prologue.py --nq 10000 --windows 23,128 > $OUT/diag_prologue.jsonl 2> $OUT/diag.err; echo "diag rc=$?" > $OUT/rc.txt
cut -c1-1500 $OUT/diag_prologue.jsonl
for rep in 1 2; do
  timeout 600 python tools/spectrum_ab.py --sets 48:0.2,0:0 > $OUT/spectrum_prod_$rep.jsonl 2>> $OUT/spectrum.err
  SVSX_LIB=variants/floor6/libsvsx.so timeout 600 python tools/spectrum_ab.py --sets 48:0.2,0:0 > $OUT/spectrum_floor6_$rep.jsonl 2>> $OUT/spectrum.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06e/spectrum_*.jsonl")):
    for l in open(f):
        j = json.loads(l)
        if j["mode"] in (2, 64, 128): print(f.split("/")[-1], j["set"], j["mode"], j["record"], j["ms"], j["ids_equal_plain"], j["kernel"])
PY
timeout 1700 python -m pytest tests/test_gpu_configs.py tests/test_gpu_bound_codes.py -m gpu -q -x -k "config4b or integer or sq_parity" > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -4 $OUT/tests.log
timeout 1500 python bench.py --steps 20 --warmup 5 --c4-n 2000000 --c5-n -1 --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?" >> $OUT/rc.txt
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06e/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "prewarm", j.get("prewarm_steps"))
c4 = j["config"]["other_configs"]["c4"]
print("c4", {k: v for k, v in c4.items() if k in ("qps", "kernel_ms", "window_for_recall_0.9", "recall_at_10", "oracle_sample")})
print("c4b", c4.get("raw_int8"))
PY
cat $OUT/rc.txt
