#!/bin/bash
# round 6, call 14: host-pointer call in its steady state (the later loops of tools/host_rate.py) for first chunks of 4 096 (production) / 3 072 / 2 048 queries and a 2 000-query third chunk; twice, alternating
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06p; mkdir -p $OUT; rm -f $OUT/*
for rep in 1 2; do
  for v in prod hostf3072 hostf2048 host3rd; do
    if [ $v = prod ]; then L=""; else L="variants/$v/libsvsx.so"; fi
    SVSX_LIB=$L timeout 300 python tools/host_rate.py --windows 128,23 > $OUT/${v}_$rep.jsonl 2> $OUT/${v}_$rep.err
    python - $v $rep <<'PY'
import json, sys
for l in open(f"gpurun_out/r06p/{sys.argv[1]}_{sys.argv[2]}.jsonl"):
    j = json.loads(l); print(sys.argv[1], sys.argv[2], j["window"], "device", round(j["device_ms"], 4), "first loop", round(j["host_c_abi_ms"], 4), "python", round(j["host_python_ms"], 4), "fresh", round(j["host_c_abi_fresh_result_arrays_ms"], 4), "again", round(j["host_c_abi_same_arrays_again_ms"], 4), j["ids_identical"])
PY
  done
done
