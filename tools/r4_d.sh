#!/bin/bash
# Round-4 call D: per-query duration distributions of the one-query kernel; sharded tests after the slot-layout change
set -u
OUT=gpurun_out/r04d
mkdir -p $OUT
SVSX_LIB=variants/diag3/libsvsx.so timeout 200 python tools/diag_prologue.py --nq 256,6144,10000 --windows 128 2>>$OUT/diag.err | tee $OUT/diag_diag3.jsonl | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -x -q --timeout 300 -k "sharded or config5 or merge" > $OUT/tests.log 2>&1
tail -n 8 $OUT/tests.log
