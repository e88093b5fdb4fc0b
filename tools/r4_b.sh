#!/bin/bash
# Round-4 call B: where a query's time goes outside the expansions (prologue / epilogue), and the two-query tests again (redo inline)
set -u
OUT=gpurun_out/r04b
mkdir -p $OUT
SVSX_LIB=variants/diag3/libsvsx.so timeout 300 python tools/diag_prologue.py --nq 256,6144,10000 --windows 128,23 2>$OUT/diag.err | tee $OUT/diag_prologue.jsonl
tail -n 3 $OUT/diag.err
timeout 600 python -m pytest tests/test_gpu_two_query.py -m gpu -x -q --timeout 300 > $OUT/tests.log 2>&1
tail -n 8 $OUT/tests.log
