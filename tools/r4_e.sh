#!/bin/bash
set -u
OUT=gpurun_out/r04e
mkdir -p $OUT
SVSX_LIB=variants/diag3/libsvsx.so timeout 200 python tools/diag_prologue.py --nq 6144,10000 --windows 128 --where 2>>$OUT/diag.err > $OUT/diag_where.jsonl
tail -n 2 $OUT/diag.err; wc -c $OUT/diag_where.jsonl
