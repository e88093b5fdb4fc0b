#!/bin/bash
# Round-4 call C: per-query duration distributions (are the stragglers issue-priority victims?) and the s_setprio pace controller A/B
set -u
OUT=gpurun_out/r04c
mkdir -p $OUT
for v in diag3 diag3p360; do
  echo "== $v"; SVSX_LIB=variants/$v/libsvsx.so timeout 200 python tools/diag_prologue.py --nq 6144,10000 --windows 128,23 2>>$OUT/diag.err | tee $OUT/diag_$v.jsonl | cut -c1-1500
done
timeout 500 python tools/ab_libs.py --libs prod=,p300=variants/prio300/libsvsx.so,p360=variants/prio360/libsvsx.so,p420=variants/prio420/libsvsx.so --script tools/time_c2.py --repeat 2 -- --nq 10000,6144 --windows 128,23 2>&1 | tee $OUT/ab_prio.txt
timeout 600 python -m pytest tests/test_gpu_two_query.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 > $OUT/tests.log 2>&1
tail -n 8 $OUT/tests.log
