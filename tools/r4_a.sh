#!/bin/bash
# Round-4 call A: the two-query kernel for the first time -- its parity tests, then 1Q vs 2Q timings on config 2.
set -u
OUT=gpurun_out/r04a
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_two_query.py -m gpu -x -q --timeout 300 > $OUT/tests.log 2>&1
tail -n 25 $OUT/tests.log
timeout 400 python tools/policy_ab.py --nq 2000,6144,10000,20000,40000 --windows 128,23 2>$OUT/ab.err | tee $OUT/ab_c2.jsonl | cut -c1-420
tail -n 5 $OUT/ab.err
