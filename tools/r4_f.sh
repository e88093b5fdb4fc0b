#!/bin/bash
# Round-4 call F: the interleaved two-query kernel (policy 3): timing first (it also checks ids and bits against the one-query kernel), then the parity tests
set -u
OUT=gpurun_out/r04f
mkdir -p $OUT
timeout 400 python tools/policy_ab.py --policy 3 --nq 2000,6144,10000,20000,40000 --windows 128,23 2>$OUT/ab.err | tee $OUT/ab_c2_x2.jsonl | cut -c1-330
tail -n 3 $OUT/ab.err
timeout 600 python -m pytest tests/test_gpu_two_query.py -m gpu -x -q --timeout 300 > $OUT/tests.log 2>&1
tail -n 12 $OUT/tests.log
