#!/bin/bash
set -u
OUT=gpurun_out/r04g
mkdir -p $OUT
timeout 400 python tools/policy_ab.py --policy 3 --config c3 --nq 2000,10000,40000 --windows 128 2>$OUT/ab.err | tee $OUT/ab_c3_x2.jsonl | cut -c1-330
tail -n 3 $OUT/ab.err
