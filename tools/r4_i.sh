#!/bin/bash
# Round-4 call I: the whole GPU suite after the hardening / spill / sharded-pipeline / row-norm changes
set -u
OUT=gpurun_out/r04i
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/tests.log 2>&1
tail -n 15 $OUT/tests.log
