#!/bin/bash
# round 6, call 6: the centred similarity records -- every bound-code, parity, config and fuzz test; spectrum check on an offset set
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06f; mkdir -p $OUT; rm -f $OUT/*
timeout 2400 python -m pytest tests/test_gpu_bound_codes.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
tail -6 $OUT/tests.log
cat $OUT/rc.txt
