#!/bin/bash
# round 6, call 2: integer-path inline records (raw int8 / uint8 rows, scalar-quantized L2)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06b; mkdir -p $OUT; rm -f $OUT/*
timeout 1700 python -m pytest tests/test_gpu_bound_codes.py -m gpu -q -x > $OUT/tests_codes.log 2>&1; echo "tests_codes rc=$?" > $OUT/rc.txt
tail -5 $OUT/tests_codes.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_iterator.py -m gpu -q -x > $OUT/tests_parity.log 2>&1; echo "tests_parity rc=$?" >> $OUT/rc.txt
tail -3 $OUT/tests_parity.log
cat $OUT/rc.txt
