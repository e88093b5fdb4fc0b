#!/bin/bash
# Round-3 fifth GPU call: predicted-next block request + worst from registers: bit-exact suites, phase profile, timings.
set -u
OUT=gpurun_out/r03e
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bound_codes.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -n 5 $OUT/tests.log
SVSX_LIB=variants/diag2/libsvsx.so timeout 200 python tools/diag_phases.py --nq 256,10000 --windows 128,23 2>&1 | grep -v amdgpu.ids > $OUT/phases.txt
cat $OUT/phases.txt
SVSX_LIB=variants/diag/libsvsx.so timeout 200 python tools/diag_early.py --nq 256,6144,10000,20000 --windows 128,23 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $OUT/diag_early.txt
cat $OUT/diag_early.txt
timeout 120 python tools/host_rate.py --nq 10000 --windows 128,23 2>&1 | grep -v amdgpu.ids > $OUT/host_rate.txt
cat $OUT/host_rate.txt
