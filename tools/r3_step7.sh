#!/bin/bash
# Round-3 seventh GPU call: lane-parallel merge insert (v2) -- bit-exact suites, A/B against the scalar-loop form, phase profile; host-path matrix.
set -u
OUT=gpurun_out/r03g
mkdir -p $OUT
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bound_codes.py tests/test_gpu_configs.py tests/test_gpu_iterator.py -m gpu -x -q > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -n 3
for L in diagm1 diag; do
  echo "== $L"
  SVSX_LIB=variants/$L/libsvsx.so timeout 200 python tools/diag_early.py --nq 256,6144,10000,20000 --windows 128,23 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > $OUT/diag_early.txt
SVSX_LIB=variants/diag2/libsvsx.so timeout 200 python tools/diag_phases.py --nq 256,10000 --windows 128 2>&1 | grep -v amdgpu.ids > $OUT/phases.txt
cat $OUT/phases.txt
for MODE in "0 0" "0 1" "1 1" "1 0" "2 0"; do
  set -- $MODE
  echo "== host in=$1 out=$2 (0 dma, 1 zero copy, 2 five chunks on five streams)"
  SVSX_HOST_IN=$1 SVSX_HOST_OUT=$2 SVSX_LIB=variants/diag/libsvsx.so timeout 120 python tools/host_rate.py --nq 10000 --windows 128,23 2>&1 | grep -v amdgpu.ids
done > $OUT/host_rate.txt
cat $OUT/host_rate.txt
