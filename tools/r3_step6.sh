#!/bin/bash
# Round-3 sixth GPU call: VALU trimming (DPP folded into adds, 9-instruction merge loop, worst from registers), zero-copy host path.
set -u
OUT=gpurun_out/r03f
mkdir -p $OUT
timeout 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bound_codes.py tests/test_gpu_configs.py tests/test_gpu_iterator.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -n 5 $OUT/tests.log
for L in diag diagE; do
  echo "== $L"
  SVSX_LIB=variants/$L/libsvsx.so timeout 200 python tools/diag_early.py --nq 256,6144,10000,20000 --windows 128,23 2>&1 | grep -v amdgpu.ids | cut -c1-330
done > $OUT/diag_early.txt
SVSX_LIB=variants/diag2/libsvsx.so timeout 200 python tools/diag_phases.py --nq 256,10000 --windows 128 2>&1 | grep -v amdgpu.ids > $OUT/phases.txt
cat $OUT/phases.txt
timeout 120 python tools/host_rate.py --nq 10000 --windows 128,23 2>&1 | grep -v amdgpu.ids > $OUT/host_rate.txt
cat $OUT/host_rate.txt
