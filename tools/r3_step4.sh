#!/bin/bash
# Round-3 fourth GPU call: phase profile of an expansion (DIAG=2), kept-block usage, host-call timeline, native sharded test.
set -u
OUT=gpurun_out/r03d
mkdir -p $OUT
for L in diag2 diag2e0; do
  echo "== $L"
  SVSX_LIB=variants/$L/libsvsx.so timeout 200 python tools/diag_phases.py --nq 256,6144,10000 --windows 128,23 2>&1 | grep -v amdgpu.ids
done > $OUT/phases.txt 2>&1
cat $OUT/phases.txt
SVSX_LIB=variants/diag/libsvsx.so timeout 200 python tools/diag_early.py --nq 256,10000 --windows 128 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $OUT/diag_early.txt
cat $OUT/diag_early.txt
SVSX_HOST_TIMELINE=1 SVSX_LIB=variants/diag/libsvsx.so timeout 120 python tools/host_rate.py --nq 10000 --windows 128 > $OUT/host_rate.txt 2>&1
grep -v amdgpu.ids $OUT/host_rate.txt | tail -n 9 | cut -c1-600
timeout 300 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "native_sharded or cancel" > $OUT/tests.log 2>&1
tail -n 6 $OUT/tests.log
