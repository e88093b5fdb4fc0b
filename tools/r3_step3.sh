#!/bin/bash
# Round-3 third GPU call: all row loads up front (svsx_device.h float_row_distance) -- diagnostics on both DIAG variants, then the whole GPU suite.
set -u
OUT=gpurun_out/r03c
mkdir -p $OUT
for L in diag diag0; do
  echo "== $L"
  SVSX_LIB=variants/$L/libsvsx.so timeout 300 python tools/diag_early.py --nq 256,6144,10000,20000 --windows 128,23 2>&1 | grep -v amdgpu.ids | cut -c1-440
done > $OUT/diag_early.txt 2>&1
timeout 120 python tools/host_rate.py --nq 10000 2>&1 | grep -v amdgpu.ids > $OUT/host_rate.txt
cat $OUT/host_rate.txt
timeout 900 python -m pytest tests -m gpu -q -x > $OUT/tests.log 2>&1
tail -n 8 $OUT/tests.log
