#!/bin/bash
# Round-3 second GPU call: (1) diagnostics of the early block (hit rate, per-expansion latency lone / full machine) on DIAG variant libraries,
# (2) residency sweep by LDS request at batch 10 000, (3) K7 + native-builder tests on the in-tree library.
set -u
OUT=gpurun_out/r03b
mkdir -p $OUT
for L in diag diag0; do
  echo "== $L"
  SVSX_LIB=variants/$L/libsvsx.so timeout 300 python tools/diag_early.py --nq 256,2048,6144,10000,20000 --windows 128,23 --lds-requests 0,6656,8192,9000,10240 2>&1 | grep -v amdgpu.ids | cut -c1-440
done > $OUT/diag_early.txt 2>&1
cat $OUT/diag_early.txt
timeout 120 python tools/host_rate.py --nq 10000,40000 2>&1 | grep -v amdgpu.ids > $OUT/host_rate.txt
cat $OUT/host_rate.txt
timeout 700 python -m pytest tests/test_gpu_native_builder.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -k "native or static or cancel or host or concurrent" > $OUT/tests.log 2>&1
tail -n 25 $OUT/tests.log
