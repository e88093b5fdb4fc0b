#!/bin/bash
# Round-3 eighth GPU call: the whole GPU suite on the production library (scalar-loop merge, two-launch host path), host-path matrix on DIAG.
set -u
OUT=gpurun_out/r03h
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -n 5
for SPLIT in 2048 4096 6144 10000; do
  echo "== dma, first chunk $SPLIT"
  SVSX_HOST_SPLIT=$SPLIT SVSX_LIB=variants/diag/libsvsx.so timeout 100 python tools/host_rate.py --nq 10000 --windows 128,23 2>&1 | grep -v amdgpu.ids
done > $OUT/host_rate.txt
for OUTM in 0 1; do
  echo "== zero-copy in (relaxed poll), out=$OUTM"
  SVSX_HOST_IN=1 SVSX_HOST_OUT=$OUTM SVSX_LIB=variants/diag/libsvsx.so timeout 100 python tools/host_rate.py --nq 10000 --windows 128,23 2>&1 | grep -v amdgpu.ids
done >> $OUT/host_rate.txt
cat $OUT/host_rate.txt
