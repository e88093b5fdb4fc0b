#!/bin/bash
# round 4, call q: threshold kernel with 128 queries per wave (one wave per SIMD) against 64 (two waves per SIMD)
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O; rm -rf $O/*
timeout 600 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_parity.py -q -k "flat" > $O/flat_tests.log 2>&1; echo "flat tests rc=$?" >> $O/out.txt
for V in base fqt2; do
  if [ $V = base ]; then unset SVSX_LIB; else export SVSX_LIB=$R/variants/$V/libsvsx.so; fi
  echo "== $V" >> $O/out.txt
  timeout 200 python tools/flat_check.py 1000000 96 10000 10 >> $O/out.txt 2>/dev/null
  timeout 200 python tools/flat_check.py 1000000 96 10000 100 l2 >> $O/out.txt 2>/dev/null
  timeout 200 python tools/flat_check.py 1000000 64 10000 10 l2 >> $O/out.txt 2>/dev/null
done
tail -3 $O/flat_tests.log; cat $O/out.txt
