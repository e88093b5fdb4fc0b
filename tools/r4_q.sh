#!/bin/bash
# round 4, call w: threshold rank kt = k + 4 / k + 8 (production) / k + 16
R=$PWD; O=$R/gpurun_out/r04w; mkdir -p $O; rm -rf $O/*
for V in fkt4 base fkt16; do
  if [ $V = base ]; then unset SVSX_LIB; else export SVSX_LIB=$R/variants/$V/libsvsx.so; fi
  echo "== $V" >> $O/out.txt
  timeout 200 python tools/flat_check.py 1000000 96 10000 10 l2 >> $O/out.txt 2>/dev/null
  timeout 200 python tools/flat_check.py 1000000 96 10000 32 l2 >> $O/out.txt 2>/dev/null
done
cat $O/out.txt
