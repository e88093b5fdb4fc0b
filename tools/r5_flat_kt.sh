#!/bin/bash
OUT=gpurun_out/r05flatkt; mkdir -p $OUT
for v in in-tree api_kt4 api_kt2; do
  if [ "$v" = in-tree ]; then unset SVSX_LIB; else export SVSX_LIB=$PWD/variants/$v/libsvsx.so; fi
  for cfg in "1000000 96 10000 10 l2" "1000000 128 10000 10 ip" "1000000 96 10000 10 cos" "1000000 96 10000 100 l2" "1000000 128 10000 1 ip"; do
    timeout 300 python tools/flat_check.py $cfg 2>>$OUT/err.log | sed "s/^/$v /" | tee -a $OUT/summary.txt
  done
done
