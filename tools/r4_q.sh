#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04h3; mkdir -p $O; rm -rf $O/*
for V in base h3c base h3c; do
  if [ $V = base ]; then unset SVSX_LIB; else export SVSX_LIB=$R/variants/$V/libsvsx.so; fi
  echo "== $V" >> $O/host_rate.jsonl
  timeout 300 python tools/host_rate.py --windows 128,23 >> $O/host_rate.jsonl 2>/dev/null
done
cut -c1-230 $O/host_rate.jsonl
