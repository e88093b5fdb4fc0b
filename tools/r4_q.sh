#!/bin/bash
# round 4, call q: timing builds of the register-staged threshold kernel (wrong results by design): which part costs what
R=$PWD; O=$R/gpurun_out/r04q; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for V in fnodma fnohits fnoepi fnolds fmfmaonly; do
  export SVSX_LIB=$R/variants/$V/libsvsx.so
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$V -o t -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > $O/$V.json 2>/dev/null
  f=$(find $O/trace_$V -name "*kernel_stats.csv" | head -1)
  echo "== $V $(cat $O/$V.json)" >> $O/summary.txt
  grep -E "flat_thresh" "$f" | cut -d, -f1-4,6,7 | cut -c1-200 >> $O/summary.txt
done
find $O -name "*_kernel_trace.csv" -delete
cat $O/summary.txt
