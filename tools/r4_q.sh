#!/bin/bash
# round 4, call q: timing builds of the threshold kernel without any vector-memory instruction in the loop (wrong results by design)
R=$PWD; O=$R/gpurun_out/r04s; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
for V in fnohits fnovmem fnovmem_noepi fnovmem_nolds; do
  export SVSX_LIB=$R/variants/$V/libsvsx.so
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$V -o t -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > $O/$V.json 2>/dev/null
  f=$(find $O/trace_$V -name "*kernel_stats.csv" | head -1)
  echo "== $V $(cat $O/$V.json)" >> $O/summary.txt
  grep -E "flat_thresh" "$f" | cut -d, -f1-4,6,7 | cut -c1-200 >> $O/summary.txt
done
find $O -name "*_kernel_trace.csv" -delete
cat $O/summary.txt
