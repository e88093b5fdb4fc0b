#!/bin/bash
# round 4, call p: what bounds the flat threshold kernel -- timing builds without appends / without new tiles (results of those builds are wrong by design)
R=$PWD; O=$R/gpurun_out/r04p; mkdir -p $O; rm -f $O/*.txt
cd /tmp && export TMPDIR=/tmp
for V in base fnohits fnoloads fneither; do
  if [ $V = base ]; then unset SVSX_LIB; else export SVSX_LIB=$R/variants/$V/libsvsx.so; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$V -o t -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > $O/$V.json 2>/dev/null
  f=$(find $O/trace_$V -name "*kernel_stats.csv" | head -1)
  echo "== $V $(cat $O/$V.json)" >> $O/summary.txt
  grep -E "flat_thresh|flat_select|rerank2|flat_certify|flat_exact|to_half" "$f" | cut -d, -f1-4,6,7 | cut -c1-200 >> $O/summary.txt
done
unset SVSX_LIB
find $O -name "*_kernel_trace.csv" -delete
cat $O/summary.txt
