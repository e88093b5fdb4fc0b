#!/bin/bash
# tools/flat_trace.sh <out dir>: kernel trace of one flat_check process per shape (10 000 x 1 M x 96 L2, x 128 IP) -> <out>/<shape>_kernel_stats.csv
R=$PWD; OUT=$R/${1:-gpurun_out/flat_trace}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "96 l2" "128 ip"; do
  set -- $cfg
  timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$1 -o t -- python $R/tools/flat_check.py 1000000 $1 10000 10 $2 > $OUT/traced_$1_$2.json 2>/dev/null
  f=$(find $OUT/t_$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/d$1_$2_kernel_stats.csv
  rm -rf $OUT/t_$1
done
cat $OUT/traced_*.json
