#!/bin/bash
# tools/flat_diag.sh <tag> variants...: timing builds of the flat threshold kernel (results wrong by design), per-kernel times from a trace of each
TAG=${1:-r05flatd}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for v in "$@"; do
  export SVSX_LIB=$PWD/variants/$v/libsvsx.so
  for cfg in "1000000 96 10000 10 l2" "1000000 128 10000 10 ip"; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o t -- python tools/flat_check.py $cfg > $OUT/$v.log 2>&1
    f=$(find $OUT/$v -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && grep "flat_thresh_wg" "$f" | awk -F'","' -v v=$v '{print v, $1, "max_ns", $7}' | tee -a $OUT/summary.txt
    rm -rf $OUT/$v
  done
done
