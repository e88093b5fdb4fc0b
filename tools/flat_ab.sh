#!/bin/bash
# tools/flat_ab.sh <tag> [variants...]: flat-search A/B on one GPU lease -- parity tests on the in-tree library, then tools/flat_check.py per library
TAG=${1:-r05flat}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 600 python -m pytest tests/test_gpu_flat_exact.py -q -x > $OUT/tests.log 2>&1; echo "flat tests rc=$?" | tee -a $OUT/summary.txt
tail -2 $OUT/tests.log | tee -a $OUT/summary.txt
for v in in-tree "$@"; do
  if [ "$v" = in-tree ]; then unset SVSX_LIB; else export SVSX_LIB=$PWD/variants/$v/libsvsx.so; fi
  for cfg in "1000000 96 10000 10 l2" "1000000 128 10000 10 ip" "1000000 96 10000 10 cos" "1000000 64 10000 10 l2"; do
    timeout 300 python tools/flat_check.py $cfg 2>>$OUT/err.log | sed "s/^/$v /" | tee -a $OUT/summary.txt
  done
done
unset SVSX_LIB
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python tools/flat_check.py 1000000 96 10000 10 l2 > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee -a $OUT/summary.txt
