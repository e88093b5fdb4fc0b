#!/bin/bash
# rocprofv3 evidence for ONE PMC target on the GPU box (run through gpurun from the repo root):
#   tools/collect_pmc.sh <tag> <pmc_target.py args...>      e.g.  tools/collect_pmc.sh r02_c2_w128 --n 1000000 --window 128 --codes 2
# 1. un-profiled run (builds + caches the graph, prints the target's JSON), 2. --kernel-trace --stats run, 3. one --pmc pass per
# counter group (never combined with other trace domains), 4. FETCH_SIZE calibration stream.  Outputs under gpurun_out/<tag>/.
set -u
TAG=$1; shift
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
ARGS="$* --graph-cache /tmp/gcache"
python tools/pmc_target.py $ARGS > $OUT/target.json 2> $OUT/target.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/pmc_target.py $ARGS > $OUT/target_traced.json 2> /dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"; do
  N=$(echo $C | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o p -- python $R/tools/pmc_target.py $ARGS > /dev/null 2>&1
done
if [ ! -d $R/gpurun_out/pmc_calib ]; then
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_calib -o p -- python $R/tools/pmc_calib.py > $R/gpurun_out/pmc_calib.log 2>&1
fi
cd $R
# keep the merged-back payload small (gpurun merges at most 64 MiB back): drop the per-dispatch kernel-trace CSVs of the PMC passes, and of the
# other CSVs keep the header and the rows of the search kernel only (a 10 M-vector target also traces the graph builder's ~10^5 dispatches)
find $OUT $R/gpurun_out/pmc_calib -name "p_kernel_trace.csv" -delete
for f in $(find $OUT -name "*counter_collection.csv" -o -name "t_kernel_trace.csv"); do
  (head -1 $f; grep search_kernel $f) > $f.tmp && mv $f.tmp $f
done
find $OUT -name "*agent_info.csv" -delete
# summarise locally after the merge: python tools/summarize_pmc.py $TAG
