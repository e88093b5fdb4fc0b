#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/collect_pmc.sh r01
# 1. bench.py once (builds + caches the graph), 2. kernel-trace/stats run, 3. one --pmc pass per counter group
# (never combined with other trace domains), 4. FETCH_SIZE calibration stream.  Outputs under gpurun_out/<round>/.
set -u
ROUND=${1:-r01}
BA=${BENCH_ARGS:-}   # extra bench.py arguments, e.g. BENCH_ARGS="--bound-codes 1"
R=$PWD
OUT=$R/gpurun_out/$ROUND
mkdir -p $OUT
python bench.py $BA --graph-cache /tmp/g1m.pt > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py $BA --graph-cache /tmp/g1m.pt --no-cpu-baseline > $OUT/bench_traced.json 2> /dev/null
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$N -o p -- python $R/bench.py $BA --graph-cache /tmp/g1m.pt --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_calib -o p -- python $R/tools/pmc_calib.py > $OUT/calib.log 2>&1
cd $R
# keep the merged-back payload small: drop the giant per-dispatch kernel-trace CSVs of the PMC passes
find $OUT -name "p_kernel_trace.csv" -delete
