#!/bin/bash
# Round-3 closing call: the driver's bench command (+ rocprofv3 kernel trace of the same command), smoke, the whole GPU suite.
set -u
OUT=gpurun_out/r03_final
mkdir -p $OUT
python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err
tail -n 4 $OUT/bench.err | cut -c1-200
cut -c1-600 $OUT/bench_1gpu.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > /dev/null 2>&1 )
find $OUT/trace -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -delete
head -n 6 $OUT/bench_kernel_stats.csv | cut -c1-200
python __graft_entry__.py smoke 2>&1 | tail -n 2 | cut -c1-300
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -n 3
