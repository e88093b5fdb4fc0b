#!/bin/bash
# Round-3 PMC evidence (one gpurun call): the three regimes of round 2 on the final kernel sources + configs 3 and 4.  Summaries afterwards, locally:
#   for t in r03_c2_w128 r03_c2_matched r03_c5shard r03_c3_w128 r03_c4_w512; do python tools/summarize_pmc.py $t; done
# `tools/r3_pmc.sh tests` first runs the static-dimension parity tests and the K7 timing of this tree.
set -u
if [ "${1:-}" = "tests" ]; then
  mkdir -p gpurun_out/r03_pmc
  timeout 600 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q --timeout 400 -k "static or partial or ragged or type_pairs or f32_l2" > gpurun_out/r03_pmc/tests.log 2>&1
  grep -E "passed|failed|error" gpurun_out/r03_pmc/tests.log | tail -n 3
  timeout 200 python tools/k7_ab.py --modes 0 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_pmc/k7_r03_plain.txt
fi
tools/collect_pmc.sh r03_c2_w128 --n 1000000 --window 128 --codes 2
tools/collect_pmc.sh r03_c2_matched --n 1000000 --window 128 --codes 2 --matched
tools/collect_pmc.sh r03_c3_w128 --config c3 --n 1000000 --window 128 --codes 2
tools/collect_pmc.sh r03_c4_w512 --config c4 --n 2000000 --nq 4096 --window 512 --codes 2
tools/collect_pmc.sh r03_c5shard --config c5 --n 12500000 --window 128 --codes 2
for t in r03_c2_w128 r03_c2_matched r03_c3_w128 r03_c4_w512 r03_c5shard; do echo "== $t"; cut -c1-160 gpurun_out/$t/target.json; done
