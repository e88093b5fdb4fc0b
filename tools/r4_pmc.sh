#!/bin/bash
# Round-4 PMC evidence (one gpurun call): config 2 at both operating points, the 12.5 M shard, config 3, config 4 (2 M, window 512 like round 3).
# Summaries afterwards, locally:  for t in r04_c2_w128 r04_c2_matched r04_c5shard r04_c3_w128 r04_c4_w512; do python tools/summarize_pmc.py $t; done
set -u
tools/collect_pmc.sh r04_c2_w128 --n 1000000 --window 128 --codes 2
tools/collect_pmc.sh r04_c2_matched --n 1000000 --window 128 --codes 2 --matched
tools/collect_pmc.sh r04_c3_w128 --config c3 --n 1000000 --window 128 --codes 2
tools/collect_pmc.sh r04_c4_w512 --config c4 --n 2000000 --nq 4096 --window 512 --codes 2
tools/collect_pmc.sh r04_c5shard --config c5 --n 12500000 --window 128 --codes 2
for t in r04_c2_w128 r04_c2_matched r04_c3_w128 r04_c4_w512 r04_c5shard; do echo "== $t"; cut -c1-200 gpurun_out/$t/target.json; cut -c1-200 gpurun_out/$t/target_traced.json; done
