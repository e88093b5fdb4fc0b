#!/bin/bash
# PMC evidence of one round in one gpurun call:  tools/pmc_round.sh <prefix, e.g. r05> [targets...]
#   config 2 at both operating points, config 3, the config-4 shape (2 M x 768, the bench's graph parameters, the bench's window), the 12.5 M shard;
#   "c4full" = BASELINE config 4's own size (10 M x 768: ~6 min for the first build alone).
# Summaries afterwards, locally:  for t in <prefix>_c2_w128 ...; do python tools/summarize_pmc.py $t; done   (-> profiles/<tag>_pmc.json)
set -u
P=$1; shift
T=${*:-"c2 c2m c3 c4 c5"}
for t in $T; do
  case $t in
    c2rb32) tools/collect_pmc.sh ${P}_c2_w128_rb32 --n 1000000 --window 128 --codes 32 ;;
    c2mrb32) tools/collect_pmc.sh ${P}_c2_matched_rb32 --n 1000000 --window 128 --codes 32 --matched ;;
    c2) tools/collect_pmc.sh ${P}_c2_w128 --n 1000000 --window 128 --codes 2 ;;
    c2m) tools/collect_pmc.sh ${P}_c2_matched --n 1000000 --window 128 --codes 2 --matched ;;
    c3) tools/collect_pmc.sh ${P}_c3_w128 --config c3 --n 1000000 --window 128 --codes 2 ;;
    c4) tools/collect_pmc.sh ${P}_c4_w285 --config c4 --n 2000000 --nq 4096 --window 285 --degree 128 --build-window 200 --codes 2 ;;
    c4full) tools/collect_pmc.sh ${P}_c4_10M --config c4 --n 10000000 --nq 4096 --window 285 --degree 128 --build-window 200 --codes 2 ;;
    c4b) tools/collect_pmc.sh ${P}_c4b_10M --config c4b --n 10000000 --nq 4096 --window 1024 --matched --degree 128 --build-window 200 --codes 2 ;;
    c4b2m) tools/collect_pmc.sh ${P}_c4b_2M --config c4b --n 2000000 --nq 4096 --window 1024 --matched --degree 128 --build-window 200 --codes 2 ;;
    c5) tools/collect_pmc.sh ${P}_c5shard --config c5 --n 12500000 --window 128 --codes 2 ;;
  esac
done
for d in gpurun_out/${P}_*; do echo "== $d"; cut -c1-240 $d/target.json; cut -c1-160 $d/target_traced.json; done
