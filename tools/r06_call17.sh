#!/bin/bash
# round 6, call 17: counters on the final kernel sources (with the 28-byte record form), flat-search counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r06s; rm -f gpurun_out/r06s/*
rm -rf gpurun_out/r06_c2_w128 gpurun_out/r06_c2_matched gpurun_out/r06_c3_w128 gpurun_out/r06_c5shard gpurun_out/r06_c4_10M gpurun_out/r06_c4b_10M gpurun_out/pmc_calib gpurun_out/r06flat
timeout 3300 tools/pmc_round.sh r06 c2 c2m c3 c5 c4full c4b > gpurun_out/r06s/pmc_round.log 2>&1; echo "pmc rc=$?" > gpurun_out/r06s/rc.txt
timeout 900 tools/flat_pmc.sh r06 > gpurun_out/r06s/flat_pmc.log 2>&1; echo "flat rc=$?" >> gpurun_out/r06s/rc.txt
cat gpurun_out/r06s/rc.txt; tail -14 gpurun_out/r06s/pmc_round.log | cut -c1-200; du -sh gpurun_out
