#!/bin/bash
# round 6, call 18: counters of the headline's 32-byte-record kernel on the final sources (the rule-sized 28-byte form: call 17), then the round-end sequence
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r06t; rm -f gpurun_out/r06t/*
rm -rf gpurun_out/r06_c2_w128_rb32 gpurun_out/r06_c2_matched_rb32 gpurun_out/pmc_calib
timeout 1500 tools/pmc_round.sh r06 c2rb32 c2mrb32 > gpurun_out/r06t/pmc_round.log 2>&1; echo "pmc rc=$?" > gpurun_out/r06t/rc.txt
python tools/summarize_pmc.py r06_c2_w128_rb32 > /dev/null 2>&1; python tools/summarize_pmc.py r06_c2_matched_rb32 > /dev/null 2>&1
ls profiles/r06_c2_w128_rb32_pmc.json profiles/r06_c2_matched_rb32_pmc.json
tools/final_check.sh r06final3
cat gpurun_out/r06t/rc.txt
