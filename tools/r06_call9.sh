#!/bin/bash
# round 6, call 9: PMC evidence on the final kernel sources (tools/pmc_round.sh r06 ...), flat-search counters, the mode-3 test
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r06i; rm -f gpurun_out/r06i/*
timeout 600 python -m pytest tests/test_gpu_bound_codes.py -m gpu -q -x -k "mode_3 or fall_back or scope or small_world" > gpurun_out/r06i/tests.log 2>&1; echo "tests rc=$?" > gpurun_out/r06i/rc.txt
tail -3 gpurun_out/r06i/tests.log
rm -rf gpurun_out/r06_c2_w128 gpurun_out/r06_c2_matched gpurun_out/r06_c3_w128 gpurun_out/r06_c5shard gpurun_out/r06_c4_10M gpurun_out/r06_c4b_10M gpurun_out/pmc_calib
timeout 3300 tools/pmc_round.sh r06 c2 c2m c3 c5 c4full c4b > gpurun_out/r06i/pmc_round.log 2>&1; echo "pmc rc=$?" >> gpurun_out/r06i/rc.txt
tail -20 gpurun_out/r06i/pmc_round.log | cut -c1-260
timeout 900 tools/flat_pmc.sh r06 > gpurun_out/r06i/flat_pmc.log 2>&1; echo "flat rc=$?" >> gpurun_out/r06i/rc.txt
timeout 600 python tools/concurrency_probe.py > gpurun_out/r06i/concurrency_probe.jsonl 2> gpurun_out/r06i/probe.err; echo "probe rc=$?" >> gpurun_out/r06i/rc.txt
cat gpurun_out/r06i/concurrency_probe.jsonl
cat gpurun_out/r06i/rc.txt
du -sh gpurun_out/r06_* gpurun_out/pmc_calib 2>/dev/null | tail -12
