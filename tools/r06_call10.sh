#!/bin/bash
# round 6, call 10: PMC evidence on the final kernel sources (again: call 9's payload exceeded what gpurun merges back), flat counters, host-path variants
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r06j; rm -f gpurun_out/r06j/*
rm -rf gpurun_out/r06_c2_w128 gpurun_out/r06_c2_matched gpurun_out/r06_c3_w128 gpurun_out/r06_c5shard gpurun_out/r06_c4_10M gpurun_out/r06_c4b_10M gpurun_out/pmc_calib gpurun_out/r06flat
timeout 3300 tools/pmc_round.sh r06 c2 c2m c3 c5 c4full c4b > gpurun_out/r06j/pmc_round.log 2>&1; echo "pmc rc=$?" > gpurun_out/r06j/rc.txt
timeout 900 tools/flat_pmc.sh r06 > gpurun_out/r06j/flat_pmc.log 2>&1; echo "flat rc=$?" >> gpurun_out/r06j/rc.txt
for v in hostf4096 hostf2048 hostf1024 hostf512; do
  SVSX_LIB=variants/$v/libsvsx.so timeout 300 python tools/host_rate.py --windows 128,23 > gpurun_out/r06j/host_$v.jsonl 2> gpurun_out/r06j/host_$v.err
  echo "== $v"; cut -c1-260 gpurun_out/r06j/host_$v.jsonl; grep "svsx host" gpurun_out/r06j/host_$v.err | tail -4
done
cat gpurun_out/r06j/rc.txt
du -sh gpurun_out/r06_* gpurun_out/r06flat gpurun_out/pmc_calib 2>/dev/null | tail -12; du -sh gpurun_out
