#!/bin/bash
# round 6, call 12: host-pointer call -- the same result arrays every call against fresh ones (what the reference's harness and idx.search do)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06n; mkdir -p $OUT; rm -f $OUT/*
SVSX_LIB=variants/hostdiag/libsvsx.so timeout 400 python tools/host_rate.py --windows 128,23 > $OUT/host_diag.jsonl 2> $OUT/host_diag.err
cat $OUT/host_diag.jsonl; grep "svsx host" $OUT/host_diag.err | head -12
timeout 400 python tools/host_rate.py --windows 128,23 > $OUT/host_prod.jsonl 2> $OUT/host_prod.err
cat $OUT/host_prod.jsonl
