#!/bin/bash
# round 6, call 16: the whole GPU suite on the sources with the 28-byte record form, then smoke
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06r; mkdir -p $OUT; rm -f $OUT/*
timeout 2700 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
tail -1 $OUT/smoke.log; cat $OUT/rc.txt
