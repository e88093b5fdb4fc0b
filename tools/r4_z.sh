#!/bin/bash
# Round-4 final check: the whole GPU suite and smoke() on the final library
set -u
OUT=gpurun_out/r04z
mkdir -p $OUT; rm -f $OUT/*
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
tail -n 4 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -1 $OUT/smoke.log
