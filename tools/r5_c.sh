#!/bin/bash
# round 5, call c: the whole GPU suite, A/B against the round-4 library (config 2, config-4 shape incl. 128-byte records), phases
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 600 python tools/ab_libs.py --libs prod=variants/prod/libsvsx.so,new= --script tools/time_c2.py --repeat 2 -- --nq 10000 --windows 128,23 > $O/ab_c2.txt 2>&1
timeout 900 python tools/ab_libs.py --libs prod=variants/prod/libsvsx.so,new= --script tools/time_c4.py --repeat 1 -- --windows 285,128 > $O/ab_c4.txt 2>&1
timeout 300 python tools/time_c4.py --windows 285 --records 128 > $O/c4_rec128.txt 2>&1
SVSX_LIB=variants/new_diag2/libsvsx.so timeout 300 python tools/time_c4.py --phases --windows 285 > $O/phases_c4_new.txt 2>&1
grep -E "passed|failed" $O/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $O/pytest.log | head -20; cat $O/ab_c2.txt $O/ab_c4.txt $O/c4_rec128.txt $O/phases_c4_new.txt | tail -12
