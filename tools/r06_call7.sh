#!/bin/bash
# round 6, call 7: similarity metrics on offset data, centred records (in-tree) against the library before them; the test files call 6 did not run
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06g; mkdir -p $OUT; rm -f $OUT/*
SVSX_LIB=variants/uncentred/libsvsx.so timeout 900 python tools/offset_ab.py > $OUT/offset_uncentred.jsonl 2> $OUT/offset.err
timeout 900 python tools/offset_ab.py > $OUT/offset_centred.jsonl 2>> $OUT/offset.err; echo "offset rc=$?" > $OUT/rc.txt
cat $OUT/offset_uncentred.jsonl $OUT/offset_centred.jsonl | cut -c1-420
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_iterator.py tests/test_gpu_flat_exact.py tests/test_gpu_native_builder.py tests/test_gpu_multi.py tests/test_abi.py -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/rc.txt
tail -5 $OUT/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
tail -1 $OUT/smoke.log
cat $OUT/rc.txt
