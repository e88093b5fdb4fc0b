#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04big2; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_parity.py -q -k "flat" > $O/flat_tests.log 2>&1; echo "flat tests rc=$?" >> $O/out.txt
timeout 200 python tools/flat_check.py 1000000 256 10000 10 l2 >> $O/out.txt 2>/dev/null
timeout 200 python tools/flat_check.py 1000000 192 10000 10 ip >> $O/out.txt 2>/dev/null
tail -3 $O/flat_tests.log; cat $O/out.txt
