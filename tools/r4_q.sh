#!/bin/bash
R=$PWD; O=$R/gpurun_out/r04big; mkdir -p $O; rm -rf $O/*
timeout 900 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_parity.py tests/test_gpu_configs.py -q -k "flat" > $O/flat_tests.log 2>&1; echo "flat tests rc=$?" >> $O/out.txt
timeout 200 python tools/flat_check.py 200000 768 4096 100 l2 >> $O/out.txt 2>/dev/null
timeout 200 python tools/flat_check.py 2000000 768 4096 10 cos >> $O/out.txt 2>/dev/null
tail -4 $O/flat_tests.log; cat $O/out.txt
