#!/bin/bash
OUT=gpurun_out/r05last; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_configs.py -q -x > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/summary.txt
tail -3 $OUT/tests.log | tee -a $OUT/summary.txt
for cfg in "1000000 96 10000 10 l2" "1000000 128 10000 10 ip" "1000000 96 10000 32 l2"; do timeout 300 python tools/flat_check.py $cfg 2>>$OUT/err.log | tee -a $OUT/summary.txt; done
