#!/bin/bash
OUT=gpurun_out/r05flat7; mkdir -p $OUT
echo "== new test on the pre-fix library (expected to fail if the hazard bites)" | tee -a $OUT/summary.txt
SVSX_LIB=$PWD/variants/flat_tr64/libsvsx.so timeout 600 python -m pytest tests/test_gpu_flat_exact.py -q -k last_sixteen 2>&1 | tail -15 | tee -a $OUT/summary.txt
echo "== all flat tests on the in-tree library" | tee -a $OUT/summary.txt
bash tools/r5_flat.sh r05flat7
