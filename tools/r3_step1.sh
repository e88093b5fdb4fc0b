#!/bin/bash
# Round-3 opening call (staging/r3/PLAN.md step 1), to be run ON THE GPU BOX through gpurun after, locally:
#   patch -p0 < staging/r3/early_block.patch && make -C scalablevectorsearch_amd/csrc -j8 && tools/build_variant.sh early0 -DSVSX_EARLY_BLOCK=0
#   gpurun --timeout 900 -- 'tools/r3_step1.sh'
# 1. parity of the patched library (results cannot depend on the early block, but prove it), 2. A/B against the same sources without it.
set -u
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_bound_codes.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/tests.log 2>&1
tail -3 $OUT/tests.log
python tools/ab_libs.py --libs off=variants/early0/libsvsx.so,early= --repeat 2 --script tools/ab_lookahead.py -- --policies off --nq 10000,20000 --windows 128,23 --steps 20 > $OUT/ab_config2.txt 2>&1
cat $OUT/ab_config2.txt
for L in variants/early0/libsvsx.so ""; do
  echo "== config 4 shape (2 M x 768 SQ-int8 cosine), library: ${L:-in-tree}"
  SVSX_LIB=$L timeout 200 python tools/window_sweep.py --n 2000000 --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64 --windows 128,512,1024 --modes 2 2>/dev/null | cut -c1-220
done > $OUT/ab_config4.txt 2>&1
cat $OUT/ab_config4.txt
