#!/bin/bash
# Round-4 call H: (1) SQ / cosine parity tests with the precomputed row norms, (2) config-4 shape A/B: base (round-3 arithmetic) vs precomputed norms
# (4 and 5 waves per SIMD), (3) config 2: base vs the one-round-trip set-up
set -u
OUT=gpurun_out/r04h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_bound_codes.py -m gpu -x -q --timeout 600 -k "sq or scalar or config4 or cos or similarity or golden or bit_exact or static" > $OUT/tests.log 2>&1
tail -n 6 $OUT/tests.log
timeout 900 python tools/ab_libs.py --libs base=variants/base/libsvsx.so,norms=,norms5=variants/sqcos5/libsvsx.so --script tools/window_sweep.py --repeat 1 -- --n 1000000 --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64 --degree 64 --windows 128,384,512 --modes 2 2>&1 | tee $OUT/ab_c4.txt
timeout 600 python tools/ab_libs.py --libs base=variants/base/libsvsx.so,fastsetup= --script tools/time_c2.py --repeat 2 -- --nq 10000,6144 --windows 128,23 2>&1 | tee $OUT/ab_c2_setup.txt
