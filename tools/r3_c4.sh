#!/bin/bash
# Config 4 shape with the reference's regression graph parameters (R = 128, prune_to 124, L = 200; tools/benchmark_inputs/vamana/static-regression.toml)
set -u
OUT=gpurun_out/r03_c4
mkdir -p $OUT
N=${1:-2000000}
timeout 900 python tools/window_sweep.py --n $N --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64 --degree 128 --build-window 200 --windows 64,96,128,192,256,384,512 --modes 2 2>/dev/null | cut -c1-330 | tee $OUT/r128_n$N.txt
python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_c4/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['config']['host_pointer'], j['roofline']['frac'])
PY
