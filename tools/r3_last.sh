#!/bin/bash
# Round-3 last call: the filtered-search / iterator tests, the driver's N = 1 bench command, K7 before / after (round-2 tree vs this tree, same graphs),
# config 4 shape at the windows around the 6-entries-per-lane merge level.
set -u
OUT=gpurun_out/r03_last
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_iterator.py tests/test_abi.py -m gpu -q --timeout 200 > $OUT/tests.log 2>&1
grep -E "passed|failed|error" $OUT/tests.log | tail -n 3
python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_last/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'], j['roofline']['frac'], j['roofline']['kernel_src_sha'], j['cpu_baseline']['gpu_over_cpu'])
PY
echo "== K7, this tree"; timeout 200 python tools/k7_ab.py --graph-cache /tmp/k7g 2>&1 | grep -v amdgpu.ids | tee $OUT/k7_r03.txt
echo "== K7, round-2 tree"; timeout 200 python tools/k7_ab.py --graph-cache /tmp/k7g --tree variants/r02tree 2>&1 | grep -v amdgpu.ids | tee $OUT/k7_r02.txt
timeout 300 python tools/window_sweep.py --n 2000000 --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64 --degree 128 --build-window 200 --windows 256,320,384,512 --modes 2 2>/dev/null | cut -c1-300 | tee $OUT/c4_epl6.txt
