#!/bin/bash
# Round-3 last call: rehearsal of bench.py's N > 1 control flow on one GPU (2 ranks, gloo, both on device 0), then the driver's N = 1 command.
set -u
OUT=gpurun_out/r03_last
mkdir -p $OUT
SVSX_BENCH_SINGLE_DEVICE=1 SVSX_BENCH_BACKEND=gloo timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/rehearsal_2ranks.json 2> $OUT/rehearsal.err
echo "rehearsal rc=$?"; cut -c1-1500 $OUT/rehearsal_2ranks.json; grep -v "amdgpu.ids\|^$" $OUT/rehearsal.err | tail -n 8 | cut -c1-300
python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r03_last/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'], j['roofline']['frac'], j['cpu_baseline']['gpu_over_cpu'], j['cpu_baseline']['gpu_over_full_host_if_linear'])
PY
