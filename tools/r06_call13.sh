#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r06o; mkdir -p $OUT; rm -f $OUT/*
timeout 900 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06o/bench.json").read().strip().splitlines()[-1])
print("value", j["value"], "kernel", j["roofline"]["kernel_ms"])
print(json.dumps(j["config"]["host_pointer"], indent=None)[:1600])
PY
