#!/bin/bash
# The driver's round-end sequence on one lease: tools/final_check.sh <out tag>  -> the whole GPU suite, smoke(), then `python bench.py` (N = 1 defaults)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/${1:-final}
mkdir -p $OUT; rm -f $OUT/*
timeout 2700 python -m pytest tests -m gpu -q > $OUT/tests.log 2>&1; echo "tests rc=$?" > $OUT/rc.txt
grep -E "passed|failed" $OUT/tests.log | tail -2; grep -E "^FAILED|^ERROR" $OUT/tests.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/rc.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4
echo "bench rc=$?" >> $OUT/rc.txt
cat $OUT/rc.txt; tail -1 $OUT/smoke.log
python - "$OUT" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench_1gpu.json").read().strip().splitlines()[-1])
print("value", j["value"], "ms/step", j["ms_per_step"], "kernel", j["roofline"]["kernel_ms"], "frac", j["roofline"]["frac"], "traffic src", j["roofline"]["traffic_source"])
c = j["config"]
print("matched", c["matched_recall_point"]["gpu_qps"], c["matched_recall_point"]["window"], "x4", c.get("batch_x4", {}).get("qps"))
print("host", c["host_pointer"]["median_ms"], c["host_pointer"]["over_device_resident"])
oc = c.get("other_configs", {})
print("c3", oc.get("c3", {}).get("qps"), "flat", {k: v for k, v in oc.get("c3", {}).get("flat_search", {}).items() if k != "what"})
print("r128", oc.get("c2_regression_graph_r128", {}).get("qps"))
print("c4", {k: v for k, v in oc.get("c4", {}).items() if k in ("qps", "recall_at_10", "window_for_recall_0.9", "oracle_sample", "error", "kernel", "vectors", "build_seconds")})
print("sharded", j.get("sharded_search"), c.get("sharded_c5", {}).get("ms_per_batch"), c.get("sharded_c5", {}).get("bound_codes"))
cb = j.get("cpu_baseline", {})
print("cpu", cb.get("value"), cb.get("cores"), cb.get("gpu_over_cpu"), cb.get("gpu_over_full_host_if_linear"))
PY
python - "$OUT" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench_1gpu.json").read().strip().splitlines()[-1])
oc = j["config"].get("other_configs", {})
print("prewarm", j.get("prewarm_steps"), "steps", [round(v, 3) for v in j["config"]["step_ms_each"][:8]])
for p in oc.get("c2_spectrum", []): print("spectrum", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items() if k in ("generator", "qps", "record_bytes", "reject_share", "speedup_over_plain_kernel", "recall_at_10", "reference_sample", "frac_traffic")})
print("u8", {k: v for k, v in oc.get("regression_uint8_r128", {}).items() if k in ("bound_scheme_chosen_by_mode_3", "qps", "recall_at_10", "reference_sample", "error")})
print("c4b", oc.get("c4", {}).get("raw_int8"))
print("phases", j["config"].get("sharded_c5", {}).get("phases_ms"))
PY
