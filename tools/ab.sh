#!/bin/bash
# tools/ab.sh <outdir> <label> <SVSX_LIB or ""> [bench args...]  -> one compact line per run
OUT=$1; L=$2; LIB=$3; shift 3
mkdir -p $OUT
SVSX_LIB=$LIB python bench.py --graph-cache /tmp/g1m.pt --no-cpu-baseline "$@" > $OUT/$L.json 2> $OUT/$L.err
python - "$OUT/$L.json" "$L" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = j["config"]
    print(f"{sys.argv[2]:28s} QPS {j['value']/1e6:6.3f} M  kern {j['roofline']['kernel_ms']:.3f} ms  matched w{c['matched_recall_point']['window']} {c['matched_recall_point']['gpu_qps']/1e6:6.2f} M  evals {c['distance_evals_per_query']:.0f} rej {c['bound_code_rejects_per_query']:.0f} ins {c['accepted_inserts_per_query']:.0f} recall {c['recall_at_10']:.4f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
