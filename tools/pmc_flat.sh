#!/bin/bash
# SQ / TCC counters of the flat-search MFMA kernel (diagnostic; run through gpurun)
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pq; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/tools/flat_bench.py > /tmp/flat.log 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flat_mfma_kernel" in r["Kernel_Name"] or "flat_thresh" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:40], int(r["Dispatch_Id"]))][r["Counter_Name"]] += float(r["Counter_Value"])
keys = sorted(acc)[:2]
for k in keys:
    for nm in sorted(acc[k]):
        print(f"{k[0]} {nm:28s} {acc[k][nm]:16.0f}")
PY
done
tail -2 /tmp/flat.log
