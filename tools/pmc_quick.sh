#!/bin/bash
# One SQ counter pass over the bench's window-128 launches (diagnostic; run through gpurun):  tools/pmc_quick.sh "--bound-codes 1"
R=$PWD; BA=${1:-}
python bench.py $BA --graph-cache /tmp/g1m.pt --no-cpu-baseline > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pq; rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pq -o p -- python $R/bench.py $BA --graph-cache /tmp/g1m.pt --no-cpu-baseline --steps 3 --warmup 1 > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "search_kernel" in r["Kernel_Name"]:
            acc[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
first = sorted(acc)[:5]
names = sorted({k for d in first for k in acc[d]})
for nm in names:
    print(f"{nm:28s} {sum(acc[d][nm] for d in first)/len(first):16.0f}")
PY
done
