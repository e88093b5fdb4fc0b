#!/bin/bash
# tools/flat_pmc_issue.sh <tag>: issue / wait counters of the flat threshold kernel's full pass (in-tree library), 10 000 x 1 M x 96 L2
TAG=${1:-r05flatp}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$i -o p -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > $OUT/pmc_$i.log 2>&1
  echo "pmc_$i rc=$?" | tee -a $OUT/rc.txt
done
cd $R
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "flat_thresh_wg" in r["Kernel_Name"]]
    # the full pass = the dispatches with the largest grid? take per dispatch sums, print the 4 largest by SQ_WAVE_CYCLES-like first counter
    by = collections.defaultdict(dict)
    for r in rows: by[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    top = sorted(by.values(), key=lambda d: -max(d.values()))[:1]
    for d in top: print(f.split("/")[-3] if "/" in f else f, {k: round(v) for k, v in d.items()})
PY
