#!/bin/bash
# Flat-search evidence (tools/flat_pmc.sh <round prefix>): kernel trace + counters of svsx_flat_search_device, 10 000 queries x 1 M x 96 float32 (L2) and x 128 float16 (IP)
set -u
R=$PWD; O=$R/gpurun_out/${1:-r05}flat; mkdir -p $O; rm -rf $O/*
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > $O/traced.json 2>/dev/null
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > /dev/null 2>&1
  echo "pmc_$i rc=$?" >> $O/rc.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -delete
cat $O/rc.txt; cat $O/traced.json
