#!/bin/bash
# round 4, call m: where flat search's time goes (kernel trace + counters), 1 M x 96 float32, 10 000 queries, k = 10 and k = 100
set -u
R=$PWD; O=$R/gpurun_out/r04m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for K in 10 100; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_k$K -o t -- python $R/tools/flat_check.py 1000000 96 10000 $K l2 > $O/traced_k$K.json 2>/dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_768 -o t -- python $R/tools/flat_check.py 200000 768 4096 10 l2 > $O/traced_768.json 2>/dev/null
i=0
for C in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$i -o p -- python $R/tools/flat_check.py 1000000 96 10000 10 l2 > /dev/null 2>&1
  echo "pmc_$i rc=$?" >> $O/rc.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -size +4M -delete
find $O -name "*stats*.csv" | head; cat $O/rc.txt
for K in k10 k100 768; do echo "== $K"; cat $O/traced_$K.json; f=$(find $O/trace_$K -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-170; done
