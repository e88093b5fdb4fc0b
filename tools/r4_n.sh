#!/bin/bash
# round 4, call n: host-pointer call, page-locked staging vs direct pageable copies (variants), and flat search after the re-rank capacity fix
cd /root/repo; O=gpurun_out/r04n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_parity.py -q -k "flat" > $O/flat_tests.log 2>&1; echo "flat tests rc=$?" >> $O/rc.txt
for K in 32 100 200; do timeout 200 python tools/flat_check.py 1000000 96 10000 $K l2 >> $O/flat_k.jsonl 2>/dev/null; done
for V in base hd4096 hd1 hd2048 hs3072; do
  if [ $V = base ]; then unset SVSX_LIB; else export SVSX_LIB=$PWD/variants/$V/libsvsx.so; fi
  echo "== $V" >> $O/host_rate.jsonl
  timeout 300 python tools/host_rate.py --windows 128,23 >> $O/host_rate.jsonl 2>/dev/null
done
unset SVSX_LIB
cat $O/rc.txt; tail -3 $O/flat_tests.log; cat $O/flat_k.jsonl; cut -c1-260 $O/host_rate.jsonl
