#!/bin/bash
# round 4, call o: flat search after the threshold kernel's epilogue rewrite
cd /root/repo; O=gpurun_out/r04o; mkdir -p $O; rm -f $O/*
timeout 600 python -m pytest tests/test_gpu_flat_exact.py tests/test_gpu_parity.py -q -k "flat" > $O/flat_tests.log 2>&1; echo "flat tests rc=$?" >> $O/rc.txt
timeout 200 python tools/flat_check.py 1000000 96 10000 10 >> $O/flat_k.jsonl 2>/dev/null
for K in 32 100; do timeout 200 python tools/flat_check.py 1000000 96 10000 $K l2 >> $O/flat_k.jsonl 2>/dev/null; done
timeout 200 python tools/flat_check.py 1000000 128 10000 10 l2 >> $O/flat_k.jsonl 2>/dev/null
cat $O/rc.txt; tail -3 $O/flat_tests.log; cat $O/flat_k.jsonl
