#!/bin/bash
# round 4, call k: flat-search exactness (proof + exhaustive pass), k > 96, golden flat vectors, builder at scale; flat timing
cd /root/repo; mkdir -p gpurun_out/r04k; O=gpurun_out/r04k
timeout 900 python -m pytest tests/test_gpu_flat_exact.py -q > $O/flat_exact.log 2>&1; echo "flat_exact rc=$?" >> $O/rc.txt
timeout 600 python -m pytest tests/test_gpu_configs.py -q -k "flat or builder_at_scale" > $O/configs.log 2>&1; echo "configs rc=$?" >> $O/rc.txt
cat $O/rc.txt; tail -8 $O/flat_exact.log; tail -5 $O/configs.log
