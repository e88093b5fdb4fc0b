#!/bin/bash
OUT=gpurun_out/r05benchq; mkdir -p $OUT
for w in 3 20 3; do
  timeout 600 python bench.py --no-extras --no-cpu-baseline --warmup $w --graph-cache /tmp/gcache > $OUT/b_w$w.json 2>$OUT/b_w$w.err || tail -5 $OUT/b_w$w.err
  python3 -c "
import json,sys
j=json.load(open('$OUT/b_w$w.json'))
print('warmup',$w,'value',round(j['value']),'kernel',j['roofline']['kernel_ms'], j['config']['step_ms_each'])
" | tee -a $OUT/summary.txt
done
