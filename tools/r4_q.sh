#!/bin/bash
set -u
OUT=gpurun_out/r04bench
mkdir -p $OUT; rm -f $OUT/*
( time python bench.py > $OUT/bench_1gpu.json 2> $OUT/bench.err ) 2>&1 | tail -n 4
echo "bench rc=$?"
tail -n 3 $OUT/bench.err
python - <<'PY'
import json
j=json.load(open('gpurun_out/r04bench/bench_1gpu.json'))
print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms'], j['roofline']['frac'])
print('flat', j['config']['other_configs']['c3'].get('flat_search'))
print('host', j['config']['host_pointer']['median_ms'], j['config']['host_pointer']['over_device_resident'])
print('c4', {k:v for k,v in j['config']['other_configs']['c4'].items() if k in ('qps','recall_at_10','window_for_recall_0.9','oracle_sample')})
print('sharded', j.get('sharded_search',{}).get('value'), 'gt', j['config']['ground_truth'].get('queries_recomputed_exhaustively'), j['config']['ground_truth'].get('id_sets_equal_to_float64_brute_force'))
PY
