#!/bin/bash
# round 5, call d: the tests that changed, config 4 with the record size the model now picks, a rehearsal of the restructured bench line
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_native_builder.py tests/test_gpu_iterator.py "tests/test_gpu_parity.py::test_argument_errors" "tests/test_gpu_parity.py::test_flat_search_k_beyond_96" "tests/test_gpu_configs.py::test_flat_search_beyond_the_shortlist_limits" "tests/test_gpu_parity.py::test_sq_cosine_row_norms_dropped_and_refreshed" -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log
timeout 300 python tools/time_c4.py --windows 285 > $O/c4_default.txt 2>&1
( time timeout 900 python bench.py --c5-n 200000 --c4-n 0 --steps 5 --warmup 2 > $O/bench_small.json 2> $O/bench_small.err ) 2>> $O/bench_small.err; echo "bench rc=$?" >> $O/rc.txt
timeout 600 python bench.py --c5-n 200000 --c4-n 0 --no-cpu-baseline --steps 3 --warmup 1 --sharded-timeout 0.5 > $O/bench_dog.json 2> $O/bench_dog.err; echo "watchdog rc=$? (3 expected)" >> $O/rc.txt
grep -E "passed|failed" $O/pytest.log | tail -2; grep -E "^FAILED|^ERROR" $O/pytest.log | head; cat $O/c4_default.txt | tail -1 | cut -c1-400; cat $O/rc.txt
python - <<'PY'
import json
for f in ("gpurun_out/r05d/bench_small.json", "gpurun_out/r05d/bench_dog.json"):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j.get("sharded_search"), j["config"].get("note"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $O/bench_small.err
