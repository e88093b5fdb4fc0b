"""The bench line contract (driver: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line): the committed
evidence line of the final round-1 kernel must carry every field the contract names, with consistent values."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    j = json.load(open(os.path.join(ROOT, "profiles", "r01g_bench_1gpu.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is not None and 0.2 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2
    # value = batch / step time; the kernel time measured with HIP events cannot exceed the step time by more than noise
    nq = 10000
    assert abs(j["value"] - nq / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert r["kernel_ms"] <= j["ms_per_step"] * 1.02
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    m = j["config"]["matched_recall_point"]
    assert m["recall_at_10"] >= 0.9 and m["ids_identical_on_sample"] is True


def test_bench_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    text = out.stdout + out.stderr  # bench.py reserves stdout for the JSON line
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in text
