"""The bench line contract (driver: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line): the committed
evidence line of the final round-3 kernels must carry every field the contract names, with consistent values."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    j = json.load(open(os.path.join(ROOT, "profiles", "r03_bench_1gpu.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in j, key
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "frac_basis", "frac_designed", "achieved_designed", "traffic", "kernel_src_sha",
                "designed_bytes_per_launch", "reference_algorithmic_bytes_per_launch"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] <= 1.0 and 0.0 < r["frac_designed"] <= 1.0
    assert r["reference_algorithmic_bytes_per_launch"] > r["designed_bytes_per_launch"]
    # round 3: the headline fraction is the counter-based one (PMC traffic of THESE kernel sources / this run's kernel time)
    assert r["traffic"] is not None and r["traffic_source"].startswith("profiles/r03_") and "PMC" in r["frac_basis"]
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert 0.5 < r["traffic"] / r["designed_bytes_per_launch"] < 1.3 and r["frac"] <= r["frac_designed"] * 1.3
    # value = batch / step time; the kernel time measured with HIP events cannot exceed the step time by more than noise
    nq = 10000
    assert abs(j["value"] - nq / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-6
    assert r["kernel_ms"] <= j["ms_per_step"] * 1.02
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "cores_present", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and 1 <= c["cores"] <= c["cores_present"]
    m = j["config"]["matched_recall_point"]
    assert m["recall_at_10"] >= 0.9 and m["ids_identical_on_sample"] is True and m["frac"] <= 1.0
    hp = j["config"]["host_pointer"]                    # SURVEY 8d wording: host pointers in / out, 1 warm-up + 5 timed
    assert hp["runs"] == 5 and hp["min_ms"] <= hp["median_ms"] <= hp["max_ms"] and hp["ids_equal_device_path"] is True
    assert hp["qps_at_median"] < j["value"] and 0.6 < hp["over_device_resident"] < 1.0
    sh = j["config"]["sharded_c5"]                      # SURVEY 8e in the same line: fixed 8 shards, RCCL communicator inside the library
    assert sh["shards_total"] == 8 and sh["shards_per_gpu"] * sh["n_gpus"] == 8 and sh["rccl_communicator"] is True and sh["recall_at_10"] > 0.99
    c3 = j["config"]["other_configs"]["c3"]
    assert ("codes=2>" in c3["kernel"] or "codes=5>" in c3["kernel"]) and c3["recall_at_10"] > 0.9


def test_pmc_summaries_are_stamped_and_consistent():
    """each committed round-6 PMC summary names its workload and the kernel sources it was measured on, its traffic agrees with the designed
    bytes of the same launches (no wasted re-reads), and the kernel trace of the profiled process agrees with that process's own HIP-event time
    (within 5 %: the tracer adds ~10-30 us per dispatch, 3 % of a 0.31 ms launch)"""
    shas = set()
    for tag in ("r06_c2_w128_rb32", "r06_c2_matched_rb32", "r06_c2_w128", "r06_c2_matched", "r06_c5shard", "r06_c3_w128", "r06_c4_10M", "r06_c4b_10M"):
        j = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_pmc.json")))
        assert len(j["kernel_src_sha"]) == 16 and j["workload_key"].count(":") >= 6
        shas.add(j["kernel_src_sha"])
        assert abs(j["fetch_size_calibration"]["factor"] - 2.0) < 0.01
        assert 0.5 < j["traffic_over_designed"] < 1.12
        assert j["frac_traffic_of_8TBs"] < 1.0 and j["kernel_trace"]["launches"] == j["launches"]
        km = j["kernel_ms"]  # warmed process: trace and HIP events within 5 % of each other, and of the un-profiled process
        assert abs(km["kernel_trace_avg_same_traced_process"] / km["hip_events_in_the_traced_process"] - 1.0) < 0.05
        assert abs(km["hip_events_in_the_traced_process"] / km["hip_events_unprofiled_process"] - 1.0) < 0.05
    assert len(shas) == 1  # one set of kernel sources
    sys.path.insert(0, ROOT)
    from bench import kernel_src_sha

    assert shas == {kernel_src_sha()}, "the committed PMC summaries belong to other kernel sources: re-collect (tools/pmc_round.sh r06)"


def test_counter_generator_is_reproducible():
    """bench.py's data generator is a pure function of (seed, stream, row, col): chunked generation equals one-shot generation"""
    sys.path.insert(0, ROOT)
    import torch

    from bench import counter_normal, deep_like_device

    a = counter_normal(1000, 8, 3, 1, "cpu")
    b = torch.cat([counter_normal(400, 8, 3, 1, "cpu"), counter_normal(600, 8, 3, 1, "cpu", row0=400)])
    assert torch.equal(a, b) and abs(float(a.mean())) < 0.05 and abs(float(a.std()) - 1.0) < 0.05
    x = deep_like_device(100, 96, 24, 1, "cpu")
    assert torch.allclose(x.norm(dim=1), torch.ones(100), atol=1e-5)
    assert float(x[0, 0]) == float(deep_like_device(100, 96, 24, 1, "cpu")[0, 0])


def test_bench_cli_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    text = out.stdout + out.stderr  # bench.py reserves stdout for the JSON line
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in text
