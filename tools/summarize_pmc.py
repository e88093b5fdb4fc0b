#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/collect_pmc.sh) -> profiles/<tag>_pmc.json (+ <tag>_kernel_stats.csv).
Per-launch means over the search_kernel dispatches BETWEEN the two marker launches of tools/pmc_target.py."""
import collections, csv, glob, json, os, statistics, sys

tag = sys.argv[1]
base = f"gpurun_out/{tag}"
target = json.loads(open(f"{base}/target.json").read().strip().splitlines()[-1])
n_launch = target["launches"]


def target_dispatches(rows):
    """rows of search_kernel dispatches in launch order -> those between the two window-1 marker launches (LDS_Block_Size smallest)"""
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    lds = {int(r["Dispatch_Id"]): int(r["LDS_Block_Size"]) for r in rows}
    grid = {int(r["Dispatch_Id"]): int(r["Grid_Size"]) for r in rows}
    marks = [i for i in ids if grid[i] == 64 * 64]
    assert len(marks) >= 2, marks
    return [i for i in ids if marks[0] < i < marks[1]][:n_launch]


def per_kernel(dirn, kname):
    rows = []
    for f in glob.glob(f"{base}/{dirn}/*counter_collection.csv"):
        rows += [r for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
    keep = set(target_dispatches(rows)) if kname == "search_kernel" else None
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in rows:
        if keep is None or int(r["Dispatch_Id"]) in keep:
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


out = {}
for d in sorted(glob.glob(f"{base}/pmc_*")):
    a = per_kernel(os.path.basename(d), "search_kernel")
    for nm in sorted({k for v in a.values() for k in v}):
        out[nm] = statistics.mean(v[nm] for v in a.values())
cal_rows = []
for f in glob.glob("gpurun_out/pmc_calib/*counter_collection.csv"):
    cal_rows += [r for r in csv.DictReader(open(f)) if "distance_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
cal = sum(float(r["Counter_Value"]) for r in cal_rows)
known = 3_000_000 * 96 * 4
factor = known / (cal * 1024) if cal else float("nan")
nq = 10000
s = dict(target)
s.update({
    "command": "tools/collect_pmc.sh %s ... : rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python tools/pmc_target.py <args> (one pass per counter group; "
               "per-launch means over the %d target launches between the marker launches)" % (tag, n_launch),
    "counters_per_launch": out,
    "fetch_size_calibration": {"known_bytes": known, "FETCH_SIZE_KB": cal, "factor": factor,
                               "pattern": "distance_kernel streaming 3e6 x 384-B rows once with the search kernel's row-group loads (4 lanes x 16 B)"},
    "hbm_read_bytes_per_launch": out["FETCH_SIZE"] * 1024 * round(factor),
    "hbm_write_bytes_per_launch": out["WRITE_SIZE"] * 1024,
    "l2_hit_rate": out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]),
    "fetch_requests_per_launch": out["FETCH_SIZE"] * 1024 / 64,
})
s["traffic_bytes_per_launch"] = s["hbm_read_bytes_per_launch"] + s["hbm_write_bytes_per_launch"]
s["traffic_over_designed"] = s["traffic_bytes_per_launch"] / s["designed_bytes_per_launch"]
if "SQ_WAVE_CYCLES" in out:
    s["sq"] = {"wait_any_frac": out["SQ_WAIT_ANY"] / out["SQ_WAVE_CYCLES"], "active_inst_frac": out["SQ_ACTIVE_INST_ANY"] / out["SQ_WAVE_CYCLES"],
               "valu_per_query": out["SQ_INSTS_VALU"] / out["SQ_WAVES"], "salu_per_query": out["SQ_INSTS_SALU"] / out["SQ_WAVES"], "lds_per_query": out["SQ_INSTS_LDS"] / out["SQ_WAVES"],
               "wave_quad_cycles_per_query": out["SQ_WAVE_CYCLES"] / out["SQ_WAVES"]}
for f in glob.glob(f"{base}/trace/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "search_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if int(r.get("Grid_Size", r.get("Grid_Size_X"))) == 64 * 64]
    sel = rows[marks[0] + 1 : marks[1]][:n_launch]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sel]
    s["kernel_trace"] = {"launches": len(dur), "avg_ms": statistics.mean(dur) / 1e6, "min_ms": min(dur) / 1e6, "max_ms": max(dur) / 1e6,
                         "vgpr": sel[0].get("VGPR_Count"), "sgpr": sel[0].get("SGPR_Count"), "lds": sel[0].get("LDS_Block_Size"),
                         "note": "rocprofv3 --kernel-trace --stats of the same command, the target launches only"}
    try:  # the HIP-event time of the SAME traced process (target_traced.json) next to the trace's own durations, and the un-profiled process's
        traced = json.loads(open(f"{base}/target_traced.json").read().strip().splitlines()[-1])
        s["kernel_ms"] = {"hip_events_unprofiled_process": target["kernel_ms_unprofiled_or_profiled"], "hip_events_in_the_traced_process": traced["kernel_ms_unprofiled_or_profiled"],
                          "kernel_trace_avg_same_traced_process": s["kernel_trace"]["avg_ms"],
                          "note": "all three after 12 warm-up launches; the traced figures carry the tracer's per-dispatch overhead, the bench line uses HIP events of its own timed region"}
    except Exception as e:  # noqa: BLE001
        s["kernel_ms"] = {"error": str(e)}
    ms = s["kernel_trace"]["avg_ms"]
    s["traffic_rate_GBs"] = s["traffic_bytes_per_launch"] / ms / 1e6
    s["frac_traffic_of_8TBs"] = s["traffic_rate_GBs"] / 8000
    s["designed_rate_GBs"] = s["designed_bytes_per_launch"] / ms / 1e6
    s["fetch_request_rate_G_per_s"] = s["fetch_requests_per_launch"] / ms / 1e6
os.makedirs("profiles", exist_ok=True)
json.dump(s, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
print(json.dumps({k: s[k] for k in s if k not in ("counters_per_launch", "command")}, indent=1))
for f in glob.glob(f"{base}/trace/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python tools/pmc_target.py ... ({tag}); top 12 kernels\n")
        w = csv.writer(o)
        w.writerow(rows[0].keys())
        for r in rows[:12]:
            w.writerow(r.values())
