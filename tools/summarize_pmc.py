#!/usr/bin/env python3
"""gpurun_out/<round>/pmc_* (tools/collect_pmc.sh) -> profiles/<round>_pmc.json + <round>_bench_kernel_stats.csv"""
import collections, csv, glob, json, os, statistics, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
visited = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bound_codes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
base = f"gpurun_out/{rnd}"


def per_kernel(dirn, kname):
    """counter sums per dispatch; for the search kernel only the window-128 launches (largest LDS block: the
    candidate buffer is sized by the window) -- bench.py also launches smaller windows while calibrating."""
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    lds = {}
    for f in glob.glob(f"{base}/{dirn}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"]:
                acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
                lds[r["Dispatch_Id"]] = int(r["LDS_Block_Size"])
    if kname == "search_kernel":
        # bench.py launches the window-128 searches FIRST (1 stats pass + warmup + steps); the calibration and
        # matched-recall launches at smaller windows follow.  The PMC passes run with --steps 3 --warmup 1 -> first 5.
        first = sorted(acc, key=lambda k: int(k))[:5]
        acc = {k: acc[k] for k in first}
    return acc


out = {}
for d in sorted(glob.glob(f"{base}/pmc_*")):
    dn = os.path.basename(d)
    if dn == "pmc_calib":
        continue
    a = per_kernel(dn, "search_kernel")
    for nm in sorted({k for v in a.values() for k in v}):
        out[nm] = statistics.mean(v[nm] for v in a.values())
cal = list(per_kernel("pmc_calib", "distance_kernel").values())[0]["FETCH_SIZE"]
known = 3_000_000 * 96 * 4
factor = known / (cal * 1024)
s = {
    "command": "rocprofv3 --pmc <COUNTERS> --kernel-trace --output-format csv -- python bench.py --graph-cache /tmp/g1m.pt --no-cpu-baseline --steps 3 --warmup 1 (one pass per counter group; means over the search_kernel launches at window 128 of each pass)",
    "workload": dict({"n": 1000000, "d": 96, "nq": 10000, "window": 128, "degree": 64, "visited": visited}, **({"bound_codes": bound_codes} if bound_codes else {})),
    "kernel": "svsx::search_kernel<0,0,0,96,false,%s>" % ("true" if bound_codes else "false"),
    "counters_per_launch": out,
    "fetch_size_calibration": {"known_bytes": known, "FETCH_SIZE_KB": cal, "factor": factor,
                               "pattern": "distance_kernel streaming 3e6 x 384-B rows once with the search kernel's row-group loads (4 lanes x 16 B)"},
    "hbm_read_bytes_per_launch": out["FETCH_SIZE"] * 1024 * round(factor),
    "hbm_write_bytes_per_launch": out["WRITE_SIZE"] * 1024,
    "l2_hit_rate": out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]),
}
s["traffic_bytes_per_launch"] = s["hbm_read_bytes_per_launch"] + s["hbm_write_bytes_per_launch"]
json.dump(s, open(f"profiles/{rnd}_pmc.json", "w"), indent=1)
print(json.dumps(s, indent=1))
for f in glob.glob(f"{base}/trace/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "search_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[:24]]  # 1 stats + 3 warmup + 20 timed, all window 128
    s["kernel_trace_window128"] = {"launches": len(dur), "avg_ms": statistics.mean(dur) / 1e6, "min_ms": min(dur) / 1e6, "max_ms": max(dur) / 1e6,
                                   "note": "rocprofv3 --kernel-trace --stats of the default bench command; the first 24 search_kernel launches = the window-128 ones (stats pass + 3 warmup + 20 timed); later launches are the calibration / matched-recall windows"}
    json.dump(s, open(f"profiles/{rnd}_pmc.json", "w"), indent=1)
    print(s["kernel_trace_window128"])
for f in glob.glob(f"{base}/trace/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    with open(f"profiles/{rnd}_bench_kernel_stats.csv", "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --graph-cache /tmp/g1m.pt --no-cpu-baseline ({rnd}); top 15 kernels\n")
        w = csv.writer(o)
        w.writerow(rows[0].keys())
        for r in rows[:15]:
            w.writerow(r.values())
