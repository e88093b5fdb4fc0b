#!/usr/bin/env python3
"""gpurun_out/<round>flat/ (tools/flat_pmc.sh) -> profiles/<round>_flat_pmc.json, <round>_flat_kernel_stats.csv, <round>_flat_traced.json.
Per kernel of the library: every counter summed over the dispatches of the process (1 warm + 3 timed searches), calls / total / max duration of
the un-instrumented trace, and the derived figures DESIGN.md 4.5 quotes.  usage: tools/summarize_flat_pmc.py gpurun_out/r05flat profiles/r05"""
import collections, csv, glob, json, os, shutil, sys

src, dst = sys.argv[1], sys.argv[2]
kern = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob(os.path.join(src, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if "svsx" in r["Kernel_Name"]:
            kern[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]] += float(r["Counter_Value"])
stats = next(iter(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)))
for r in csv.DictReader(open(stats)):
    name = r["Name"].split("(")[0].replace("void ", "")
    if name in kern:
        kern[name]["calls"] = int(r["Calls"])
        kern[name]["total_ms"] = int(r["TotalDurationNs"]) / 1e6
        kern[name]["max_ms"] = int(r["MaxNs"]) / 1e6
searches = 4
for name, c in kern.items():
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        c["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if c.get("SQ_WAVE_CYCLES", 0) > 0:
        c["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
        c["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_INSTS_MFMA", 0) > 0:
        c["mfma_busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]
        c["insts_per_mfma"] = {k: c.get("SQ_INSTS_" + k, 0) / c["SQ_INSTS_MFMA"] for k in ("VALU", "SALU", "LDS")}
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy cycles over the 1 024 SIMDs
        if c.get("GRBM_GUI_ACTIVE", 0) > 0:
            c["mfma_pipe_busy_frac_of_kernel_cycles"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["GRBM_GUI_ACTIVE"] / 8.0)
            c["shader_clock_GHz_over_the_kernels"] = (c["GRBM_GUI_ACTIVE"] / 8.0) / (c["total_ms"] * 1e-3) / 1e9 if c.get("total_ms") else None
out = {"what": "svsx_flat_search_device, 10 000 queries x 1 000 000 x 96 float32, L2, k = 10 (tools/flat_check.py: 1 warm + 3 timed searches in the process; "
               "counters summed over the 4 searches, one --pmc pass per group; tools/flat_pmc.sh)",
       "traced_process": json.loads(open(os.path.join(src, "traced.json")).read().strip().splitlines()[-1]),
       "searches_in_process": searches, "kernels": {k: dict(v) for k, v in kern.items()}}
json.dump(out, open(dst + "_flat_pmc.json", "w"), indent=1)
shutil.copy(stats, dst + "_flat_kernel_stats.csv")
shutil.copy(os.path.join(src, "traced.json"), dst + "_flat_traced.json")
for k, v in kern.items():
    if "thresh" in k:
        print(k, {x: (round(y, 3) if isinstance(y, float) else y) for x, y in v.items() if x in ("calls", "total_ms", "max_ms", "l2_hit_rate", "wait_any_frac", "active_inst_frac", "mfma_pipe_busy_frac_of_kernel_cycles", "shader_clock_GHz_over_the_kernels", "insts_per_mfma")})
