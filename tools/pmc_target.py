#!/usr/bin/env python3
"""One PMC target: build (or load) one shard of the bench workload and launch the search kernel `--launches` times at ONE
configuration, nothing else -- so every search_kernel dispatch a rocprofv3 pass sees belongs to the workload named by
`workload_key` (the key bench.py looks profiles/*_pmc.json up by).  Prints one JSON line with the key, the kernel source
hash and the designed lines of a launch.
   python tools/pmc_target.py --n 1000000 --window 128 --codes 2 [--matched] --graph-cache /tmp/g"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from bench import LINE, brute_force_topk, build_or_load_graph, deep_like_device, designed_lines, kernel_src_sha, recall_at

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2"); ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", type=int, default=10_000)
ap.add_argument("--window", type=int, default=128); ap.add_argument("--codes", type=int, default=2); ap.add_argument("--visited", type=int, default=1)
ap.add_argument("--degree", type=int, default=64); ap.add_argument("--launches", type=int, default=5); ap.add_argument("--graph-cache", default="")
ap.add_argument("--matched", action="store_true", help="use the smallest window with recall@10 >= 0.9 instead of --window")
ap.add_argument("--build-window", type=int, default=0, help="build window L (0 = 2 R; the bench's config-4 graph uses 200)"); ap.add_argument("--sid", type=int, default=0); ap.add_argument("--warmup", type=int, default=12, help="un-marked launches before the first marker: the traced launches then run in the warmed state the bench times")
a = ap.parse_args()
dev = torch.device("cuda", 0)
k, R = 10, a.degree
cache = f"{a.graph_cache}.{a.config}.s{a.sid}.n{a.n}.R{a.degree}.L{a.build_window}" if a.graph_cache else ""
if a.config == "c3":    # BASELINE config 3: sift-1M-like, float16 rows, float32 queries, inner product
    d, fn, ddt, sq = 128, svs.DistanceIP(), np.float16, None
    x = deep_like_device(a.n, d, 32, seed=31, device=dev, normalize=False).half()
    q = deep_like_device(a.nq, d, 32, seed=32, device=dev, normalize=False)
    graph, entry, _ = build_or_load_graph(svs, x, a.n, d, R, 0, cache, distance=fn, alpha=0.95, data_dtype=ddt)
elif a.config in ("c4", "c4b"):  # BASELINE config 4's row type: scalar-quantized int8 codes, float32 queries, cosine (c4b: the same rows as raw int8, int8 queries)
    d, fn, ddt = 768, svs.DistanceCosineSimilarity(), np.int8
    q = deep_like_device(a.nq, d, 64, seed=2, device=dev, normalize=False)
    mn = mx = None  # two passes over the generator, 1 M rows at a time: the float32 rows of 10 M x 768 never exist at once (bench.config4_point)
    for lo in range(0, a.n, 1 << 20):
        xf = deep_like_device(min(1 << 20, a.n - lo), d, 64, seed=1, device=dev, normalize=False, row0=lo)
        mn = float(xf.min()) if mn is None else min(mn, float(xf.min())); mx = float(xf.max()) if mx is None else max(mx, float(xf.max()))
    scale = (mx - mn) / 255.0; bias = mn + 128.0 * scale
    x = torch.empty((a.n, d), dtype=torch.int8, device=dev)
    for lo in range(0, a.n, 1 << 20):
        xf = deep_like_device(min(1 << 20, a.n - lo), d, 64, seed=1, device=dev, normalize=False, row0=lo)
        x[lo:lo + xf.shape[0]] = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8)
    del xf
    sq = (scale, bias)
    graph, entry, _ = build_or_load_graph(svs, x, a.n, d, R, 0, cache.replace(".c4b.", ".c4."), distance=fn, alpha=0.95, data_dtype=ddt, sq=sq, build_window=a.build_window)
    if a.config == "c4b":  # SURVEY 8d C4(b), as bench.config4b_point runs it: config 4's graph, the codes read as plain int8 vectors, int8 queries
        q = torch.clamp(torch.round((q - bias) / scale), -128, 127).to(torch.int8)
        sq = None
else:
    d, fn, ddt, sq = 96, svs.DistanceL2(), np.float32, None
    x = deep_like_device(a.n, d, 24, seed=1 + 1000 * a.sid, device=dev)
    q = deep_like_device(a.nq, d, 24, seed=2, device=dev)
    graph, entry, _ = build_or_load_graph(svs, x, a.n, d, R, 0, cache)
idx = svs.VamanaIndex.from_device(graph, x, fn, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, ddt, sq=sq)
if a.codes:
    idx.build_bound_codes(a.codes)
ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dist = torch.empty((a.nq, k), device=dev)
window = a.window
if a.matched:
    gt = idx.flat_search_device(q, k)[0] if a.config != "c2" and a.config != "c5" else brute_force_topk(x, q, k)
    lo, hi = k, a.window
    while lo < hi:
        mid = (lo + hi) // 2
        idx.search_device(q, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(mid), a.visited), d_ids64=ids, d_dists=dist); torch.cuda.synchronize()
        if recall_at(gt, ids, k) >= 0.9: hi = mid
        else: lo = mid + 1
    window = lo
sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(window), a.visited)
stats = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
# marker launches bracket the target launches: window 1 searches (tiny LDS block) are easy to tell apart in the dispatch list
mark = svs.VamanaSearchParameters(svs.SearchBufferConfig(1), 0)
for _ in range(a.warmup): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
torch.cuda.synchronize()
idx.search_device(q[:64], 1, mark, d_ids64=ids[:64, :1], d_dists=dist[:64, :1]); torch.cuda.synchronize()
for i in range(a.launches):
    idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist, d_stats=stats if i == 0 else None)
torch.cuda.synchronize()
idx.search_device(q[:64], 1, mark, d_ids64=ids[:64, :1], d_dists=dist[:64, :1]); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.launches): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
e1.record(); torch.cuda.synchronize()
rb = {"codes=2>": 32, "codes=3>": 64, "codes=4>": 128, "codes=5>": 28}.get(idx.last_kernel_info()[-8:], 32)
lines = designed_lines(stats.to(torch.int64), (R + 15) // 16 * 16, x.shape[1] * x.element_size(), min(a.codes, 2), d, k, rb)
print(json.dumps({"workload_key": f"{a.config}:n={a.n}:nq={a.nq}:window={window}:R={R}:visited={a.visited}:codes={a.codes}", "kernel_src_sha": kernel_src_sha(),
                  "window": window, "launches": a.launches, "designed_lines_per_launch": lines, "designed_bytes_per_launch": lines["total"] * LINE,
                  "kernel_ms_unprofiled_or_profiled": e0.elapsed_time(e1) / a.launches, "kernel_ms_hip_events_this_process": e0.elapsed_time(e1) / a.launches, "kernel": idx.last_kernel_info(),
                  "stats_per_query": {"expansions": float(stats[:, 0].sum()) / a.nq, "evals": float(stats[:, 2].sum()) / a.nq, "bound_rejects": float(stats[:, 5].sum()) / a.nq}}))
