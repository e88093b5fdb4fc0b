#!/usr/bin/env python3
"""Kernel time of the config-4 shape (N x 768 scalar-quantized int8 rows, float32 queries, cosine, batch 4 096, graph R = 128 L = 200 prune_to 124),
for tools/ab_libs.py:
   python tools/time_c4.py [--n 2000000] [--windows 285] [--phases]   -> one JSON line per window with "ms" (median of 5 groups of 3 launches), id checksum,
   per-query statistics; with --phases (a -DSVSX_DIAG=2 library) the shader-clock cycles per expansion spent in
   next | adjacency-block wait | bound + filter + compaction | row wait + distance | insert  instead of the statistics."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000); ap.add_argument("--nq", type=int, default=4096); ap.add_argument("--windows", default="285")
ap.add_argument("--degree", type=int, default=128); ap.add_argument("--phases", action="store_true"); ap.add_argument("--records", type=int, default=2)
ap.add_argument("--graph-cache", default="/tmp/time_c4_graph.pt")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 768, 10, a.degree
fn = svs.DistanceCosineSimilarity()
mn = mx = None
for lo in range(0, a.n, 1 << 20):
    xf = deep_like_device(min(1 << 20, a.n - lo), d, 64, seed=1, device=dev, normalize=False, row0=lo)
    mn = float(xf.min()) if mn is None else min(mn, float(xf.min())); mx = float(xf.max()) if mx is None else max(mx, float(xf.max()))
scale = (mx - mn) / 255.0; bias = mn + 128.0 * scale
x = torch.empty((a.n, d), dtype=torch.int8, device=dev)
for lo in range(0, a.n, 1 << 20):
    xf = deep_like_device(min(1 << 20, a.n - lo), d, 64, seed=1, device=dev, normalize=False, row0=lo)
    x[lo:lo + xf.shape[0]] = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8)
del xf
q = deep_like_device(a.nq, d, 64, seed=2, device=dev, normalize=False)
sq = (scale, bias)
cache = f"{a.graph_cache}.{a.n}.{R}"
if os.path.exists(cache):
    blob = torch.load(cache, map_location=dev); graph, entry = blob["graph"], blob["entry"]
else:
    graph, entry = build_vamana_device(x, a.n, d, fn, VamanaBuildParameters(alpha=0.95, graph_max_degree=R, window_size=200, prune_to=R - 4), data_dtype=np.int8, sq=sq)
    torch.save({"graph": graph, "entry": entry}, cache)
idx = svs.VamanaIndex.from_device(graph, x, fn, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.int8, sq=sq)
idx.build_bound_codes(a.records)
ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dists = torch.empty((a.nq, k), device=dev); stats = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
for w in [int(v) for v in a.windows.split(",")]:
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
    for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats)
    torch.cuda.synchronize()
    samples = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        e1.record(); torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / 3)
    st = stats.cpu().numpy().view(np.uint32).astype(np.float64)
    out = {"n": a.n, "nq": a.nq, "window": w, "ms": float(np.median(samples)), "min_ms": float(min(samples)), "checksum": int(ids.sum().item()), "kernel": idx.last_kernel_info()}
    if a.phases:
        per = st[:, 1:6].sum(0) / st[:, 0].sum()
        out["cycles_per_expansion"] = {n_: round(float(v), 1) for n_, v in zip(["next", "block_wait", "bound_compact", "rows_distance", "insert"], per)}
        out["sum"] = round(float(per.sum()), 1)
    else:
        out["per_query"] = {"expansions": st[:, 0].mean(), "slots": st[:, 1].mean(), "evals": st[:, 2].mean(), "inserts": st[:, 3].mean(), "bound_rejects": st[:, 5].mean()}
    print(json.dumps(out), flush=True)
