#!/usr/bin/env python3
"""Kernel time of the config-2 workload (one query per wavefront unless --policy says otherwise), for tools/ab_libs.py:
   python tools/time_c2.py [--nq 10000] [--windows 128,23] [--policy 1]   -> one JSON line per (nq, window) with "ms" (median of 7 groups of 5 launches)"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="10000"); ap.add_argument("--windows", default="128,23"); ap.add_argument("--policy", type=int, default=1)
ap.add_argument("--graph-cache", default="/tmp/time_c2_graph.pt")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
if os.path.exists(a.graph_cache):
    blob = torch.load(a.graph_cache, map_location=dev); graph, entry = blob["graph"], blob["entry"]
else:
    graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
    torch.save({"graph": graph, "entry": entry}, a.graph_cache)
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
if hasattr(idx, "set_kernel_policy"):
    idx.set_kernel_policy(a.policy)
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        for _ in range(5): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        samples = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
            e1.record(); torch.cuda.synchronize()
            samples.append(e0.elapsed_time(e1) / 5)
        print(json.dumps({"nq": nq, "window": w, "ms": float(np.median(samples)), "min_ms": float(min(samples)), "checksum": int(ids.sum().item())}), flush=True)
