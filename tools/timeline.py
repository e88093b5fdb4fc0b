#!/usr/bin/env python3
"""Per-query start / end times of one launch (TIMELINE variant library: stats words 4, 5 = wall_clock64 at wave start / end, 100 MHz):
   SVSX_LIB=variants/libsvsx.so python tools/timeline.py out.npz [--nq 10000] [--window 128]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("out"); ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="10000,20000"); ap.add_argument("--windows", default="128,23"); ap.add_argument("--visited", default="1")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
out = {}
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev); stats = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        for vis in [int(v) for v in a.visited.split(",")]:
            sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), vis)
            for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
            e1.record(); torch.cuda.synchronize()
            print(f"nq {nq} window {w} visited {vis}: {e0.elapsed_time(e1) / 10:.4f} ms", flush=True)
            out[f"nq{nq}_w{w}_v{vis}"] = stats.cpu().numpy()
np.savez_compressed(a.out, **out)
