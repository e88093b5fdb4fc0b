#!/usr/bin/env python3
"""In-process A/B of the look-ahead policies of a variant library (staging/r3/01_tail_lookahead.patch):
   SVSX_LIB=variants/libsvsx.so python tools/ab_lookahead.py [--n 1000000] [--nq 10000] [--windows 128,23]
One JSON line per (window, policy): kernel ms, QPS, ids identical to policy off."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="10000"); ap.add_argument("--windows", default="128,23")
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--policies", default="off,tail,all,off")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        base = None
        for policy in a.policies.split(","):
            os.environ["SVSX_LOOKAHEAD"] = policy
            for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.steps
            if base is None: base = ids.clone()
            print(json.dumps({"nq": nq, "window": w, "policy": policy, "ms": ms, "qps": nq / ms * 1e3, "ids_identical": bool((base == ids).all()), "kernel": idx.last_kernel_info()}), flush=True)
