#!/usr/bin/env python3
"""Phase profile of an expansion from a DIAG=2 variant library (tools/build_variant.sh diag2 -DSVSX_DIAG=2): shader-clock cycles per expansion spent
in  next | adjacency-block wait | bound + filter + compaction | row wait + distance | insert,  lone waves and a full machine:
   SVSX_LIB=variants/diag2/libsvsx.so python tools/diag_phases.py [--nq 256,10000] [--windows 128,23]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="256,10000"); ap.add_argument("--windows", default="128,23")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev); stats = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats); e1.record(); torch.cuda.synchronize()
        st = stats.cpu().numpy().view(np.uint32).astype(np.float64)
        per = st[:, 1:6].sum(0) / st[:, 0].sum()
        names = ["next", "block_wait", "bound_compact", "rows_distance", "insert"]
        print(json.dumps({"nq": nq, "window": w, "ms": e0.elapsed_time(e1), "cycles_per_expansion": {n: round(float(v), 1) for n, v in zip(names, per)}, "sum": round(float(per.sum()), 1)}), flush=True)
