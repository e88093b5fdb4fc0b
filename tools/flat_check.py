"""Flat search at scale: time per batch and how many queries the exactness proof sends to the exhaustive float32 pass.
usage: python tools/flat_check.py [n] [d] [nq] [k] [metrics, e.g. l2,ip,cos]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import scalablevectorsearch_amd as svs
from bench import deep_like_device

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 96
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
wanted = (sys.argv[5] if len(sys.argv) > 5 else "l2,ip,cos").split(",")
dev = torch.device("cuda", 0)
x = deep_like_device(n, d, 64, 1, dev)
q = deep_like_device(nq, d, 64, 2, dev)
graph = torch.full((n, 16), -1, dtype=torch.int32, device=dev)  # flat search never reads it
entry = torch.zeros(1, dtype=torch.int32, device=dev)
for metric, fn in (("l2", svs.DistanceL2()), ("ip", svs.DistanceIP()), ("cos", svs.DistanceCosineSimilarity())):
    if metric not in wanted:
        continue
    idx = svs.VamanaIndex.from_device(graph, x, fn, entry, n, d, np.float32, device=0)
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    idx.flat_search_device(q, k, recomputed=cnt)
    torch.cuda.synchronize()
    first = int(cnt.item())
    t0 = time.perf_counter()
    for _ in range(3):
        ids, dists = idx.flat_search_device(q, k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"metric": metric, "n": n, "d": d, "nq": nq, "k": k, "ms_per_batch": round(ms, 2), "recomputed": first,
                      "tflops_f16": round(2.0 * n * d * nq / ms / 1e9, 1), "ids_sum": int(ids.to(torch.int64).sum().item()),
                      "dist_sum": float(dists.double().sum().item()), "lib": os.environ.get("SVSX_LIB", "in-tree")}), flush=True)
    del idx
