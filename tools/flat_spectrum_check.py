#!/usr/bin/env python3
"""How the two bound-code schemes do on a FLAT spectrum (isotropic Gaussian data, nothing for PCA to find): 200 k x 96."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import counter_normal
dev = torch.device("cuda", 0)
n, d, nq, k = 200_000, 96, 4096, 10
x = counter_normal(n, d, 5, 0, dev); q = counter_normal(nq, d, 6, 0, dev)
g, e = build_vamana_device(x, n, d, svs.DistanceL2(), VamanaBuildParameters(graph_max_degree=64, window_size=128, prune_to=60))
idx = svs.VamanaIndex.from_device(g, x, svs.DistanceL2(), torch.tensor([e], dtype=torch.int32, device=dev), n, d, np.float32)
ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dist = torch.empty((nq, k), device=dev); st = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
base = None
for mode in (0, 1, 2):
    idx.build_bound_codes(mode)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(128), 1)
    idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist, d_stats=st); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
    e1.record(); torch.cuda.synchronize()
    if base is None: base = ids.clone()
    print(json.dumps({"mode": mode, "kernel": idx.last_kernel_info(), "ms": e0.elapsed_time(e1) / 10, "evals": float(st[:, 2].sum()) / nq, "rejects": float(st[:, 5].sum()) / nq, "ids_equal": bool((ids == base).all())}))
