"""How many neighbour-slot vector fetches are re-fetches?  (analysis for the visited-filter design; GPU box)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs
from bench import deep_like_device
from oracle.oracle import L2, Oracle
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = deep_like_device(n, 96, 24, 1, dev); q = deep_like_device(256, 96, 24, 2, dev)
if True:
    graph, entry = build_vamana_device(x, n, 96, svs.DistanceL2(), VamanaBuildParameters(graph_max_degree=64, window_size=128, prune_to=60), device=0)
o = Oracle()
xh, qh, gh, eh = x.cpu().numpy(), q.cpu().numpy(), device_graph_to_svs(graph, 64), np.array([entry], np.uint32)
for w in (24, 128):
    _, _, s0 = o.search(L2, xh, gh, eh, qh, 10, w, stats=True)
    _, _, s1 = o.search(L2, xh, gh, eh, qh, 10, w, visited_set=True, stats=True)
    print(f"window {w}: candidates whose first-64-dims partial L2 already exceeds the worst (visited set on): {s1[:,8].sum()/s1[:,2].sum():.2%} of evaluated, (visited set off): {s0[:,8].sum()/s0[:,2].sum():.2%}")
    small = ", ".join("%.1f%%" % (100.0 * s0[:, 4 + i].sum() / s0[:, 1].sum()) for i in range(3))
    print(f"window {w}: slots/query {s0[:,1].mean():.0f}  in-buffer {s0[:,3].sum()/s0[:,1].sum():.2%}  evals with 64K visited filter {s1[:,2].sum()/s1[:,1].sum():.2%} of slots  small exact filters 512/1k/2k hit {small}")
