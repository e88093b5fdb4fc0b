"""PCIe-inclusive rate of the host-pointer entry point svsx_search (DESIGN.md 'Measurement'); GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import device_graph_to_svs
from bench import deep_like_device
dev = torch.device("cuda", 0)
b = torch.load("/tmp/g1m.pt.r0", map_location=dev)
x = deep_like_device(1_000_000, 96, 24, 1, dev); q = deep_like_device(10_000, 96, 24, 2, dev)
t0 = time.perf_counter()
idx = svs.VamanaIndex(device_graph_to_svs(b["graph"], 64), x.cpu().numpy(), svs.DistanceL2(), [b["entry"]])
t_create = time.perf_counter() - t0
qh = q.cpu().numpy()
sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(128), True)
if len(sys.argv) > 1 and sys.argv[1] == "codes":
    idx.build_bound_codes()
idx.search(qh, 10, sp)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); idx.search(qh, 10, sp); ts.append(time.perf_counter() - t0)
print(f"svsx_index_create (H2D + re-layout of 1M x 96 f32 + graph): {t_create:.2f} s; svsx_search host pointers, 10k queries window 128: best {min(ts)*1e3:.2f} ms -> {10000/min(ts):.0f} QPS (median {sorted(ts)[2]*1e3:.2f} ms)")
