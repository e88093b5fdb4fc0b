#!/usr/bin/env python3
"""Do search launches on different streams share the GPU?  (round 6, host-pointer path: DESIGN.md section 5.2)  Config 2's index and batch, device-resident
queries: one launch of 10 000; the same batch as 4 096 + 5 904 and as 4 x 2 500 on ONE stream and on SEPARATE streams.  Wall clock around a
device synchronisation, median of 15.   python tools/concurrency_probe.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from bench import build_or_load_graph, deep_like_device

dev = torch.device("cuda", 0)
n, d, nq, k, R = 1_000_000, 96, 10_000, 10, 64
x = deep_like_device(n, d, 24, 1, dev)
q = deep_like_device(nq, d, 24, 2, dev)
graph, entry, _ = build_or_load_graph(svs, x, n, d, R, 0, "")
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), n, d, np.float32)
idx.build_bound_codes(2)
sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(128), 1)
ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dd = torch.empty((nq, k), device=dev)
streams = [torch.cuda.Stream() for _ in range(4)]


def run(bounds, separate):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(len(bounds) - 1):
        lo, hi = bounds[i], bounds[i + 1]
        s = streams[i] if separate else streams[0]
        idx.search_device(q[lo:hi], k, sp, d_ids64=ids[lo:hi], d_dists=dd[lo:hi], stream=s.cuda_stream)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


for name, bounds in (("1 x 10000", [0, nq]), ("4096 + 5904", [0, 4096, nq]), ("4 x 2500", [0, 2500, 5000, 7500, nq]), ("1024 + 8976", [0, 1024, nq])):
    for separate in (False, True):
        if len(bounds) == 2 and separate:
            continue
        for _ in range(25): run(bounds, separate)
        ms = sorted(run(bounds, separate) for _ in range(15))
        print(json.dumps({"launches": name, "streams": "separate" if separate else "one", "ms_median": round(ms[7], 4), "ms_min": round(ms[0], 4)}), flush=True)
