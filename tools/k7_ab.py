#!/usr/bin/env python3
"""QPS of the static-dimension kernels of K7 (the reference's supported_dim_list, distance_core.h:31-32) for d = 64, 100, 160, 200, 512:
   python tools/k7_ab.py [--tree <dir holding another scalablevectorsearch_amd/ (e.g. the round-2 sources + library)>]
One JSON line per dimension: which instantiation served the call and how fast (200 000 x d float32 L2, R = 64, 10 000 queries, window 128,
graph-inline records).  Run once on this tree and once with --tree on the round-2 tree for before / after."""
import argparse, json, os, sys
ap = argparse.ArgumentParser()
ap.add_argument("--tree", default=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); ap.add_argument("--dims", default="64,100,160,200,512")
ap.add_argument("--n", type=int, default=200_000); ap.add_argument("--modes", default="2,0")
ap.add_argument("--graph-cache", default="", help="directory: graphs are saved there / loaded from there, so that two trees search the SAME graph")
a = ap.parse_args()
sys.path.insert(0, a.tree)
import numpy as np
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device

assert os.path.dirname(os.path.dirname(os.path.abspath(svs.__file__))) == os.path.abspath(a.tree), svs.__file__
dev = torch.device("cuda", 0)
nq, k, R = 10_000, 10, 64
for d in [int(v) for v in a.dims.split(",")]:
    g = torch.Generator(device=dev); g.manual_seed(d)
    proj = torch.randn((24, d), device=dev, generator=g)
    x = torch.randn((a.n, 24), device=dev, generator=g) @ proj + 0.1 * torch.randn((a.n, d), device=dev, generator=g)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    q = torch.randn((nq, 24), device=dev, generator=g) @ proj + 0.1 * torch.randn((nq, d), device=dev, generator=g)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    dpad = (d + 15) // 16 * 16
    xp = torch.zeros((a.n, dpad), device=dev); xp[:, :d] = x           # device layout: rows padded to 16 elements
    cache = os.path.join(a.graph_cache, f"k7_d{d}_n{a.n}.pt") if a.graph_cache else ""
    if cache and os.path.exists(cache):
        blob = torch.load(cache, map_location=dev); graph, entry = blob["graph"], blob["entry"]
    else:
        graph, entry = build_vamana_device(xp, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
        if cache:
            os.makedirs(a.graph_cache, exist_ok=True); torch.save({"graph": graph, "entry": entry}, cache)
    idx = svs.VamanaIndex.from_device(graph, xp, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dist = torch.empty((nq, k), device=dev)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(128), 1)
    for mode in [int(v) for v in a.modes.split(",")]:
        try:
            idx.build_bound_codes(mode)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"d": d, "mode": mode, "error": str(e)[:100]})); continue
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(json.dumps({"d": d, "mode": mode, "ms": round(ms, 4), "qps": round(nq / ms * 1e3), "kernel": idx.last_kernel_info(), "ids_checksum": int(ids.sum())}), flush=True)
