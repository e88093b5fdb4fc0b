#!/usr/bin/env python3
"""Similarity metrics on data with a common offset (round 6: the inner-product / cosine records are centred): 1 M rows, the headline's graph parameters,
window 128.  Sets: sift-like non-negative float16 rows under inner product (SURVEY 8d C3's optional variant), float32 unit rows around an offset under
cosine, uint8 rows under inner product.  Run once per library (SVSX_LIB=...): plain kernel, inline records, rejects, ids equal.
   python tools/offset_ab.py [--n 1000000]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from bench import build_or_load_graph, deep_like_device, recall_at

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", type=int, default=10_000); ap.add_argument("--window", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)
k, R = 10, 64


def sets():
    x = deep_like_device(a.n, 128, 32, 31, dev, normalize=False).abs().half()
    q = deep_like_device(a.nq, 128, 32, 32, dev, normalize=False).abs()
    yield "sift-like non-negative float16 x float32, inner product, d=128", x, q, svs.DistanceIP(), np.float16, 0.95
    x = deep_like_device(a.n, 96, 24, 1, dev) + 0.15
    q = deep_like_device(a.nq, 96, 24, 2, dev) + 0.15
    yield "unit rows + 0.15 offset, float32, cosine, d=96", x.contiguous(), q.contiguous(), svs.DistanceCosineSimilarity(), np.float32, 0.95
    xf = deep_like_device(a.n, 128, 32, 31, dev, normalize=False)
    qf = deep_like_device(a.nq, 128, 32, 32, dev, normalize=False)
    g = 100.0 / float(xf.abs().max())
    yield "uint8 rows x uint8 queries (values around 128), inner product, d=128", torch.clamp(torch.round(xf * g + 128), 0, 255).to(torch.uint8), \
        torch.clamp(torch.round(qf * g + 128), 0, 255).to(torch.uint8), svs.DistanceIP(), np.uint8, 0.95


for name, x, q, fn, ddt, alpha in sets():
    d = x.shape[1]
    graph, entry, tb = build_or_load_graph(svs, x, a.n, d, R, 0, "", distance=fn, alpha=alpha, data_dtype=ddt)
    idx = svs.VamanaIndex.from_device(graph, x, fn, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, ddt)
    gt, _ = idx.flat_search_device(q, k)
    ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dd = torch.empty((a.nq, k), device=dev); st = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(a.window), 1)
    base = None
    for mode in (0, 2):
        idx.build_bound_codes(mode)
        idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd, d_stats=st)
        for _ in range(15): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for e0, e1 in ev:
            e0.record(); idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd); e1.record()
        torch.cuda.synchronize()
        ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
        if mode == 0: base = ids.clone()
        s_ = st.to(torch.int64)
        print(json.dumps({"lib": os.environ.get("SVSX_LIB", "in-tree"), "set": name, "mode": mode, "ms": round(ms, 4), "qps_M": round(a.nq / ms / 1e3, 3), "recall": round(recall_at(gt, ids, k), 4),
                          "evals": float(s_[:, 2].sum()) / a.nq, "rejects": float(s_[:, 5].sum()) / a.nq, "ids_equal_plain": bool((ids == base).all()), "kernel": idx.last_kernel_info()}), flush=True)
    del idx, x, graph
    torch.cuda.empty_cache()
