#!/usr/bin/env python3
"""Record size against spectrum (round 6): 1 M x 96 unit rows of G(latent, sigma) / isotropic rows, the headline's graph parameters, window 128 (and the
window given): plain kernel, the size the rule picks, and every forced size.  python tools/spectrum_ab.py [--sets 24:0.1,48:0.2,0:0] [--windows 128]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from bench import build_or_load_graph, deep_like_device, recall_at

ap = argparse.ArgumentParser()
ap.add_argument("--sets", default="24:0.1,48:0.2,0:0"); ap.add_argument("--windows", default="128"); ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", type=int, default=10_000)
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
for spec in a.sets.split(","):
    latent, sigma = int(spec.split(":")[0]), float(spec.split(":")[1])
    x = deep_like_device(a.n, d, latent, 1, dev, sigma=sigma)
    q = deep_like_device(a.nq, d, latent, 2, dev, sigma=sigma)
    graph, entry, tb = build_or_load_graph(svs, x, a.n, d, R, 0, "")
    idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
    gt, _ = idx.flat_search_device(q, k)
    ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dd = torch.empty((a.nq, k), device=dev); st = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        base = None
        for mode in (0, 2, 28, 32, 64, 128):
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
            print(json.dumps({"set": spec, "window": w, "mode": mode, "record": int(idx.memory()["record_size"]), "ms": round(ms, 4), "qps_M": round(a.nq / ms / 1e3, 3),
                              "recall": round(recall_at(gt, ids, k), 4), "evals": float(s_[:, 2].sum()) / a.nq, "rejects": float(s_[:, 5].sum()) / a.nq,
                              "expansions": float(s_[:, 0].sum()) / a.nq, "ids_equal_plain": bool((ids == base).all()), "kernel": idx.last_kernel_info()}), flush=True)
    del idx, x, graph
    torch.cuda.empty_cache()
