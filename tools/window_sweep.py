#!/usr/bin/env python3
"""Recall / QPS versus search window for one of the side configs, one build:
   tools/window_sweep.py --n 2000000 --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64 --windows 64,128,256,512,1024
Prints one JSON line per (bound-code mode, window)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device, recall_at
from tools.bench_config import exact_topk

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2_000_000); ap.add_argument("--d", type=int, default=768)
ap.add_argument("--dtype", default="sq8", choices=["f32", "f16", "sq8"]); ap.add_argument("--metric", default="cos", choices=["l2", "ip", "cos"])
ap.add_argument("--nq", type=int, default=4096); ap.add_argument("--degree", type=int, default=64); ap.add_argument("--build-window", type=int, default=0)
ap.add_argument("--latent", type=int, default=64); ap.add_argument("--windows", default="64,128,256,512,1024"); ap.add_argument("--modes", default="0,2")
ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
fn = {"l2": svs.DistanceL2(), "ip": svs.DistanceIP(), "cos": svs.DistanceCosineSimilarity()}[a.metric]
xf = deep_like_device(a.n, a.d, a.latent, 1, dev, normalize=a.metric == "l2")
qf = deep_like_device(a.nq, a.d, a.latent, 2, dev, normalize=a.metric == "l2")
sq, np_dt = None, np.float32
if a.dtype == "f32": x = xf
elif a.dtype == "f16": x, np_dt = xf.half(), np.float16; xf = x.float()
else:
    mn, mx = float(xf.min()), float(xf.max()); scale = (mx - mn) / 255.0; bias = mn + 128.0 * scale
    x = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8); sq, np_dt = (scale, bias), np.int8
    xf = x.float() * scale + bias
alpha = 1.2 if a.metric == "l2" else 0.95
t0 = time.time()
L = a.build_window or 2 * a.degree
graph, entry = build_vamana_device(x, a.n, a.d, fn, VamanaBuildParameters(alpha=alpha, graph_max_degree=a.degree, window_size=L, prune_to=a.degree - 4), data_dtype=np_dt, sq=sq)
t_build = time.time() - t0
k = 10
gt = exact_topk(xf, qf, k, a.metric)
if a.dtype != "f32":
    del xf  # the float copy is only needed for the ground truth: give the memory back before the inline records are built
torch.cuda.empty_cache()
idx = svs.VamanaIndex.from_device(graph, x, fn, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, a.d, np_dt, sq=sq)
ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dists = torch.empty((a.nq, k), device=dev); stats = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
base = None
for mode in [int(m) for m in a.modes.split(",")]:
    idx.build_bound_codes(mode)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        idx.search_device(qf, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats); torch.cuda.synchronize()
        rec = recall_at(gt, ids, k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps): idx.search_device(qf, k, sp, d_ids64=ids, d_dists=dists)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        key = (w,)
        same = None
        if mode == 0: base = base or {}; base[w] = ids.clone()
        elif base and w in base: same = bool((base[w] == ids).all())
        print(json.dumps({"n": a.n, "d": a.d, "dtype": a.dtype, "metric": a.metric, "build_s": t_build, "build_window": L, "mode": mode, "window": w, "recall_at_10": rec, "ms_per_batch": ms, "ms": ms, "ids_checksum": int(ids.sum().item()),
                          "qps": a.nq / ms * 1e3, "evals": float(stats[:, 2].sum()) / a.nq, "bound_rejects": float(stats[:, 5].sum()) / a.nq, "expansions": float(stats[:, 0].sum()) / a.nq,
                          "kernel": idx.last_kernel_info(), "ids_equal_mode0": same}), flush=True)
