#!/usr/bin/env python3
"""Measured points for the other BASELINE configs (parity-test cases, not the bench line):
   tools/bench_config.py --n 1000000 --d 128 --dtype f16 --metric ip --nq 10000            (config 3)
   tools/bench_config.py --n 2000000 --d 768 --dtype sq8 --metric cos --nq 4096 --latent 64  (config 4, scaled)
Prints one JSON line (QPS, recall, algorithmic bytes -> GB/s, CPU reference QPS on the same box)."""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs
from bench import deep_like_device, host_cores, recall_at


def gen(n, d, latent, seed, dev, normalize):
    ga = torch.Generator(device=dev).manual_seed(7)
    a = torch.randn((latent, d), generator=ga, device=dev)
    g = torch.Generator(device=dev).manual_seed(seed)
    out = torch.empty((n, d), device=dev)
    for lo in range(0, n, 1 << 18):
        m = min(1 << 18, n - lo)
        x = torch.randn((m, latent), generator=g, device=dev) @ a + 0.1 * torch.randn((m, d), generator=g, device=dev)
        out[lo:lo + m] = x / x.norm(dim=1, keepdim=True) if normalize else x
    return out


def exact_topk(xf, qf, k, metric, chunk=1 << 17):
    best_d = torch.full((qf.shape[0], k), float("inf"), device=qf.device)
    best_i = torch.zeros((qf.shape[0], k), dtype=torch.int64, device=qf.device)
    qn = qf / qf.norm(dim=1, keepdim=True) if metric == "cos" else qf
    for lo in range(0, xf.shape[0], chunk):
        xs = xf[lo:lo + chunk]
        if metric == "l2":
            dist = (qf * qf).sum(1, keepdim=True) + (xs * xs).sum(1)[None] - 2 * qf @ xs.T
        elif metric == "ip":
            dist = -(qf @ xs.T)
        else:
            dist = -(qn @ (xs / xs.norm(dim=1, keepdim=True)).T)
        dd, ii = torch.topk(dist, k, dim=1, largest=False)
        cd, ci = torch.cat([best_d, dd], 1), torch.cat([best_i, ii + lo], 1)
        sel = torch.topk(cd, k, dim=1, largest=False)
        best_d, best_i = sel.values, torch.gather(ci, 1, sel.indices)
    return best_i


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--dtype", default="f16", choices=["f32", "f16", "sq8", "i8"]); ap.add_argument("--metric", default="ip", choices=["l2", "ip", "cos"])
    ap.add_argument("--nq", type=int, default=10_000); ap.add_argument("--window", type=int, default=128); ap.add_argument("--degree", type=int, default=64)
    ap.add_argument("--latent", type=int, default=32); ap.add_argument("--steps", type=int, default=10); ap.add_argument("--cpu", type=int, default=1); ap.add_argument("--visited", type=int, default=1); ap.add_argument("--bound-codes", type=int, default=0)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fn = {"l2": svs.DistanceL2(), "ip": svs.DistanceIP(), "cos": svs.DistanceCosineSimilarity()}[a.metric]
    xf = gen(a.n, a.d, a.latent, 1, dev, normalize=a.metric == "l2")
    qf = gen(a.nq, a.d, a.latent, 2, dev, normalize=a.metric == "l2")
    sq, q, np_dt = None, qf, np.float32
    if a.dtype == "f32":
        x = xf
    elif a.dtype == "f16":
        x, np_dt = xf.half(), np.float16
        xf = x.float()  # the database IS the f16 values
    elif a.dtype == "sq8":  # SQDataset<int8>: global scale/bias (quantization/scalar/scalar.h:454-462), f32 queries
        mn, mx = float(xf.min()), float(xf.max())
        scale = (mx - mn) / 255.0; bias = mn + 128.0 * scale
        x = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8); sq, np_dt = (scale, bias), np.int8
    else:  # raw int8 data and int8 queries (VNNI pairs)
        s = 127.0 / float(xf.abs().max())
        x = torch.clamp(torch.round(xf * s), -128, 127).to(torch.int8); q = torch.clamp(torch.round(qf * s), -128, 127).to(torch.int8); np_dt = np.int8
        xf, qf = x.float(), q.float()
    if a.d % 16:  # device layout: rows zero padded to a multiple of 16 elements (include/svsx.h)
        x = torch.nn.functional.pad(x, (0, 16 - a.d % 16)).contiguous()
    alpha = 1.2 if a.metric == "l2" else 0.95
    t0 = time.time()
    graph, entry = build_vamana_device(x, a.n, a.d, fn, VamanaBuildParameters(alpha=alpha, graph_max_degree=a.degree, window_size=2 * a.degree, prune_to=a.degree - 4),
                                       data_dtype=np_dt, sq=sq)
    t_build = time.time() - t0
    idx = svs.VamanaIndex.from_device(graph, x, fn, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, a.d, np_dt, sq=sq)
    if a.bound_codes:
        idx.build_bound_codes(a.bound_codes)
    k = 10
    ids = torch.empty((a.nq, k), dtype=torch.int64, device=dev); dists = torch.empty((a.nq, k), device=dev)
    stats = torch.zeros((a.nq, 6), dtype=torch.int32, device=dev)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(a.window), search_buffer_visited_set=a.visited)
    idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats); torch.cuda.synchronize()
    gt = exact_topk(xf if sq is None else xf, qf, k, a.metric)
    rec = recall_at(gt, ids, k)
    es = {"f32": 4, "f16": 2, "sq8": 1, "i8": 1}[a.dtype]; qs = 1 if a.dtype == "i8" else 4
    st = stats.to(torch.int64); exp, slots = int(st[:, 0].sum()), int(st[:, 1].sum())
    alg = 4 * (exp + slots) + slots * a.d * es + a.nq * (a.d * es + a.d * qs + k * 12)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
    e0.record()
    for _ in range(a.steps): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    ms_plain = None
    if a.bound_codes:  # same index without the bound, for the A/B in one process
        ids_b = ids.clone()
        stats_b = stats.clone()
        idx.build_bound_codes(0)
        for _ in range(2): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        e0.record()
        for _ in range(a.steps): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        e1.record(); torch.cuda.synchronize()
        ms_plain = e0.elapsed_time(e1) / a.steps
        assert bool((ids == ids_b).all()), "bound codes changed the result"
        idx.build_bound_codes(a.bound_codes)
    out = dict(config=vars(a), build_s=t_build, mean_degree=float((graph[:, :a.degree] >= 0).sum(1).float().mean()), recall_at_10=rec, ms_per_batch=ms,
               qps=a.nq / ms * 1e3, ms_per_batch_without_bound=ms_plain, bound_rejects_per_query=float(stats[:, 5].sum()) / a.nq, evals_per_query=float(stats[:, 2].sum()) / a.nq, expansions=exp / a.nq, slots=slots / a.nq, alg_GB=alg / 1e9, achieved_GBs=alg / ms / 1e6, frac_of_8TBs=alg / ms / 1e6 / 8000)
    if a.cpu:
        from oracle.oracle import COS, IP, L2, Oracle, Ref
        m = {"l2": L2, "ip": IP, "cos": COS}[a.metric]
        xh, qh, gh = np.ascontiguousarray(x[:, :a.d].cpu().numpy()), q.cpu().numpy(), device_graph_to_svs(graph, a.degree)
        ns = min(a.nq, 2000); th = host_cores()
        if sq is None and Ref.available():
            r = Ref(); run = lambda: r.search(m, xh, gh, np.array([entry], np.uint32), qh[:ns], k, a.window, nthreads=th)[0]; kind = f"reference x{th}"
        else:
            o = Oracle(); ns = min(ns, 200); run = lambda: o.search(m, xh, gh, np.array([entry], np.uint32), qh[:ns], k, a.window, sq=sq)[0]; kind = "port x1 (SQ is not in oracle/_ref)"
        run(); t0 = time.perf_counter(); ci = run(); dt = time.perf_counter() - t0
        out.update(cpu_qps=ns / dt, cpu_kind=kind, gpu_ids_identical_on_sample=bool((ids[:ns].cpu().numpy() == ci.astype(np.int64)).all()))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
