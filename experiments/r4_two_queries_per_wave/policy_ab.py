#!/usr/bin/env python3
"""One query per wavefront vs two (svsx_index_set_kernel_policy 1 / 2) on the config-2 workload, same process, same graph:
   python tools/policy_ab.py [--nq 2000,6144,10000,40000] [--windows 128,23] [--config c2|c3]
One JSON line per (nq, window): ms per launch under either policy (median of --reps timed groups of 5 launches), ids identical, per-query statistics."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="2000,4096,6144,8192,10000,20000,40000"); ap.add_argument("--windows", default="128,23")
ap.add_argument("--config", default="c2"); ap.add_argument("--reps", type=int, default=5); ap.add_argument("--visited", type=int, default=1); ap.add_argument("--policy", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
P2 = a.policy  # the policy compared with the one-query kernel (3 interleaved, 2 SIMT halves)
k, R = 10, 64
if a.config == "c2":
    d, dist, dt = 96, svs.DistanceL2(), np.float32
    x = deep_like_device(a.n, d, 24, 1, dev)
    mk_q = lambda nq: deep_like_device(nq, d, 24, 2, dev)
    alpha = 1.2
else:  # config 3: float16 rows, float32 queries, inner product
    d, dist, dt = 128, svs.DistanceIP(), np.float16
    x = deep_like_device(a.n, d, 32, 31, dev, normalize=False).half()
    mk_q = lambda nq: deep_like_device(nq, d, 32, 32, dev, normalize=False)
    alpha = 0.95
graph, entry = build_vamana_device(x, a.n, d, dist, VamanaBuildParameters(alpha=alpha, graph_max_degree=R, window_size=2 * R, prune_to=R - 4), data_dtype=dt)
idx = svs.VamanaIndex.from_device(graph, x, dist, torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, dt)
idx.build_bound_codes(2)
for nq in [int(v) for v in a.nq.split(",")]:
    q = mk_q(nq)
    out = {p: (torch.empty((nq, k), dtype=torch.int64, device=dev), torch.empty((nq, k), device=dev), torch.zeros((nq, 6), dtype=torch.int32, device=dev)) for p in (1, P2)}
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), a.visited)
        ms, info, st = {}, {}, {}
        for p in (1, P2):
            idx.set_kernel_policy(p)
            ids, dd, stats = out[p]
            idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd, d_stats=stats)
            torch.cuda.synchronize()
            info[p] = idx.last_kernel_info()
            st[p] = stats.cpu().numpy().astype(np.int64)
        samples = {1: [], P2: []}
        for _ in range(a.reps):
            for p in (1, P2):  # alternate: clocks drift within a call
                idx.set_kernel_policy(p)
                ids, dd, _ = out[p]
                idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
                e1.record(); torch.cuda.synchronize()
                samples[p].append(e0.elapsed_time(e1) / 5)
        for p in (1, P2): ms[p] = float(np.median(samples[p]))
        same = bool((out[1][0] == out[P2][0]).all() and (out[1][1].view(torch.int32) == out[P2][1].view(torch.int32)).all())
        print(json.dumps({"config": a.config, "nq": nq, "window": w, "ms_1q": ms[1], "ms_2q": ms[P2], "speedup": ms[1] / ms[P2], "ids_and_bits_identical": same,
                          "kernel_2q": info[P2], "expansions": float(st[1][:, 0].mean()), "same_expansions": bool((st[1][:, 0] == st[P2][:, 0]).all() and (st[1][:, 1] == st[P2][:, 1]).all()),
                          "rows_fetched_1q": float((st[1][:, 2] - st[1][:, 5]).mean()), "rows_fetched_2q": float((st[P2][:, 2] - st[P2][:, 5]).mean()),
                          "qps_2q": nq / ms[P2] * 1e3}), flush=True)
idx.set_kernel_policy(0)
