#!/usr/bin/env python3
"""Per-expansion latency and early-block hit rate of one launch, from a DIAG variant library (tools/build_variant.sh diag -DSVSX_DIAG=1:
stats words 3 / 4 / 5 = expansions whose next block was requested early, wall_clock64 (100 MHz) at wave start / end):
   SVSX_LIB=variants/diag/libsvsx.so python tools/diag_early.py [--nq 256,2048,10000] [--windows 128,23]
One JSON line per (nq, window): kernel ms, expansions, early share, per-expansion microseconds for the first-generation waves and for the late starters."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="256,2048,10000"); ap.add_argument("--windows", default="128,23")
ap.add_argument("--d", type=int, default=96); ap.add_argument("--lds-requests", default="0"); ap.add_argument("--visited", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = a.d, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev); stats = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
    for w, req in [(int(v), r) for v in a.windows.split(",") for r in a.lds_requests.split(",")]:
        os.environ["SVSX_LDS_REQUEST"] = req
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), a.visited)
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats)
        torch.cuda.synchronize()
        st = stats.cpu().numpy().view(np.uint32).astype(np.int64)
        nexp, early = st[:, 0], st[:, 3]
        t0, t1 = st[:, 4], st[:, 5]
        dur = ((t1 - t0) & 0xFFFFFFFF) * 0.01  # microseconds
        start = ((t0 - t0.min()) & 0xFFFFFFFF) * 0.01
        first = start < 20.0   # the first resident generation
        late = start > np.percentile(start, 90) if (~first).any() else first
        per = dur / np.maximum(nexp, 1)
        print(json.dumps({"nq": nq, "window": w, "lds_request": int(req), "ms": ms, "kernel": idx.last_kernel_info(), "expansions": float(nexp.mean()), "early_share": float(early.sum() / max(nexp.sum(), 1)), "kept_used_share": float(st[:, 2].sum() / max(nexp.sum(), 1)),
                          "first_generation": int(first.sum()), "us_per_expansion_first": float(per[first].mean()), "us_per_expansion_late": float(per[late].mean()),
                          "query_us_first": float(dur[first].mean()), "query_us_late": float(dur[late].mean()), "span_us": float((start + dur).max())}), flush=True)
