#!/usr/bin/env python3
"""Where a query's time goes outside its expansions (DIAG variant: tools/build_variant.sh diag3 -DSVSX_DIAG=3; stats words 2 / 3 = wall_clock64 at
main-loop start / end, 4 / 5 = wave start / end, 100 MHz):  SVSX_LIB=variants/diag3/libsvsx.so python tools/diag_prologue.py [--nq 256,10000] [--windows 128,23]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="256,10000"); ap.add_argument("--windows", default="128,23"); ap.add_argument("--where", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev); stats = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats); e1.record(); torch.cuda.synchronize()
        st = stats.cpu().numpy().view(np.uint32).astype(np.int64)
        t0, tl, te, t1 = st[:, 4], st[:, 2], st[:, 3], st[:, 5]
        us = lambda a_, b_: ((b_ - a_) & 0xFFFFFFFF) * 0.01
        start = us(np.full_like(t0, t0.min()), t0)
        first = start < 20.0
        row = {"nq": nq, "window": w, "ms": e0.elapsed_time(e1), "expansions": float(st[:, 0].mean()), "first_generation": int(first.sum()),
               "span_us": float((start + us(t0, t1)).max())}
        dur = us(t0, t1)
        per = dur / np.maximum(st[:, 0], 1)
        row["duration_us_quantiles_p1_p10_p50_p90_p99_max"] = [float(v) for v in np.percentile(dur, [1, 10, 50, 90, 99, 100])]
        row["us_per_expansion_quantiles_p1_p10_p50_p90_p99_max"] = [float(v) for v in np.percentile(per, [1, 10, 50, 90, 99, 100])]
        row["expansions_quantiles_p1_p50_p99_max"] = [float(v) for v in np.percentile(st[:, 0], [1, 50, 99, 100])]
        row["corr_duration_expansions"] = float(np.corrcoef(dur, st[:, 0])[0, 1])
        row["corr_duration_slots"] = float(np.corrcoef(dur, st[:, 1])[0, 1])
        hw = st[:, 1]
        if a.where:  # mean us per expansion by where the wave ran (st[1] = HW_ID | XCC_ID << 28 in DIAG builds)
            fields = {"xcc": (hw >> 28) & 0xF, "se": (hw >> 13) & 0x3, "sh": (hw >> 12) & 1, "cu": (hw >> 8) & 0xF, "simd": (hw >> 4) & 0x3, "wave_slot": hw & 0xF,
                      "launch_order_decile": np.minimum(np.arange(nq) * 10 // max(nq, 1), 9)}
            row["where"] = {name: {int(v): [round(float(per[f == v].mean()), 3), int((f == v).sum())] for v in np.unique(f)} for name, f in fields.items()}
            key = ((hw >> 28) & 0xF) * 4096 + ((hw >> 13) & 3) * 1024 + ((hw >> 12) & 1) * 512 + ((hw >> 8) & 0xF) * 16 + ((hw >> 4) & 3)  # one SIMD
            simd_mean = np.array([per[key == v].mean() for v in np.unique(key)])
            row["per_simd_mean_us_per_expansion_quantiles_p1_p10_p50_p90_p99"] = [float(v) for v in np.percentile(simd_mean, [1, 10, 50, 90, 99])]
            row["n_simds"] = int(len(simd_mean))
            cu_key = key // 4
            cu_mean = np.array([per[cu_key == v].mean() for v in np.unique(cu_key)])
            row["per_cu_mean_quantiles_p1_p10_p50_p90_p99"] = [float(v) for v in np.percentile(cu_mean, [1, 10, 50, 90, 99])]
            # within a SIMD: spread of its waves around the SIMD's mean
            dev = np.concatenate([per[key == v] - per[key == v].mean() for v in np.unique(key)])
            row["within_simd_std"] = float(dev.std()); row["between_simd_std"] = float(simd_mean.std()); row["between_cu_std"] = float(cu_mean.std())
        endt = start + dur
        row["end_us_quantiles_p10_p50_p90_p99_max"] = [float(v) for v in np.percentile(endt, [10, 50, 90, 99, 100])]
        for name, sel in (("first", first), ("late", ~first)):
            if sel.any():
                row[name] = {"prologue_us": float(us(t0, tl)[sel].mean()), "loop_us": float(us(tl, te)[sel].mean()), "epilogue_us": float(us(te, t1)[sel].mean()),
                             "start_us_mean": float(start[sel].mean()), "start_us_max": float(start[sel].max()), "end_us_p50": float(np.median((start + us(t0, t1))[sel])),
                             "end_us_max": float((start + us(t0, t1))[sel].max())}
        print(json.dumps(row), flush=True)
