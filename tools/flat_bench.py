"""Timing of the MFMA flat search (FlatIndex::search analogue); GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from bench import deep_like_device, brute_force_topk
dev = torch.device("cuda", 0)
for (n, d, nq, dt) in ((1_000_000, 96, 10_000, "f32"), (1_000_000, 128, 10_000, "f16")):
    x = deep_like_device(n, d, 24, 1, dev); q = deep_like_device(nq, d, 24, 2, dev)
    xd = x if dt == "f32" else x.half()
    idx = svs.VamanaIndex.from_device(torch.full((n, 16), -1, dtype=torch.int32, device=dev), xd, svs.DistanceL2(), torch.zeros(1, dtype=torch.int32, device=dev), n, d,
                                      np.float32 if dt == "f32" else np.float16)
    ids, dists = idx.flat_search_device(q, 10); torch.cuda.synchronize()   # builds the f16 shadow
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): idx.flat_search_device(q, 10, ids, dists)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    t0 = time.perf_counter(); gt = brute_force_topk(xd.float(), q, 10); torch.cuda.synchronize(); t_torch = time.perf_counter() - t0
    agree = (ids == gt).float().mean().item()
    flops = 2.0 * n * nq * ((d + 15) // 16 * 16)
    print(f"{dt} n={n} d={d} nq={nq}: flat search {ms:.1f} ms ({flops/ms/1e9:.0f} TFLOP/s f16 MFMA incl. shortlist+merge+rerank), torch GEMM+topk {t_torch*1e3:.0f} ms, id agreement {agree:.5f}")
