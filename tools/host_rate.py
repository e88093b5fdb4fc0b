#!/usr/bin/env python3
"""Host-pointer rate of svsx_search (host queries in, host results out) next to the device-resident rate, same index, 1 warm-up + 5 timed
(benchmark/include/svs-benchmark/search.h:209-219):   python tools/host_rate.py [--nq 10000] [--windows 128,23]"""
import argparse, ctypes as C, json, os, statistics, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd import _lib
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1_000_000); ap.add_argument("--nq", default="10000"); ap.add_argument("--windows", default="128,23")
a = ap.parse_args()
dev = torch.device("cuda", 0)
d, k, R = 96, 10, 64
x = deep_like_device(a.n, d, 24, 1, dev)
graph, entry = build_vamana_device(x, a.n, d, svs.DistanceL2(), VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), a.n, d, np.float32)
idx.build_bound_codes(2)
L = _lib.lib()
for nq in [int(v) for v in a.nq.split(",")]:
    q = deep_like_device(nq, d, 24, 2, dev)
    qh = q.cpu().numpy()
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dists = torch.empty((nq, k), device=dev)
    h64 = np.empty((nq, k), np.uint64); hd = np.empty((nq, k), np.float32)
    for w in [int(v) for v in a.windows.split(",")]:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        spc = sp._c()
        for _ in range(3): idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        dev_ms = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); idx.search_device(q, k, sp, d_ids64=ids, d_dists=dists); e1.record(); torch.cuda.synchronize()
            dev_ms.append(e0.elapsed_time(e1))
        aux = _lib.SearchAux()
        host_ms, py_ms = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            _lib.check(L.svsx_search(idx._h, qh.ctypes.data, 0, nq, k, C.byref(spc), h64.ctypes.data, None, hd.ctypes.data, C.byref(aux)))
            host_ms.append(1e3 * (time.perf_counter() - t0))
        for _ in range(6):
            t0 = time.perf_counter(); r = idx.search(qh, k, sp); py_ms.append(1e3 * (time.perf_counter() - t0))
        # round 6: does it matter whether the result arrays are the same ones every call (as above) or fresh ones (as the reference's harness has them:
        # Traits::search returns a new QueryResult per call, benchmark/include/svs-benchmark/search.h:209-219; idx.search allocates too)?
        fresh_ms, again_ms = [], []
        for _ in range(6):
            t0 = time.perf_counter()
            f64_, fd_ = np.empty((nq, k), np.uint64), np.empty((nq, k), np.float32)
            _lib.check(L.svsx_search(idx._h, qh.ctypes.data, 0, nq, k, C.byref(spc), f64_.ctypes.data, None, fd_.ctypes.data, C.byref(aux)))
            fresh_ms.append(1e3 * (time.perf_counter() - t0))
        for _ in range(6):
            t0 = time.perf_counter()
            _lib.check(L.svsx_search(idx._h, qh.ctypes.data, 0, nq, k, C.byref(spc), h64.ctypes.data, None, hd.ctypes.data, C.byref(aux)))
            again_ms.append(1e3 * (time.perf_counter() - t0))
        same = bool((h64.astype(np.int64) == ids.cpu().numpy()).all() and (r.indices == h64).all())
        med = lambda v: statistics.median(v[1:])
        print(json.dumps({"nq": nq, "window": w, "device_ms": med(dev_ms), "host_c_abi_ms": med(host_ms), "host_c_abi_min_ms": min(host_ms[1:]), "host_python_ms": med(py_ms), "host_c_abi_fresh_result_arrays_ms": med(fresh_ms), "host_c_abi_same_arrays_again_ms": med(again_ms),
                          "host_over_device": med(dev_ms) / med(host_ms), "ids_identical": same}), flush=True)
