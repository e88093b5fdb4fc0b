"""Batched BatchIterator at the bench scale (1 M x 96 f32 L2, 10 000 queries): time per next(), resumed search vs a
fresh search of the same window, ids identical to the reference-header CPU iterator on a sample.  GPU box."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs
from bench import deep_like_device, host_cores
from oracle.oracle import L2, Ref, Oracle

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq, bs, nb = 10_000, int(sys.argv[2]) if len(sys.argv) > 2 else 10, 10
x = deep_like_device(n, 96, 24, 1, dev); q = deep_like_device(nq, 96, 24, 2, dev)
graph, entry = build_vamana_device(x, n, 96, svs.DistanceL2(), VamanaBuildParameters(graph_max_degree=64, window_size=128, prune_to=60), device=0)
idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), n, 96, np.float32)
qh = q.cpu().numpy()
it = idx.batch_iterator(qh)
d_ids = torch.empty((nq, bs), dtype=torch.int64, device=dev); d_d = torch.empty((nq, bs), dtype=torch.float32, device=dev)
d_c = torch.empty(nq, dtype=torch.int32, device=dev)
s_ids = torch.empty((nq, bs), dtype=torch.int64, device=dev); s_d = torch.empty((nq, bs), dtype=torch.float32, device=dev)
all_ids = []
for b in range(nb):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it.next_device(bs, d_ids, d_d, d_c)
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    all_ids.append(d_ids.cpu().numpy().copy())
    w = bs * (b + 1)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w, w + 100), True)
    idx.search_device(q, bs, sp, d_ids64=s_ids, d_dists=s_d); torch.cuda.synchronize(); t0 = time.perf_counter()
    idx.search_device(q, bs, sp, d_ids64=s_ids, d_dists=s_d); torch.cuda.synchronize(); ms2 = (time.perf_counter() - t0) * 1e3
    print(f"next #{b+1}: window {w:4d} capacity {w+100:4d}: {ms:7.2f} ms ({nq/ms/1e3:.2f} M queries/s); fresh search of the same window/capacity: {ms2:7.2f} ms", flush=True)
# CPU: the reference's buffer/search driven with iterator.h's control flow, sample of queries
ref = Ref() if Ref.available() else None
chk = ref if ref is not None else Oracle()
xh, gh, eh = x.cpu().numpy(), device_graph_to_svs(graph, 64), np.array([entry], np.uint32)
ns = 64
t0 = time.perf_counter()
same = True
for i in range(ns):
    res = chk.iterate(L2, xh, gh, eh, qh[i], [bs] * nb)[0]
    for b in range(nb):
        same &= bool((all_ids[b][i] == res[b][0].astype(np.int64)).all())
cpu_s = time.perf_counter() - t0
print(f"CPU {'reference headers' if ref else 'oracle'} iterator, 1 thread: {cpu_s/ns*1e3:.2f} ms per query for {nb} x next({bs}) -> {ns/cpu_s:.0f} queries/s/core; ids identical on the sample: {same}")
