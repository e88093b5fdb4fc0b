import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device
from bench import deep_like_device
dev = torch.device("cuda", 0)
n, d, nq, k = 1_000_000, 768, 4096, 10
fn = svs.DistanceCosineSimilarity()
xf = deep_like_device(n, d, 64, 1, dev, normalize=False); qf = deep_like_device(nq, d, 64, 2, dev, normalize=False)
mn, mx = float(xf.min()), float(xf.max()); scale = (mx - mn) / 255.0; bias = mn + 128.0 * scale
x = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8); del xf
g, e = build_vamana_device(x, n, d, fn, VamanaBuildParameters(alpha=0.95, graph_max_degree=64, window_size=128, prune_to=60), data_dtype=np.int8, sq=(scale, bias))
torch.cuda.empty_cache()
idx = svs.VamanaIndex.from_device(g, x, fn, torch.tensor([e], dtype=torch.int32, device=dev), n, d, np.int8, sq=(scale, bias))
ids = torch.empty((nq, k), dtype=torch.int64, device=dev); dist = torch.empty((nq, k), device=dev); st = torch.zeros((nq, 6), dtype=torch.int32, device=dev)
for rb in (32, 64, 128):
    idx.build_bound_codes(rb)  # inline records of exactly this size
    for w in (128, 512):
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w), 1)
        idx.search_device(qf, k, sp, d_ids64=ids, d_dists=dist, d_stats=st); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): idx.search_device(qf, k, sp, d_ids64=ids, d_dists=dist)
        e1.record(); torch.cuda.synchronize()
        print("rb", rb, "window", w, "ms %.3f" % (e0.elapsed_time(e1) / 5), "rejects %.0f of %.0f" % (float(st[:, 5].sum()) / nq, float(st[:, 2].sum()) / nq), idx.last_kernel_info())
