"""The in-library builder (svsx_vamana_build_device, csrc/svsx_builder.hip = scalablevectorsearch_amd.build.build_vamana_device) against the
ORACLE's sequential builder (oracle/svs_oracle_build.c: vamana_build.h / prune.h semantics).  The reference's own build depends on its thread
schedule, so graphs are compared by validity (core/graph/graph.h:169-172) and by search quality at equal windows, not byte for byte.
(Round 5: the checker used to be this repo's round-2 torch-glue builder -- a self-comparison.)"""
import numpy as np
import pytest
import torch

import scalablevectorsearch_amd as svs
from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device

pytestmark = pytest.mark.gpu
L2, IP, COS = 0, 1, 2


def _recall(index, q, gt, k, window):
    res = index.search(q, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(window)))
    return float(np.mean([len(set(a) & set(b)) / k for a, b in zip(res.indices.astype(np.int64), gt)]))


@pytest.mark.parametrize("metric", ["l2", "ip", "cos"])
def test_native_builder_valid_graph_and_quality(oracle, metric):
    from conftest import deep_like

    n, d, R, k = 8000, 48, 32, 10
    x = deep_like(n, d, 12, seed=3)
    q = deep_like(300, d, 12, seed=4)
    fn = {"l2": svs.DistanceL2(), "ip": svs.DistanceIP(), "cos": svs.DistanceCosineSimilarity()}[metric]
    om = {"l2": L2, "ip": IP, "cos": COS}[metric]
    alpha = 1.2 if metric == "l2" else 0.95
    dx = torch.from_numpy(x).cuda()
    bp = VamanaBuildParameters(alpha=alpha, graph_max_degree=R, window_size=2 * R, prune_to=R - 4)
    g_nat, e_nat = build_vamana_device(dx, n, d, fn, bp)
    gn = g_nat.cpu().numpy()
    # valid: ids in range, packed from slot 0, no self loops, no duplicates, padding beyond R untouched
    ids = gn[:, :R]
    valid = ids >= 0
    assert (ids[valid] < n).all()
    assert (valid[:, :-1] | ~valid[:, 1:]).all()          # packed
    assert (gn[:, R:] == -1).all()
    assert not (ids == np.arange(n)[:, None]).any()
    srt = np.sort(np.where(valid, ids, -np.arange(1, R + 1)[None, :]), axis=1)
    assert (srt[:, 1:] != srt[:, :-1]).all()              # no duplicate neighbours
    deg = valid.sum(1)
    assert deg.min() >= 1 and deg.mean() > R / 3
    # quality gate: recall vs exhaustive ground truth within 1 % of the ORACLE builder's graph at the same window, both searched on the GPU
    g_o, e_o = oracle.build_vamana(om, x, R=R, L=2 * R, prune_to=R - 4, alpha=alpha)
    assert abs(float(g_o[:, 0].mean()) - float(deg.mean())) < 0.15 * float(g_o[:, 0].mean())
    ix_ref = svs.VamanaIndex(g_o, x, fn, e_o)
    ix_nat = svs.VamanaIndex.from_device(g_nat, dx, fn, torch.tensor([e_nat], dtype=torch.int32).cuda(), n, d, np.float32)
    gi, _ = ix_nat.flat_search_device(torch.from_numpy(q).cuda(), k)
    gt = gi.cpu().numpy()
    for w in (20, 40):
        r_ref, r_nat = _recall(ix_ref, q, gt, k, w), _recall(ix_nat, q, gt, k, w)
        assert r_nat >= r_ref - 0.01, (metric, w, r_ref, r_nat)


def test_native_medoid_matches_torch():
    from conftest import deep_like
    from scalablevectorsearch_amd.build import find_medoid

    n, d = 30000, 48  # device views take rows padded to 16 elements
    x = deep_like(n, d, 8, seed=9)
    dx = torch.from_numpy(x).cuda()
    g = torch.full((n, 16), -1, dtype=torch.int32).cuda()
    ix = svs.VamanaIndex.from_device(g, dx, svs.DistanceL2(), torch.zeros(1, dtype=torch.int32).cuda(), n, d, np.float32)
    m_nat, m_ref = ix.find_medoid(), find_medoid(dx)
    mean = x.astype(np.float64).mean(0)
    d_nat, d_ref = ((x[m_nat] - mean) ** 2).sum(), ((x[m_ref] - mean) ** 2).sum()
    assert m_nat == m_ref or abs(d_nat - d_ref) <= 1e-6 * d_ref  # a different row only on a float32 near-tie


def test_native_builder_sq_and_small_batches(oracle):
    """scalar-quantized rows (decompressed float queries) and n smaller than the minimum batch count"""
    from conftest import deep_like

    n, d, R = 1500, 64, 16
    xf = deep_like(n, d, 8, seed=5)
    mn, mx = float(xf.min()), float(xf.max())
    scale = (mx - mn) / 255.0
    bias = mn + 128.0 * scale
    xq = np.clip(np.round((xf - bias) / scale), -128, 127).astype(np.int8)
    dx = torch.from_numpy(xq).cuda()
    bp = VamanaBuildParameters(alpha=0.95, graph_max_degree=R, window_size=2 * R, prune_to=R - 2)
    fn = svs.DistanceIP()
    g_nat, e_nat = build_vamana_device(dx, n, d, fn, bp, data_dtype=np.int8, sq=(scale, bias))
    gn = g_nat.cpu().numpy()[:, :R]
    assert ((gn >= -1) & (gn < n)).all() and (gn >= 0).sum(1).min() >= 1
    # searched on the GPU against the exhaustive answer over the same codes, the graph must navigate (the oracle builder takes no codes)
    ix = svs.VamanaIndex.from_device(g_nat, dx, fn, torch.tensor([e_nat], dtype=torch.int32).cuda(), n, d, np.int8, sq=(scale, bias))
    qf = deep_like(100, d, 8, seed=6)
    gi, _ = ix.flat_search_device(torch.from_numpy(qf).cuda(), 10)
    r = _recall(ix, qf, gi.cpu().numpy(), 10, 48)
    assert r > 0.9, r


def test_builder_without_the_full_search_history(oracle):
    """VamanaBuildParameters::use_full_search_history = false (build_params.h:28-75, vamana_build.h:407-421): a vertex's candidates are the search
    buffer's contents instead of the nodes its search expanded.  Checked like the default: valid graph, and search quality within 1 % of the ORACLE
    builder's graph built with the same flag -- and the flag must change the graph."""
    from conftest import deep_like

    n, d, R, k = 8000, 48, 32, 10
    x = deep_like(n, d, 12, seed=3)
    q = deep_like(300, d, 12, seed=4)
    fn = svs.DistanceL2()
    dx = torch.from_numpy(x).cuda()
    g_full, _ = build_vamana_device(dx, n, d, fn, VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4))
    g_buf, e_buf = build_vamana_device(dx, n, d, fn, VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, prune_to=R - 4, use_full_search_history=False))
    gn = g_buf.cpu().numpy()[:, :R]
    valid = gn >= 0
    assert (gn[valid] < n).all() and not (gn == np.arange(n)[:, None]).any() and valid.sum(1).min() >= 1
    assert float((g_buf == g_full).all(1).float().mean()) < 1.0
    g_o, e_o = oracle.build_vamana(L2, x, R=R, L=2 * R, prune_to=R - 4, alpha=1.2, use_full_search_history=False)
    ix_ref = svs.VamanaIndex(g_o, x, fn, e_o)
    ix_nat = svs.VamanaIndex.from_device(g_buf, dx, fn, torch.tensor([e_buf], dtype=torch.int32).cuda(), n, d, np.float32)
    gi, _ = ix_nat.flat_search_device(torch.from_numpy(q).cuda(), k)
    gt = gi.cpu().numpy()
    for w in (20, 40):
        r_ref, r_nat = _recall(ix_ref, q, gt, k, w), _recall(ix_nat, q, gt, k, w)
        assert r_nat >= r_ref - 0.01, (w, r_ref, r_nat)
