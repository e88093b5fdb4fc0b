"""The two-queries-per-wavefront kernels (csrc/svsx_search_x2.h: interleaved, policy 3 and the auto policy's choice for large batches;
csrc/svsx_search_2q.h: SIMT halves, policy 2; svsx_index_set_kernel_policy): ids, distance bits and the tracker counts
(expansions, neighbour slots) must equal the oracle's and the one-query kernel's on everything the kernel is eligible for -- every buffer
level (1 / 2 / 4 entries per lane), split buffers, both filter settings, odd batches, half-filled adjacency rows, several entry
points, exact ties and duplicate vectors, and non-finite distances (which hand the query back to the one-query routine)."""
import numpy as np
import pytest

from conftest import deep_like

pytestmark = pytest.mark.gpu

L2, IP = 0, 1


@pytest.fixture(scope="module")
def svs():
    import scalablevectorsearch_amd as svs

    return svs


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same_dists(got, want):
    return ((_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))).all()


# (window, capacity, k, visited filter)
CONFIGS = [(10, 10, 10, 0), (23, 23, 10, 1), (32, 32, 10, 1), (33, 33, 10, 1), (20, 40, 10, 0), (64, 64, 10, 1), (65, 100, 10, 1),
           (128, 128, 10, 1), (5, 5, 5, 0), (64, 64, 64, 0), (100, 128, 100, 1), (128, 128, 10, 0), (1, 1, 1, 1), (96, 128, 10, 256)]


def _check(svs, oracle, metric, x, graph, entry, q, configs=CONFIGS, expect_2q=True, record_mode=32):
    dist = svs.DistanceL2() if metric == L2 else svs.DistanceIP()
    idx = svs.VamanaIndex(graph, x, dist, entry)
    idx.build_bound_codes(record_mode)
    n2q = 0
    for w, c, k, vis in configs:
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c), vis)
        ids, dists, ost = oracle.search(metric, x, graph, entry, q, k, w, c, visited_set=bool(vis), stats=True)
        for policy in (3, 2, 1):
            idx.set_kernel_policy(policy)
            res, st = idx.search(q, k, sp, stats=True)
            info = idx.last_kernel_info()
            if policy == 3:  # the interleaved kernel takes everything the SIMT form takes, and any number (<= 16) of entry points
                assert info.startswith("search_kernel_x2<") and info.endswith("codes=2>"), info
            if policy == 2 and expect_2q:
                assert info.startswith("search_kernel_2q<") and info.endswith("codes=2>"), info
                n2q += 1
            if policy == 1:
                assert info.startswith("search_kernel<"), info
            assert (res.indices == ids.astype(np.uint64)).all(), (w, c, k, vis, policy, info)
            assert _same_dists(res.distances, dists), (w, c, k, vis, policy, info)
            assert (st[:, 0] == ost[:, 0]).all() and (st[:, 1] == ost[:, 1]).all(), (w, c, k, vis, policy, info)
            if not vis:
                assert (st[:, 2] == ost[:, 2]).all(), (w, c, k, vis, policy)  # every candidate accounted for (bound rejects included)
    idx.set_kernel_policy(0)
    return n2q


def test_two_query_f32_l2_d96_half_rows(svs, oracle, small_world):
    """R = 32: the second adjacency slot of every lane is empty"""
    sw = small_world
    assert _check(svs, oracle, L2, sw["x"], sw["graph"], sw["entry"], sw["q"]) == len(CONFIGS)


@pytest.mark.parametrize("nq", [1, 2, 3, 33])
def test_two_query_odd_batches(svs, oracle, small_world, nq):
    sw = small_world
    _check(svs, oracle, L2, sw["x"], sw["graph"], sw["entry"], sw["q"][:nq], [(10, 10, 10, 0), (64, 64, 10, 1), (128, 128, 10, 1)])


@pytest.mark.parametrize("R", [48, 64])
def test_two_query_full_rows(svs, oracle, R):
    x = deep_like(3000, 96, 24, seed=11)
    q = deep_like(50, 96, 24, seed=12)
    g, e = oracle.build_vamana(L2, x, R=R, L=R + 16)
    _check(svs, oracle, L2, x, g, e, q)


@pytest.mark.parametrize("metric", [L2, IP])
@pytest.mark.parametrize("d", [96, 128])
@pytest.mark.parametrize("xdt,qdt", [(np.float32, np.float32), (np.float16, np.float32), (np.float16, np.float16), (np.float32, np.float16)])
def test_two_query_type_pairs(svs, oracle, metric, d, xdt, qdt):
    x = deep_like(2000, d, 24, seed=21 + d, normalize=(metric == L2)).astype(xdt)
    q = deep_like(40, d, 24, seed=22 + d, normalize=(metric == L2)).astype(qdt)
    g, e = oracle.build_vamana(metric, x, R=40, L=56, alpha=1.2 if metric == L2 else 0.95)
    _check(svs, oracle, metric, x, g, e, q, [(10, 10, 10, 0), (32, 48, 10, 1), (64, 64, 10, 1), (128, 128, 20, 1), (100, 100, 10, 0)])


def test_two_query_ties_and_duplicates(svs, oracle):
    """every vector stored four times (exact distance ties, candidates that duplicate buffer entries once the filter is off or evicts),
    coarse-grained coordinates (ties between different vectors)"""
    rng = np.random.default_rng(5)
    base = np.round(deep_like(400, 96, 8, seed=31, normalize=False) * 2.0) / 2.0
    x = np.ascontiguousarray(np.concatenate([base, base, base, base])[rng.permutation(1600)], np.float32)
    q = np.ascontiguousarray(np.concatenate([x[rng.integers(0, 1600, 20)], np.round(deep_like(20, 96, 8, seed=32, normalize=False) * 2.0) / 2.0]), np.float32)
    g, e = oracle.build_vamana(L2, x, R=32, L=48)
    _check(svs, oracle, L2, x, g, e, q, [(10, 10, 10, 0), (40, 40, 10, 0), (64, 64, 10, 64), (128, 128, 30, 0), (128, 128, 30, 1), (30, 60, 10, 0)], record_mode=32)
    g, e = oracle.build_vamana(IP, x, R=32, L=48, alpha=0.95)
    _check(svs, oracle, IP, x, g, e, q, [(10, 10, 10, 0), (64, 64, 10, 64), (128, 128, 30, 0)], record_mode=32)


def test_two_query_entry_points(svs, oracle, small_world):
    sw = small_world
    for ne in (2, 3, 8):
        entry = np.arange(7, 7 + 13 * ne, 13, dtype=np.uint32)
        _check(svs, oracle, L2, sw["x"], sw["graph"], entry, sw["q"][:20], [(10, 10, 10, 0), (64, 64, 10, 1), (128, 128, 10, 1)])
    # more entry points than the kernel takes, or than the buffer holds: the one-query kernel serves the call
    entry = np.arange(3, 3 + 9 * 11, 11, dtype=np.uint32)
    assert _check(svs, oracle, L2, sw["x"], sw["graph"], entry, sw["q"][:20], [(10, 10, 10, 0), (64, 64, 10, 1)], expect_2q=False) == 0
    entry = np.array([5, 50, 500], np.uint32)
    _check(svs, oracle, L2, sw["x"], sw["graph"], entry, sw["q"][:20], [(2, 2, 2, 0), (1, 1, 1, 1)], expect_2q=False)


def test_two_query_non_finite_distances(svs, oracle, small_world):
    """NaN queries (every distance NaN), queries that overflow float32 (+inf distances), and an infinite coordinate next to ordinary
    queries: the halves that meet a NaN are answered again by the one-query routine inside the same wave"""
    sw = small_world
    q = sw["q"][:24].copy()
    q[1, 5] = np.nan
    q[4, :] = 3.0e19          # squares overflow: +inf distances
    q[7, 0] = np.inf          # inf - finite = inf, squared inf: +inf distances; under IP +-inf
    q[10, 3] = -np.inf
    q[13, :] = np.nan
    q[14, :] = np.nan         # both halves of one wave
    q[15, :] = np.nan
    for metric in (L2, IP):
        g, e = (sw["graph"], sw["entry"]) if metric == L2 else oracle.build_vamana(IP, sw["x"], R=32, L=48, alpha=0.95)
        _check(svs, oracle, metric, sw["x"], g, e, q, [(10, 10, 10, 0), (32, 32, 10, 1), (64, 100, 10, 0), (128, 128, 10, 1)])
    x = sw["x"].copy()
    x[400, :] = 3.0e19        # a row whose squared distance overflows: +inf meets only the queries that reach it (records need finite DATA, so no NaN rows)
    x[17, :] *= np.float32(1.0e18)
    try:
        _check(svs, oracle, L2, x, sw["graph"], sw["entry"], sw["q"][:24], [(32, 32, 10, 1), (128, 128, 10, 0)])
    except svs.SvsxError as e:  # the record builder may refuse a spectrum like this one: then there is no inline-record kernel to compare
        assert "bound codes" in str(e) or "record" in str(e), e


@pytest.mark.parametrize("seed", range(6))
def test_two_query_fuzz(svs, oracle, seed):
    rng = np.random.default_rng(300 + seed)
    d = 96 if seed % 2 == 0 else 128
    metric = L2 if seed % 3 != 2 else IP
    n = int(rng.integers(300, 2500))
    xdt = np.float32 if seed % 4 < 2 else np.float16
    scale = float(10.0 ** rng.uniform(-2, 2))
    x = (deep_like(n, d, int(rng.integers(4, 40)), seed=400 + seed, normalize=False) * np.float32(scale) + np.float32(rng.uniform(-1, 1) * scale)).astype(xdt)
    q = (deep_like(37, d, 12, seed=500 + seed, normalize=False) * np.float32(scale)).astype(np.float32)
    R = int(rng.choice([16, 24, 40, 64]))
    g, e = oracle.build_vamana(metric, x, R=R, L=R + 16, alpha=1.2 if metric == L2 else 0.95)
    configs = [(int(w), int(w + rng.integers(0, 2) * rng.integers(0, 128 - w + 1)), int(min(w, rng.integers(1, 30))), int(rng.integers(0, 2))) for w in rng.integers(1, 129, 6)]
    _check(svs, oracle, metric, x, g, e, q, configs)


def test_auto_policy_picks_two_query_kernel_for_large_batches(svs, oracle, small_world):
    """policy 0: batches that exceed one resident generation of the one-query kernel (6 144 queries) go to the two-query kernel; the
    answers equal the one-query kernel's on all of them and the oracle's on a sample"""
    import torch

    sw = small_world
    rng = np.random.default_rng(9)
    q = np.ascontiguousarray(sw["q"][rng.integers(0, 64, 7001)] + 0.01 * rng.standard_normal((7001, 96)).astype(np.float32), np.float32)
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    idx.build_bound_codes(32)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(64), True)
    dq = torch.from_numpy(q).cuda()
    out = {}
    for policy in (0, 1):
        idx.set_kernel_policy(policy)
        ids = torch.empty((7001, 10), dtype=torch.int64, device="cuda")
        dd = torch.empty((7001, 10), dtype=torch.float32, device="cuda")
        idx.search_device(dq, 10, sp, d_ids64=ids, d_dists=dd)
        torch.cuda.synchronize()
        out[policy] = (ids.cpu().numpy(), dd.cpu().numpy(), idx.last_kernel_info())
    assert out[0][2].startswith("search_kernel_x2<") and out[1][2].startswith("search_kernel<"), (out[0][2], out[1][2])
    assert (out[0][0] == out[1][0]).all() and (_bits(out[0][1]) == _bits(out[1][1])).all()
    oi, od = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], q[:300], 10, 64, 64, visited_set=True)
    assert (out[0][0][:300] == oi.astype(np.int64)).all() and (_bits(out[0][1][:300]) == _bits(od)).all()
    idx.set_kernel_policy(0)
    small = idx.search(q[:100], 10, sp)  # a small batch under the auto policy: the one-query kernel
    assert idx.last_kernel_info().startswith("search_kernel<") and (small.indices == out[0][0][:100].astype(np.uint64)).all()
