"""Flat search is exact by proof, not by margin (svsx_flat.hip stages 4 and 5): the float16 shortlist is re-ranked in the reference's float32
arithmetic, every query's answer is then proven against the float16 rounding bounds, and a query that cannot be proven is recomputed by the
exhaustive float32 kernel.  Inputs here are built to defeat the shortlist; the answers still equal the oracle's exhaustive search bit for bit
(ids, distance bits, ties by lower id)."""
import numpy as np
import pytest

from conftest import deep_like

pytestmark = pytest.mark.gpu

F32, F16, I8, U8 = 0, 1, 2, 3
L2, IP, COS = 0, 1, 2


@pytest.fixture(scope="module")
def svs():
    import scalablevectorsearch_amd as svs

    return svs


def _fn(svs, metric):
    return {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]


def _search_counted(idx, q, k):
    import torch

    dq = torch.from_numpy(q).cuda()
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ids, dists = idx.flat_search_device(dq, k, recomputed=cnt)
    torch.cuda.synchronize()
    return ids.cpu().numpy().view(np.uint64), dists.cpu().numpy(), int(cnt.item())


def _assert_equal(got_ids, got_d, want_ids, want_d, what):
    assert (got_d.view(np.uint32) == want_d.view(np.uint32)).all(), what
    assert (got_ids == want_ids.astype(np.uint64)).all(), what


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("n", [6000, 70_000])
def test_clusters_below_float16_resolution(svs, oracle, metric, n):
    """four clusters of 300 rows each 1e-5 apart: their float16 keys coincide and a shortlist of 40 holds an arbitrary tenth of a cluster.
    The proof fails for the queries aimed at a cluster, the exhaustive pass answers them; queries elsewhere are proven and never recomputed.
    n = 6000: shortlist pipeline; n = 70 000: threshold pipeline, which re-ranks EVERY row under its threshold (about 16 shortlists' worth): a
    whole cluster fits and the proof holds without the exhaustive pass."""
    rng = np.random.default_rng(5)
    d, k = 96, 10
    x = deep_like(n, d, 12, seed=51, normalize=False)
    centre = x[:4].copy()
    for c in range(4):
        x[100 + 300 * c : 400 + 300 * c] = centre[c] + np.float32(1e-5) * rng.standard_normal((300, d)).astype(np.float32)
    aimed = (centre * np.float32(1.5 if metric == IP else 1.0) + np.float32(1e-3) * rng.standard_normal((4, d)).astype(np.float32)).astype(np.float32)
    plain = deep_like(60, d, 12, seed=52, normalize=False)
    idx = svs.FlatIndex(x, _fn(svs, metric))
    ids, dists, redone = _search_counted(idx, aimed, k)
    want_i, want_d = oracle.flat_search(metric, x, aimed, k)
    _assert_equal(ids, dists, want_i, want_d, (metric, n, "aimed"))
    if metric != IP:  # under inner product the best rows are the longest ones, wherever the query points
        assert redone == 4 if n == 6000 else redone <= 4, redone
        assert (((want_i >= 100) & (want_i < 1300)) | (want_i < 4)).all()  # the answers do come from inside the clusters (or are their centres)
    ids, dists, redone = _search_counted(idx, plain, k)
    want_i, want_d = oracle.flat_search(metric, x, plain, k)
    _assert_equal(ids, dists, want_i, want_d, (metric, n, "plain"))
    assert redone <= 3, redone  # ordinary queries are proven from the shortlist alone (unless a cluster happens to be their k-th neighbour)


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_rows_float16_cannot_hold(svs, oracle, metric):
    """rows beyond the float16 range (their shadow is +-inf) and rows below it (shadow zero: no direction under cosine) make every proof
    impossible: all queries take the exhaustive float32 pass and still equal the oracle."""
    rng = np.random.default_rng(9)
    x = rng.standard_normal((5000, 64)).astype(np.float32)
    x[17] *= np.float32(1.0e6)
    x[23] = np.float32(1.0e-12) * rng.standard_normal(64).astype(np.float32)
    q = rng.standard_normal((20, 64)).astype(np.float32)
    idx = svs.FlatIndex(x, _fn(svs, metric))
    ids, dists, redone = _search_counted(idx, q, 10)
    want_i, want_d = oracle.flat_search(metric, x, q, 10)
    _assert_equal(ids, dists, want_i, want_d, metric)
    assert redone == 20


@pytest.mark.parametrize("metric,d", [(L2, 256), (COS, 200), (IP, 768)])
def test_clusters_in_more_than_128_dimensions(svs, oracle, metric, d):
    """the chunked threshold kernel (128 < d <= 768) under the same attack: a cluster of 400 rows 1e-5 apart, queries aimed at it and elsewhere."""
    rng = np.random.default_rng(31)
    n = 66_000
    x = deep_like(n, d, 12, seed=81, normalize=False)
    x[500:900] = x[7] + np.float32(1e-5) * rng.standard_normal((400, d)).astype(np.float32)
    q = np.concatenate([x[7:8] * np.float32(1.5 if metric == IP else 1.0) + np.float32(1e-3) * rng.standard_normal((1, d)).astype(np.float32),
                        deep_like(40, d, 12, seed=82, normalize=False)]).astype(np.float32)
    idx = svs.FlatIndex(x, _fn(svs, metric))
    ids, dists, redone = _search_counted(idx, q, 10)
    want_i, want_d = oracle.flat_search(metric, x, q, 10)
    _assert_equal(ids, dists, want_i, want_d, (metric, d))
    assert redone <= 4, redone


@pytest.mark.parametrize("n", [3000, 70_000])
def test_queries_float16_cannot_hold(svs, oracle, n):
    """a query component beyond the float16 range turns the query's shadow into +-inf and every float16 key into NaN / inf: nothing can be
    proven for that query (an EMPTY shortlist must not read as 'every row was re-ranked'), the exhaustive pass answers it."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal((n, 32)).astype(np.float32)
    q = rng.standard_normal((6, 32)).astype(np.float32)
    q[1, 3] = np.float32(1.0e6)
    q[4] *= np.float32(3.0e5)
    for metric in (L2, IP, COS):
        idx = svs.FlatIndex(x, _fn(svs, metric))
        ids, dists, redone = _search_counted(idx, q, 10)
        want_i, want_d = oracle.flat_search(metric, x, q, 10)
        _assert_equal(ids, dists, want_i, want_d, (metric, n))
        assert 2 <= redone <= 3, redone


def test_exhaustive_pass_orders_ties_and_short_sets(svs, oracle):
    """duplicates of one row (equal float32 distances: lowest ids win) spread over the whole set, more of them than the shortlist holds; and a
    set smaller than k."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((4000, 48)).astype(np.float32)
    x[::7] = x[0]
    q = (x[0] + np.float32(0.01) * rng.standard_normal((5, 48))).astype(np.float32)
    idx = svs.FlatIndex(x, svs.DistanceL2())
    for k in (10, 100, 224):
        ids, dists, redone = _search_counted(idx, q, k)
        want_i, want_d = oracle.flat_search(L2, x, q, k)
        _assert_equal(ids, dists, want_i, want_d, k)
        assert redone == 5
    small = x[:6].copy()
    ids, dists, _ = _search_counted(svs.FlatIndex(small, svs.DistanceL2()), q, 6)
    want_i, want_d = oracle.flat_search(L2, small, q, 6)
    _assert_equal(ids, dists, want_i, want_d, "six rows")


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("code", [I8, U8])
def test_scalar_quantized_sets(svs, oracle, metric, code):
    """SQ data sets: under L2 the reference compresses the query as well (EuclideanCompressed, scalar.h:88-94), so the distance the proof bounds
    is measured from the query's decompressed code; clusters of identical codes force the exhaustive pass for the queries aimed at them."""
    rng = np.random.default_rng(13)
    xs = deep_like(8000, 64, 16, seed=61, normalize=False)
    xs[200:500] = xs[3] + np.float32(1e-4) * rng.standard_normal((300, 64)).astype(np.float32)
    qs = np.concatenate([deep_like(30, 64, 16, seed=62, normalize=False), xs[3:4] + np.float32(1e-3)]).astype(np.float32)
    codes, scale, bias = oracle.sq_compress(xs, code)
    idx = svs.FlatIndex(codes, _fn(svs, metric), sq=(scale, bias))
    ids, dists, redone = _search_counted(idx, qs, 10)
    want_i, want_d = oracle.flat_search(metric, codes, qs, 10, sq=(float(scale), float(bias)))
    _assert_equal(ids, dists, want_i, want_d, (metric, code))
    assert redone <= 4, redone


@pytest.mark.parametrize("ddt,qdt", [(F16, F16), (I8, I8), (U8, U8), (F16, F32)])
def test_other_element_types(svs, oracle, ddt, qdt):
    def conv(a, dt):
        if dt == F32:
            return a
        if dt == F16:
            return a.astype(np.float16)
        if dt == I8:
            return np.clip(np.round(a * 300), -128, 127).astype(np.int8)
        return np.clip(np.round(a * 300 + 128), 0, 255).astype(np.uint8)

    x = conv(deep_like(70_000, 100, 16, seed=71), ddt)
    x[1000:1200] = x[5]  # 200 copies: ties the shortlist cannot hold
    q = conv(deep_like(40, 100, 16, seed=72), qdt)
    q[0] = x[5].astype(q.dtype)
    for metric in (L2, COS):
        idx = svs.FlatIndex(x, _fn(svs, metric))
        ids, dists, redone = _search_counted(idx, q, 10)
        want_i, want_d = oracle.flat_search(metric, x, q, 10)
        _assert_equal(ids, dists, want_i, want_d, (ddt, qdt, metric))
        assert redone <= 3, redone  # 70 000 rows: the threshold pipeline re-ranks every row under its threshold, the 200 copies included


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("d", [96, 128, 80])
def test_everything_that_matters_sits_in_the_last_sixteen_dimensions(svs, oracle, metric, d):
    """All rows share their first d - 16 coordinates; who is near whom is decided by the last 16 alone -- the LAST matrix instruction of a
    block's accumulation chain.  A threshold test that reads the accumulators before that instruction has retired (the matrix core's results
    are not interlocked against the vector ALU on gfx950; svsx_flat.hip pads its inline-assembly screen with wait states) sees every row at the
    same distance and drops true neighbours; the proof of stage 4 cannot notice a row that was never appended.  256 queries = one full
    four-wave workgroup, 70 000 rows = the threshold pipeline, a ragged last block."""
    n, nq, k = 70_001, 256, 10
    rng = np.random.default_rng(77 + d + metric)
    x = np.empty((n, d), np.float32)
    x[:, : d - 16] = 0.05 * rng.standard_normal(d - 16).astype(np.float32)  # small: the float16 proof keeps its margin
    x[:, d - 16 :] = rng.standard_normal((n, 16)).astype(np.float32)
    q = np.empty((nq, d), np.float32)
    q[:, : d - 16] = 0.05 * rng.standard_normal(d - 16).astype(np.float32)
    q[:, d - 16 :] = 2.0 * rng.standard_normal((nq, 16)).astype(np.float32)
    idx = svs.FlatIndex(x, _fn(svs, metric))
    ids, dists, redone = _search_counted(idx, q, k)
    want_i, want_d = oracle.flat_search(metric, x, q, k)
    _assert_equal(ids, dists, want_i, want_d, (metric, d))
    assert redone <= nq // 4, redone  # the answers come from the threshold pipeline, not from the exhaustive fallback
