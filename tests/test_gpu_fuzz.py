"""Randomised parity campaign: many small random indexes (random graphs with degree 0..R, tie-heavy or continuous data,
all AVX-512 type pairs, all metrics, split buffers, k > capacity, visited filter sizes, several entry points) searched on
the GPU and by the oracle; ids, distance bits and tracker counts must agree exactly."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32, F16, I8, U8 = 0, 1, 2, 3
L2, IP, COS = 0, 1, 2
PAIRS = [(F32, F32), (F32, F16), (F16, F32), (F16, F16), (F32, I8), (F32, U8), (I8, I8), (U8, U8)]


def _random_graph(rng, n, R):
    g = np.zeros((n, R + 1), np.uint32)
    for i in range(n):
        deg = int(rng.integers(0, R + 1)) if rng.random() < 0.9 else 0
        nb = rng.choice(n, size=min(deg, n), replace=False)
        g[i, 0] = len(nb)
        g[i, 1 : 1 + len(nb)] = nb
    # make the graph reasonably connected from node 0: a ring through slot 1
    for i in range(n):
        if g[i, 0] == 0:
            g[i, 0] = 1
        g[i, 1] = (i + 1) % n
        # keep ids unique within the row
        row = g[i, 1 : 1 + g[i, 0]]
        _, first = np.unique(row, return_index=True)
        row = row[np.sort(first)]
        g[i, 0] = len(row)
        g[i, 1 : 1 + len(row)] = row
    return g


def _data(rng, dt, n, d, ties):
    if dt == F32:
        return rng.integers(-2, 3, (n, d)).astype(np.float32) if ties else rng.standard_normal((n, d)).astype(np.float32)
    if dt == F16:
        return rng.integers(-2, 3, (n, d)).astype(np.float16) if ties else rng.standard_normal((n, d)).astype(np.float16)
    if dt == I8:
        return rng.integers(-3, 4, (n, d)).astype(np.int8) if ties else rng.integers(-128, 128, (n, d)).astype(np.int8)
    return rng.integers(0, 5, (n, d)).astype(np.uint8) if ties else rng.integers(0, 256, (n, d)).astype(np.uint8)


@pytest.mark.parametrize("seed", range(48))
def test_random_index_parity(seed, oracle):
    import scalablevectorsearch_amd as svs

    rng = np.random.default_rng(1000 + seed)
    qdt, ddt = PAIRS[seed % len(PAIRS)]
    metric = [L2, IP, COS][(seed // len(PAIRS)) % 3]
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]
    n = int(rng.integers(20, 900))
    d = int(rng.choice([1, 3, 16, 31, 64, 65, 96, 100, 128, 130, 200]))
    R = int(rng.choice([1, 4, 17, 32, 64, 70, 128]))
    ties = bool(rng.random() < 0.4)
    x = _data(rng, ddt, n, d, ties)
    if ties and rng.random() < 0.5:
        x[n // 2 :] = x[: n - n // 2]  # exact duplicate vectors
    q = _data(rng, qdt, 24, d, ties)
    if metric == COS:  # avoid 0/0
        x[(x.astype(np.float32) ** 2).sum(1) == 0, 0] = 1
        q[(q.astype(np.float32) ** 2).sum(1) == 0, 0] = 1
    g = _random_graph(rng, n, min(R, n - 1) if n > 1 else 1)
    n_entry = int(rng.choice([1, 1, 1, 2, 5]))
    entry = rng.choice(n, size=min(n_entry, n), replace=False).astype(np.uint32)
    idx = svs.VamanaIndex(g, x, fn, entry)
    for _ in range(3):
        cap = int(rng.choice([1, 2, 7, 33, 64, 65, 130, 257]))
        win = int(rng.integers(1, cap + 1))
        k = int(rng.choice([1, 5, 10, cap, cap + 3]))
        vs = int(rng.choice([0, 0, 1, 64, 128, 4096]))
        res, st = idx.search(q, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(win, cap), search_buffer_visited_set=vs), stats=True)
        ids, dists, ost = oracle.search(metric, x, g, entry, q, k, win, cap, stats=True)
        want = np.where(ids == 0xFFFFFFFF, np.uint64(2**64 - 1), ids.astype(np.uint64))
        ctx = dict(seed=seed, n=n, d=d, R=R, ties=ties, metric=metric, qdt=qdt, ddt=ddt, win=win, cap=cap, k=k, vs=vs, n_entry=len(entry))
        assert (res.indices == want).all(), ctx
        same = (res.distances.view(np.uint32) == dists.view(np.uint32)) | (np.isnan(res.distances) & np.isnan(dists))
        assert same.all(), ctx
        assert (st[:, :2] == ost[:, :2]).all(), ctx


@pytest.mark.parametrize("seed", range(12))
def test_random_sq_index_parity(seed, oracle):
    """Same campaign for SQDataset<int8/uint8> indexes with f32 / f16 queries (quantization/scalar/scalar.h)."""
    import scalablevectorsearch_amd as svs

    rng = np.random.default_rng(5000 + seed)
    metric = [L2, IP, COS][seed % 3]
    code = [I8, U8][(seed // 3) % 2]
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]
    n = int(rng.integers(50, 700))
    d = int(rng.choice([8, 33, 64, 100, 128, 768]))
    xf = (rng.standard_normal((n, d)) * rng.uniform(0.1, 5)).astype(np.float32) + np.float32(rng.uniform(-2, 2))
    codes, scale, bias = oracle.sq_compress(xf, code)
    q = (rng.standard_normal((16, d)) * 2).astype(np.float32)
    if seed % 2:
        q = q.astype(np.float16)
    g = _random_graph(rng, n, int(rng.choice([8, 32, 64])))
    entry = np.array([int(rng.integers(0, n))], np.uint32)
    idx = svs.VamanaIndex(g, codes, fn, entry, sq=(scale, bias))
    for (win, cap, k, vs) in ((5, 9, 4, 0), (40, 40, 10, 1), (64, 100, 70, 256)):
        res = idx.search(q, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(win, cap), search_buffer_visited_set=vs))
        ids, dists = oracle.search(metric, codes, g, entry, q, k, win, cap, sq=(scale, bias))
        want = np.where(ids == 0xFFFFFFFF, np.uint64(2**64 - 1), ids.astype(np.uint64))
        ctx = dict(seed=seed, n=n, d=d, metric=metric, code=code, win=win, cap=cap, k=k)
        assert (res.indices == want).all(), ctx
        assert ((res.distances.view(np.uint32) == dists.view(np.uint32)) | (np.isnan(res.distances) & np.isnan(dists))).all(), ctx
