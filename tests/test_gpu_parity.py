"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit for bit."""
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


def _gen(rng, dt, n, d):
    if dt == F32:
        return rng.standard_normal((n, d)).astype(np.float32)
    if dt == F16:
        return rng.standard_normal((n, d)).astype(np.float16)
    if dt == I8:
        return rng.integers(-128, 128, (n, d)).astype(np.int8)
    return rng.integers(0, 256, (n, d)).astype(np.uint8)


PAIRS = [(F32, F32), (F32, F16), (F16, F32), (F16, F16), (F32, I8), (F32, U8), (I8, I8), (U8, U8)]


@pytest.mark.parametrize("d", [1, 7, 16, 17, 63, 64, 65, 96, 100, 128, 160, 223, 768])
def test_distance_bits_match_oracle(svs, oracle, d):
    """distance plugin surface: every AVX-512 specialised type pair x metric, incl. ragged tails (dims of
    tests/svs/core/distances/*.cpp: 160 even, 223 ragged) -- bit-exact, stricter than the reference's Approx."""
    rng = np.random.default_rng(d)
    for qa, qb in PAIRS:
        a = _gen(rng, qa, 9, d)
        b = _gen(rng, qb, 70, d)
        for m, f in ((L2, svs.DistanceL2()), (IP, svs.DistanceIP()), (COS, svs.DistanceCosineSimilarity())):
            got = svs.compute(f, a, b)
            want = oracle.distance_batch(m, a, b)
            bad = got.view(np.uint32) != want.view(np.uint32)
            # NaN (0/0 cosine) compares by bit pattern too, but allow any NaN payload
            bad &= ~(np.isnan(got) & np.isnan(want))
            assert not bad.any(), (d, qa, qb, m, got[bad][:4], want[bad][:4])


# the (query, data) pairs utils/search_index.cpp:147-162 offers that have NO AVX-512 specialisation: the reference answers them with its scalar
# loops (generic_l2 / generic_ip / generic_cosine_similarity); round 5 builds them (QM_SCALAR) against the oracle's restatement of those loops
# (one fma per element, float16 elements through the reference's scalar conversion: pinned by a run against oracle/ref_scalar_loops.cpp, tests/test_oracle_vs_reference.py)
SCALAR_PAIRS = [(I8, U8), (I8, F32), (I8, F16), (U8, I8), (U8, F32), (U8, F16)]


@pytest.mark.parametrize("d", [1, 16, 17, 96, 100, 223])
def test_scalar_loop_pairs_distance_bits_match_oracle(svs, oracle, d):
    rng = np.random.default_rng(1000 + d)
    for qa, qb in SCALAR_PAIRS:
        a = _gen(rng, qa, 5, d)
        b = _gen(rng, qb, 130, d)   # more than one 64-row pass
        for m, f in ((L2, svs.DistanceL2()), (IP, svs.DistanceIP()), (COS, svs.DistanceCosineSimilarity())):
            got = svs.compute(f, a, b)
            want = oracle.distance_batch(m, a, b)
            bad = got.view(np.uint32) != want.view(np.uint32)
            bad &= ~(np.isnan(got) & np.isnan(want))
            assert not bad.any(), (d, qa, qb, m, got[bad][:4], want[bad][:4])


def _spice_f16(a):
    """float16 array with denormals, the largest finite value, an infinity and a NaN pattern written into its first rows"""
    b = a.view(np.uint16).copy()
    d = b.shape[1]
    b[0, 0] = 0x0001
    b[0, d - 1] = 0x83FF
    if b.shape[0] > 3 and d > 2:
        b[1, 0] = 0x7BFF
        b[2, 1] = 0x7C00
        b[3, 2] = 0xFE01
    return b.view(np.float16)


@pytest.mark.parametrize("d", [16, 96, 100])
def test_float16_scalar_conversion_cases(svs, oracle, small_world, d):
    """Where the reference widens float16 through Float16::operator float (lib/float16.h:45-52: denormals flushed, inf / NaN patterns finite)
    and where through vcvtph2ps (exact): scalar-loop pairs with float16 rows, the cosine norm of a float16 query, float16 queries of a
    scalar-quantized index.  The oracle is pinned on exactly these inputs against the reference's own class (tests/test_oracle_vs_reference.py)."""
    rng = np.random.default_rng(4200 + d)
    x16 = _spice_f16((rng.standard_normal((70, d)) * 1e-3).astype(np.float16))
    q16 = _spice_f16((rng.standard_normal((6, d)) * 1e-3).astype(np.float16))
    fns = ((L2, svs.DistanceL2()), (IP, svs.DistanceIP()), (COS, svs.DistanceCosineSimilarity()))
    for qa in (I8, U8):  # generic_* loops over float16 rows
        a = _gen(rng, qa, 5, d)
        for m, f in fns:
            got, want = svs.compute(f, a, x16), oracle.distance_batch(m, a, x16)
            assert (((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))).all(), (d, qa, m)
    for xb in (x16, x16.astype(np.float32)):  # SIMD kernels, float16 queries: exact loads, scalar norm
        for m, f in fns:
            got, want = svs.compute(f, q16, xb), oracle.distance_batch(m, q16, xb)
            assert (((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))).all(), (d, m)
    if d == 96:  # scalar-quantized index searched with float16 queries (fix_argument widens them element by element)
        sw = small_world
        q = _spice_f16((sw["q"][:16] * 1e-3).astype(np.float16))
        for code in (I8, U8):
            codes, scale, bias = oracle.sq_compress(sw["x"], code)
            for m, f in fns:
                res, st, ids, dists, ost = _search_both(svs, oracle, m, f, codes, sw["graph"], sw["entry"], q, 10, 24, 24, sq=(scale, bias))
                assert (res.indices == ids.astype(np.uint64)).all(), (code, m)
                assert (((res.distances.view(np.uint32) == dists.view(np.uint32)) | (np.isnan(res.distances) & np.isnan(dists)))).all(), (code, m)


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("qdt,ddt", SCALAR_PAIRS)
def test_search_scalar_loop_pairs(svs, oracle, small_world, metric, qdt, ddt):
    """graph search, BatchIterator and get_distance of the six scalar-loop pairs: ids, distance bits and tracker counts equal to the oracle;
    svsx_last_kernel_info names the scalar loop and says that its parity is unpinned"""
    sw = small_world
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]

    def conv(a, dt):
        if dt == F32:
            return a
        if dt == F16:
            return a.astype(np.float16)
        if dt == I8:
            return np.clip(np.round(a * 400), -128, 127).astype(np.int8)
        return np.clip(np.round(a * 400 + 128), 0, 255).astype(np.uint8)

    x = conv(sw["x"], ddt)
    q = conv(sw["q"], qdt)
    idx = svs.VamanaIndex(sw["graph"], x, fn, sw["entry"])
    for (w, c) in [(16, 16), (12, 30), (260, 260)]:
        res, st = idx.search(q, 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c)), stats=True)
        ids, dists, ost = oracle.search(metric, x, sw["graph"], sw["entry"], q, 10, w, c, stats=True)
        assert (res.indices == ids.astype(np.uint64)).all(), (metric, qdt, ddt, w)
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
        assert (st[:, :2] == ost[:, :2]).all()
        assert "qmode=5" in idx.last_kernel_info() and "scalar loop" in idx.last_kernel_info()
    res = idx.search(q[:4], 3, svs.VamanaSearchParameters(svs.SearchBufferConfig(16)))
    for i in range(4):
        for j in range(3):
            assert np.float32(idx.get_distance(int(res.indices[i, j]), q[i])).view(np.uint32) == res.distances[i, j].view(np.uint32)
    with pytest.raises(svs.SvsxError) as e:   # the exhaustive search keeps to the pairs with a SIMD kernel
        svs.FlatIndex(x, fn).search(q[:2], 3)
    assert e.value.code == 3


def _search_both(svs, oracle, metric, fn, x, graph, entry, q, k, w, c, sq=None):
    idx = svs.VamanaIndex(graph, x, fn, entry, sq=sq)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c))
    res, st = idx.search(q, k, sp, stats=True)
    ids, dists, ost = oracle.search(metric, x, graph, entry, q, k, w, c, sq=sq, stats=True)
    return res, st, ids, dists, ost


@pytest.mark.parametrize("w,c,k", [(10, 10, 10), (32, 32, 10), (20, 40, 10), (128, 128, 10), (5, 5, 10), (64, 64, 64), (1, 1, 1),
                                   (129, 129, 10), (200, 256, 20), (300, 300, 10), (384, 384, 10), (260, 385, 10), (512, 512, 50), (513, 600, 10), (1000, 1024, 100), (1025, 1100, 10)])
def test_search_f32_l2_bit_exact(svs, oracle, small_world, w, c, k):
    sw = small_world
    res, st, ids, dists, ost = _search_both(svs, oracle, L2, svs.DistanceL2(), sw["x"], sw["graph"], sw["entry"], sw["q"], k, w, c)
    assert (res.indices == ids.astype(np.uint64)).all()
    assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
    # tracker counts (greedy_search.h:165): expansions and neighbour slots identical
    assert (st[:, 0] == ost[:, 0]).all() and (st[:, 1] == ost[:, 1]).all()


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("qdt,ddt", PAIRS)
def test_search_all_type_pairs(svs, oracle, small_world, metric, qdt, ddt):
    sw = small_world
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]

    def conv(a, dt):
        if dt == F32:
            return a
        if dt == F16:
            return a.astype(np.float16)
        if dt == I8:
            return np.clip(np.round(a * 400), -128, 127).astype(np.int8)
        return np.clip(np.round(a * 400 + 128), 0, 255).astype(np.uint8)

    x = conv(sw["x"], ddt)
    q = conv(sw["q"], qdt)
    for (w, c) in [(16, 16), (12, 30), (260, 260), (700, 700)]:  # the last two: 4- / 8- and 16-entries-per-lane merge instantiations
        res, st, ids, dists, ost = _search_both(svs, oracle, metric, fn, x, sw["graph"], sw["entry"], q, 10, w, c)
        assert (res.indices == ids.astype(np.uint64)).all(), (metric, qdt, ddt)
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
        assert (st[:, :2] == ost[:, :2]).all()


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("code", [I8, U8])
def test_search_scalar_quantized(svs, oracle, small_world, metric, code):
    """SQDataset<int8/uint8> + compressed distances (quantization/scalar/scalar.h:58-195)."""
    sw = small_world
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]
    codes, scale, bias = oracle.sq_compress(sw["x"], code)
    for q in (sw["q"], sw["q"].astype(np.float16)):
        res, st, ids, dists, ost = _search_both(svs, oracle, metric, fn, codes, sw["graph"], sw["entry"], q, 10, 24, 24, sq=(scale, bias))
        assert (res.indices == ids.astype(np.uint64)).all()
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()


@pytest.mark.parametrize("code", [I8, U8])
def test_sq_cosine_row_norms_dropped_and_refreshed(svs, oracle, small_world, code):
    """svsx_index_drop_row_norms / _refresh_row_norms (include/svsx.h): without the precomputed norms the kernel takes the norm of the decompressed
    row per candidate (scalar.h:175-184) -- same ids, same distance bits -- and a refresh brings the fast path back with the same answer."""
    sw = small_world
    codes, scale, bias = oracle.sq_compress(sw["x"], code)
    idx = svs.VamanaIndex(sw["graph"], codes, svs.DistanceCosineSimilarity(), sw["entry"], sq=(scale, bias))
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(24, 24))
    ids, dists, _ = oracle.search(COS, codes, sw["graph"], sw["entry"], sw["q"], 10, 24, 24, sq=(scale, bias), stats=True)
    for step in ("norms", "dropped", "refreshed"):
        if step == "dropped":
            idx.drop_row_norms()
        if step == "refreshed":
            idx.refresh_row_norms()
        res = idx.search(sw["q"], 10, sp)
        assert (res.indices == ids.astype(np.uint64)).all(), step
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all(), step
        assert ("general=1" in idx.last_kernel_info()) == (step == "dropped"), (step, idx.last_kernel_info())


@pytest.mark.parametrize("d", [100, 223, 40])
def test_search_ragged_dims(svs, oracle, d):
    from oracle.oracle import L2 as OL2

    x = deep_like(600, d, 16, seed=3, normalize=False)
    q = deep_like(32, d, 16, seed=4, normalize=False)
    graph, entry = oracle.build_vamana(OL2, x, R=24, L=32)
    for fn, m in ((svs.DistanceL2(), L2), (svs.DistanceCosineSimilarity(), COS)):
        res, st, ids, dists, ost = _search_both(svs, oracle, m, fn, x, graph, entry, q, 8, 20, 20)
        assert (res.indices == ids.astype(np.uint64)).all()
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()


def test_search_wide_graph_and_duplicates(svs, oracle):
    """max degree 128 (two adjacency chunks), heavy distance ties (integer grid data) and duplicate vectors."""
    from oracle.oracle import L2 as OL2

    rng = np.random.default_rng(5)
    x = rng.integers(-2, 3, (800, 32)).astype(np.float32)
    x[100:140] = x[0:40]  # exact duplicates -> equal distances, different ids
    q = rng.integers(-2, 3, (40, 32)).astype(np.float32)
    graph, entry = oracle.build_vamana(OL2, x, R=128, L=140)
    assert graph[:, 0].max() > 64
    for (w, c) in [(30, 30), (150, 150), (10, 64)]:
        res, st, ids, dists, ost = _search_both(svs, oracle, L2, svs.DistanceL2(), x, graph, entry, q, 10, w, c)
        assert (res.indices == ids.astype(np.uint64)).all()
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
        assert (st[:, :2] == ost[:, :2]).all()


def test_fewer_than_k_results_are_padded(svs, oracle):
    """vamana_index_impl.h:205-209: id = all-ones, dist = +inf."""
    x = deep_like(6, 16, 4, seed=6)
    graph = np.zeros((6, 4), np.uint32)
    graph[0, :3] = (2, 1, 2)
    graph[1, :2] = (1, 0)
    graph[2, :2] = (1, 0)  # nodes 3..5 unreachable
    q = deep_like(3, 16, 4, seed=7)
    idx = svs.VamanaIndex(graph, x, svs.DistanceL2(), [0])
    res = idx.search(q, 5, svs.VamanaSearchParameters(svs.SearchBufferConfig(8)))
    ids, dists = oracle.search(L2, x, graph, np.array([0], np.uint32), q, 5, 8)
    assert (res.indices[:, :3] == ids[:, :3]).all()
    assert (res.indices[:, 3:] == np.uint64(2**64 - 1)).all()
    assert np.isposinf(res.distances[:, 3:]).all()
    assert (res.distances[:, :3].view(np.uint32) == dists[:, :3].view(np.uint32)).all()


def test_argument_errors(svs, small_world):
    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    with pytest.raises(svs.SvsxError) as e:
        idx.search(sw["q"], 0)
    assert e.value.code == 2 and "k must be greater than 0" in e.value.message
    with pytest.raises(svs.SvsxError):
        svs.SearchBufferConfig(10, 9)
    with pytest.raises(svs.SvsxError) as e:  # a scalar-quantized index takes float queries only (scalar.h:58-195); plain int8 queries are a scalar-loop pair (served)
        codes = np.clip(np.round(sw["x"] * 400), -128, 127).astype(np.int8)
        svs.VamanaIndex(sw["graph"], codes, svs.DistanceL2(), sw["entry"], sq=(0.0025, 0.0)).search(sw["q"].astype(np.int8), 5)
    assert e.value.code == 2
    # empty batch is a no-op (vamana_index_impl.h:125-127)
    r = idx.search(np.zeros((0, 96), np.float32), 5)
    assert r.indices.shape == (0, 5)


def test_multiple_entry_points(svs, oracle, small_world):
    sw = small_world
    entry = np.array([int(sw["entry"][0]), 3, 700], np.uint32)
    res, st, ids, dists, ost = _search_both(svs, oracle, L2, svs.DistanceL2(), sw["x"], sw["graph"], entry, sw["q"], 10, 20, 20)
    assert (res.indices == ids.astype(np.uint64)).all()


# ---- committed golden vectors (generated from the reference's own code, tests/golden/make_golden.py) -----------
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {F32: "f32", F16: "f16", I8: "i8", U8: "u8"}


def test_golden_distances_on_gpu(svs):
    g = np.load(os.path.join(GOLD, "distances.npz"))
    fns = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}
    for d in (96, 100, 223):
        for qa, qb in PAIRS:
            for m in (L2, IP, COS):
                got = svs.compute(fns[m], g[f"a_{NAMES[qa]}_{d}"], g[f"b_{NAMES[qb]}_{d}"])
                want = g[f"dist_{NAMES[qa]}_{NAMES[qb]}_{m}_{d}"]
                assert ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))).all(), (d, qa, qb, m)


def test_golden_search_on_gpu(svs):
    g = np.load(os.path.join(GOLD, "search.npz"))
    fns = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}
    for qa, qb in PAIRS:
        for m in (L2, IP, COS):
            idx = svs.VamanaIndex(g["graph"], g[f"x_{NAMES[qb]}"], fns[m], g["entry"])
            for ci, (w, c, k) in enumerate(g["configs"]):
                res, st = idx.search(g[f"q_{NAMES[qa]}"], int(k), svs.VamanaSearchParameters(svs.SearchBufferConfig(int(w), int(c))), stats=True)
                key = f"{NAMES[qa]}_{NAMES[qb]}_{m}_{ci}"
                assert (res.indices == g["ids_" + key].astype(np.uint64)).all(), key
                assert (res.distances.view(np.uint32) == g["dists_" + key].view(np.uint32)).all(), key
                assert (st[:, :2].astype(np.uint64) == g["stats_" + key]).all(), key


def test_device_resident_search_and_u32_ids(svs, oracle, small_world):
    """svsx_search_device: queries/results in HBM, enqueued on torch's current stream; 32- and 64-bit ids agree."""
    import torch

    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    q = torch.from_numpy(sw["q"]).cuda()
    ids64 = torch.empty((64, 10), dtype=torch.int64, device="cuda")
    ids32 = torch.empty((64, 10), dtype=torch.int32, device="cuda")
    dists = torch.empty((64, 10), dtype=torch.float32, device="cuda")
    idx.search_device(q, 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(20, 32)), d_ids64=ids64, d_ids32=ids32, d_dists=dists)
    torch.cuda.synchronize()
    ids, d = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"], 10, 20, 32)
    assert (ids64.cpu().numpy() == ids).all() and (ids32.cpu().numpy() == ids.astype(np.int32)).all()
    assert (dists.cpu().numpy().view(np.uint32) == d.view(np.uint32)).all()


def test_full_size_properties(svs):
    """BASELINE-size invariants that do not need the oracle: 200k x 96 index, 4096 queries, window 128:
    rows sorted best-first, ids unique and in range, result independent of batch composition (idempotence /
    permutation of the query batch), distances recomputable from the ids."""
    import torch

    from bench import brute_force_topk, deep_like_device
    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device

    dev = torch.device("cuda", 0)
    n, d, nq, k = 200_000, 96, 4096, 10
    x = deep_like_device(n, d, 24, 1, dev)
    q = deep_like_device(nq, d, 24, 2, dev)
    graph, entry = build_vamana_device(x, n, d, svs.DistanceL2(), VamanaBuildParameters(graph_max_degree=32, window_size=64, prune_to=28), device=0)
    idx = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=dev), n, d, np.float32)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(128))
    ids = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dist = torch.empty((nq, k), dtype=torch.float32, device=dev)
    idx.search_device(q, k, sp, d_ids64=ids, d_dists=dist)
    perm = torch.randperm(nq, device=dev)
    ids2 = torch.empty_like(ids)
    dist2 = torch.empty_like(dist)
    idx.search_device(q[perm].contiguous(), k, sp, d_ids64=ids2, d_dists=dist2)
    torch.cuda.synchronize()
    assert (ids[perm] == ids2).all() and (dist[perm] == dist2).all()
    assert (dist[:, 1:] >= dist[:, :-1]).all()
    assert int(ids.min()) >= 0 and int(ids.max()) < n
    assert (torch.sort(ids, 1).values[:, 1:] != torch.sort(ids, 1).values[:, :-1]).all()
    recomputed = ((x[ids] - q[:, None, :]) ** 2).sum(-1)
    assert torch.allclose(recomputed, dist, rtol=1e-4, atol=1e-6)
    gt = brute_force_topk(x, q, k)
    recall = (ids[:, :, None] == gt[:, None, :]).any(2).float().mean().item()
    assert recall > 0.95, recall


def test_shard_merge_kernel_matches_rule(svs):
    """svsx_merge_topk_device against the merge rule checker (ties -> lower shard, padding skipped), L2 and IP."""
    import torch

    from oracle.oracle import merge_topk_ref
    from scalablevectorsearch_amd.sharded import hip_merge

    rng = np.random.default_rng(9)
    S, nq, k = 8, 300, 10
    for metric in (L2, IP):
        d = rng.integers(0, 40, (S, nq, k)).astype(np.float32)  # heavy ties across shards
        d.sort(axis=2)
        if metric != L2:
            d = d[:, :, ::-1].copy()
        i = rng.integers(0, 1 << 40, (S, nq, k)).astype(np.int64)
        i[rng.random((S, nq)) < 0.2, 7:] = -1  # some shards return fewer than k
        d[i == -1] = np.inf
        gd, gi = hip_merge(metric, torch.from_numpy(d).cuda(), torch.from_numpy(i).cuda(), k)
        torch.cuda.synchronize()
        wd, wi = merge_topk_ref(metric, d, i, k)
        assert (gi.cpu().numpy() == wi).all()
        assert (gd.cpu().numpy().view(np.uint32) == wd.view(np.uint32)).all()


@pytest.mark.parametrize("entries", [1, 64, 256, 4096])
def test_visited_filter_is_result_neutral(svs, oracle, small_world, entries):
    """search_buffer_visited_set (greedy_search.h:191-193, filter.h): same ids, distance bits, expansions and
    neighbour slots; only the number of distance evaluations may drop."""
    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    for (w, c) in [(16, 16), (48, 64), (128, 128)]:
        off, st0 = idx.search(sw["q"], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c)), stats=True)
        on, st1 = idx.search(sw["q"], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c), search_buffer_visited_set=entries), stats=True)
        assert (off.indices == on.indices).all() and (off.distances.view(np.uint32) == on.distances.view(np.uint32)).all()
        assert (st0[:, :2] == st1[:, :2]).all()
        assert (st1[:, 2] <= st0[:, 2]).all() and st1[:, 2].sum() < st0[:, 2].sum()
        ids, dists = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"], 10, w, c, visited_set=True)
        assert (on.indices == ids.astype(np.uint64)).all()


@pytest.mark.parametrize("metric,alpha", [(L2, 1.2), (L2, 1.0), (IP, 0.95), (COS, 0.95)])
def test_prune_kernel_matches_oracle(svs, oracle, small_world, metric, alpha):
    """svsx_prune_device (index/vamana/prune.h:100-246) against the C oracle's prune on identical pools."""
    import torch

    sw = small_world
    x = sw["x"]
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]
    idx = svs.VamanaIndex(sw["graph"], x, fn, sw["entry"])
    rng = np.random.default_rng(4)
    B, P, R = 48, 70, 16
    nodes = rng.integers(0, 1500, B).astype(np.int32)
    pools = np.full((B, P), -1, np.int32)
    dists = np.zeros((B, P), np.float32)
    for b in range(B):
        npool = int(rng.integers(1, P + 1))
        ids = rng.choice(1500, npool, replace=False).astype(np.int32)
        if b % 5 == 0:
            ids[0] = nodes[b]  # the vertex itself may be in its pool (excluded by the prune)
        dd = oracle.distance_batch(metric, x[nodes[b] : nodes[b] + 1], x[ids])[0]
        order = np.lexsort((ids, -dd if metric != L2 else dd))
        pools[b, :npool], dists[b, :npool] = ids[order], dd[order]
        dists[b, npool:] = -np.inf if metric != L2 else np.inf
    out = idx.prune_device(torch.from_numpy(nodes).cuda(), torch.from_numpy(pools).cuda(), torch.from_numpy(dists).cuda(), R, alpha, 16)
    torch.cuda.synchronize()
    out = out.cpu().numpy()
    for b in range(B):
        npool = int((pools[b] >= 0).sum())
        want = oracle.prune(metric, x, nodes[b], pools[b, :npool], dists[b, :npool], R, alpha)
        got = out[b][out[b] >= 0]
        assert (got == want.astype(np.int32)).all(), (b, got, want)


def test_calibrate_reaches_target_recall(svs, oracle, small_world):
    """calibrate(): smallest window reaching the target on the GPU index; one less misses it (calibrate.h:286-300)."""
    from scalablevectorsearch_amd.calibrate import calibrate

    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    gt, _ = oracle.flat_search(L2, sw["x"], sw["q"], 10)
    best = calibrate(idx, sw["q"], gt, 10, 0.95)
    w = best.buffer_config.search_window_size
    assert idx.get_search_parameters() == best
    r = svs.k_recall_at_n(gt, idx.search(sw["q"], 10).indices, 10, 10)
    assert r >= 0.95
    if w > 10:
        r1 = svs.k_recall_at_n(gt, idx.search(sw["q"], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(w - 1))).indices, 10, 10)
        assert r1 < 0.95


def test_config1_plumbing_shape(svs, oracle):
    """BASELINE config 1 / SURVEY 8d C1: 10 000 x 128 f32 L2, graph with the build parameters of
    data/test_dataset/vamana_config.toml:19-27 (R=128, L=200, alpha=1.2, C=1000, prune_to=128; built here by the GPU
    builder because the reference's base vectors/graph are missing blobs), 1 000 queries searched as ONE batch and as
    a batch=1 loop at windows {10, 20, 50, 100}: ids and distance bits equal the oracle's either way."""
    import torch

    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs

    x = deep_like(10_000, 128, 32, seed=1, normalize=False)
    q = deep_like(1_000, 128, 32, seed=2, normalize=False)
    dx = torch.from_numpy(x).cuda()
    bp = VamanaBuildParameters(alpha=1.2, graph_max_degree=128, window_size=200, max_candidate_pool_size=1000, prune_to=128)
    graph, entry = build_vamana_device(dx, 10_000, 128, svs.DistanceL2(), bp)
    g = device_graph_to_svs(graph, 128)
    assert g[:, 0].max() <= 128 and g[:, 0].mean() > 32
    idx = svs.VamanaIndex(g, x, svs.DistanceL2(), [entry])
    gt, _ = oracle.flat_search(L2, x, q[:100], 10)
    for w in (10, 20, 50, 100):
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(w))
        res = idx.search(q, 10, sp)
        ids, dists = oracle.search(L2, x, g, np.array([entry], np.uint32), q, 10, w)
        assert (res.indices == ids.astype(np.uint64)).all() and (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
        for i in range(0, 1000, 97):  # batch = 1
            one = idx.search(q[i : i + 1], 10, sp)
            assert (one.indices[0] == res.indices[i]).all() and (one.distances[0] == res.distances[i]).all()
    assert oracle.recall(gt, idx.search(q[:100], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(100))).indices.astype(np.uint32), 10, 10) > 0.95


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("ddt,qdt", [(F32, F32), (F16, F32), (F16, F16), (I8, F32), (I8, I8), (U8, U8)])
def test_flat_search_matches_exhaustive_oracle(svs, oracle, metric, ddt, qdt):
    """FlatIndex::search (index/flat/flat.h:322-460): MFMA f16 shortlist + exact re-rank == the oracle's exhaustive
    search (ids and distance bits; ties by lower id)."""
    rng = np.random.default_rng(17)
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]

    def conv(a, dt):
        if dt == F32:
            return a
        if dt == F16:
            return a.astype(np.float16)
        if dt == I8:
            return np.clip(np.round(a * 300), -128, 127).astype(np.int8)
        return np.clip(np.round(a * 300 + 128), 0, 255).astype(np.uint8)

    x = conv(deep_like(3000, 100, 16, seed=5), ddt)
    q = conv(deep_like(70, 100, 16, seed=6), qdt)
    idx = svs.FlatIndex(x, fn)
    for k in (1, 10, 32):
        res = idx.search(q, k)
        ids, dists = oracle.flat_search(metric, x, q, k)
        # exact by proof (svsx_flat.hip stage 4; tests/test_gpu_flat_exact.py): no shortlist miss, ties by lower id
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all(), (metric, ddt, qdt, k)
        assert (res.indices == ids.astype(np.uint64)).all(), (metric, ddt, qdt, k)


def test_flat_search_large_and_sq(svs, oracle):
    import torch

    from bench import brute_force_topk, deep_like_device

    dev = torch.device("cuda", 0)
    n, d, nq, k = 150_000, 96, 3000, 10
    x = deep_like_device(n, d, 24, 1, dev)
    q = deep_like_device(nq, d, 24, 2, dev)
    idx = svs.FlatIndex(x.cpu().numpy(), svs.DistanceL2())
    ids, dists = idx.flat_search_device(q, k)
    torch.cuda.synchronize()
    gt = brute_force_topk(x, q, k)
    assert (ids == gt).float().mean().item() > 0.9995
    assert (dists[:, 1:] >= dists[:, :-1]).all()
    # scalar-quantized database, cosine
    xs = deep_like(4000, 64, 16, seed=8, normalize=False)
    qs = deep_like(50, 64, 16, seed=9, normalize=False)
    codes, scale, bias = oracle.sq_compress(xs, I8)
    f = svs.FlatIndex(codes, svs.DistanceCosineSimilarity(), sq=(scale, bias))
    r = f.search(qs, 5)
    oi, od = oracle.flat_search(COS, codes, qs, 5, sq=(scale, bias))
    assert (r.indices == oi.astype(np.uint64)).all() and (r.distances.view(np.uint32) == od.view(np.uint32)).all()


def test_concurrent_host_callers(svs, oracle, small_world):
    """index/vamana/index.h:541-545: search is const / re-entrant -- several host threads share one index."""
    import threading

    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    want = {}
    for w in (8, 24, 64, 100):
        want[w] = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"], 10, w)
    errors = []

    def worker(w, reps):
        try:
            for _ in range(reps):
                r = idx.search(sw["q"], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(w), search_buffer_visited_set=bool(w % 16)))
                assert (r.indices == want[w][0].astype(np.uint64)).all()
                assert (r.distances.view(np.uint32) == want[w][1].view(np.uint32)).all()
        except Exception as e:  # noqa: BLE001
            errors.append((w, repr(e)))

    threads = [threading.Thread(target=worker, args=(w, 15)) for w in (8, 24, 64, 100) for _ in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_host_pointer_pipeline_equals_one_launch(svs, oracle, small_world):
    """svsx_search cuts batches of >= 4096 queries into chunks, each on its own stream (H2D / kernel / D2H of neighbouring chunks overlap).
    Queries are independent (index/vamana/index.h:571-574): every output -- 64- and 32-bit ids, distances, statistics, search history, with
    dense and strided query rows -- must equal ONE launch over device-resident queries, and the oracle on a sample."""
    import ctypes as C

    import torch

    from scalablevectorsearch_amd import _lib

    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    rng = np.random.default_rng(5)
    nq, k, d = 5003, 10, sw["x"].shape[1]
    q = (sw["q"][rng.integers(0, sw["q"].shape[0], nq)] + np.float32(1e-3) * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(24, 40), True)
    dq = torch.from_numpy(q).cuda()
    ids64 = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    dd = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    dst = torch.zeros((nq, 6), dtype=torch.int32, device="cuda")
    idx.search_device(dq, k, sp, d_ids64=ids64, d_dists=dd, d_stats=dst)
    torch.cuda.synchronize()
    want_i, want_d, want_s = ids64.cpu().numpy(), dd.cpu().numpy(), dst.cpu().numpy()
    oi, od = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], q[:64], k, 24, 40, visited_set=True)
    assert (want_i[:64] == oi).all() and (want_d[:64].view(np.uint32) == od.view(np.uint32)).all()
    for stride_pad in (0, 16):  # dense rows, then rows of a wider host matrix
        host = np.zeros((nq, d + stride_pad), np.float32)
        host[:, :d] = q
        h64 = np.zeros((nq, k), np.uint64); h32 = np.zeros((nq, k), np.uint32); hd = np.zeros((nq, k), np.float32)
        hst = np.zeros((nq, 6), np.uint32); hcap = 8
        hhi = np.zeros((nq, hcap), np.uint32); hhd = np.zeros((nq, hcap), np.float32); hhl = np.zeros(nq, np.uint32)
        aux = _lib.SearchAux()
        aux.stats, aux.history_ids, aux.history_dists, aux.history_len = hst.ctypes.data, hhi.ctypes.data, hhd.ctypes.data, hhl.ctypes.data
        aux.history_cap = hcap
        aux.query_row_stride_bytes = host.strides[0]
        spc = sp._c()
        _lib.check(_lib.lib().svsx_search(idx._h, host.ctypes.data, 0, nq, k, C.byref(spc), h64.ctypes.data, h32.ctypes.data, hd.ctypes.data, C.byref(aux)))
        assert (h64.astype(np.int64) == want_i).all() and (h32 == want_i.astype(np.uint32)).all()
        assert (hd.view(np.uint32) == want_d.view(np.uint32)).all()
        assert (hst.astype(np.int64) == want_s.astype(np.int64)).all()
        assert (hhl == want_s[:, 0].astype(np.uint32)).all()          # history_len = expansions
        assert (hhi[:, 0] == np.uint32(sw["entry"][0] if np.ndim(sw["entry"]) else sw["entry"])).all()  # every search expands the entry point first
    # a second call on the same index reuses the staging context; concurrent callers get their own (test_concurrent_host_callers)
    r = idx.search(q, k, sp)
    assert (r.indices.astype(np.int64) == want_i).all()


def test_partial_distance_rejection_matches_analysis(svs, oracle, small_world):
    """f32 L2, 64 < d <= 128: the kernel stops after the first 64 dimensions when the partial sum already exceeds the
    worst buffer entry (result-neutral; parity is covered by every f32-L2 test above).  The number of such rejections
    must stay below the oracle's count of candidates whose 64-dim partial exceeds the worst (the kernel's threshold is
    per pass, hence slightly staler) and be substantial."""
    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    res, st = idx.search(sw["q"], 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(48)), stats=True)
    _, _, so = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"], 10, 48, stats=True)
    assert (st[:, 2] == so[:, 2]).all()  # both count the entry-point evaluation
    assert st[:, 4].sum() <= so[:, 8].sum() and st[:, 4].sum() > 0.5 * so[:, 8].sum()


def test_search_index_cli_round_trip(svs, oracle, small_world, tmp_path):
    """utils/search_index.cpp:54-178: positional CLI over files written in the reference's formats (native v1 blobs,
    svs_config.toml entry point, fvecs queries) -> <prefix>_idx.ivecs with the oracle's ids."""
    import subprocess
    import sys

    from scalablevectorsearch_amd import io

    sw = small_world
    os.makedirs(tmp_path / "data"); os.makedirs(tmp_path / "graph"); os.makedirs(tmp_path / "config")
    io.write_svs_native(tmp_path / "data" / "data_0.svs", sw["x"])
    io.write_svs_native(tmp_path / "graph" / "graph_0.svs", sw["graph"])
    io.write_fvecs(tmp_path / "q.fvecs", sw["q"])
    (tmp_path / "config" / "svs_config.toml").write_text(
        '__version__ = "v0.0.3"\n[object]\n__schema__ = "vamana_index_parameters"\nentry_point = %d\nname = "vamana config parameters"\n' % int(sw["entry"][0]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "scalablevectorsearch_amd.search_index", "float", "float", str(tmp_path / "q.fvecs"), "32", "10", "4",
                          str(tmp_path / "config"), str(tmp_path / "graph"), str(tmp_path / "data"), str(tmp_path / "res"), "L2"],
                         cwd=root, capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "Global search time" in out.stdout
    got = io.read_ivecs(str(tmp_path / "res_idx.ivecs"))
    ids, _ = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"], 10, 32)
    assert (got == ids).all()
    bad = subprocess.run([sys.executable, "-m", "scalablevectorsearch_amd.search_index", "float"], cwd=root, capture_output=True, text=True)
    assert bad.returncode == 1 and "Expected 11 arguments" in bad.stdout
    # the same program in C++ over the C ABI and include/svsx_io.hpp (tools/search_index.cpp): the same answer file
    exe = str(tmp_path / "search_index")
    pkg = os.path.join(root, "scalablevectorsearch_amd")
    subprocess.check_call(["g++", "-std=c++20", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "search_index.cpp"),
                           "-o", exe, "-L", pkg, "-lsvsx", f"-Wl,-rpath,{pkg}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe, "float", "float", str(tmp_path / "q.fvecs"), "32", "10", "4", str(tmp_path / "config"), str(tmp_path / "graph"),
                          str(tmp_path / "data"), str(tmp_path / "res_cxx"), "L2"], capture_output=True, text=True)
    assert out.returncode == 0 and "Global search time" in out.stdout, out.stdout + out.stderr
    assert (io.read_ivecs(str(tmp_path / "res_cxx_idx.ivecs")) == ids).all()


def test_malformed_graph_is_rejected(svs):
    x = np.zeros((4, 8), np.float32)
    g = np.zeros((4, 3), np.uint32)
    g[:, 0] = 1
    g[:, 1] = (1, 2, 3, 9)  # id 9 >= n
    with pytest.raises(svs.SvsxError) as e:
        svs.VamanaIndex(g, x, svs.DistanceL2(), [0])
    assert e.value.code == 2 and "malformed graph" in e.value.message
    g[3, 1] = 0
    g[2, 0] = 7  # degree larger than the row
    with pytest.raises(svs.SvsxError):
        svs.VamanaIndex(g, x, svs.DistanceL2(), [0])


@pytest.mark.parametrize("metric,d", [(L2, 96), (IP, 100), (COS, 32), (L2, 128), (L2, 160), (IP, 256), (COS, 300), (L2, 768), (COS, 700)])
def test_flat_search_threshold_pipeline(svs, oracle, metric, d):
    """n >= 65536 and d <= 768 take the sample-threshold-append pipeline (svsx_flat.hip stage 1b / 3b; beyond d = 128 the kernel that takes the
    dimension in chunks of 128, rows of the float16 shadow padded to a multiple of 128): same answers as the oracle's exhaustive search,
    ties by lower id."""
    fn = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[metric]
    n, nq = 70_000, 100
    x = deep_like(n, d, 12, seed=31, normalize=metric == L2)
    q = deep_like(nq, d, 12, seed=32, normalize=metric == L2)
    idx = svs.FlatIndex(x, fn)
    for k in (10, 32):
        res = idx.search(q, k)
        ids, dists = oracle.flat_search(metric, x, q, k)
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all(), (metric, d, k)
        assert (res.indices == ids.astype(np.uint64)).all(), (metric, d, k)


def test_flat_search_threshold_overflow_falls_back(svs, oracle):
    """three distinct vectors repeated 70 000 times: every key ties with the threshold, every candidate list overflows, the
    gated shortlist pass takes over; the answer is still the exhaustive one (equal distances -> lowest ids)."""
    rng = np.random.default_rng(3)
    base = rng.standard_normal((3, 96)).astype(np.float32)
    x = base[np.arange(70_000) % 3].copy()
    q = base + np.float32(0.01) * rng.standard_normal((3, 96)).astype(np.float32)
    idx = svs.FlatIndex(x, svs.DistanceL2())
    res = idx.search(q, 10)
    ids, dists = oracle.flat_search(L2, x, q, 10)
    assert (res.indices == ids.astype(np.uint64)).all() and (res.distances.view(np.uint32) == dists.view(np.uint32)).all()
    # unrepresentative sample: the sampled tiles (every 16th tile of 32 rows in a slice) hold far rows, the rest near ones
    x2 = deep_like(80_000, 96, 12, seed=41)
    q2 = deep_like(40, 96, 12, seed=42)
    x2[(np.arange(80_000) // 32) % 4 == 0] *= np.float32(25.0)  # a superset of every possible sample tile pattern
    idx2 = svs.FlatIndex(x2, svs.DistanceL2())
    res2 = idx2.search(q2, 10)
    ids2, dists2 = oracle.flat_search(L2, x2, q2, 10)
    assert (res2.indices == ids2.astype(np.uint64)).all() and (res2.distances.view(np.uint32) == dists2.view(np.uint32)).all()


@pytest.mark.parametrize("w,c,k,vis", [(17000, 17000, 10, 0), (100, 20000, 10, 1), (16001, 16001, 200, 0), (5, 40000, 10, 0)])
def test_search_buffers_beyond_one_waves_lds(svs, oracle, small_world, w, c, k, vis):
    """plain searches whose buffer exceeds 16 000 entries work on a state table in HBM (svsx_api.hip search_spilled): the reference only
    requires window <= capacity (search_buffer.h:84-93).  ids, distance bits and the tracker counts equal the oracle's; inline records are
    simply not used there."""
    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    for mode in (0, 2):
        if mode:
            idx.build_bound_codes(mode)
        res, st = idx.search(sw["q"][:9], k, svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c), vis), stats=True)
        ids, dists, ost = oracle.search(L2, sw["x"], sw["graph"], sw["entry"], sw["q"][:9], k, w, c, visited_set=bool(vis), stats=True)
        assert (res.indices == ids.astype(np.uint64)).all(), (w, c, k, mode)
        assert (res.distances.view(np.uint32) == dists.view(np.uint32)).all(), (w, c, k, mode)
        assert (st[:, 0] == ost[:, 0]).all() and (st[:, 1] == ost[:, 1]).all()
        assert "general=1" in idx.last_kernel_info(), idx.last_kernel_info()


def test_contraction_sensitive_inputs(svs, oracle):
    """the inputs of tests/test_contraction_choices.py (scalar-quantization decompress, inner-product epilogue and cosine query norm where a
    contracted multiply-add would give other float32 bits): the GPU path follows the oracle's -- the reference build's -- un-contracted reading"""
    from test_contraction_choices import contraction_sensitive_sq

    scale, bias, codes = contraction_sensitive_sq()
    rng = np.random.default_rng(1)
    d = 64
    x = rng.choice(codes, size=(600, d)).astype(np.int8)
    q = rng.standard_normal((16, d)).astype(np.float32)
    sq = (float(scale), float(bias))
    for metric, fn in ((COS, svs.DistanceCosineSimilarity()), (IP, svs.DistanceIP())):
        g, e = oracle.build_vamana(metric, x.astype(np.float32) * scale + bias, R=16, L=32, alpha=0.95)
        idx = svs.VamanaIndex(g, x, fn, e, sq=sq)
        for mode in (0, 2):
            if mode:
                idx.build_bound_codes(mode)
            res = idx.search(q, 10, svs.VamanaSearchParameters(svs.SearchBufferConfig(40), True))
            ids, dists = oracle.search(metric, x, g, e, q, 10, 40, sq=sq)
            assert (res.indices == ids.astype(np.uint64)).all() and (res.distances.view(np.uint32) == dists.view(np.uint32)).all(), (metric, mode)
    rngq = np.random.default_rng(7)  # cosine query norms on plain float rows
    xf = rngq.standard_normal((400, 96)).astype(np.float32)
    qf = rngq.standard_normal((64, 96)).astype(np.float32)
    got = svs.compute(svs.DistanceCosineSimilarity(), qf, xf)
    assert (got.view(np.uint32) == oracle.distance_batch(COS, qf, xf).view(np.uint32)).all()


@pytest.mark.parametrize("k", [100, 128, 224])
def test_flat_search_k_beyond_96(svs, oracle, k):
    """the reference's ground-truth files hold 100 neighbours per query (data/test_dataset/groundtruth_*.ivecs); the shortlist kernel's general
    insert path keeps k + 32 <= 256 candidates.  Small set (shortlist kernel alone) and a set large enough for the threshold pipeline."""
    import torch

    for n, d, nq in ((3000, 96, 40), (70_000, 64, 33)):
        x = deep_like(n, d, 24, seed=41)
        q = deep_like(nq, d, 24, seed=42)
        idx = svs.FlatIndex(x, svs.DistanceL2())
        r = idx.search(q, k)
        oi, od = oracle.flat_search(L2, x, q, k)
        assert (r.indices == oi.astype(np.uint64)).all(), (n, k)
        assert (r.distances.view(np.uint32) == od.view(np.uint32)).all(), (n, k)
        assert (np.diff(r.distances, axis=1) >= 0).all()
    r = svs.FlatIndex(x, svs.DistanceL2()).search(q[:3], 225)   # beyond the shortlist's 224: the exhaustive float32 pass alone (round 5)
    oi, od = oracle.flat_search(L2, x, q[:3], 225)
    assert (r.indices == oi.astype(np.uint64)).all() and (r.distances.view(np.uint32) == od.view(np.uint32)).all()


def test_golden_flat_search_on_gpu(svs):
    """tests/golden/flat.npz, generated from the reference's own BulkInserter / HeapInserter (oracle/_ref): tie-free queries give the same ids and
    distance bits at k = 1 / 10 / 100; with exact ties the distances are the reference's and every id is a real row at its distance."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "flat.npz"))
    names = {F32: "f32", F16: "f16", I8: "i8"}
    fns = {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}
    for qa, qb in [(F32, F32), (F32, F16), (F16, F16), (I8, I8), (F32, I8)]:
        x, q = g[f"x_{names[qb]}"], g[f"q_{names[qa]}"]
        for m in (L2, IP, COS):
            full = svs.compute(fns[m], q, x)
            idx = svs.FlatIndex(x, fns[m])
            for k in (int(v) for v in g["ks"]):
                r = idx.search(q, k)
                gi, gd = g[f"ids_{names[qa]}_{names[qb]}_{m}_{k}"], g[f"dists_{names[qa]}_{names[qb]}_{m}_{k}"]
                assert ((r.distances.view(np.uint32) == gd.view(np.uint32)) | (np.isnan(r.distances) & np.isnan(gd))).all(), (qa, qb, m, k)
                for i in range(len(q)):
                    if len(np.unique(full[i])) == len(full[i]):
                        assert (r.indices[i] == gi[i]).all(), (qa, qb, m, k, i)
    idx = svs.FlatIndex(g["x_tied"], svs.DistanceL2())
    full = svs.compute(svs.DistanceL2(), g["q_tied"], g["x_tied"])
    for k in (3, 10):
        r = idx.search(g["q_tied"], k)
        assert (r.distances.view(np.uint32) == g[f"dists_tied_{k}"].view(np.uint32)).all()
        for i in range(len(r.indices)):
            ids = r.indices[i].astype(np.int64)
            assert len(set(ids.tolist())) == k and (full[i][ids].view(np.uint32) == r.distances[i].view(np.uint32)).all()
