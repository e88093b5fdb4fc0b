"""GPU parity of the batched BatchIterator (svsx_iterator_*, index/vamana/iterator.h:81-339) against the oracle's
svso_iterator and the golden vectors generated from the reference's own SearchBuffer / greedy_search / std::sort."""
import os

import numpy as np
import pytest

from conftest import deep_like

pytestmark = pytest.mark.gpu

L2, IP, COS = 0, 1, 2
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iterator.npz")
ALL_ONES = np.iinfo(np.uint64).max


@pytest.fixture(scope="module")
def svs():
    import scalablevectorsearch_amd as svs

    return svs


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _fn(svs, m):
    return {L2: svs.DistanceL2(), IP: svs.DistanceIP(), COS: svs.DistanceCosineSimilarity()}[m]


def _check_schedule(svs, oracle, metric, x, graph, entry, q, sched, extra=None, sq=None, per_query=None):
    """Drive the GPU iterator over all queries and one oracle iterator per query through `sched`; everything must agree."""
    idx = svs.VamanaIndex(graph, x, _fn(svs, metric), entry, sq=sq)
    it = idx.batch_iterator(q, extra)
    from oracle.oracle import PLACEHOLDER

    oits = [oracle.iterator(metric, x, graph, entry, qv, PLACEHOLDER if extra is None else extra, sq) for qv in q]
    for step, bs in enumerate(sched):
        sizes = None if per_query is None else per_query[step]
        ids, dists, counts = it.next(bs, sizes)
        meta = it.state()
        done = it.done()
        for i, oi in enumerate(oits):
            b = bs if sizes is None else int(sizes[i])
            if b == 0xFFFFFFFF:
                assert counts[i] == 0
                continue
            wi, wd = oi.next(b)
            c = int(counts[i])
            assert c == len(wi), (step, i, c, len(wi))
            assert (ids[i, :c] == wi.astype(np.uint64)).all(), (step, i)
            assert (_bits(dists[i, :c]) == _bits(wd)).all(), (step, i)
            assert (ids[i, c:] == ALL_ONES).all() and np.isinf(dists[i, c:]).all()
            assert bool(done[i]) == oi.done(), (step, i)
            assert int(meta[i, 5]) == oi.batch_number()
            assert (int(meta[i, 2]), int(meta[i, 3])) == oi.config()
            assert int(meta[i, 4]) == oi.n_yielded()
    return it


def test_golden_iterator_on_gpu(svs):
    g = np.load(GOLD)
    q = g["q"]
    ties = 0
    for m, xv, tag in ((L2, g["x"], "l2_f32"), (IP, g["x"].astype(np.float16), "ip_f16")):
        idx = svs.VamanaIndex(g["graph"], xv, _fn(svs, m), g["entry"])
        for si in range(int(g["n_schedules"])):
            sched = [int(v) for v in g[f"sched_{si}"]]
            it = idx.batch_iterator(q)
            got_ids = [[] for _ in range(len(q))]
            got_d = [[] for _ in range(len(q))]
            for b, bs in enumerate(sched):
                ids, dists, counts = it.next(bs)
                done = it.done()
                for qi in range(len(q)):
                    key = f"{tag}_{si}_{qi}"
                    assert counts[qi] == g[key + "_counts"][b], (key, b)
                    assert int(done[qi]) == g[key + "_done"][b], (key, b)
                    got_ids[qi].append(ids[qi, :counts[qi]])
                    got_d[qi].append(dists[qi, :counts[qi]])
            for qi in range(len(q)):
                key = f"{tag}_{si}_{qi}"
                assert (np.concatenate(got_ids[qi]) == g[key + "_ids"].astype(np.uint64)).all(), key
                dd = np.concatenate(got_d[qi])
                assert (_bits(dd) == _bits(g[key + "_dists"])).all(), key
                ties += int((np.diff(dd) == 0).sum())
        # restart_next_search before calls 2 and 4 (query 0)
        it = idx.batch_iterator(q[:1])
        ids_all, d_all = [], []
        for b, r in enumerate([0, 0, 1, 0, 1, 0]):
            if r:
                it.restart_next_search()
            ids, dists, counts = it.next(15)
            ids_all.append(ids[0, :counts[0]])
            d_all.append(dists[0, :counts[0]])
        assert (np.concatenate(ids_all) == g[f"{tag}_restart_ids"].astype(np.uint64)).all()
        assert (_bits(np.concatenate(d_all)) == _bits(g[f"{tag}_restart_dists"])).all()
    assert ties > 100


@pytest.mark.parametrize("sched", [[10] * 10, [20, 5, 50, 1, 0, 7, 100], [7] * 40, [500, 500, 500, 500]])
def test_iterator_matches_oracle(svs, oracle, small_world, sched):
    sw = small_world
    _check_schedule(svs, oracle, L2, sw["x"], sw["graph"], sw["entry"], sw["q"][:24], sched)


def test_iterator_ties_and_extra_capacities(svs, oracle, small_world):
    sw = small_world
    x2 = np.concatenate([sw["x"][:600], sw["x"][:600], sw["x"][:600]])  # every vector three times: ties everywhere
    g2, e2 = oracle.build_vamana(L2, x2, R=24, L=40)
    for extra in (None, 0, 1, 37, 300):
        _check_schedule(svs, oracle, L2, x2, g2, e2, sw["q"][:12], [16, 16, 64, 3, 200, 200], extra=extra)


def test_iterator_type_pairs_and_metrics(svs, oracle, small_world):
    sw = small_world
    x, q = sw["x"], sw["q"][:10]
    sched = [12, 30, 5, 64]
    _check_schedule(svs, oracle, IP, x.astype(np.float16), sw["graph"], sw["entry"], q, sched)
    _check_schedule(svs, oracle, COS, x.astype(np.float16), sw["graph"], sw["entry"], q.astype(np.float16), sched)
    scale = 127.0 / np.abs(x).max()
    xi = np.clip(np.round(x * scale), -128, 127).astype(np.int8)
    qi = np.clip(np.round(q * scale), -128, 127).astype(np.int8)
    _check_schedule(svs, oracle, L2, xi, sw["graph"], sw["entry"], qi, sched)
    _check_schedule(svs, oracle, COS, xi, sw["graph"], sw["entry"], q, sched)
    codes, sc, bias = oracle.sq_compress(x)
    for m in (L2, IP, COS):
        _check_schedule(svs, oracle, m, codes, sw["graph"], sw["entry"], q, sched, sq=(sc, bias))
    xr = deep_like(900, 100, 20, seed=9)  # ragged dimension
    gr, er = oracle.build_vamana(L2, xr, R=16, L=32)
    _check_schedule(svs, oracle, L2, xr, gr, er, deep_like(6, 100, 20, seed=10), sched)


def test_iterator_per_query_batch_sizes_and_skip(svs, oracle, small_world):
    sw = small_world
    nq = 16
    rng = np.random.default_rng(11)
    sched, per_query = [], []
    for step in range(8):
        sizes = rng.integers(0, 40, nq).astype(np.uint32)
        sizes[rng.random(nq) < 0.3] = 0xFFFFFFFF
        sched.append(40)
        per_query.append(sizes)
    _check_schedule(svs, oracle, L2, sw["x"], sw["graph"], sw["entry"], sw["q"][:nq], sched, per_query=per_query)


def test_iterator_exhaustion_and_update(svs, oracle):
    x = deep_like(50, 32, 8, seed=5)
    q = deep_like(5, 32, 8, seed=6)
    g, e = oracle.build_vamana(L2, x, R=8, L=16)
    it = _check_schedule(svs, oracle, L2, x, g, e, q, [30, 30, 30, 30, 30], extra=7)
    assert it.done().all() and (it.state()[:, 4] == 50).all()
    frozen = it.state()
    ids, dists, counts = it.next(30)
    assert (counts == 0).all() and (ids == ALL_ONES).all() and (it.state() == frozen).all()  # iterator.h:253-256
    q2 = deep_like(5, 32, 8, seed=7)
    it.update(q2)
    m = it.state()
    assert (m[:, [0, 1, 2, 4, 5]] == 0).all() and (m[:, 3] == 7).all() and (m[:, 6] == 1).all()
    ids, dists, counts = it.next(9)
    for i in range(5):
        wi, wd = oracle.iterate(L2, x, g, e, q2[i], [9], extra=7)[0][0]
        assert (ids[i, :counts[i]] == wi.astype(np.uint64)).all() and (_bits(dists[i, :counts[i]]) == _bits(wd)).all()


def test_iterator_device_pointers_and_errors(svs, oracle, small_world):
    import torch

    sw = small_world
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    q = sw["q"][:8]
    it = idx.batch_iterator(q)
    dev = torch.device("cuda", 0)
    for bs in (10, 25):
        d_ids = torch.empty((8, bs), dtype=torch.int64, device=dev)
        d_d = torch.empty((8, bs), dtype=torch.float32, device=dev)
        d_c = torch.empty(8, dtype=torch.int32, device=dev)
        it.next_device(bs, d_ids, d_d, d_c)
        torch.cuda.synchronize()
        assert (d_c.cpu().numpy() == bs).all()
    want = [oracle.iterate(L2, sw["x"], sw["graph"], sw["entry"], q[i], [10, 25])[0][1] for i in range(8)]
    for i in range(8):
        assert (d_ids[i].cpu().numpy() == want[i][0].astype(np.int64)).all()
        assert (_bits(d_d[i].cpu().numpy()) == _bits(want[i][1])).all()
    with pytest.raises(svs.SvsxError):  # detail::checkdims (iterator.h:66-74)
        idx.batch_iterator(np.zeros((2, 5), np.float32))
    it8 = idx.batch_iterator(np.clip(np.round(q[:2] * 400), -128, 127).astype(np.int8))  # a scalar-loop type pair (int8 queries, float32 rows): served since round 5
    ids8, dists8, counts8 = it8.next(10)
    want8 = [oracle.iterate(L2, sw["x"], sw["graph"], sw["entry"], np.clip(np.round(q[i] * 400), -128, 127).astype(np.int8), [10])[0][0] for i in range(2)]
    for i in range(2):
        assert (ids8[i] == want8[i][0].astype(np.uint64)).all() and (_bits(dists8[i]) == _bits(want8[i][1])).all()
    it2 = idx.batch_iterator(q[:2], 15000)
    ids, dists, counts = it2.next(1400)  # 16 400 entries: beyond one wave's LDS, the buffer continues in HBM (1 500 points: everything comes out)
    want2 = [oracle.iterate(L2, sw["x"], sw["graph"], sw["entry"], q[i], [1400], extra=15000) for i in range(2)]
    for i in range(2):
        got_n = int(counts[i])
        assert got_n == len(want2[i][0][0][0]) and (ids[i, :got_n] == want2[i][0][0][0].astype(np.uint64)).all()
    assert (it2.batch_number() == 1).all()


def test_many_entry_points_use_std_sort_order(svs, oracle, small_world):
    """> 16 entry points: EntryPointInitializer's buffer.sort() (greedy_search.h:91) is an introsort, not a stable
    insertion sort; with duplicated entry vectors the tie order must still match (oracle = libstdc++ order)."""
    sw = small_world
    x2 = np.concatenate([sw["x"][:700], sw["x"][:700]])
    g2, e2 = oracle.build_vamana(L2, x2, R=24, L=40)
    entry = np.concatenate([np.arange(0, 20), np.arange(700, 720), [int(e2[0])]]).astype(np.uint32)  # 41 entries, 20 tied pairs
    idx = svs.VamanaIndex(g2, x2, svs.DistanceL2(), entry)
    for w, c, k in ((41, 41, 41), (30, 45, 30), (64, 64, 40), (8, 8, 8)):
        res = idx.search(sw["q"][:16], k, svs.VamanaSearchParameters(svs.SearchBufferConfig(w, c)))
        ids, dists = oracle.search(L2, x2, g2, entry, sw["q"][:16], k, w, c)
        assert (res.indices == ids.astype(np.uint64)).all(), (w, c, k)
        assert (_bits(res.distances) == _bits(dists)).all()


def test_range_and_filtered_search_on_gpu(svs, oracle, small_world):
    from test_iterator_oracle import _FakeIndex, _OracleBatchIterator

    sw = small_world
    q = sw["q"][:12]
    idx = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"])
    idx.set_search_parameters(svs.VamanaSearchParameters(svs.SearchBufferConfig(20)))
    fake = _FakeIndex(oracle, L2, sw["x"], sw["graph"], sw["entry"], window=20)
    member = (np.arange(sw["x"].shape[0]) % 3) == 0
    for radius in (0.9, 1.1):
        for mem in (member, None):
            a = svs.range_search(idx, q, radius, is_member=mem)                                      # svsx_range_search: the loop inside the library
            a2 = svs.range_search(idx, q, radius, is_member=mem, make_iterator=svs.BatchIterator)    # the Python mirror over the library's iterator
            b = svs.range_search(fake, q, radius, is_member=mem, make_iterator=_OracleBatchIterator)  # the Python mirror over the oracle's iterator
            for r in (a, a2):
                assert (r[0] == b[0]).all() and (r[1] == b[1]).all() and (_bits(r[2]) == _bits(b[2])).all()
            assert a[0][-1] > 0
    with pytest.raises(svs.SvsxError):
        svs.range_search(idx, q, 0.0)
    with pytest.raises(svs.SvsxError, match="Incompatible dimensions"):  # detail::checkdims before the library sees the pointer
        svs.range_search(idx, q[:, :-3], 1.0)
    e = svs.range_search(idx, q[:0], 1.0)  # no queries: the library never calls the allocator (as the reference)
    assert e[0].tolist() == [0] and e[1].size == 0 and e[2].size == 0
    # an index whose labels start at id_offset: the filter is indexed by local id on the library path and on its Python mirror alike
    idx_off = svs.VamanaIndex(sw["graph"], sw["x"], svs.DistanceL2(), sw["entry"], id_offset=1000)
    idx_off.set_search_parameters(svs.VamanaSearchParameters(svs.SearchBufferConfig(20)))
    a = svs.range_search(idx_off, q, 1.1, is_member=member)
    a2 = svs.range_search(idx_off, q, 1.1, is_member=member, make_iterator=svs.BatchIterator)
    b = svs.range_search(fake, q, 1.1, is_member=member, make_iterator=_OracleBatchIterator)
    assert (a[0] == b[0]).all() and (a[1] == b[1] + 1000).all() and (a2[1] == a[1]).all() and (_bits(a2[2]) == _bits(a[2])).all()
    # VamanaIndex::get_distance (index.h:915-933): the value a search reports for that row, bit for bit
    res = idx.search(q, 5, svs.VamanaSearchParameters(svs.SearchBufferConfig(20)))
    for i in range(4):
        for j in range(5):
            dd = idx.get_distance(int(res.indices[i, j]), q[i])
            assert np.float32(dd).view(np.uint32) == res.distances[i, j].view(np.uint32)
    with pytest.raises(svs.SvsxError):
        idx.get_distance(sw["x"].shape[0], q[0])
    for k, stop in ((10, 0.0), (25, 0.0), (10, 0.5)):
        a = svs.filtered_search(idx, q, k, member, filter_stop=stop, filter_estimate_batch=False)
        b = svs.filtered_search(fake, q, k, member, filter_stop=stop, make_iterator=_OracleBatchIterator, filter_estimate_batch=False)
        assert (a.indices == b.indices).all() and (_bits(a.distances) == _bits(b.distances)).all()
        # the same loop inside the library (svsx_search_filtered: the IDFilter argument of the runtime's batch operator)
        c = idx.search_filtered(q, k, member, svs.VamanaSearchParameters(svs.SearchBufferConfig(20)), filter_stop=stop, filter_estimate_batch=False)
        assert (c.indices == b.indices).all() and (_bits(c.distances) == _bits(b.distances)).all()
    # SearchParams::filter_estimate_batch = true, the runtime's default (vamana_index_impl.h:153-175): the sampled hit rate (std::mt19937(42))
    # sets the first batch size -- or answers everything with padding when it falls below filter_stop
    sparse = (np.arange(sw["x"].shape[0]) % 40) == 0    # ~2.5 % of the ids: the estimate asks for a first batch of hundreds
    for mem, k, stop in ((member, 10, 0.0), (sparse, 10, 0.0), (sparse, 5, 0.01), (sparse, 5, 0.2), (member, 25, 0.5)):
        b = svs.filtered_search(fake, q, k, mem, filter_stop=stop, make_iterator=_OracleBatchIterator, filter_estimate_batch=True)
        c = idx.search_filtered(q, k, mem, svs.VamanaSearchParameters(svs.SearchBufferConfig(20)), filter_stop=stop)   # default: estimate
        assert (c.indices == b.indices).all() and (_bits(c.distances) == _bits(b.distances)).all(), (k, stop)
    assert (c.indices == np.uint64(2**64 - 1)).all()     # 1/3 < 0.5: the sample alone stops the search
    # a filter nothing passes: every row padded; a filter with a handful of members: found through growing batches
    none = idx.search_filtered(q, 5, np.zeros(sw["x"].shape[0], bool), svs.VamanaSearchParameters(svs.SearchBufferConfig(20)))
    assert (none.indices == np.uint64(2**64 - 1)).all() and np.isposinf(none.distances).all()
    few = np.zeros(sw["x"].shape[0], bool)
    few[[3, 500, 900, 1400]] = True
    a = svs.filtered_search(idx, q, 3, few, filter_estimate_batch=False)
    c = idx.search_filtered(q, 3, few, svs.VamanaSearchParameters(svs.SearchBufferConfig(20)), filter_estimate_batch=False)
    assert (a.indices == c.indices).all() and (c.indices != np.uint64(2**64 - 1)).all()


def test_filtered_search_with_buffers_beyond_lds(svs, oracle):
    """A filter almost nothing passes: the runtime's loop (vamana_index_impl.h:176-213) asks the iterator for batches of up to n,
    so the per-query search buffer outgrows one wave's LDS (16 000 entries) and the kernel continues in place on the HBM state
    table.  Same answers as the loop over the oracle's iterator, for the Python mirror and for svsx_search_filtered."""
    import torch

    from test_iterator_oracle import _FakeIndex, _OracleBatchIterator

    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs

    n, d = 24_000, 32
    x = deep_like(n, d, 8, seed=51)
    q = deep_like(2, d, 8, seed=52)
    g, entry = build_vamana_device(torch.from_numpy(x).cuda(), n, d, svs.DistanceL2(), VamanaBuildParameters(graph_max_degree=32, window_size=64, prune_to=28))
    graph = device_graph_to_svs(g, 32)
    ep = np.array([entry], np.uint32)
    idx = svs.VamanaIndex(graph, x, svs.DistanceL2(), ep)
    # window 4000: the loop starts with batches of 4000 (max(k, search_window_size)), so a handful of rounds cross the 16 000-entry line
    idx.set_search_parameters(svs.VamanaSearchParameters(svs.SearchBufferConfig(4000)))
    fake = _FakeIndex(oracle, L2, x, graph, ep, window=4000)
    member = np.zeros(n, bool)
    far = np.argsort(((x - q[0]) ** 2).sum(1))[-4:]  # the members are the points FARTHEST from the first query
    member[far] = True
    want = svs.filtered_search(fake, q, 3, member, make_iterator=_OracleBatchIterator, filter_estimate_batch=False)
    got = svs.filtered_search(idx, q, 3, member, filter_estimate_batch=False)
    assert (got.indices == want.indices).all() and (_bits(got.distances) == _bits(want.distances)).all()
    lib = idx.search_filtered(q, 3, member, svs.VamanaSearchParameters(svs.SearchBufferConfig(4000)), filter_estimate_batch=False)
    assert (lib.indices == want.indices).all() and (_bits(lib.distances) == _bits(want.distances)).all()
    assert (want.indices[0] != np.uint64(2**64 - 1)).all()  # found, through buffers far beyond the LDS budget
