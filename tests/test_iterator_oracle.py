"""BatchIterator (index/vamana/iterator.h:81-339): pins the C restatement (oracle/svs_oracle.c svso_iterator_*) against
the reference's own SearchBuffer / greedy_search / std::sort (oracle/_ref, golden vectors generated from it) and restates
the properties the reference's own test checks (tests/svs/index/vamana/iterator.cpp:79-200).  Also covers the host logic
of range / filtered search (runtime vamana_index_impl.h:140-326) on an oracle-backed iterator.  No GPU needed."""
import os

import numpy as np
import pytest

from oracle.oracle import IP, L2
from conftest import deep_like

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "iterator.npz")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _golden_cases(g):
    x, q = g["x"], g["q"]
    for m, xv, tag in ((L2, x, "l2_f32"), (IP, x.astype(np.float16), "ip_f16")):
        for si in range(int(g["n_schedules"])):
            for qi in range(q.shape[0]):
                yield m, xv, tag, si, qi


def test_golden_iterator(oracle):
    g = np.load(GOLD)
    ties = 0
    for m, xv, tag, si, qi in _golden_cases(g):
        sched = [int(v) for v in g[f"sched_{si}"]]
        res, done, iters, cfg = oracle.iterate(m, xv, g["graph"], g["entry"], g["q"][qi], sched)
        key = f"{tag}_{si}_{qi}"
        assert [len(r[0]) for r in res] == list(g[key + "_counts"]), key
        ids = np.concatenate([r[0] for r in res])
        dd = np.concatenate([r[1] for r in res])
        assert (ids == g[key + "_ids"]).all(), key
        assert (_bits(dd) == _bits(g[key + "_dists"])).all(), key
        assert [int(v) for v in done] == list(g[key + "_done"]), key
        ties += int((np.diff(dd) == 0).sum())
    assert ties > 100  # the fixture really exercises the std::sort tie order
    for m, xv, tag in ((L2, g["x"], "l2_f32"), (IP, g["x"].astype(np.float16), "ip_f16")):
        res, _, _, _ = oracle.iterate(m, xv, g["graph"], g["entry"], g["q"][0], [15] * 6, restart_before=[0, 0, 1, 0, 1, 0])
        assert (np.concatenate([r[0] for r in res]) == g[f"{tag}_restart_ids"]).all()
        assert (_bits(np.concatenate([r[1] for r in res])) == _bits(g[f"{tag}_restart_dists"])).all()


def test_golden_std_sort_order(oracle):
    """SearchBuffer::sort is std::sort (search_buffer.h:408): unstable, so the tie order is libstdc++'s."""
    g = np.load(GOLD)
    for t in range(12):
        vals = g[f"sort_{t}_in"]
        b = oracle.buffer(len(vals), len(vals), bool(g[f"sort_{t}_greater"]))
        for i, v in enumerate(vals):
            b.push_back(i, float(v))
        b.sort()
        assert [e[0] for e in b.state()] == list(g[f"sort_{t}_ids"]), t
    assert oracle.lib.svso_sort_depth_exhausted() == 0


def test_live_sort_vs_real_std_sort(oracle, ref):
    rng = np.random.default_rng(3)
    for trial in range(200):
        n = int(rng.integers(1, 600))
        greater = bool(trial & 1)
        vals = rng.integers(0, max(2, n // int(rng.integers(1, 6))), n).astype(np.float32)
        if trial % 3:
            vals = np.sort(vals)[::-1 if greater else 1]
        a, b = oracle.buffer(n, n, greater), ref.buffer(n, n, greater)
        for i, v in enumerate(vals):
            a.push_back(i, float(v))
            b.push_back(i, float(v))
        a.sort()
        b.sort()
        assert a.state() == b.state(), trial


def test_live_iterator_vs_reference_headers(oracle, ref, small_world):
    w = small_world
    x2 = np.concatenate([w["x"][:700], w["x"][:700]])  # duplicates -> ties
    g2, e2 = oracle.build_vamana(L2, x2, R=24, L=40)
    for x, g, e in ((w["x"], w["graph"], w["entry"]), (x2, g2, e2)):
        for sched in ([10] * 10, [20, 5, 50, 1, 0, 7, 100], [700, 700, 700]):
            for qi in range(6):
                a = oracle.iterate(L2, x, g, e, w["q"][qi], sched)
                b = ref.iterate(L2, x, g, e, w["q"][qi], sched)
                assert a[1] == b[1]
                for (ia, da), (ib, db) in zip(a[0], b[0]):
                    assert (ia == ib).all() and (_bits(da) == _bits(db)).all()


def test_reference_iterator_properties(oracle, small_world):
    """tests/svs/index/vamana/iterator.cpp:79-200: batches of exactly batchsize unique ids, batch_number, distances equal to
    those of a full search, >= 98 % of the first 100 yielded ids are the window-100 search result."""
    w = small_world
    num_neighbors = 100
    for qi in range(10):
        ids, dists = oracle.search(L2, w["x"], w["graph"], w["entry"], w["q"][qi:qi + 1], num_neighbors, num_neighbors)
        id_to_distance = dict(zip(ids[0].tolist(), dists[0].tolist()))
        for batchsize in (10, 20, 25, 50, 100):
            it = oracle.iterator(L2, w["x"], w["graph"], w["entry"], w["q"][qi])
            seen, similar = set(), 0
            for batch in range(num_neighbors // batchsize):
                bi, bd = it.next(batchsize)
                assert it.batch_number() == batch + 1
                assert len(bi) == batchsize
                for i, dd in zip(bi.tolist(), bd.tolist()):
                    assert i not in seen
                    if i in id_to_distance:
                        assert id_to_distance[i] == dd
                        similar += 1
                    seen.add(i)
            assert len(seen) == num_neighbors
            assert similar >= 0.98 * num_neighbors


def test_iterator_exhaustion_update_and_config(oracle):
    x = deep_like(50, 32, 8, seed=5)
    q = deep_like(3, 32, 8, seed=6)
    g, e = oracle.build_vamana(L2, x, R=8, L=16)
    it = oracle.iterator(L2, x, g, e, q[0], extra=7)
    assert it.config() == (0, 7)  # initialize_buffer: SearchBufferConfig{0, extra} (iterator.h:127-131)
    total = []
    for _ in range(3):
        total += it.next(30)[0].tolist()
    assert len(total) == len(set(total)) == 50 and it.done()  # every reachable id exactly once
    cfg, iters = it.config(), it.batch_number()
    assert it.next(30)[0].size == 0 and it.config() == cfg and it.batch_number() == iters  # done: state frozen (:253-256)
    it.update(q[1])
    assert it.config() == (0, 7) and it.batch_number() == 0 and not it.done()
    a = it.next(5)
    b = oracle.iterate(L2, x, g, e, q[1], [5], extra=7)[0][0]
    assert (a[0] == b[0]).all()
    it0 = oracle.iterator(L2, x, g, e, q[2])
    assert it0.config() == (0, 100)  # ITERATOR_EXTRA_BUFFER_CAPACITY_DEFAULT (lib/preprocessor.h:187)
    assert it0.next(0)[0].size == 0 and not it0.done() and it0.batch_number() == 1  # next(0) does not exhaust (:313)


# ---- host logic of filtered / range search on an oracle-backed iterator -------------------------------------------------
class _OracleBatchIterator:
    """Same interface as scalablevectorsearch_amd.BatchIterator, one svso_iterator per query (the checker)."""

    def __init__(self, index, queries, extra=None):
        from oracle.oracle import PLACEHOLDER

        self.its = [index.oracle.iterator(index.metric, index.x, index.graph, index.entry, qv, PLACEHOLDER if extra is None else extra)
                    for qv in np.atleast_2d(queries)]

    def next(self, batch_size, batch_sizes=None):
        nq = len(self.its)
        ids = np.full((nq, batch_size), np.iinfo(np.uint64).max, np.uint64)
        dists = np.full((nq, batch_size), np.inf, np.float32)
        counts = np.zeros(nq, np.uint32)
        for i, it in enumerate(self.its):
            bs = batch_size if batch_sizes is None else int(batch_sizes[i])
            if bs == 0xFFFFFFFF:
                continue
            bi, bd = it.next(bs)
            ids[i, :len(bi)], dists[i, :len(bi)], counts[i] = bi, bd, len(bi)
        return ids, dists, counts

    def done(self):
        return np.array([it.done() for it in self.its])


class _FakeIndex:
    def __init__(self, oracle, metric, x, graph, entry, window):
        import scalablevectorsearch_amd as svs

        self.oracle, self.metric, self.x, self.graph, self.entry = oracle, metric, x, graph, entry
        self.distance = svs.DistanceL2() if metric == L2 else svs.DistanceIP()
        self._p = svs.VamanaSearchParameters(svs.SearchBufferConfig(window))

    def get_search_parameters(self): return self._p
    def size(self): return self.x.shape[0]


def test_range_and_filtered_search_host_logic(oracle, small_world):
    import scalablevectorsearch_amd as svs
    from scalablevectorsearch_amd.iterator import predict_further_processing, should_stop_filtered_search

    w = small_world
    ix = _FakeIndex(oracle, L2, w["x"], w["graph"], w["entry"], window=20)
    q = w["q"][:8]
    # range search: everything the iterator yields before its first out-of-range neighbour (vamana_index_impl.h:262-320)
    radius = 1.0
    lims, labels, dists = svs.range_search(ix, q, radius, make_iterator=_OracleBatchIterator)
    assert lims[-1] == labels.size == dists.size and (dists < radius).all()
    for i in range(len(q)):
        it = oracle.iterator(L2, w["x"], w["graph"], w["entry"], q[i])
        want = []
        while True:
            bi, bd = it.next(20)
            stop = False
            for a, b in zip(bi.tolist(), bd.tolist()):
                if not b < radius:
                    stop = True
                    break
                want.append(a)
            if stop or it.done():
                break
        assert labels[lims[i]:lims[i + 1]].tolist() == want
    assert lims[-1] > 0
    with pytest.raises(svs.SvsxError):
        svs.range_search(ix, q, 0.0, make_iterator=_OracleBatchIterator)
    # detail::checkdims: a query array of the wrong width is refused before anything reads it (the library path only gets a pointer)
    ix.d = w["x"].shape[1]
    with pytest.raises(svs.SvsxError, match="Incompatible dimensions"):
        svs.range_search(ix, q[:, :-1], radius, make_iterator=_OracleBatchIterator)
    # the filter is indexed by LOCAL id (label - id_offset) on the mirror path too
    ix.id_offset = 0
    mem = (np.arange(w["x"].shape[0]) % 2) == 0
    l2, lab2, _ = svs.range_search(ix, q, radius, is_member=mem, make_iterator=_OracleBatchIterator)
    assert (lab2 % 2 == 0).all() and 0 < l2[-1] < lims[-1]
    # filtered search: k members of the filter, best first, == the first k members in iterator order
    member = (np.arange(w["x"].shape[0]) % 3) == 0
    res = svs.filtered_search(ix, q, 10, member, make_iterator=_OracleBatchIterator)
    full_ids, full_d = oracle.search(L2, w["x"], w["graph"], w["entry"], q, 200, 200)
    for i in range(len(q)):
        got = res.indices[i]
        assert member[got.astype(np.int64)].all() and len(set(got.tolist())) == 10
        assert (np.diff(res.distances[i]) >= 0).all()
        want = [v for v in full_ids[i].tolist() if member[v]][:10]
        assert len(set(want) & set(got.tolist())) >= 9
    # no member at all: padded (vamana_index_impl.h:203-208)
    res = svs.filtered_search(ix, q[:2], 5, np.zeros(w["x"].shape[0], bool), make_iterator=_OracleBatchIterator)
    assert (res.indices == np.iinfo(np.uint64).max).all() and np.isinf(res.distances).all()
    # svs_runtime_utils.h:411-434
    assert predict_further_processing(0, 0, 10, 50, 1000) == 50
    assert predict_further_processing(100, 2, 10, 50, 1000) == 400
    assert predict_further_processing(100, 2, 10, 50, 300) == 300
    assert predict_further_processing(3, 2, 3, 50, 300) == 1
    assert not should_stop_filtered_search(100, 0, 0.0) and should_stop_filtered_search(100, 0, 0.1)
    assert should_stop_filtered_search(100, 5, 0.1) and not should_stop_filtered_search(100, 20, 0.1)
