"""Pins the C restatement bit-for-bit against the reference's own code (oracle/_ref) and against the committed
golden vectors that were generated from it (tests/golden/make_golden.py).  No GPU needed."""
import os

import numpy as np
import pytest

from oracle.oracle import COS, F16, F32, I8, IP, L2, U8

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = {F32: "f32", F16: "f16", I8: "i8", U8: "u8"}
PAIRS = [(F32, F32), (F32, F16), (F16, F32), (F16, F16), (F32, I8), (F32, U8), (I8, I8), (U8, U8)]


def _bits_equal(a, b):
    return ((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all()


def test_golden_distances(oracle):
    g = np.load(os.path.join(GOLD, "distances.npz"))
    for d in (96, 100, 223):
        for qa, qb in PAIRS:
            for m in (L2, IP, COS):
                got = oracle.distance_batch(m, g[f"a_{NAMES[qa]}_{d}"], g[f"b_{NAMES[qb]}_{d}"])
                assert _bits_equal(got, g[f"dist_{NAMES[qa]}_{NAMES[qb]}_{m}_{d}"]), (d, qa, qb, m)


def test_golden_search(oracle):
    g = np.load(os.path.join(GOLD, "search.npz"))
    for qa, qb in PAIRS:
        for m in (L2, IP, COS):
            for ci, (w, c, k) in enumerate(g["configs"]):
                ids, dists, st = oracle.search(m, g[f"x_{NAMES[qb]}"], g["graph"], g["entry"], g[f"q_{NAMES[qa]}"], int(k), int(w), int(c), stats=True)
                key = f"{NAMES[qa]}_{NAMES[qb]}_{m}_{ci}"
                assert (ids == g["ids_" + key]).all(), key
                assert _bits_equal(dists, g["dists_" + key]), key
                assert (st[:, :2] == g["stats_" + key]).all(), key


def test_golden_buffer_trace(oracle):
    g = np.load(os.path.join(GOLD, "buffer_trace.npz"))
    for name in ("less", "greater"):
        w, c, greater = (int(v) for v in g[f"cfg_{name}"])
        b = oracle.buffer(w, c, bool(greater))
        ops, snaps = g[f"ops_{name}"], g[f"snaps_{name}"]
        b.push_back(int(ops[0][1]), float(ops[0][2]))
        b.sort()
        for (op, i, d), snap in zip(ops[1:], snaps):
            if op == 1:
                got = b.next()
                assert got == (int(i), float(d))
            else:
                b.insert(int(i), float(d))
            assert (b.size(), b.best_unvisited(), float(b.done())) == tuple(snap[:3])
            st = np.full((c, 3), -1.0)
            for j, e in enumerate(b.state()):
                st[j] = e
            assert (st.ravel() == snap[3:]).all()


@pytest.mark.parametrize("d", [1, 17, 64, 96, 100, 160, 223, 768])
def test_live_distances_vs_reference_headers(oracle, ref, d):
    rng = np.random.default_rng(d + 1)

    def gen(dt, n):
        if dt == F32:
            return (rng.standard_normal((n, d)) * 10.0 ** int(rng.integers(-3, 3))).astype(np.float32)
        if dt == F16:
            return rng.standard_normal((n, d)).astype(np.float16)
        if dt == I8:
            return rng.integers(-128, 128, (n, d)).astype(np.int8)
        return rng.integers(0, 256, (n, d)).astype(np.uint8)

    for qa, qb in PAIRS:
        a, b = gen(qa, 12), gen(qb, 40)
        for m in (L2, IP, COS):
            assert _bits_equal(oracle.distance_batch(m, a, b), ref.distance_batch(m, a, b)), (d, qa, qb, m)


def test_live_search_vs_reference_headers(oracle, ref, small_world):
    sw = small_world
    for m in (L2, IP, COS):
        for w, c, k in ((8, 8, 8), (30, 45, 10), (100, 100, 10)):
            a = oracle.search(m, sw["x"], sw["graph"], sw["entry"], sw["q"], k, w, c, stats=True)
            b = ref.search(m, sw["x"], sw["graph"], sw["entry"], sw["q"], k, w, c, nthreads=2, stats=True)
            assert (a[0] == b[0]).all() and _bits_equal(a[1], b[1]) and (a[2][:, :2] == b[2]).all()


def test_live_buffer_fuzz_vs_reference_headers(oracle, ref):
    rng = np.random.default_rng(11)
    for greater in (False, True):
        for w, c in ((5, 5), (8, 20), (32, 32), (1, 1)):
            ob, rb = oracle.buffer(w, c, greater), ref.buffer(w, c, greater)
            for _ in range(10):
                ob.clear(); rb.clear()
                ob.push_back(1000, 5.0); rb.push_back(1000, 5.0)
                ob.sort(); rb.sort()
                while not ob.done():
                    assert not rb.done()
                    assert ob.next() == rb.next()
                    for _ in range(int(rng.integers(0, 10))):
                        i = int(rng.integers(0, 150))
                        dd = float((i * 7) % 19)
                        assert ob.insert(i, dd) == rb.insert(i, dd)
                    assert ob.state() == rb.state() and ob.best_unvisited() == rb.best_unvisited()
                assert rb.done()


def test_filter_vs_reference_headers(oracle, ref):
    import ctypes as C

    f = (C.c_uint16 * 65536)()
    oracle.lib.svso_filter_reset(f)
    rf = ref.lib.svsref_filter_new()
    rng = np.random.default_rng(5)
    for key in rng.integers(0, 1 << 20, 5000):
        key = int(key)
        assert bool(oracle.lib.svso_filter_contains(f, key)) == bool(ref.lib.svsref_filter_contains(C.c_void_p(rf), key))
        assert bool(oracle.lib.svso_filter_emplace(f, key)) == bool(ref.lib.svsref_filter_emplace(C.c_void_p(rf), key))
    ref.lib.svsref_filter_free(C.c_void_p(rf))


def test_live_buffer_nan_distances_vs_reference_headers(oracle, ref):
    """NaN distances (cosine of a zero query): every comparison is false, lower_bound lands at the end and the duplicate scan
    of insert_inner (search_buffer.h:377-389) degenerates to an ids-only scan.  Mixed NaN / ordinary inserts, op by op."""
    rng = np.random.default_rng(0)

    def same(u, v):
        return u == v or (u != u and v != v)

    for trial in range(40):
        greater = bool(trial & 1)
        w = int(rng.integers(2, 40))
        c = w + int(rng.integers(0, 20))
        a, b = oracle.buffer(w, c, greater), ref.buffer(w, c, greater)
        for step in range(250):
            if not a.done() and rng.random() < 0.25:
                assert a.done() == b.done()
                x, y = a.next(), b.next()
                assert x[0] == y[0] and same(x[1], y[1])
            else:
                i = int(rng.integers(0, 60))
                dd = float("nan") if rng.random() < 0.5 else float(i % 7)
                assert a.insert(i, dd) == b.insert(i, dd), (trial, step)
            sa, sb = a.state(), b.state()
            assert len(sa) == len(sb)
            for u, v in zip(sa, sb):
                assert u[0] == v[0] and u[2] == v[2] and same(u[1], v[1]), (trial, step, u, v)
