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


def test_live_flat_search_vs_reference_inserters(oracle, ref):
    """FlatIndex::search keeps every query's k best in a HeapInserter (index/flat/inserters.h:76-116 through BulkInserter, flat.h:173) fed in
    data order and returns what std::sort_heap leaves.  Without ties that is THE top-k: oracle == reference, ids and distance bits.  With exact
    ties the reference's ids are whatever libstdc++'s heap leaves (an equal candidate replaces the heap's top, inserters.h:96-106); the oracle
    and the GPU path order ties by lower id -- distances are still identical row by row, ids can only differ inside a group of equal
    distance.  Both facts are pinned here against the reference's own inserter."""
    rng = np.random.default_rng(11)
    for m in (L2, IP, COS):
        for qa, qb in [(F32, F32), (F32, F16), (F16, F16), (I8, I8), (F32, U8)]:
            x = (rng.standard_normal((500, 40)) * (20 if qb in (I8, U8) else 1)).astype({F32: np.float32, F16: np.float16, I8: np.int8, U8: np.uint8}[qb])
            if qb == U8:
                x = np.abs(x.astype(np.int16)).astype(np.uint8)
            q = (rng.standard_normal((7, 40)) * (20 if qa == I8 else 1)).astype({F32: np.float32, F16: np.float16, I8: np.int8}[qa])
            oi, od = oracle.flat_search(m, x, q, 10)
            ri, rd = ref.flat_search(m, x, q, 10)
            assert _bits_equal(od, rd), (m, qa, qb)
            tie_free = np.array([len(np.unique(ref.distance_batch(m, q[i : i + 1], x)[0])) == 500 for i in range(len(q))])
            assert (oi[tie_free] == ri[tie_free].astype(np.uint32)).all(), (m, qa, qb)
    # exact ties: every vector four times, k cuts through a tie group
    base = rng.standard_normal((60, 24)).astype(np.float32)
    x = np.concatenate([base, base, base, base])[rng.permutation(240)]
    q = rng.standard_normal((9, 24)).astype(np.float32)
    differ = 0
    for k in (1, 3, 6, 10):
        oi, od = oracle.flat_search(L2, x, q, k)
        ri, rd = ref.flat_search(L2, x, q, k)
        assert _bits_equal(od, rd)                       # the same distances in the same order ...
        full = ref.distance_batch(L2, q, x)
        for i in range(len(q)):
            for ids in (oi[i].astype(np.int64), ri[i].astype(np.int64)):
                assert len(set(ids.tolist())) == k and (full[i][ids].view(np.uint32) == rd[i].view(np.uint32)).all()  # ... of real, distinct rows
            differ += int((oi[i].astype(np.int64) != ri[i].astype(np.int64)).any())
    assert differ > 0  # the heap really does not pick the lowest ids: a documented difference under exact ties


def test_live_vecs_readers_vs_reference(ref, tmp_path):
    """[fbi]vecs (core/io/vecs.h): this repo's readers / writer against the reference's get_dims (with its element-size deduction),
    VecsReader and VecsWriter, on float / int / byte files incl. the reference-held known_f32.fvecs, a one-vector file, and malformed files."""
    from scalablevectorsearch_amd import io

    rng = np.random.default_rng(5)
    f = rng.standard_normal((13, 7)).astype(np.float32)
    io.write_fvecs(str(tmp_path / "a.fvecs"), f)
    assert ref.vecs_dims(str(tmp_path / "a.fvecs")) == (13, 7) and ref.vecs_dims(str(tmp_path / "a.fvecs"), 4) == (13, 7)
    assert (ref.vecs_read(str(tmp_path / "a.fvecs"), np.float32).view(np.uint32) == f.view(np.uint32)).all()
    assert (io.read_fvecs(str(tmp_path / "a.fvecs")).view(np.uint32) == f.view(np.uint32)).all()
    iv = rng.integers(0, 2**31, (5, 100)).astype(np.uint32)   # the layout of the reference's groundtruth_*.ivecs (1000 x 100)
    ref.vecs_write_u32(str(tmp_path / "b.ivecs"), iv)          # the reference WRITES, we read
    assert (io.read_ivecs(str(tmp_path / "b.ivecs")).view(np.uint32) == iv).all() and ref.vecs_dims(str(tmp_path / "b.ivecs")) == (5, 100)
    bv = rng.integers(0, 256, (9, 16)).astype(np.uint8)
    raw = np.empty((9, 4 + 16), np.uint8)
    raw[:, :4] = np.array([16], np.uint32).view(np.uint8)
    raw[:, 4:] = bv
    raw.tofile(str(tmp_path / "c.bvecs"))
    assert ref.vecs_dims(str(tmp_path / "c.bvecs")) == (9, 16)   # element size 1 deduced
    assert (ref.vecs_read(str(tmp_path / "c.bvecs"), np.uint8) == bv).all() and (io.read_bvecs(str(tmp_path / "c.bvecs")) == bv).all()
    known = os.path.join(GOLD, "known_f32.fvecs")
    assert ref.vecs_dims(known) is not None
    assert (ref.vecs_read(known, np.float32).view(np.uint32) == io.read_fvecs(known).view(np.uint32)).all()
    io.write_fvecs(str(tmp_path / "one.fvecs"), f[:1])
    assert ref.vecs_dims(str(tmp_path / "one.fvecs")) == (1, 7) and io.read_fvecs(str(tmp_path / "one.fvecs")).shape == (1, 7)
    open(tmp_path / "bad.fvecs", "wb").write(open(tmp_path / "a.fvecs", "rb").read()[:-3])  # truncated: both sides refuse
    assert ref.vecs_dims(str(tmp_path / "bad.fvecs")) is None
    with pytest.raises(Exception):
        io.read_fvecs(str(tmp_path / "bad.fvecs"))


def test_golden_flat_search(oracle):
    """tests/golden/flat.npz (generated from the reference's own inserters): where a query's distances are tie-free the oracle returns the same
    ids and distance bits, k = 1 / 10 / 100; with exact ties the distances agree and every returned id is a real row at that distance."""
    g = np.load(os.path.join(GOLD, "flat.npz"))
    for qa, qb in [(F32, F32), (F32, F16), (F16, F16), (I8, I8), (F32, I8)]:
        x, q = g[f"x_{NAMES[qb]}"], g[f"q_{NAMES[qa]}"]
        for m in (L2, IP, COS):
            full = oracle.distance_batch(m, q, x)
            for k in (int(v) for v in g["ks"]):
                oi, od = oracle.flat_search(m, x, q, k)
                gi, gd = g[f"ids_{NAMES[qa]}_{NAMES[qb]}_{m}_{k}"], g[f"dists_{NAMES[qa]}_{NAMES[qb]}_{m}_{k}"]
                assert _bits_equal(od, gd), (qa, qb, m, k)
                for i in range(len(q)):
                    if len(np.unique(full[i])) == len(full[i]):
                        assert (oi[i] == gi[i].astype(np.uint32)).all(), (qa, qb, m, k, i)
    for k in (3, 10):
        oi, od = oracle.flat_search(L2, g["x_tied"], g["q_tied"], k)
        assert _bits_equal(od, g[f"dists_tied_{k}"])
        full = oracle.distance_batch(L2, g["q_tied"], g["x_tied"])
        for i in range(len(od)):
            for ids in (oi[i].astype(np.int64), g[f"ids_tied_{k}"][i].astype(np.int64)):
                assert len(set(ids.tolist())) == k and (full[i][ids].view(np.uint32) == od[i].view(np.uint32)).all()


SCALAR_PAIRS = [(I8, U8), (I8, F32), (I8, F16), (U8, I8), (U8, F32), (U8, F16)]


def _cast(v, dt, rng):
    if dt == F32:
        return v.astype(np.float32)
    if dt == F16:
        return v.astype(np.float16)
    if dt == I8:
        return rng.integers(-128, 128, v.shape).astype(np.int8)
    return rng.integers(0, 256, v.shape).astype(np.uint8)


@pytest.mark.parametrize("d", [1, 7, 96, 100, 223])
def test_scalar_loops_follow_the_compiler(oracle, ref, d):
    """The six pairs the reference answers with generic_l2 / generic_ip / generic_cosine_similarity: the C restatement hard-codes
    one fmaf per element; oracle/ref_scalar_loops.cpp holds the three loop bodies and is compiled with the flags of the reference's kernel
    translation unit (cmake/multi-arch.cmake:17-35: -O3 -march=cascadelake, the compiler's own -ffp-contract default) -- so whether GCC
    contracts them is observed here, not assumed.  float16 elements go through the reference's own Float16 class (compiled from
    lib/float16.h): its scalar conversion flushes denormals and gives inf / NaN finite values, unlike the SIMD kernels' vcvtph2ps."""
    rng = np.random.default_rng(1000 + d)
    for qa, qb in SCALAR_PAIRS:
        a = _cast(rng.standard_normal((6, d)) * 3.0, qa, rng)
        b = _cast(rng.standard_normal((9, d)) * 3.0, qb, rng)
        if qb == F16:  # float16 rows with denormals, the largest finite value, an infinity and a NaN pattern in them
            bb = b.view(np.uint16).copy()
            bb[0, 0] = 0x0001
            bb[1, d - 1] = 0x83FF
            if d > 1:
                bb[2, 0] = 0x7BFF
                bb[3, 1] = 0x7C00
                bb[4, 0] = 0xFE01
            b = bb.view(np.float16)
        norms = np.array([oracle.norm(a[i]) for i in range(a.shape[0])], np.float32)
        for m in (L2, IP, COS):
            got = oracle.distance_batch(m, a, b)
            want = ref.scalar_loop_batch(m, a, b, norms)
            assert _bits_equal(got, want), (d, qa, qb, m, got.ravel()[:4], want.ravel()[:4])


def test_float16_query_norm_uses_the_scalar_conversion(oracle, ref):
    """norm() of a float16 query (distance_core.h:45-58, called by DistanceCosineSimilarity::fix_argument, cosine.h:117-119) converts every
    element through Float16::operator float -- denormals flushed, inf / NaN patterns finite -- while the SIMD kernels load the same elements
    exactly (vcvtph2ps).  Both sides of that are pinned here: the norm against the reference's class, the cosine of a float16 query with
    denormal, inf and NaN-pattern elements against the reference's kernels."""
    rng = np.random.default_rng(77)
    for d in (16, 96, 100):
        q = (rng.standard_normal((5, d)) * 1e-4).astype(np.float16).view(np.uint16).copy()
        q[0, :4] = (0x0001, 0x03FF, 0x8200, 0x0000)  # denormals
        q[1, 0] = 0x7BFF
        q[2, 1] = 0x7C00  # +inf: 65536 to the scalar conversion
        q[3, 2] = 0xFC01  # a NaN pattern: a finite value to the scalar conversion
        q = q.view(np.float16)
        for i in range(q.shape[0]):
            assert np.float32(oracle.norm(q[i])).view(np.uint32) == np.float32(ref.norm(q[i])).view(np.uint32), (d, i)
        x = rng.standard_normal((7, d)).astype(np.float32)
        for xb in (x, x.astype(np.float16)):
            assert _bits_equal(oracle.distance_batch(COS, q, xb), ref.distance_batch(COS, q, xb)), d
    # the conversion itself, all 65 536 patterns, against the reference's class (through the one-element L2 loop: (0 - x)^2 = x * x exactly
    # representable or not, the comparison is of bits)
    allh = np.arange(65536, dtype=np.uint32).astype(np.uint16).view(np.float16).reshape(-1, 1)
    zero = np.zeros((1, 1), np.int8)
    got = oracle.distance_batch(IP, np.ones((1, 1), np.int8), allh)  # 1 * float(x) + 0 = float(x)
    want = ref.scalar_loop_batch(IP, np.ones((1, 1), np.int8), allh, [0.0])
    assert _bits_equal(got, want) and zero.size == 1
