#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE'S OWN CODE (oracle/_ref: svs::index::vamana::greedy_search,
SearchBuffer, VisitedFilter and simd::generic_simd_op compiled from /root/reference/include).

Run in the build container (needs /root/reference and an AVX-512 host):  python tests/golden/make_golden.py
(`python tests/golden/make_golden.py iterator` regenerates only iterator.npz.)
The .npz files hold only inputs and expected outputs (data), no reference source.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle.oracle import COS, F16, F32, I8, IP, L2, U8, Oracle, Ref  # noqa: E402

PAIRS = [(F32, F32), (F32, F16), (F16, F32), (F16, F16), (F32, I8), (F32, U8), (I8, I8), (U8, U8)]
NAMES = {F32: "f32", F16: "f16", I8: "i8", U8: "u8"}


def gen(rng, dt, n, d):
    if dt == F32:
        return rng.standard_normal((n, d)).astype(np.float32)
    if dt == F16:
        return rng.standard_normal((n, d)).astype(np.float16)
    if dt == I8:
        return rng.integers(-128, 128, (n, d)).astype(np.int8)
    return rng.integers(0, 256, (n, d)).astype(np.uint8)


SCHEDULES = [[10] * 10, [20, 5, 50, 1, 0, 7, 100], [64, 64, 64], [400, 400]]


def make_iterator_golden():
    """BatchIterator known answers: the reference's SearchBuffer / greedy_search / EntryPointInitializer / std::sort driven
    with iterator.h's control flow (oracle/ref_driver.cpp svsref_iterate), on a set with duplicated vectors so that
    exactly tied distances occur in every buffer."""
    ref, orc = Ref(), Oracle()
    rng = np.random.default_rng(20260927)
    n, d, nq = 600, 40, 6
    a = np.random.default_rng(7).standard_normal((12, d)).astype(np.float32)
    base = (rng.standard_normal((n // 2, 12)).astype(np.float32) @ a + 0.1 * rng.standard_normal((n // 2, d)).astype(np.float32)).astype(np.float32)
    x = np.concatenate([base, base[rng.permutation(n // 2)]])[rng.permutation(n)]  # every vector appears twice
    q = (rng.standard_normal((nq, 12)).astype(np.float32) @ a + 0.1 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    graph, entry = orc.build_vamana(L2, x, R=16, L=32)
    out = dict(x=x, q=q, graph=graph, entry=entry, n_schedules=np.array(len(SCHEDULES)))
    for m, xv, tag in ((L2, x, "l2_f32"), (IP, x.astype(np.float16), "ip_f16")):
        for si, sched in enumerate(SCHEDULES):
            out[f"sched_{si}"] = np.array(sched, np.int64)
            for qi in range(nq):
                res, done = ref.iterate(m, xv, graph, entry, q[qi], sched)
                out[f"{tag}_{si}_{qi}_counts"] = np.array([len(r[0]) for r in res], np.int64)
                out[f"{tag}_{si}_{qi}_ids"] = np.concatenate([r[0] for r in res]) if res else np.zeros(0, np.uint32)
                out[f"{tag}_{si}_{qi}_dists"] = np.concatenate([r[1] for r in res]) if res else np.zeros(0, np.float32)
                out[f"{tag}_{si}_{qi}_done"] = np.array(done, np.uint8)
        # restart_next_search before calls 2 and 4
        rb = [0, 0, 1, 0, 1, 0]
        res, done = ref.iterate(m, xv, graph, entry, q[0], [15] * 6, restart_before=rb)
        out[f"{tag}_restart_ids"] = np.concatenate([r[0] for r in res])
        out[f"{tag}_restart_dists"] = np.concatenate([r[1] for r in res])
    # std::sort on sorted-with-ties and unsorted buffers (SearchBuffer::sort, search_buffer.h:408)
    for t in range(12):
        m_ = int(rng.integers(17, 400))
        vals = rng.integers(0, max(2, m_ // int(rng.integers(2, 8))), m_).astype(np.float32)
        greater = bool(t & 1)
        if t % 3:
            vals = np.sort(vals)[::-1 if greater else 1].copy()
        b = ref.buffer(m_, m_, greater)
        for i, v in enumerate(vals):
            b.push_back(i, float(v))
        b.sort()
        out[f"sort_{t}_in"] = vals
        out[f"sort_{t}_greater"] = np.array(int(greater))
        out[f"sort_{t}_ids"] = np.array([e[0] for e in b.state()], np.uint32)
    np.savez_compressed(os.path.join(HERE, "iterator.npz"), **out)
    print("iterator.npz", os.path.getsize(os.path.join(HERE, "iterator.npz")))


def make_flat_golden():
    """flat.npz: exhaustive search through the reference's own BulkInserter / HeapInserter (index/flat/inserters.h; oracle/_ref svsref_flat_search):
    a tie-free set (THE top-k: ids and distance bits are the answer) for three metrics / several type pairs, k up to 100 like the reference's
    groundtruth_*.ivecs, and a set in which every vector occurs four times (what libstdc++'s heap does with exact ties)."""
    ref = Ref()
    rng = np.random.default_rng(20260927)
    out = {}
    for dt in (F32, F16, I8):
        out[f"x_{NAMES[dt]}"] = gen(rng, dt, 700, 48)
        out[f"q_{NAMES[dt]}"] = gen(rng, dt, 6, 48)
    ks = [1, 10, 100]
    out["ks"] = np.array(ks)
    for qa, qb in [(F32, F32), (F32, F16), (F16, F16), (I8, I8), (F32, I8)]:
        for m in (L2, IP, COS):
            for k in ks:
                ids, dd = ref.flat_search(m, out[f"x_{NAMES[qb]}"], out[f"q_{NAMES[qa]}"], k)
                out[f"ids_{NAMES[qa]}_{NAMES[qb]}_{m}_{k}"] = ids
                out[f"dists_{NAMES[qa]}_{NAMES[qb]}_{m}_{k}"] = dd
    base = rng.standard_normal((50, 24)).astype(np.float32)
    out["x_tied"] = np.concatenate([base, base, base, base])[rng.permutation(200)]
    out["q_tied"] = rng.standard_normal((5, 24)).astype(np.float32)
    for k in (3, 10):
        ids, dd = ref.flat_search(L2, out["x_tied"], out["q_tied"], k)
        out[f"ids_tied_{k}"] = ids
        out[f"dists_tied_{k}"] = dd
    np.savez_compressed(os.path.join(HERE, "flat.npz"), **out)
    print("flat.npz", os.path.getsize(os.path.join(HERE, "flat.npz")))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "iterator":
        return make_iterator_golden()
    if len(sys.argv) > 1 and sys.argv[1] == "flat":
        return make_flat_golden()
    ref, orc = Ref(), Oracle()
    rng = np.random.default_rng(20260926)

    # ---- distance known answers (K1-K6) ----------------------------------------------------------------------
    out = {}
    for d in (96, 100, 223):
        for dt in (F32, F16, I8, U8):
            out[f"a_{NAMES[dt]}_{d}"] = gen(rng, dt, 5, d)
            out[f"b_{NAMES[dt]}_{d}"] = gen(rng, dt, 7, d)
        for qa, qb in PAIRS:
            for m in (L2, IP, COS):
                out[f"dist_{NAMES[qa]}_{NAMES[qb]}_{m}_{d}"] = ref.distance_batch(m, out[f"a_{NAMES[qa]}_{d}"], out[f"b_{NAMES[qb]}_{d}"])
    np.savez_compressed(os.path.join(HERE, "distances.npz"), **out)

    # ---- search known answers: one graph, all type pairs / metrics / buffer configs ---------------------------
    n, d, nq = 1000, 100, 24
    a = np.random.default_rng(7).standard_normal((16, d)).astype(np.float32)
    x = (rng.standard_normal((n, 16)).astype(np.float32) @ a + 0.1 * rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    q = (rng.standard_normal((nq, 16)).astype(np.float32) @ a + 0.1 * rng.standard_normal((nq, d)).astype(np.float32)).astype(np.float32)
    graph, entry = orc.build_vamana(L2, x, R=24, L=40)
    s = dict(graph=graph, entry=entry)
    scale = 127.0 / np.abs(x).max()
    variants = {
        F32: (x, q), F16: (x.astype(np.float16), q.astype(np.float16)),
        I8: (np.clip(np.round(x * scale), -128, 127).astype(np.int8), np.clip(np.round(q * scale), -128, 127).astype(np.int8)),
        U8: (np.clip(np.round(x * scale + 128), 0, 255).astype(np.uint8), np.clip(np.round(q * scale + 128), 0, 255).astype(np.uint8)),
    }
    for dt, (xv, qv) in variants.items():
        s[f"x_{NAMES[dt]}"] = xv
        s[f"q_{NAMES[dt]}"] = qv
    configs = [(10, 10, 10), (24, 24, 10), (12, 30, 8), (3, 3, 10), (64, 64, 20)]
    s["configs"] = np.array(configs)
    for qa, qb in PAIRS:
        for m in (L2, IP, COS):
            for ci, (w, c, k) in enumerate(configs):
                ids, dists, st = ref.search(m, variants[qb][0], graph, entry, variants[qa][1], k, w, c, stats=True)
                # the reference's own visited-set / multi-thread variants must agree (result-neutral)
                ids2, d2 = ref.search(m, variants[qb][0], graph, entry, variants[qa][1], k, w, c, visited_set=True, nthreads=3)
                assert (ids == ids2).all() and (dists.view(np.uint32) == d2.view(np.uint32)).all()
                key = f"{NAMES[qa]}_{NAMES[qb]}_{m}_{ci}"
                s["ids_" + key] = ids
                s["dists_" + key] = dists
                s["stats_" + key] = st
    np.savez_compressed(os.path.join(HERE, "search.npz"), **s)

    # ---- SearchBuffer op trace (split window/capacity, ties, duplicate ids) -------------------------------------
    t = {}
    for name, greater, w, c in (("less", False, 6, 10), ("greater", True, 5, 5)):
        rb = ref.buffer(w, c, greater)
        ops, snaps = [], []
        rb.push_back(7, 3.0)
        rb.sort()
        ops.append((0, 7, 3.0))
        for step in range(400):
            if not rb.done() and rng.random() < 0.25:
                i, dd = rb.next()
                ops.append((1, i, dd))
            else:
                i = int(rng.integers(0, 60))
                dd = float((i * 5) % 11)
                r = rb.insert(i, dd)
                ops.append((2, i, dd))
            st = rb.state()
            snap = np.full((c, 3), -1.0)
            for j, (ii, d2, v) in enumerate(st):
                snap[j] = (ii, d2, v)
            snaps.append(np.concatenate([[rb.size(), rb.best_unvisited(), float(rb.done())], snap.ravel()]))
        t[f"ops_{name}"] = np.array(ops, dtype=np.float64)
        t[f"snaps_{name}"] = np.array(snaps)
        t[f"cfg_{name}"] = np.array([w, c, int(greater)])
    np.savez_compressed(os.path.join(HERE, "buffer_trace.npz"), **t)
    for f in ("distances.npz", "search.npz", "buffer_trace.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)))
    make_iterator_golden()
    make_flat_golden()


if __name__ == "__main__":
    main()
