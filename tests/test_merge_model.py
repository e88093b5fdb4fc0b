"""The batched merge insert of the search kernel (csrc/svsx_device.h wave_merge_insert), as an algorithm, against the oracle's SearchBuffer
(index/vamana/search_buffer.h:342-403) on the CPU: a pass's candidates merged at once -- old entry i moves to i + #(new strictly better),
new entry j goes to #(old not worse) + its rank among the new (ties: adjacency order), whatever lands at or beyond the capacity is dropped,
best_unvisited = min(old, position of the best new entry) -- equals inserting them one after the other, and the round-3 tie test ("the last
old entry that stays before the candidate has the candidate's distance bits") fires whenever a candidate duplicates a buffer entry, so the
merge is only ever used where no duplicate can be involved.  The GPU fuzz suites check the HIP code itself; this pins the reasoning."""
import numpy as np
import pytest


def _merge(buf, cands, capacity, best_unvisited):
    """buf: list of (dist, id, visited) sorted; cands: list of (dist, id) in adjacency order.  Returns (new buffer, new best_unvisited, tie)."""
    old_d = np.array([e[0] for e in buf], np.float32)
    tie = False
    placed = []
    for j, (dj, idj) in enumerate(cands):
        le = int((~(np.float32(dj) < old_d)).sum()) if len(buf) else 0          # old entries that stay before j (upper-bound position)
        if le > 0 and old_d[le - 1].view(np.uint32) == np.float32(dj).view(np.uint32):
            tie = True
        rank = sum(1 for jj, (dd, _) in enumerate(cands) if (np.float32(dd), jj) < (np.float32(dj), j))
        placed.append((le + rank, (np.float32(dj), idj, False)))
    out = [None] * (len(buf) + len(cands))
    for i, e in enumerate(buf):
        c = sum(1 for (dj, _) in cands if np.float32(dj) < e[0])
        out[i + c] = e
    for pos, e in placed:
        assert out[pos] is None
        out[pos] = e
    assert all(e is not None for e in out)
    minpos = min(p for p, _ in placed) if placed else 1 << 30
    return out[:capacity], min(best_unvisited, minpos), tie


@pytest.mark.parametrize("seed", range(6))
def test_merge_equals_sequential_inserts_and_ties_cover_duplicates(oracle, seed):
    rng = np.random.default_rng(seed)
    for trial in range(60):
        capacity = int(rng.integers(4, 40))
        window = int(rng.integers(1, capacity + 1))
        b = oracle.buffer(window, capacity)
        b.clear()
        # a buffer state reached by sequential inserts (some entries visited), distances drawn from a small set so that exact ties are common
        pool = rng.random(200).astype(np.float32)
        if trial % 3 == 0:  # every third trial: distances from a coarse grid, so that exact ties between DIFFERENT ids are common
            pool = np.round(pool * 8) / 8
        next_id = 0
        for _ in range(int(rng.integers(0, capacity + 5))):
            b.insert(next_id, float(pool[rng.integers(0, 200)]))
            next_id += 1
        for _ in range(int(rng.integers(0, 4))):
            if not b.done():
                b.next()
        state = b.state()
        buf = [(np.float32(d), i, v) for (i, d, v) in state]
        worst = buf[-1][0] if len(buf) == capacity else None
        # one pass of candidates: new ids, plus (sometimes) a duplicate of a buffer entry with its own distance
        cands = []
        dup_used = False
        for _ in range(int(rng.integers(1, 9))):
            if buf and rng.random() < 0.04:
                e = buf[int(rng.integers(0, len(buf)))]
                cands.append((e[0], e[1]))
                dup_used = True
            else:
                cands.append((np.float32(pool[rng.integers(0, 200)]), next_id))
                next_id += 1
        # can_skip (search_buffer.h:338-340) against the worst entry at the START of the pass, as the kernel's pass does
        cands = [c for c in cands if worst is None or not (worst < c[0])]
        if not cands:
            continue
        merged, bu, tie = _merge(buf, cands, capacity, b.best_unvisited())
        if dup_used and any(c[1] in {e[1] for e in buf} for c in cands):
            assert tie, "a duplicate of a buffer entry must trip the tie test"
        for (dj, idj) in cands:
            b.insert(idj, float(dj))
        if tie:
            continue  # the kernel takes the sequential path for this pass
        want = b.state()
        assert [(i, np.float32(d)) for (i, d, v) in want] == [(e[1], e[0]) for e in merged], (seed, trial)
        assert [v for (i, d, v) in want] == [e[2] for e in merged]
        assert b.best_unvisited() == min(bu, len(merged)) or b.best_unvisited() == bu
