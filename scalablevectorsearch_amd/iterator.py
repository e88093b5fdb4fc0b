"""Batched BatchIterator and the two searches the reference's runtime builds on it.

Mirrors (host side, over the C ABI `svsx_iterator_*` of include/svsx.h):
  svs::index::vamana::BatchIterator                index/vamana/iterator.h:81-339
  VamanaIndex::make_batch_iterator                 index/vamana/index.h:936
  runtime filtered search (IDFilter)               bindings/cpp/src/vamana_index_impl.h:140-218
  runtime range_search                             bindings/cpp/src/vamana_index_impl.h:220-326
One iterator object drives one independent reference-style iterator PER QUERY of the batch; every next() is one
kernel launch that resumes all live queries from the search buffers kept in HBM.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .index import QueryResult, VamanaSearchParameters, dtype_code

SKIP = 0xFFFFFFFF  # per-query batch size meaning "leave this query untouched"
UNSIGNED_INTEGER_PLACEHOLDER = (1 << 64) - 1  # lib/preprocessor.h:177


class BatchIterator:
    """`index.batch_iterator(queries)`; `next(batch_size)` yields, per query, the next best neighbours not yet yielded."""

    def __init__(self, index, queries, extra_search_buffer_capacity: int | None = None):
        queries = np.ascontiguousarray(np.atleast_2d(queries))
        if queries.shape[1] != index.d:  # detail::checkdims (iterator.h:66-74)
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Incompatible dimensions. Query has {queries.shape[1]} while the index expects {index.d}.")
        self.index = index
        self.nq = queries.shape[0]
        self._qdtype = queries.dtype
        self._h = C.c_void_p()
        extra = UNSIGNED_INTEGER_PLACEHOLDER if extra_search_buffer_capacity is None else int(extra_search_buffer_capacity)
        _lib.check(_lib.lib().svsx_iterator_create(index._h, queries.ctypes.data, dtype_code(queries), self.nq, extra, C.byref(self._h)))
        self._ids = np.zeros((self.nq, 0), np.uint64)
        self._dists = np.zeros((self.nq, 0), np.float32)
        self._counts = np.zeros(self.nq, np.uint32)
        self._extra = 100 if extra_search_buffer_capacity is None else int(extra_search_buffer_capacity)  # ITERATOR_EXTRA_BUFFER_CAPACITY_DEFAULT
        self._cap_bound = self._extra

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().svsx_iterator_destroy(h)
            except Exception:  # interpreter shutdown
                pass
            self._h = None

    def next(self, batch_size: int, batch_sizes=None):
        """BatchIterator::next for every query.  Returns (ids uint64 [nq, batch_size], dists [nq, batch_size], counts [nq]);
        row i holds counts[i] results (best first), the rest is padded with all-ones / +inf."""
        batch_size = int(batch_size)
        ids = np.empty((self.nq, batch_size), np.uint64)
        dists = np.empty((self.nq, batch_size), np.float32)
        counts = np.empty(self.nq, np.uint32)
        bs = None
        if batch_sizes is not None:
            bs = np.ascontiguousarray(batch_sizes, dtype=np.uint32)
            if bs.shape != (self.nq,):
                raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "batch_sizes must have one entry per query")
            if np.any((bs != SKIP) & (bs > batch_size)):
                raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "per-query batch sizes must not exceed batch_size")
        _lib.check(_lib.lib().svsx_iterator_next(self._h, batch_size, None if bs is None else bs.ctypes.data, ids.ctypes.data,
                                                 dists.ctypes.data, counts.ctypes.data))
        self._ids, self._dists, self._counts = ids, dists, counts
        self._cap_bound += batch_size
        return ids, dists, counts

    def next_device(self, batch_size: int, d_ids64, d_dists, d_counts, d_batch_sizes=None, stream=None):
        """Same with torch CUDA tensors (ids int64/uint64 [nq, batch_size], dists f32, counts int32/uint32 [nq])."""
        _lib.check(_lib.lib().svsx_iterator_next_device(
            self._h, int(batch_size), None if d_batch_sizes is None else d_batch_sizes.data_ptr(), d_ids64.data_ptr(),
            d_dists.data_ptr(), d_counts.data_ptr(), None if stream is None else stream.cuda_stream))

    def contents(self, i: int):
        """Current batch of query i (BatchIterator::contents, iterator.h:221)."""
        c = int(self._counts[i])
        return self._ids[i, :c], self._dists[i, :c]

    def size(self) -> np.ndarray:
        return self._counts.copy()

    def state(self) -> np.ndarray:
        meta = np.empty((self.nq, 8), np.uint32)
        _lib.check(_lib.lib().svsx_iterator_state(self._h, meta.ctypes.data))
        return meta

    def done(self) -> np.ndarray:  # iterator.h:230
        m = self.state()
        return (m[:, 4] == self.index.n) | ((m[:, 6] & 2) != 0)

    def batch_number(self) -> np.ndarray:  # iterator.h:227
        return self.state()[:, 5].copy()

    def parameters_for_current_iteration(self):  # iterator.h:236-245: (search window, capacity) per query
        m = self.state()
        return m[:, 2].copy(), m[:, 3].copy()

    def restart_next_search(self) -> None:
        _lib.check(_lib.lib().svsx_iterator_restart_next_search(self._h))

    def update(self, queries) -> None:
        queries = np.ascontiguousarray(np.atleast_2d(queries))
        if queries.shape != (self.nq, self.index.d) or queries.dtype != self._qdtype:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Incompatible dimensions. Query has {queries.shape[1]} while the index expects {self.index.d}.")
        _lib.check(_lib.lib().svsx_iterator_update(self._h, queries.ctypes.data))
        self._counts[:] = 0
        self._cap_bound = self._extra


def predict_further_processing(processed, hits, goal, hint, max_value):
    """bindings/cpp/src/svs_runtime_utils.h:411-419 (float arithmetic as there)."""
    if hits == 0 or hits >= goal:
        return min(int(hint), int(max_value))
    batch = np.float32(goal - hits) * np.float32(processed) / np.float32(hits)
    return min(max(int(batch), 1), int(max_value))


def should_stop_filtered_search(total_checked, found, filter_stop):
    """svs_runtime_utils.h:424-434"""
    if filter_stop <= 0 or total_checked == 0:
        return False
    if found == 0:
        return True
    return np.float32(found) / np.float32(total_checked) < np.float32(filter_stop)


class _MT19937:
    """std::mt19937 (the 32-bit Mersenne Twister with its standard seeding), for the mirror of sample_filter_hits below"""

    def __init__(self, seed: int):
        self.mt = [0] * 624
        self.mt[0] = seed & 0xFFFFFFFF
        for i in range(1, 624):
            self.mt[i] = (1812433253 * (self.mt[i - 1] ^ (self.mt[i - 1] >> 30)) + i) & 0xFFFFFFFF
        self.idx = 624

    def __call__(self) -> int:
        if self.idx >= 624:
            mt = self.mt
            for i in range(624):
                y = (mt[i] & 0x80000000) | (mt[(i + 1) % 624] & 0x7FFFFFFF)
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ (0x9908B0DF if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]
        self.idx += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        return y & 0xFFFFFFFF


def _uniform_below(rng, n: int) -> int:
    """std::uniform_int_distribution<size_t>(0, n - 1) over a 32-bit generator as libstdc++ >= 11 draws it for n <= 2^32 (bits/uniform_int_dist.h:
    Lemire's nearly divisionless method on a 64-bit product) -- the library itself uses the std classes; tests/test_host_logic.py checks this
    restatement against them"""
    if n > (1 << 32):
        raise NotImplementedError("more than 2^32 ids: not restated (the library uses the standard classes directly)")
    if n == (1 << 32):
        return rng()
    product = rng() * n
    low = product & 0xFFFFFFFF
    if low < n:
        threshold = ((1 << 32) - n) % n
        while low < threshold:
            product = rng() * n
            low = product & 0xFFFFFFFF
    return product >> 32


def sample_filter_hits(member, total_ids: int, sample_size: int = 200):
    """svs_runtime_utils.h:445-471 with is_valid == true: (checked, hits) over ids drawn by std::mt19937(42)"""
    if total_ids == 0:
        return 0, 0
    target = min(sample_size, total_ids)
    rng = _MT19937(42)
    hits = checked = 0
    for _ in range(target * 4):
        if checked >= target:
            break
        hits += int(bool(member[_uniform_below(rng, total_ids)]))
        checked += 1
    return checked, hits


def sample_size_for_filter_stop(filter_stop: float) -> int:
    """svs_runtime_utils.h:477-482"""
    return 200 if filter_stop <= 0 else max(200, int(np.float32(1.0) / np.float32(filter_stop)))


def filtered_search(index, queries, k: int, is_member, search_parameters: VamanaSearchParameters | None = None,
                    filter_stop: float = 0.0, extra_search_buffer_capacity: int | None = None, make_iterator=None,
                    filter_estimate_batch: bool = True) -> QueryResult:
    """Runtime VamanaIndex::search with an IDFilter (vamana_index_impl.h:140-218): per query, pull adaptive batches from the iterator until
    k members of the filter are found.  filter_estimate_batch (default True, like the runtime, VamanaIndex.search_filtered and the C++ facade): the sampled hit rate decides whether to search at
    all and the first batch size (:153-175).  `is_member`: boolean array over ids.  Unfilled slots: id all-ones, distance +inf."""
    queries = np.ascontiguousarray(np.atleast_2d(queries))
    nq = queries.shape[0]
    member = np.asarray(is_member, dtype=bool)
    sp = search_parameters or index.get_search_parameters()
    sws = sp.buffer_config.get_search_window_size()
    max_batch = index.size()
    id_offset = int(getattr(index, "id_offset", 0))  # `is_member` is indexed by LOCAL id (label - id_offset), as svsx_search_filtered's bitmap is
    it = (make_iterator or BatchIterator)(index, queries, extra_search_buffer_capacity)
    ids = np.full((nq, k), np.iinfo(np.uint64).max, np.uint64)
    dists = np.full((nq, k), np.inf, np.float32)
    found = np.zeros(nq, np.int64)
    checked = np.zeros(nq, np.int64)
    initial = max(k, sws)
    if filter_estimate_batch:
        sampled, sample_hits = sample_filter_hits(member, max_batch, sample_size_for_filter_stop(filter_stop))
        if should_stop_filtered_search(sampled, sample_hits, filter_stop):
            return QueryResult(ids, dists)
        initial = predict_further_processing(sampled, sample_hits, k, initial, max_batch)
    batch = np.full(nq, initial, np.int64)
    active = np.ones(nq, bool)
    while active.any():
        for i in np.flatnonzero(active):
            batch[i] = predict_further_processing(checked[i], found[i], k, batch[i], max_batch)
        per_query = np.where(active, batch, SKIP).astype(np.uint32)
        step = batch
        b_ids, b_dists, counts = it.next(int(step[active].max()), per_query)
        done = it.done()
        for i in np.flatnonzero(active):
            c = int(counts[i])
            checked[i] += c
            for j in range(c):
                if member[int(b_ids[i, j]) - id_offset]:
                    ids[i, found[i]] = b_ids[i, j]
                    dists[i, found[i]] = b_dists[i, j]
                    found[i] += 1
                    if found[i] == k:
                        break
            if should_stop_filtered_search(checked[i], found[i], filter_stop):
                found[i] = 0
                ids[i, :] = np.iinfo(np.uint64).max
                dists[i, :] = np.inf
                active[i] = False
            elif found[i] >= k or done[i]:
                active[i] = False
    return QueryResult(ids, dists)


def range_search(index, queries, radius: float, search_parameters: VamanaSearchParameters | None = None, is_member=None,
                 extra_search_buffer_capacity: int | None = None, make_iterator=None):
    """Runtime VamanaIndex::range_search (vamana_index_impl.h:220-326): everything closer than `radius` the iterator
    reaches before its first out-of-range neighbour.  Returns FAISS-style (lims [nq+1], labels, distances)."""
    if radius <= 0:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "radius must be greater than 0")
    queries = np.ascontiguousarray(np.atleast_2d(queries))
    nq = queries.shape[0]
    sp = search_parameters or index.get_search_parameters()
    if hasattr(index, "d") and queries.shape[1] != index.d:  # detail::checkdims (index/vamana/iterator.h): the C entry point only sees a pointer
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Incompatible dimensions. Query has {queries.shape[1]} while the index expects {index.d}.")
    # `is_member` is indexed by LOCAL id (label - index.id_offset, position in the index's own arrays) on both paths below
    id_offset = int(getattr(index, "id_offset", 0))
    if make_iterator is None and extra_search_buffer_capacity is None and hasattr(index, "_h"):
        # the library's own loop (svsx_range_search, include/svsx.h); the Python loop below is its mirror and runs over any iterator class
        import ctypes as C

        from .index import dtype_code

        out = {}

        def allocate(_user, counts, n, storage):
            c = np.array([counts[i] for i in range(n)], np.int64)
            out["lims"] = np.concatenate([[0], np.cumsum(c)]).astype(np.int64)
            out["labels"] = np.empty(int(c.sum()), np.uint64)
            out["distances"] = np.empty(int(c.sum()), np.float32)
            storage[0].labels = out["labels"].ctypes.data
            storage[0].distances = out["distances"].ctypes.data

        bitmap = None
        if is_member is not None:
            bits = np.zeros((index.size() + 63) // 64 * 64, np.uint8)
            bits[: index.size()] = np.asarray(is_member, dtype=bool)[: index.size()]
            bitmap = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64)
        spc = sp._c()
        _lib.check(_lib.lib().svsx_range_search(index._h, queries.ctypes.data, dtype_code(queries), nq, float(radius), C.byref(spc),
                                               None if bitmap is None else bitmap.ctypes.data, _lib.RESULTS_ALLOCATOR(allocate), None))
        if "lims" not in out:  # nq == 0: the library returns SUCCESS without calling the allocator, as the reference does
            return np.zeros(nq + 1, np.int64), np.empty(0, np.uint64), np.empty(0, np.float32)
        return out["lims"], out["labels"], out["distances"]
    batch = max(sp.buffer_config.get_search_window_size(), 10)
    greater = int(index.distance.metric) != 0
    it = (make_iterator or BatchIterator)(index, queries, extra_search_buffer_capacity)
    member = None if is_member is None else np.asarray(is_member, dtype=bool)
    res = [[] for _ in range(nq)]
    active = np.ones(nq, bool)
    while active.any():
        per_query = np.where(active, batch, SKIP).astype(np.uint32)
        b_ids, b_dists, counts = it.next(batch, per_query)
        done = it.done()
        for i in np.flatnonzero(active):
            in_range = True
            for j in range(int(counts[i])):
                dd = b_dists[i, j]
                in_range = bool(dd > radius) if greater else bool(dd < radius)
                if not in_range:
                    break
                if member is None or member[int(b_ids[i, j]) - id_offset]:
                    res[i].append((int(b_ids[i, j]), dd))
            if not in_range or done[i]:
                active[i] = False
    lims = np.zeros(nq + 1, np.int64)
    lims[1:] = np.cumsum([len(r) for r in res])
    labels = np.array([p[0] for r in res for p in r], np.uint64)
    distances = np.array([p[1] for r in res for p in r], np.float32)
    return lims, labels, distances
