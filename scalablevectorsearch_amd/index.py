"""Host-side mirror of svs::index::vamana::VamanaIndex for the batch-search path, over the C ABI.

Reference interfaces mirrored (paths relative to the reference root):
  VamanaIndex::search(QueryResultView, queries, VamanaSearchParameters, cancel)   index/vamana/index.h:564-611
  VamanaSearchParameters                                                           index/vamana/search_params.h:28-61
  SearchBufferConfig                                                               index/vamana/search_buffer.h:39-96
  QueryResult<size_t>                                                              core/query_result.h:53-172
  k_recall_at_n                                                                    core/recall.h:47-58
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .distance import as_functor, dtype_code


class SearchBufferConfig:
    """search_buffer.h:39-96: invariant search_window_size <= total_capacity (throws ANNException otherwise)."""

    def __init__(self, search_window_size: int = 0, total_capacity: int | None = None):
        total_capacity = search_window_size if total_capacity is None else total_capacity
        if search_window_size > total_capacity:
            raise _lib.SvsxError(
                _lib.INVALID_ARGUMENT,
                f"Improper configuration for search buffer! Effective size ({search_window_size}) cannot be less than capacity ({total_capacity}).",
            )
        self.search_window_size = int(search_window_size)
        self.total_capacity = int(total_capacity)

    def get_search_window_size(self) -> int:
        return self.search_window_size

    def get_total_capacity(self) -> int:
        return self.total_capacity

    def __eq__(self, o):
        return isinstance(o, SearchBufferConfig) and (self.search_window_size, self.total_capacity) == (o.search_window_size, o.total_capacity)

    def __repr__(self):
        return f"SearchBufferConfig({self.search_window_size}, {self.total_capacity})"


@dataclass
class VamanaSearchParameters:
    """search_params.h:28-61.  prefetch_* are accepted and ignored (occupancy hides latency on the GPU)."""

    buffer_config: SearchBufferConfig = field(default_factory=SearchBufferConfig)
    search_buffer_visited_set: bool = False
    prefetch_lookahead: int = 4
    prefetch_step: int = 1

    def _c(self) -> _lib.SearchParams:
        return _lib.SearchParams(self.buffer_config.search_window_size, self.buffer_config.total_capacity,
                                 int(self.search_buffer_visited_set), self.prefetch_lookahead, self.prefetch_step)


@dataclass
class QueryResult:
    """core/query_result.h:53-172: row-major nq x k ids (size_t) and f32 distances."""

    indices: np.ndarray
    distances: np.ndarray

    def n_queries(self) -> int:
        return self.indices.shape[0]

    def n_neighbors(self) -> int:
        return self.indices.shape[1]


def k_recall_at_n(groundtruth: np.ndarray, results: np.ndarray, k: int = 0, n: int = 0) -> float:
    """core/recall.h:47-58,181-185: |gt[:k] ∩ res[:n]| summed over queries / (k * nq); k, n default to all columns."""
    groundtruth = np.asarray(groundtruth)
    results = np.asarray(results.indices if isinstance(results, QueryResult) else results)
    nmax = results.shape[1]
    k = k or nmax  # recall.h:68-69: defaults to n-recall@n ("not a typo")
    n = n or nmax
    if groundtruth.shape[0] != results.shape[0]:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Groundtruth contains {groundtruth.shape[0]} points while the result has {results.shape[0]} points!")
    if k > n:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Argument k ({k}) must be less than n ({n})")
    if k > groundtruth.shape[1]:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Argument k ({k}) must be less than groundtruth entries ({groundtruth.shape[1]})")
    if n > nmax:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Argument n ({n}) must be less than result entries ({nmax})")
    hits = 0
    for g, r in zip(groundtruth[:, :k].astype(np.int64), results[:, :n].astype(np.int64)):
        hits += np.intersect1d(g, r).size
    return hits / (k * groundtruth.shape[0])


class VamanaIndex:
    """A read-only Vamana index resident in one GPU's HBM.

    ``graph`` uses the SVS row layout (core/graph/graph.h:32-48): ``graph[i, 0]`` = degree,
    ``graph[i, 1:1+deg]`` = neighbour ids.  ``data`` is row-major n x d (float32/float16/int8/uint8);
    ``sq=(scale, bias)`` declares int8/uint8 data to be SQDataset codes (quantization/scalar/scalar.h).
    """

    def __init__(self, graph, data, distance, entry_points, device: int = 0, sq=None, id_offset: int = 0, num_threads: int = 1):
        self.distance = as_functor(distance)
        data = np.ascontiguousarray(data)
        graph = np.ascontiguousarray(graph, dtype=np.uint32)
        entry = np.ascontiguousarray(np.atleast_1d(entry_points), dtype=np.uint32)
        if data.ndim != 2 or graph.ndim != 2 or graph.shape[0] != data.shape[0]:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "Wrong sizes!")  # index.h:420-422
        desc = _lib.IndexDesc(
            int(self.distance.metric), dtype_code(data), data.shape[0], data.shape[1], data.ctypes.data, 0,
            graph.ctypes.data, graph.shape[1], graph.shape[1] - 1, entry.ctypes.data, entry.size,
            int(sq is not None), float(sq[0]) if sq else 1.0, float(sq[1]) if sq else 0.0, device, id_offset,
        )
        self._h = C.c_void_p()
        self._keep = None
        _lib.check(_lib.lib().svsx_index_create(C.byref(desc), C.byref(self._h)))
        self.device = device
        self.id_offset = int(id_offset)
        self.n, self.d = data.shape
        self.data_dtype = data.dtype
        self.entry_points = entry.copy()
        # construct_default_search_parameters (index.h:243-250) leaves window 0 -> raised to k at search time
        self._params = VamanaSearchParameters()

    @classmethod
    def from_device(cls, d_graph, d_data, distance, d_entry_points, n: int, d: int, data_dtype, device: int = 0, sq=None, id_offset: int = 0):
        """Adopt torch tensors already in the device layout (svsx_index_create_device_view): data rows padded to
        16 elements, graph rows of 16k ids padded with 0xFFFFFFFF.  The tensors are kept alive by the index."""
        self = cls.__new__(cls)
        self.distance = as_functor(distance)
        code = dtype_code(np.empty(0, data_dtype))
        desc = _lib.DeviceIndexDesc(
            int(self.distance.metric), code, n, d, d_data.data_ptr(), d_data.stride(0) * d_data.element_size(),
            d_graph.data_ptr(), d_graph.stride(0), d_entry_points.data_ptr(), d_entry_points.numel(),
            int(sq is not None), float(sq[0]) if sq else 1.0, float(sq[1]) if sq else 0.0, device, id_offset,
        )
        self._h = C.c_void_p()
        self._keep = (d_graph, d_data, d_entry_points)
        _lib.check(_lib.lib().svsx_index_create_device_view(C.byref(desc), C.byref(self._h)))
        self.device = device
        self.id_offset = int(id_offset)
        self.n, self.d = n, d
        self.data_dtype = np.dtype(data_dtype)
        self.entry_points = None
        self._params = VamanaSearchParameters()
        if sq is not None:
            # the library took the row norms of a scalar-quantized cosine index on the NULL stream; tensors produced on another torch stream
            # are only guaranteed complete in THAT stream's order: take the norms again there (include/svsx.h, svsx_index_refresh_row_norms)
            import torch

            cur = torch.cuda.current_stream(device)
            if cur != torch.cuda.default_stream(device):
                self.refresh_row_norms(cur.cuda_stream)
        return self

    def get_distance(self, id: int, query) -> float:
        """VamanaIndex::get_distance (index/vamana/index.h:915-933): the index's own distance between stored vector `id` and `query`."""
        q = np.ascontiguousarray(np.asarray(query).reshape(-1))
        if q.shape[0] != self.d:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Incompatible dimensions. Query has {q.shape[0]} while the index expects {self.d}.")
        out = C.c_double()
        _lib.check(_lib.lib().svsx_get_distance(self._h, int(id), q.ctypes.data, dtype_code(q), C.byref(out)))
        return out.value

    def refresh_row_norms(self, stream: int = 0) -> None:
        """Recompute the per-row norms of a scalar-quantized cosine index on `stream` (after rewriting rows of a device view)."""
        _lib.check(_lib.lib().svsx_index_refresh_row_norms(self._h, C.c_void_p(stream)))

    def drop_row_norms(self) -> None:
        """Release the per-row norms: searches then take the norm per candidate (same results, slower; the reference's own form)."""
        _lib.check(_lib.lib().svsx_index_drop_row_norms(self._h))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().svsx_index_destroy(h)
            except Exception:  # interpreter shutdown: module globals may already be gone
                pass
            self._h = None

    # -- parameters (index.h:318, get/set_search_parameters) ---------------------------------------------------
    def get_search_parameters(self) -> VamanaSearchParameters:
        return self._params

    def set_search_parameters(self, p: VamanaSearchParameters) -> None:
        self._params = p

    @property
    def search_window_size(self) -> int:
        return self._params.buffer_config.search_window_size

    @search_window_size.setter
    def search_window_size(self, w: int) -> None:
        self._params.buffer_config = SearchBufferConfig(w)

    def size(self) -> int:
        return self.n

    def dimensions(self) -> int:
        return self.d

    def info(self) -> dict:
        arr = (C.c_uint64 * 6)()
        _lib.check(_lib.lib().svsx_index_info(self._h, arr))
        return dict(n=arr[0], d=arr[1], data_bytes=arr[2], graph_bytes=arr[3], graph_row_stride=arr[4], data_row_bytes=arr[5])

    def memory(self) -> dict:
        """svsx_index_memory: the HBM the index holds (bytes) and the bound scheme in effect"""
        m = (C.c_uint64 * 10)()
        _lib.check(_lib.lib().svsx_index_memory(self._h, m))
        return dict(data_bytes=m[0], graph_bytes=m[1], inline_record_bytes=m[2], record_size=m[3], code_row_bytes=m[4], flat_shadow_bytes=m[5],
                    sq_row_norm_bytes=m[6], allocated_by_library=m[7], bound_mode=m[8], owns_rows_and_graph=bool(m[9]))

    # -- search ---------------------------------------------------------------------------------------------------
    def search(self, queries, num_neighbors: int, search_parameters: VamanaSearchParameters | None = None, stats: bool = False):
        """Batch search with host arrays (svs.Vamana.search(queries, k) -> (ids, distances))."""
        queries = np.ascontiguousarray(np.atleast_2d(queries))
        if queries.shape[1] != self.d:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Query dimension {queries.shape[1]} does not match index dimension {self.d}")
        nq = queries.shape[0]
        ids = np.empty((nq, num_neighbors), np.uint64)
        dists = np.empty((nq, num_neighbors), np.float32)
        sp = (search_parameters or self._params)._c()
        aux = _lib.SearchAux()
        st = None
        if stats:
            st = np.zeros((nq, 6), np.uint32)
            aux.stats = st.ctypes.data
        _lib.check(_lib.lib().svsx_search(self._h, queries.ctypes.data, dtype_code(queries), nq, num_neighbors, C.byref(sp),
                                          ids.ctypes.data, None, dists.ctypes.data, C.byref(aux)))
        res = QueryResult(ids, dists)
        return (res, st) if stats else res

    def search_filtered(self, queries, num_neighbors: int, is_member, search_parameters: VamanaSearchParameters | None = None, filter_stop: float = 0.0,
                        filter_estimate_batch: bool = True):
        """Runtime VamanaIndex::search with an IDFilter (svsx_search_filtered; vamana_index_impl.h:140-218): `is_member` = boolean
        array over the ids.  Unfilled slots: id all-ones, distance +inf."""
        queries = np.ascontiguousarray(np.atleast_2d(queries))
        member = np.asarray(is_member, dtype=bool)
        if member.shape != (self.n,):
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "is_member must have one entry per vector")
        bits = np.zeros((self.n + 63) // 64 * 64, np.uint8)
        bits[: self.n] = member
        bitmap = np.packbits(bits.reshape(-1, 8), axis=1, bitorder="little").reshape(-1).view(np.uint64).copy()
        nq = queries.shape[0]
        ids = np.empty((nq, num_neighbors), np.uint64)
        dists = np.empty((nq, num_neighbors), np.float32)
        sp = (search_parameters or self._params)._c()
        # filter_estimate_batch: SearchParams::filter_estimate_batch of the runtime (default true there and here)
        _lib.check(_lib.lib().svsx_search_filtered_ex(self._h, queries.ctypes.data, dtype_code(queries), nq, num_neighbors, C.byref(sp),
                                                      bitmap.ctypes.data, float(filter_stop), int(bool(filter_estimate_batch)), ids.ctypes.data, dists.ctypes.data))
        return QueryResult(ids, dists)

    def build_bound_codes(self, mode: int | bool = 1) -> int:
        """svsx_index_build_bound_codes: result-neutral bound rejection.  mode 2: graph-inline records per neighbour, size chosen from the
        data's spectrum (float32 / float16 rows, scalar-quantized rows, raw int8 / uint8 rows; any metric; 32 <= d <= 4096); 3: the same, but only
        where the records are expected to pay; 28 / 32 / 64 / 128: mode 2 with that record size (28: float32 / float16 rows); mode 1 (True): 128-byte code row per vector (float rows, L2, d = 96 / 128); 0 (False): off.
        Returns the scheme in effect (0 / 1 / 2): records that do not fit the free HBM shrink or fall back with a RuntimeWarning, not an error."""
        rc = _lib.lib().svsx_index_build_bound_codes(self._h, int(mode))
        if rc == _lib.FELL_BACK:  # not a failure: a smaller scheme than asked for was built (the one asked for does not fit the free HBM)
            import warnings

            warnings.warn(_lib.lib().svsx_last_error().decode(), RuntimeWarning, stacklevel=2)
        else:
            _lib.check(rc)
        return self.memory()["bound_mode"]

    def last_kernel_info(self) -> str:
        """svsx_last_kernel_info: which search-kernel instantiation the last search on this thread launched."""
        return _lib.lib().svsx_last_kernel_info().decode()

    def batch_iterator(self, queries, extra_search_buffer_capacity: int | None = None):
        """VamanaIndex::make_batch_iterator (index/vamana/index.h:936), one iterator per query of the batch."""
        from .iterator import BatchIterator

        return BatchIterator(self, queries, extra_search_buffer_capacity)

    def flat_search_device(self, d_queries, num_neighbors: int, d_ids64=None, d_dists=None, q_dtype=None, stream=None, recomputed=None):
        """Exhaustive search (FlatIndex::search, index/flat/flat.h:322-460) on torch CUDA tensors: MFMA f16 shortlist +
        exact re-rank + per-query proof of exactness (unproven queries are recomputed exhaustively in float32).
        Returns (ids int64 [nq,k], dists f32 [nq,k]).  `recomputed`: optional int32 CUDA tensor of one element, incremented
        by the number of queries that needed the exhaustive float32 pass."""
        import torch

        nq = d_queries.shape[0]
        code = q_dtype if q_dtype is not None else {torch.float32: 0, torch.float16: 1, torch.int8: 2, torch.uint8: 3}[d_queries.dtype]
        if d_ids64 is None:
            d_ids64 = torch.empty((nq, num_neighbors), dtype=torch.int64, device=d_queries.device)
        if d_dists is None:
            d_dists = torch.empty((nq, num_neighbors), dtype=torch.float32, device=d_queries.device)
        s = stream if stream is not None else torch.cuda.current_stream(d_queries.device).cuda_stream
        _lib.check(_lib.lib().svsx_flat_search_device_counted(
            self._h, d_queries.data_ptr(), code, nq, num_neighbors, d_ids64.data_ptr(), d_dists.data_ptr(),
            recomputed.data_ptr() if recomputed is not None else None, C.c_void_p(s)))
        return d_ids64, d_dists

    def vamana_build(self, alpha: float, graph_max_degree: int, window_size: int, max_candidate_pool_size: int = 750, prune_to: int = 0, stream=None,
                     use_full_search_history: bool = True) -> None:
        """VamanaIndex build constructor (index/vamana/index.h:398-439) on this device-view index: fills its (all 0xFFFFFFFF) graph in place."""
        import torch

        bp = _lib.BuildParams(float(alpha), int(graph_max_degree), int(window_size), int(max_candidate_pool_size), int(prune_to), 1 if use_full_search_history else 0)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().svsx_vamana_build_device(self._h, C.byref(bp), C.c_void_p(s)))

    def find_medoid(self) -> int:
        """utils::find_medioid (core/medioid.h:292-353): the vector closest to the component-wise mean."""
        m = C.c_uint32(0)
        _lib.check(_lib.lib().svsx_find_medoid_device(self._h, C.byref(m)))
        return int(m.value)

    def prune_device(self, d_node_ids, d_pool_ids, d_pool_dists, max_out: int, alpha: float, out_stride: int, stream=None):
        """heuristic_prune_neighbors (index/vamana/prune.h:100-246) for a batch of vertices, on torch CUDA tensors:
        node ids int32 [B], pool ids int32 [B,P] (sorted best-first, -1 padded), pool dists f32 [B,P] -> int32 [B,out_stride]."""
        import torch

        B, P = d_pool_ids.shape
        out = torch.empty((B, out_stride), dtype=torch.int32, device=d_pool_ids.device)
        s = stream if stream is not None else torch.cuda.current_stream(d_pool_ids.device).cuda_stream
        _lib.check(_lib.lib().svsx_prune_device(self._h, d_node_ids.data_ptr(), d_pool_ids.data_ptr(), d_pool_dists.data_ptr(), B, P, max_out,
                                                float(alpha), out.data_ptr(), out_stride, C.c_void_p(s)))
        return out

    def search_device(self, d_queries, num_neighbors: int, search_parameters: VamanaSearchParameters | None = None,
                      d_ids64=None, d_ids32=None, d_dists=None, d_stats=None, history=None, q_dtype=None, stream=None):
        """Enqueue a batch search on torch CUDA tensors (no synchronisation).  ``history`` = (ids, dists, len) tensors."""
        import torch

        nq = d_queries.shape[0]
        code = q_dtype if q_dtype is not None else {torch.float32: 0, torch.float16: 1, torch.int8: 2, torch.uint8: 3}[d_queries.dtype]
        sp = (search_parameters or self._params)._c()
        aux = _lib.SearchAux()
        aux.query_row_stride_bytes = d_queries.stride(0) * d_queries.element_size()
        if d_stats is not None:
            aux.stats = d_stats.data_ptr()
        if history is not None:
            aux.history_ids = history[0].data_ptr()
            aux.history_dists = history[1].data_ptr() if history[1] is not None else None
            aux.history_len = history[2].data_ptr()
            aux.history_cap = history[0].shape[1]
        s = stream if stream is not None else torch.cuda.current_stream(d_queries.device).cuda_stream
        _lib.check(_lib.lib().svsx_search_device(
            self._h, d_queries.data_ptr(), code, nq, num_neighbors, C.byref(sp),
            d_ids64.data_ptr() if d_ids64 is not None else None, d_ids32.data_ptr() if d_ids32 is not None else None,
            d_dists.data_ptr() if d_dists is not None else None, C.byref(aux), C.c_void_p(s)))


class FlatIndex:
    """svs::index::flat::FlatIndex (index/flat/flat.h:322-460): exhaustive search over a dataset resident in HBM.
    The dense query x database products run on the matrix cores (float16 MFMA shortlist), the returned distances are
    recomputed with the exact reference-order kernels."""

    def __init__(self, data, distance, device: int = 0, sq=None):
        self.distance = as_functor(distance)
        data = np.ascontiguousarray(data)
        desc = _lib.IndexDesc(
            int(self.distance.metric), dtype_code(data), data.shape[0], data.shape[1], data.ctypes.data, 0,
            None, 0, 0, None, 0, int(sq is not None), float(sq[0]) if sq else 1.0, float(sq[1]) if sq else 0.0, device, 0,
        )
        self._h = C.c_void_p()
        _lib.check(_lib.lib().svsx_index_create(C.byref(desc), C.byref(self._h)))
        self.device = device
        self.id_offset = 0
        self.n, self.d = data.shape

    __del__ = VamanaIndex.__del__
    flat_search_device = VamanaIndex.flat_search_device

    def search(self, queries, num_neighbors: int) -> QueryResult:
        import torch

        queries = np.ascontiguousarray(np.atleast_2d(queries))
        if queries.shape[1] != self.d:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"Query dimension {queries.shape[1]} does not match index dimension {self.d}")
        nq = queries.shape[0]
        ids = np.empty((nq, num_neighbors), np.uint64)
        dists = np.empty((nq, num_neighbors), np.float32)
        _lib.check(_lib.lib().svsx_flat_search(self._h, queries.ctypes.data, dtype_code(queries), nq, num_neighbors, ids.ctypes.data, dists.ctypes.data))
        return QueryResult(ids, dists)
