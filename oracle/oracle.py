"""ctypes bindings for the CPU checkers under oracle/ -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
  * ``Oracle``  -> oracle/libsvs_oracle.so   (plain-C restatement, svs_oracle.c)
  * ``Ref``     -> oracle/_ref/libsvs_ref.so (the reference's own headers, ref_driver.cpp); optional.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32, F16, I8, U8 = 0, 1, 2, 3
L2, IP, COS = 0, 1, 2
NP_DTYPES = {F32: np.float32, F16: np.float16, I8: np.int8, U8: np.uint8}


def dtype_code(arr: np.ndarray) -> int:
    for code, dt in NP_DTYPES.items():
        if arr.dtype == dt:
            return code
    raise TypeError(f"unsupported dtype {arr.dtype}")


def build(force: bool = False) -> None:
    """Compile the C restatement (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(HERE, "libsvs_oracle.so")
    srcs = [os.path.join(HERE, f) for f in ("svs_oracle.c", "svs_oracle_build.c", "svs_oracle.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", HERE, "libsvs_oracle.so"])
    ref_so = os.path.join(HERE, "_ref", "libsvs_ref.so")
    if os.path.isdir("/root/reference/include/svs"):
        drv = os.path.join(HERE, "ref_driver.cpp")
        if force or not os.path.exists(ref_so) or os.path.getmtime(drv) > os.path.getmtime(ref_so):
            subprocess.check_call(["make", "-C", HERE, "ref"])


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def merge_topk_ref(metric, gathered_d, gathered_i, k):
    """Checker for the shard merge (SURVEY 8e rule): per query, best k of the S sorted lists by the metric's
    comparator; ties -> lower shard, then position; padded entries (id == -1 / all-ones) are skipped."""
    gd = np.asarray(gathered_d)
    gi = np.asarray(gathered_i).astype(np.int64)
    S, nq, kk = gd.shape
    out_d = np.full((nq, k), np.inf, np.float32)
    out_i = np.full((nq, k), -1, np.int64)
    greater = metric != L2
    for q in range(nq):
        cur = [0] * S
        for j in range(k):
            best = -1
            for s_ in range(S):
                while cur[s_] < kk and gi[s_, q, cur[s_]] == -1:
                    cur[s_] = kk
                if cur[s_] >= kk:
                    continue
                dd = gd[s_, q, cur[s_]]
                if best < 0 or (dd > bd if greater else dd < bd):
                    best, bd = s_, dd
            if best < 0:
                break
            out_d[q, j] = bd
            out_i[q, j] = gi[best, q, cur[best]]
            cur[best] += 1
    return out_d, out_i


class _Index(C.Structure):
    _fields_ = [
        ("metric", C.c_int), ("data_dtype", C.c_int), ("n", C.c_size_t), ("d", C.c_size_t),
        ("data", C.c_void_p), ("graph", C.c_void_p), ("graph_stride", C.c_size_t),
        ("entry_points", C.c_void_p), ("n_entry", C.c_size_t),
        ("sq_enabled", C.c_int), ("sq_scale", C.c_float), ("sq_bias", C.c_float),
    ]


class _Neighbor(C.Structure):
    _fields_ = [("id", C.c_uint32), ("dist", C.c_float), ("visited", C.c_uint8)]


class _Iterator(C.Structure):
    pass  # fields set after _Buffer


class _Buffer(C.Structure):
    _fields_ = [
        ("greater", C.c_int), ("size", C.c_size_t), ("best_unvisited", C.c_size_t), ("window", C.c_size_t),
        ("capacity", C.c_size_t), ("e", C.POINTER(_Neighbor)), ("filter", C.c_void_p),
    ]


_Iterator._fields_ = [
    ("ix", C.c_void_p), ("query", C.c_void_p), ("q_dtype", C.c_int), ("extra", C.c_size_t), ("buf", _Buffer),
    ("yielded", C.c_void_p), ("n_yielded", C.c_size_t), ("yielded_cap", C.c_size_t), ("iteration", C.c_size_t),
    ("restart", C.c_int), ("exhausted", C.c_int),
]

PLACEHOLDER = (1 << 64) - 1  # svs::UNSIGNED_INTEGER_PLACEHOLDER (lib/preprocessor.h:177)


class Oracle:
    def __init__(self):
        build()
        self.lib = C.CDLL(os.path.join(HERE, "libsvs_oracle.so"))
        L = self.lib
        L.svso_norm.restype = C.c_float
        L.svso_norm.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.svso_distance.restype = C.c_float
        L.svso_distance.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.POINTER(C.c_int)]
        L.svso_distance_batch.restype = C.c_int
        L.svso_distance_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.svso_search_batch.restype = C.c_int
        L.svso_search_batch.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svso_flat_search.restype = C.c_int
        L.svso_flat_search.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.svso_build_vamana.restype = C.c_int
        L.svso_build_vamana.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.POINTER(C.c_uint32)]
        L.svso_k_recall_at_n.restype = C.c_double
        L.svso_k_recall_at_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t]
        L.svso_sq_params.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.svso_sq_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.svso_half_to_float.restype = C.c_float
        L.svso_half_to_float.argtypes = [C.c_uint16]
        L.svso_float_to_half.restype = C.c_uint16
        L.svso_float_to_half.argtypes = [C.c_float]
        L.svso_buffer_init.argtypes = [C.POINTER(_Buffer), C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        L.svso_buffer_insert.restype = C.c_size_t
        L.svso_buffer_insert.argtypes = [C.POINTER(_Buffer), C.c_uint32, C.c_float]
        L.svso_buffer_push_back.argtypes = [C.POINTER(_Buffer), C.c_uint32, C.c_float]
        L.svso_buffer_next.restype = _Neighbor
        L.svso_buffer_next.argtypes = [C.POINTER(_Buffer)]
        for f in ("svso_buffer_free", "svso_buffer_clear", "svso_buffer_sort"):
            getattr(L, f).argtypes = [C.POINTER(_Buffer)]
        L.svso_buffer_done.argtypes = [C.POINTER(_Buffer)]
        L.svso_filter_emplace.argtypes = [C.c_void_p, C.c_uint32]
        L.svso_filter_contains.argtypes = [C.c_void_p, C.c_uint32]
        L.svso_filter_reset.argtypes = [C.c_void_p]

    # -- distances -------------------------------------------------------------------------------------------
    def norm(self, x: np.ndarray) -> float:
        x = np.ascontiguousarray(x)
        return float(self.lib.svso_norm(_ptr(x), dtype_code(x), x.size))

    def distance_batch(self, metric: int, a: np.ndarray, b: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        assert a.shape[1] == b.shape[1]
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        rc = self.lib.svso_distance_batch(metric, _ptr(a), dtype_code(a), a.shape[0], _ptr(b), dtype_code(b), b.shape[0], a.shape[1], _ptr(out))
        if rc:
            raise ValueError(f"unsupported type pair ({a.dtype}, {b.dtype})")
        return out

    # -- index helpers ---------------------------------------------------------------------------------------
    def _index(self, metric, data, graph, entry, sq=None):
        data = np.ascontiguousarray(data)
        graph = np.ascontiguousarray(graph, dtype=np.uint32)
        entry = np.ascontiguousarray(entry, dtype=np.uint32)
        ix = _Index(metric, dtype_code(data), data.shape[0], data.shape[1], data.ctypes.data, graph.ctypes.data,
                    graph.shape[1], entry.ctypes.data, entry.size, 0, 0.0, 0.0)
        if sq is not None:
            ix.sq_enabled, ix.sq_scale, ix.sq_bias = 1, sq[0], sq[1]
        return ix, (data, graph, entry)

    def search(self, metric, data, graph, entry, queries, k, window, capacity=None, visited_set=False, sq=None, stats=False):
        capacity = window if capacity is None else capacity
        ix, keep = self._index(metric, data, graph, entry, sq)
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        st = np.zeros((nq, 9), np.uint64)
        rc = self.lib.svso_search_batch(C.byref(ix), _ptr(queries), dtype_code(queries), nq, k, window, capacity, int(visited_set), _ptr(ids), _ptr(dists), _ptr(st))
        if rc:
            raise ValueError(f"svso_search_batch failed rc={rc}")
        return (ids, dists, st) if stats else (ids, dists)

    def flat_search(self, metric, data, queries, k, sq=None):
        graph = np.zeros((1, 1), np.uint32)
        ix, keep = self._index(metric, data, graph, np.zeros(1, np.uint32), sq)
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        rc = self.lib.svso_flat_search(C.byref(ix), _ptr(queries), dtype_code(queries), nq, k, _ptr(ids), _ptr(dists))
        if rc:
            raise ValueError(f"svso_flat_search failed rc={rc}")
        return ids, dists

    def build_vamana(self, metric, data, R, L, C_pool=750, prune_to=None, alpha=None, use_full_search_history=True):
        data = np.ascontiguousarray(data)
        n, d = data.shape
        prune_to = R if prune_to is None else prune_to
        if alpha is None:
            alpha = 1.2 if metric == L2 else 0.95
        graph = np.zeros((n, R + 1), np.uint32)
        entry = C.c_uint32(0)
        self.lib.svso_build_vamana_ex.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float,
                                                  C.c_int, C.c_void_p, C.POINTER(C.c_uint32)]
        rc = self.lib.svso_build_vamana_ex(metric, dtype_code(data), _ptr(data), n, d, R, L, C_pool, prune_to, alpha, 1 if use_full_search_history else 0,
                                           _ptr(graph), C.byref(entry))
        if rc:
            raise ValueError("svso_build_vamana failed")
        return graph, np.array([entry.value], np.uint32)

    def prune(self, metric, data, node, pool_ids, pool_dists, max_out, alpha):
        data = np.ascontiguousarray(data)
        pool_ids = np.ascontiguousarray(pool_ids, np.uint32)
        pool_dists = np.ascontiguousarray(pool_dists, np.float32)
        out = np.zeros(max_out, np.uint32)
        self.lib.svso_prune.restype = C.c_size_t
        self.lib.svso_prune.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p]
        r = self.lib.svso_prune(metric, dtype_code(data), _ptr(data), data.shape[0], data.shape[1], int(node), _ptr(pool_ids), _ptr(pool_dists), pool_ids.size, max_out, alpha, _ptr(out))
        return out[:r]

    def recall(self, gt, res, k, n):
        gt = np.ascontiguousarray(gt, dtype=np.uint32)
        res = np.ascontiguousarray(res, dtype=np.uint32)
        return float(self.lib.svso_k_recall_at_n(_ptr(gt), gt.shape[1], _ptr(res), res.shape[1], gt.shape[0], k, n))

    def sq_compress(self, data: np.ndarray, code_dtype=I8):
        data = np.ascontiguousarray(data, dtype=np.float32)
        scale, bias = C.c_float(), C.c_float()
        self.lib.svso_sq_params(_ptr(data), data.shape[0], data.shape[1], code_dtype, C.byref(scale), C.byref(bias))
        codes = np.empty(data.shape, NP_DTYPES[code_dtype])
        self.lib.svso_sq_compress(_ptr(data), data.size, code_dtype, scale, bias, _ptr(codes))
        return codes, float(scale.value), float(bias.value)

    def iterator(self, metric, data, graph, entry, query, extra=PLACEHOLDER, sq=None):
        return OracleIterator(self, metric, data, graph, entry, query, extra, sq)

    def iterate(self, metric, data, graph, entry, query, batch_sizes, extra=PLACEHOLDER, restart_before=None, sq=None):
        """BatchIterator (index/vamana/iterator.h) over one query: returns (list of (ids, dists) per next() call,
        done-after flags, iteration count, final (window, capacity))."""
        it = self.iterator(metric, data, graph, entry, query, extra, sq)
        out, done = [], []
        for b, bs in enumerate(batch_sizes):
            if restart_before is not None and restart_before[b]:
                it.restart_next_search()
            out.append(it.next(bs))
            done.append(it.done())
        return out, done, it.batch_number(), it.config()

    # -- search buffer (op-by-op) ----------------------------------------------------------------------------
    def buffer(self, window, capacity, greater=False, visited_set=False):
        return OracleBuffer(self, window, capacity, greater, visited_set)


class OracleIterator:
    """svso_iterator: BatchIterator (index/vamana/iterator.h:81-339) over one query."""

    def __init__(self, oracle, metric, data, graph, entry, query, extra=PLACEHOLDER, sq=None):
        L = self.L = oracle.lib
        L.svso_iterator_init.argtypes = [C.POINTER(_Iterator), C.POINTER(_Index), C.c_void_p, C.c_int, C.c_size_t]
        L.svso_iterator_next.restype = C.c_long
        L.svso_iterator_next.argtypes = [C.POINTER(_Iterator), C.c_size_t, C.c_void_p, C.c_void_p]
        L.svso_iterator_update.argtypes = [C.POINTER(_Iterator), C.c_void_p]
        for f in ("svso_iterator_free", "svso_iterator_done", "svso_iterator_restart_next_search"):
            getattr(L, f).argtypes = [C.POINTER(_Iterator)]
        self.ix, self._keep = oracle._index(metric, data, graph, entry, sq)
        self.query = np.ascontiguousarray(query)
        self.it = _Iterator()
        self._live = False
        if L.svso_iterator_init(C.byref(self.it), C.byref(self.ix), _ptr(self.query), dtype_code(self.query), extra):
            raise ValueError("svso_iterator_init failed")
        self._live = True

    def __del__(self):
        if getattr(self, "_live", False):
            self.L.svso_iterator_free(C.byref(self.it))
            self._live = False

    def next(self, batch_size):
        ids = np.empty(max(batch_size, 1), np.uint32)
        dd = np.empty(max(batch_size, 1), np.float32)
        got = self.L.svso_iterator_next(C.byref(self.it), batch_size, _ptr(ids), _ptr(dd))
        if got < 0:
            raise ValueError(f"svso_iterator_next rc={got}")
        return ids[:got].copy(), dd[:got].copy()

    def update(self, query):
        self.query = np.ascontiguousarray(query)
        if self.L.svso_iterator_update(C.byref(self.it), _ptr(self.query)):
            raise ValueError("svso_iterator_update failed")

    def done(self): return bool(self.L.svso_iterator_done(C.byref(self.it)))
    def restart_next_search(self): self.L.svso_iterator_restart_next_search(C.byref(self.it))
    def batch_number(self): return int(self.it.iteration)
    def config(self): return int(self.it.buf.window), int(self.it.buf.capacity)
    def n_yielded(self): return int(self.it.n_yielded)


class OracleBuffer:
    def __init__(self, oracle, window, capacity, greater, visited_set):
        self.L = oracle.lib
        self.b = _Buffer()
        if self.L.svso_buffer_init(C.byref(self.b), window, capacity, int(greater), int(visited_set)):
            raise ValueError("window > capacity")

    def __del__(self):
        self.L.svso_buffer_free(C.byref(self.b))

    def clear(self): self.L.svso_buffer_clear(C.byref(self.b))
    def push_back(self, i, d): self.L.svso_buffer_push_back(C.byref(self.b), i, d)
    def sort(self): self.L.svso_buffer_sort(C.byref(self.b))
    def insert(self, i, d): return self.L.svso_buffer_insert(C.byref(self.b), i, d)
    def done(self): return bool(self.L.svso_buffer_done(C.byref(self.b)))
    def next(self):
        n = self.L.svso_buffer_next(C.byref(self.b))
        return n.id, n.dist
    def size(self): return self.b.size
    def best_unvisited(self): return self.b.best_unvisited
    def get(self, i): return self.b.e[i].id, self.b.e[i].dist, bool(self.b.e[i].visited)
    def emplace_visited(self, i): return bool(self.b.filter and self.L.svso_filter_emplace(self.b.filter, i))
    def state(self): return [self.get(i) for i in range(self.size())]


class Ref:
    """The reference's own headers (oracle/_ref/libsvs_ref.so).  ``Ref.available()`` is False when the prebuilt
    file is missing or the host CPU lacks AVX-512 (F/BW/VNNI)."""

    _path = os.path.join(HERE, "_ref", "libsvs_ref.so")

    @classmethod
    def available(cls) -> bool:
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            return False
        if not all(f in flags for f in ("avx512f", "avx512bw", "avx512vl", "avx512_vnni")):
            return False
        build()
        return os.path.exists(cls._path)

    def __init__(self):
        if not self.available():
            raise RuntimeError("oracle/_ref/libsvs_ref.so unavailable")
        self.lib = C.CDLL(self._path)
        L = self.lib
        L.svsref_norm.restype = C.c_float
        L.svsref_norm.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.svsref_distance_batch.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p]
        L.svsref_search_batch.restype = C.c_int
        L.svsref_search_batch.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsref_buffer_new.restype = C.c_void_p
        L.svsref_buffer_new.argtypes = [C.c_size_t, C.c_size_t, C.c_int, C.c_int]
        for f in ("free", "clear", "sort"):
            getattr(L, "svsref_buffer_" + f).argtypes = [C.c_void_p]
        L.svsref_buffer_push_back.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        L.svsref_buffer_insert.restype = C.c_size_t
        L.svsref_buffer_insert.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        L.svsref_buffer_done.argtypes = [C.c_void_p]
        L.svsref_buffer_next.restype = C.c_uint32
        L.svsref_buffer_next.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.svsref_buffer_size.restype = C.c_size_t
        L.svsref_buffer_size.argtypes = [C.c_void_p]
        L.svsref_buffer_best_unvisited.restype = C.c_size_t
        L.svsref_buffer_best_unvisited.argtypes = [C.c_void_p]
        L.svsref_buffer_get.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_float), C.POINTER(C.c_int)]
        L.svsref_buffer_emplace_visited.argtypes = [C.c_void_p, C.c_uint32]
        L.svsref_filter_new.restype = C.c_void_p
        for f in ("free", "reset"):
            getattr(L, "svsref_filter_" + f).argtypes = [C.c_void_p]
        L.svsref_filter_emplace.argtypes = [C.c_void_p, C.c_uint32]
        L.svsref_filter_contains.argtypes = [C.c_void_p, C.c_uint32]
        L.svsref_half_to_float_hw.restype = C.c_float
        L.svsref_half_to_float_hw.argtypes = [C.c_uint16]
        L.svsref_scalar_loop.restype = C.c_float  # oracle/ref_scalar_loops.cpp: the reference's scalar loops under the reference's compile flags
        L.svsref_scalar_loop.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.POINTER(C.c_int)]

    def norm(self, x):
        x = np.ascontiguousarray(x)
        return float(self.lib.svsref_norm(_ptr(x), dtype_code(x), x.size))

    def scalar_loop_batch(self, metric, a, b, a_norms):
        """generic_l2 / generic_ip / generic_cosine_similarity as the compiler builds them with the reference's flags (ref_scalar_loops.cpp);
        a_norms: the query norms handed to the cosine loop (one per row of a)."""
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        ok = C.c_int(0)
        isz, jsz = a.shape[1] * a.itemsize, b.shape[1] * b.itemsize
        for i in range(a.shape[0]):
            for j in range(b.shape[0]):
                out[i, j] = self.lib.svsref_scalar_loop(metric, a.ctypes.data + i * isz, dtype_code(a), b.ctypes.data + j * jsz, dtype_code(b), a.shape[1],
                                                        float(a_norms[i]), C.byref(ok))
                if not ok.value:
                    raise ValueError("not one of the six scalar-loop pairs")
        return out

    def distance_batch(self, metric, a, b):
        a = np.ascontiguousarray(a)
        b = np.ascontiguousarray(b)
        out = np.empty((a.shape[0], b.shape[0]), np.float32)
        self.lib.svsref_distance_batch(metric, _ptr(a), dtype_code(a), a.shape[0], _ptr(b), dtype_code(b), b.shape[0], a.shape[1], _ptr(out))
        return out

    def search(self, metric, data, graph, entry, queries, k, window, capacity=None, visited_set=False, nthreads=1, stats=False):
        capacity = window if capacity is None else capacity
        data = np.ascontiguousarray(data)
        graph = np.ascontiguousarray(graph, dtype=np.uint32)
        entry = np.ascontiguousarray(entry, dtype=np.uint32)
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint32)
        dists = np.empty((nq, k), np.float32)
        st = np.zeros((nq, 2), np.uint64)
        rc = self.lib.svsref_search_batch(metric, dtype_code(data), _ptr(data), data.shape[0], data.shape[1], _ptr(graph), graph.shape[1],
                                          _ptr(entry), entry.size, _ptr(queries), dtype_code(queries), nq, k, window, capacity,
                                          int(visited_set), nthreads, _ptr(ids), _ptr(dists), _ptr(st))
        if rc:
            raise ValueError(f"svsref_search_batch rc={rc}")
        return (ids, dists, st) if stats else (ids, dists)

    def iterate(self, metric, data, graph, entry, query, batch_sizes, extra=PLACEHOLDER, restart_before=None):
        """The reference's SearchBuffer / greedy_search driven with BatchIterator's control flow (ref_driver.cpp)."""
        data = np.ascontiguousarray(data)
        graph = np.ascontiguousarray(graph, dtype=np.uint32)
        entry = np.ascontiguousarray(entry, dtype=np.uint32)
        query = np.ascontiguousarray(query)
        bs = np.ascontiguousarray(batch_sizes, dtype=np.uint64)
        rb = np.zeros(bs.size, np.uint8) if restart_before is None else np.ascontiguousarray(restart_before, dtype=np.uint8)
        total = int(bs.sum())
        ids = np.empty(max(total, 1), np.uint32)
        dd = np.empty(max(total, 1), np.float32)
        counts = np.zeros(bs.size, np.uint64)
        done = np.zeros(bs.size, np.uint8)
        self.lib.svsref_iterate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                            C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = self.lib.svsref_iterate(metric, dtype_code(data), _ptr(data), data.shape[0], data.shape[1], _ptr(graph), graph.shape[1],
                                     _ptr(entry), entry.size, _ptr(query), dtype_code(query), extra, _ptr(bs), _ptr(rb), bs.size,
                                     _ptr(ids), _ptr(dd), _ptr(counts), _ptr(done))
        if rc:
            raise ValueError(f"svsref_iterate rc={rc}")
        out, off = [], 0
        for b in range(bs.size):
            c = int(counts[b])
            out.append((ids[off:off + c].copy(), dd[off:off + c].copy()))
            off += int(bs[b])
        return out, [bool(x) for x in done]

    def buffer(self, window, capacity, greater=False, visited_set=False):
        return RefBuffer(self, window, capacity, greater, visited_set)

    def flat_search(self, metric, data, queries, k):
        """FlatIndex::search's loop around the reference's own BulkInserter / HeapInserter (index/flat/inserters.h) and distance kernels:
        ids (u64) and distances in the order std::sort_heap leaves them."""
        data = np.ascontiguousarray(data)
        queries = np.ascontiguousarray(queries)
        nq = queries.shape[0]
        ids = np.empty((nq, k), np.uint64)
        dists = np.empty((nq, k), np.float32)
        self.lib.svsref_flat_search.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        rc = self.lib.svsref_flat_search(metric, _ptr(data), dtype_code(data), data.shape[0], _ptr(queries), dtype_code(queries), nq, data.shape[1], k, _ptr(ids), _ptr(dists))
        if rc:
            raise ValueError(f"svsref_flat_search rc={rc}")
        return ids, dists

    def vecs_dims(self, path, elsize_hint=0):
        """svs::io::vecs::get_dims (core/io/vecs.h:97-124): (vectors, dimensions), or None where the reference throws."""
        n, d = C.c_size_t(), C.c_size_t()
        self.lib.svsref_vecs_dims.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        rc = self.lib.svsref_vecs_dims(os.fsencode(path), elsize_hint, C.byref(n), C.byref(d))
        return None if rc else (n.value, d.value)

    def vecs_read(self, path, dtype):
        """VecsReader<T> (core/io/vecs.h:140-190): the whole file as an [n, d] array of `dtype` (4-byte or 1-byte elements)."""
        dt = np.dtype(dtype)
        nd = self.vecs_dims(path, dt.itemsize)
        if nd is None:
            raise ValueError(f"{path}: the reference's reader rejects it")
        out = np.empty(nd, np.uint32 if dt.itemsize == 4 else np.uint8)
        self.lib.svsref_vecs_read.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_size_t]
        rc = self.lib.svsref_vecs_read(os.fsencode(path), dt.itemsize, _ptr(out), out.size)
        if rc:
            raise ValueError(f"svsref_vecs_read rc={rc}")
        return out.view(dt)

    def vecs_write_u32(self, path, rows):
        """VecsWriter<uint32_t> (core/io/vecs.h:196-222); float rows go through as their bit patterns."""
        rows = np.ascontiguousarray(rows).view(np.uint32)
        self.lib.svsref_vecs_write_u32.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t]
        if self.lib.svsref_vecs_write_u32(os.fsencode(path), _ptr(rows), rows.shape[0], rows.shape[1]):
            raise ValueError("svsref_vecs_write_u32 failed")


class RefBuffer:
    def __init__(self, ref, window, capacity, greater, visited_set):
        self.L = ref.lib
        self.p = self.L.svsref_buffer_new(window, capacity, int(greater), int(visited_set))
        if not self.p:
            raise ValueError("window > capacity")

    def __del__(self):
        if getattr(self, "p", None):
            self.L.svsref_buffer_free(self.p)

    def clear(self): self.L.svsref_buffer_clear(self.p)
    def push_back(self, i, d): self.L.svsref_buffer_push_back(self.p, i, d)
    def sort(self): self.L.svsref_buffer_sort(self.p)
    def insert(self, i, d): return self.L.svsref_buffer_insert(self.p, i, d)
    def done(self): return bool(self.L.svsref_buffer_done(self.p))
    def next(self):
        d = C.c_float()
        i = self.L.svsref_buffer_next(self.p, C.byref(d))
        return i, d.value
    def size(self): return self.L.svsref_buffer_size(self.p)
    def best_unvisited(self): return self.L.svsref_buffer_best_unvisited(self.p)
    def get(self, i):
        a, b, c = C.c_uint32(), C.c_float(), C.c_int()
        self.L.svsref_buffer_get(self.p, i, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, bool(c.value)
    def emplace_visited(self, i): return bool(self.L.svsref_buffer_emplace_visited(self.p, i))
    def state(self): return [self.get(i) for i in range(self.size())]
