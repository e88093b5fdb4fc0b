"""Loader for libsvsx.so (the C ABI of include/svsx.h).  There is no Python or CPU fallback: if the HIP
library is missing this raises, and every compute entry point fails loudly without a GPU."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVSX_LIB") or os.path.join(_HERE, "libsvsx.so")  # SVSX_LIB: A/B builds of the same ABI

SUCCESS, UNKNOWN_ERROR, INVALID_ARGUMENT, NOT_IMPLEMENTED, NOT_INITIALIZED, RUNTIME_ERROR = range(6)
FELL_BACK = 64  # svsx.h: not a failure -- svsx_index_build_bound_codes built a smaller scheme than asked for (HBM)


class SvsxError(RuntimeError):
    """Mirrors svs::ANNException / runtime Status{code, message} (bindings/cpp/include/svs/runtime/api_defs.h:110-184)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[svsx status {code}] {message}")
        self.code = code
        self.message = message


class IndexDesc(C.Structure):
    _fields_ = [
        ("metric", C.c_int), ("data_dtype", C.c_int), ("n", C.c_size_t), ("d", C.c_size_t),
        ("data", C.c_void_p), ("data_row_stride_bytes", C.c_size_t),
        ("graph", C.c_void_p), ("graph_row_stride_u32", C.c_size_t), ("max_degree", C.c_uint32),
        ("entry_points", C.c_void_p), ("n_entry", C.c_size_t),
        ("sq_enabled", C.c_int), ("sq_scale", C.c_float), ("sq_bias", C.c_float),
        ("device", C.c_int), ("id_offset", C.c_uint64),
    ]


class DeviceIndexDesc(C.Structure):
    _fields_ = [
        ("metric", C.c_int), ("data_dtype", C.c_int), ("n", C.c_size_t), ("d", C.c_size_t),
        ("d_data", C.c_void_p), ("data_row_stride_bytes", C.c_size_t),
        ("d_graph", C.c_void_p), ("graph_row_stride_u32", C.c_size_t),
        ("d_entry_points", C.c_void_p), ("n_entry", C.c_size_t),
        ("sq_enabled", C.c_int), ("sq_scale", C.c_float), ("sq_bias", C.c_float),
        ("device", C.c_int), ("id_offset", C.c_uint64),
    ]


class SearchParams(C.Structure):
    _fields_ = [
        ("search_window_size", C.c_size_t), ("search_buffer_capacity", C.c_size_t), ("visited_set", C.c_int),
        ("prefetch_lookahead", C.c_size_t), ("prefetch_step", C.c_size_t),
    ]


class SearchAux(C.Structure):
    _fields_ = [
        ("struct_size", C.c_size_t), ("stats", C.c_void_p), ("history_ids", C.c_void_p), ("history_dists", C.c_void_p), ("history_len", C.c_void_p),
        ("history_cap", C.c_size_t), ("query_row_stride_bytes", C.c_size_t), ("cancel_flag", C.c_void_p),
    ]


def _aux_init(self, *a, **kw):
    C.Structure.__init__(self, *a, **kw)
    if not self.struct_size:
        self.struct_size = C.sizeof(SearchAux)


SearchAux.__init__ = _aux_init


class RangeStorage(C.Structure):
    """svsx_range_storage: what the results allocator of svsx_range_search hands back (runtime SearchResultsStorage, api_defs.h:194-197)."""

    _fields_ = [("labels", C.c_void_p), ("distances", C.c_void_p)]


RESULTS_ALLOCATOR = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_size_t), C.c_size_t, C.POINTER(RangeStorage))


class BuildParams(C.Structure):
    """svsx_build_params (index/vamana/build_params.h:28-75)."""

    _fields_ = [
        ("alpha", C.c_float), ("graph_max_degree", C.c_size_t), ("window_size", C.c_size_t), ("max_candidate_pool_size", C.c_size_t),
        ("prune_to", C.c_size_t), ("use_full_search_history", C.c_int),
    ]


EXPORTS = [
    "svsx_index_create", "svsx_index_create_device_view", "svsx_index_destroy", "svsx_search", "svsx_search_device",
    "svsx_distance_batch", "svsx_merge_topk_device", "svsx_prune_device", "svsx_flat_search_device", "svsx_index_info", "svsx_last_error", "svsx_version",
    "svsx_iterator_create", "svsx_iterator_destroy", "svsx_iterator_update", "svsx_iterator_restart_next_search", "svsx_iterator_next",
    "svsx_iterator_next_device", "svsx_iterator_state", "svsx_index_build_bound_codes", "svsx_merge_topk_strided_device",
    "svsx_last_kernel_info", "svsx_search_filtered", "svsx_search_filtered_ex", "svsx_vamana_build_device", "svsx_find_medoid_device",
    "svsx_sharded_unique_id", "svsx_sharded_create", "svsx_sharded_create_with_comm", "svsx_sharded_destroy", "svsx_sharded_info",
    "svsx_sharded_search_device", "svsx_abi_version", "svsx_flat_search_device_counted", "svsx_flat_search",
    "svsx_index_refresh_row_norms", "svsx_index_drop_row_norms", "svsx_range_search", "svsx_get_distance", "svsx_index_memory",
    "svsx_sharded_plan", "svsx_sharded_set_timing", "svsx_sharded_phase_times",
]

_lib = None


def build(force: bool = False) -> str:
    """Compile libsvsx.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "svsx.h"))
    stale = force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", csrc, "-j4"])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950).  scalablevectorsearch_amd has no CPU fallback."
            )
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7.  Import torch FIRST so that
        # libsvsx.so binds to the runtime torch already loaded (same SONAME) instead of pulling in /opt/rocm's copy,
        # which would leave torch unable to see the GPU afterwards.  Without torch the RUNPATH (/opt/rocm) is used.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - plain C/C++ hosts
            pass
        L = C.CDLL(LIB_PATH)
        L.svsx_last_error.restype = C.c_char_p
        L.svsx_version.restype = C.c_char_p
        L.svsx_last_kernel_info.restype = C.c_char_p
        if L.svsx_abi_version() != 3:  # include/svsx.h SVSX_ABI_VERSION: the structures below are laid out for ABI 3
            raise ImportError(f"{LIB_PATH} speaks ABI {L.svsx_abi_version()}, this binding ABI 3")
        L.svsx_index_create.argtypes = [C.POINTER(IndexDesc), C.POINTER(C.c_void_p)]
        L.svsx_index_create_device_view.argtypes = [C.POINTER(DeviceIndexDesc), C.POINTER(C.c_void_p)]
        L.svsx_index_destroy.argtypes = [C.c_void_p]
        L.svsx_index_destroy.restype = None
        L.svsx_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(SearchParams),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SearchAux)]
        L.svsx_search_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(SearchParams),
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(SearchAux), C.c_void_p]
        L.svsx_distance_batch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
        L.svsx_merge_topk_device.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_merge_topk_strided_device.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_prune_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_size_t, C.c_void_p]
        L.svsx_flat_search_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_flat_search_device_counted.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_flat_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_void_p, C.c_void_p]
        L.svsx_index_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.svsx_index_memory.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.svsx_sharded_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.svsx_sharded_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        L.svsx_iterator_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]
        L.svsx_iterator_destroy.argtypes = [C.c_void_p]
        L.svsx_iterator_destroy.restype = None
        L.svsx_iterator_update.argtypes = [C.c_void_p, C.c_void_p]
        L.svsx_iterator_restart_next_search.argtypes = [C.c_void_p]
        L.svsx_iterator_next.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_iterator_next_device.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.svsx_iterator_state.argtypes = [C.c_void_p, C.c_void_p]
        L.svsx_index_build_bound_codes.argtypes = [C.c_void_p, C.c_int]
        L.svsx_search_filtered.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(SearchParams), C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]
        L.svsx_search_filtered_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(SearchParams), C.c_void_p, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        try:  # entry points added within ABI 3 (round 5): an older build of the library (SVSX_LIB=variants/...: A/B timing) simply lacks them
            L.svsx_index_refresh_row_norms.argtypes = [C.c_void_p, C.c_void_p]
            L.svsx_index_drop_row_norms.argtypes = [C.c_void_p]
            L.svsx_range_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.POINTER(SearchParams), C.c_void_p, RESULTS_ALLOCATOR, C.c_void_p]
            L.svsx_get_distance.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
        except AttributeError:
            if not os.environ.get("SVSX_LIB"):
                raise
        L.svsx_vamana_build_device.argtypes = [C.c_void_p, C.POINTER(BuildParams), C.c_void_p]
        L.svsx_find_medoid_device.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.svsx_sharded_unique_id.argtypes = [C.c_void_p]
        L.svsx_sharded_create.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.svsx_sharded_create_with_comm.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.svsx_sharded_destroy.argtypes = [C.c_void_p]
        L.svsx_sharded_destroy.restype = None
        L.svsx_sharded_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.svsx_sharded_search_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(SearchParams), C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != SUCCESS:
        raise SvsxError(rc, lib().svsx_last_error().decode())
