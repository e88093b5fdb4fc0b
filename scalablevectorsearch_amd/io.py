"""Readers for the on-disk formats that feed this path (host-side, no compute).

  [fbi]vecs                 core/io/vecs.h:37-273   each vector = int32 dimension + d elements (f32 / u8 / i32)
  SVS native v1 (".svs")    core/io/native.h:307-344  1024-byte header {u64 magic 0xcad4a6b2579980fe, 16-byte UUID,
                            u64 num_vectors, u64 dimensions, zero padding} followed by the dense row-major payload.
                            VamanaIndex::save writes the dataset and the graph (n x (R+1) u32, slot 0 = degree,
                            core/graph/graph.h:300-305) as two such blobs.
"""
from __future__ import annotations

import struct

import numpy as np

SVS_V1_MAGIC = 0xCAD4A6B2579980FE
SVS_V1_HEADER = 1024


def _read_vecs(path, dtype):
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size == 0:
        return np.zeros((0, 0), dtype)
    d = int(raw[:4].view(np.int32)[0])
    es = np.dtype(dtype).itemsize
    row = 4 + d * es
    if d <= 0 or raw.size % row:
        raise ValueError(f"{path}: not a well-formed vecs file (d={d}, {raw.size} bytes)")
    m = raw.reshape(-1, row)
    if not (m[:, :4].view(np.int32) == d).all():
        raise ValueError(f"{path}: vectors of differing dimensionality")
    return np.ascontiguousarray(m[:, 4:]).view(dtype).reshape(-1, d)


def read_fvecs(path):
    return _read_vecs(path, np.float32)


def read_ivecs(path):
    return _read_vecs(path, np.uint32)


def read_bvecs(path):
    return _read_vecs(path, np.uint8)


def write_fvecs(path, x):
    x = np.ascontiguousarray(x, np.float32)
    n, d = x.shape
    out = np.empty((n, 4 + 4 * d), np.uint8)
    out[:, :4] = np.array([d], np.int32).view(np.uint8)
    out[:, 4:] = x.view(np.uint8).reshape(n, 4 * d)
    out.tofile(path)


def read_svs_native(path, dtype):
    """Dense SVS v1 blob -> [num_vectors, dimensions] array of `dtype` (f32/f16/i8/u8 data, u32 graph rows)."""
    with open(path, "rb") as f:
        head = f.read(SVS_V1_HEADER)
        if len(head) != SVS_V1_HEADER:
            raise ValueError(f"{path}: truncated header")
        magic, = struct.unpack_from("<Q", head, 0)
        if magic != SVS_V1_MAGIC:
            raise ValueError(f"{path}: bad magic {magic:#x}")
        n, d = struct.unpack_from("<QQ", head, 24)
        data = np.fromfile(f, dtype=dtype, count=n * d)
    if data.size != n * d:
        raise ValueError(f"{path}: payload shorter than {n} x {d}")
    return data.reshape(n, d)


def write_svs_native(path, x):
    x = np.ascontiguousarray(x)
    head = bytearray(SVS_V1_HEADER)
    struct.pack_into("<Q", head, 0, SVS_V1_MAGIC)
    struct.pack_into("<QQ", head, 24, x.shape[0], x.shape[1])
    with open(path, "wb") as f:
        f.write(head)
        f.write(x.tobytes())


def read_index_config(path):
    """svs_config.toml as VamanaIndex::save writes it (index/vamana/index.h:88-97,141-170): entry point, build parameters (build_params.h:88-101)
    and the DEFAULT SEARCH PARAMETERS the index is reloaded with (search_params.h:79-126); also the flat legacy layout of versions <= v0.0.2
    (index.h:107-150).  Returns dict(entry_point, build=dict|None, search=dict|None)."""
    import os

    import tomli

    if os.path.isdir(path):
        path = os.path.join(path, "svs_config.toml")
    with open(path, "rb") as f:
        cfg = tomli.load(f)
    obj = cfg.get("object", cfg)
    out = dict(entry_point=int(obj["entry_point"]), build=None, search=None)
    if "build_parameters" in obj or "search_parameters" in obj:
        b, s_ = obj.get("build_parameters"), obj.get("search_parameters")
        if b is not None:
            out["build"] = dict(alpha=float(b["alpha"]), graph_max_degree=int(b["graph_max_degree"]), window_size=int(b["window_size"]),
                                max_candidate_pool_size=int(b["max_candidate_pool_size"]), prune_to=int(b.get("prune_to", b["graph_max_degree"])),
                                use_full_search_history=bool(b.get("use_full_search_history", True)))
        if s_ is not None:
            out["search"] = dict(search_window_size=int(s_["search_window_size"]), search_buffer_capacity=int(s_["search_buffer_capacity"]),
                                 search_buffer_visited_set=bool(s_["search_buffer_visited_set"]), prefetch_lookahead=int(s_.get("prefetch_lookahead", 4)),
                                 prefetch_step=int(s_.get("prefetch_step", 1)))
    elif "max_out_degree" in obj:  # load_legacy: prune_to defaults to the degree, full search history to true, prefetch to (4, 1)
        r = int(obj["max_out_degree"])
        out["build"] = dict(alpha=float(obj["alpha"]), graph_max_degree=r, window_size=int(obj["construction_window_size"]),
                            max_candidate_pool_size=int(obj["max_candidates"]), prune_to=int(obj.get("prune_to", r)),
                            use_full_search_history=bool(obj.get("use_full_search_history", True)))
        w = int(obj["default_search_window_size"])
        out["search"] = dict(search_window_size=w, search_buffer_capacity=w, search_buffer_visited_set=bool(obj["visited_set"]), prefetch_lookahead=4, prefetch_step=1)
    if out["search"] and out["search"]["search_window_size"] > out["search"]["search_buffer_capacity"]:
        raise ValueError(f"{path}: search window larger than the buffer capacity (search_buffer.h:84-93)")
    return out
