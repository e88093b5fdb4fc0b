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
