"""svs::distance plugin surface (concepts/distance.h, core/distance/*.h) over svsx_distance_batch."""
from __future__ import annotations

import ctypes as C
import enum

import numpy as np

from . import _lib

F32, F16, I8, U8 = 0, 1, 2, 3
_NP = {np.dtype(np.float32): F32, np.dtype(np.float16): F16, np.dtype(np.int8): I8, np.dtype(np.uint8): U8}


def dtype_code(arr) -> int:
    try:
        return _NP[np.dtype(arr.dtype)]
    except KeyError:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"unsupported element type {arr.dtype}") from None


class DistanceType(enum.IntEnum):
    """svs::DistanceType (core/distance.h)."""
    L2 = 0
    MIP = 1
    Cosine = 2


class _Functor:
    metric: DistanceType
    #: ``compare`` of the reference functor: True when larger is more similar (std::greater<>)
    greater: bool

    def compute(self, a: np.ndarray, b: np.ndarray, device: int = 0) -> np.ndarray:
        """distance::compute(f, a_i, b_j) for every pair, after f.fix_argument(a_i); returns [na, nb] f32."""
        return compute(self, a, b, device)


class DistanceL2(_Functor):
    """svs::distance::DistanceL2 (core/distance/euclidean.h:142-166); compare = std::less<>."""
    metric, greater, name = DistanceType.L2, False, "squared_l2"


class DistanceIP(_Functor):
    """svs::distance::DistanceIP (core/distance/inner_product.h:103-127); compare = std::greater<>."""
    metric, greater, name = DistanceType.MIP, True, "inner_product"


class DistanceCosineSimilarity(_Functor):
    """svs::distance::DistanceCosineSimilarity (core/distance/cosine.h:102-144); must_fix_argument."""
    metric, greater, name = DistanceType.Cosine, True, "cosine_similarity"


def as_functor(x) -> _Functor:
    if isinstance(x, _Functor):
        return x
    if isinstance(x, type) and issubclass(x, _Functor):
        return x()
    return {DistanceType.L2: DistanceL2, DistanceType.MIP: DistanceIP, DistanceType.Cosine: DistanceCosineSimilarity}[DistanceType(x)]()


def compute(distance, a: np.ndarray, b: np.ndarray, device: int = 0) -> np.ndarray:
    f = as_functor(distance)
    a = np.ascontiguousarray(np.atleast_2d(a))
    b = np.ascontiguousarray(np.atleast_2d(b))
    if a.shape[1] != b.shape[1]:
        raise _lib.SvsxError(_lib.INVALID_ARGUMENT, f"dimension mismatch {a.shape[1]} vs {b.shape[1]}")
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    _lib.check(_lib.lib().svsx_distance_batch(int(f.metric), dtype_code(a), dtype_code(b), a.shape[1], a.ctypes.data, a.shape[0],
                                               b.ctypes.data, b.shape[0], out.ctypes.data, device))
    return out
