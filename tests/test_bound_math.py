"""The arithmetic of the bound codes (csrc/svsx_distance.hip codes_build_kernel, csrc/svsx_search.hip CODES path), restated in
numpy float32 and checked against the oracle's exact distances: with the threshold set to a candidate's OWN exact distance the
bound must never reject it (a rejection needs distance > worst).  CPU-only; the GPU parity tests check the kernel itself."""
import numpy as np
import pytest

from conftest import deep_like
from oracle.oracle import L2

f32 = np.float32


def build_codes(x, dc):
    lo, hi = f32(x[:, :dc].min()), f32(x[:, :dc].max())
    scale = f32((hi - lo) / f32(255.0))
    if not (scale > 0):
        scale = f32(1.0)
    inv = f32(1.0) / scale
    c = np.clip(np.rint((x[:, :dc] - lo) * inv), 0, 255).astype(f32)
    err = x[:, :dc].astype(np.float64) - (np.float64(lo) + np.float64(scale) * c.astype(np.float64))
    ex = (np.sqrt((err * err).sum(1)) * (1.0 + 1e-9)).astype(f32)
    ex = (ex.view(np.uint32) + np.uint32(2)).view(f32)  # rounded up, as stored
    return c.astype(np.int64), scale, lo, ex


def query_codes(q, dc, scale, bias):
    inv = f32(1.0) / scale
    c = np.clip(np.rint((q[:dc] - bias) * inv), 0, 255).astype(f32)
    r = q[:dc] - (bias + scale * c)
    e2 = f32(0.0)
    for v in r:  # f32 accumulation (the kernel's order differs; the 1.001 margin covers any order)
        e2 = f32(e2 + f32(v * v))
    # round 3: + the absolute term of the residual's own rounding, which happens at the magnitude of the offset (svsx_index_build_bound_codes)
    eq_abs = f32(np.sqrt(np.float64(dc)) * 2.4e-7 * (abs(np.float64(bias)) + 255.0 * np.float64(scale)) * 1.01)
    eq = f32(f32(np.sqrt(e2)) * f32(1.001) + f32(1.0e-7)) + eq_abs
    return c.astype(np.int64), eq


def rejected(i_sum, worst, eq, ex, scale):
    sthr = f32(np.sqrt(f32(worst * f32(1.0001)))) * f32(1.00001) + eq
    tscale = f32(f32(1.00002) / scale) * f32(1.000001)
    t = f32(f32(sthr + ex) * tscale)
    return f32(i_sum) > f32(f32(t * t) * f32(1.00001))


def _datasets():
    rng = np.random.default_rng(0)
    yield deep_like(400, 96, 24, seed=1), deep_like(40, 96, 24, seed=2)                       # the bench distribution
    yield deep_like(400, 128, 24, seed=3), deep_like(40, 128, 24, seed=4)                      # 120 of 128 dims coded
    x = deep_like(300, 96, 8, seed=5, normalize=False) * f32(1e3) + f32(5e3)                     # large offset
    yield x.astype(f32), (x[:30] + rng.standard_normal((30, 96)).astype(f32)).astype(f32)
    x = deep_like(300, 96, 8, seed=6, normalize=False) * f32(1e-3)
    x[::50] *= f32(100.0)                                                                         # outliers stretch the range
    yield x.astype(f32), (x[:30] * f32(30.0)).astype(f32)                                          # queries far outside it
    x = rng.standard_normal((300, 96)).astype(f32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    yield x, np.zeros((4, 96), f32) + f32(1e-4)                                                    # all candidates equidistant
    yield np.concatenate([x[:100], x[:100]]), x[:20] + f32(1e-3)                                   # exact duplicates
    for off, spread in ((1e3, 1e-2), (1e5, 1e-1)):                                                # far offset band: |x| / spread up to 10^6
        xb = (deep_like(300, 96, 8, seed=7, normalize=False) * f32(spread) + f32(off)).astype(f32)
        yield xb, (xb[:16] + f32(spread) * rng.standard_normal((16, 96)).astype(f32)).astype(f32)


@pytest.mark.parametrize("case", range(8))
def test_bound_never_exceeds_the_exact_distance(oracle, case):
    x, q = list(_datasets())[case]
    d = x.shape[1]
    dc = min(d, 120)
    cx, scale, bias, ex = build_codes(x, dc)
    exact = oracle.distance_batch(L2, q, x)  # the reference's f32 values, bit for bit
    tight, useful = 0, 0
    for qi in range(q.shape[0]):
        cq, eq = query_codes(q[qi], dc, scale, bias)
        i_sum = ((cq[None, :] - cx) ** 2).sum(1)
        for j in range(x.shape[0]):
            assert not rejected(i_sum[j], exact[qi, j], eq, ex[j], scale), (case, qi, j)      # never against its own distance
            tight += int(rejected(i_sum[j], f32(exact[qi, j] * f32(0.5)), eq, ex[j], scale))
            useful += 1
    if case < 2:
        assert tight > 0.9 * useful  # and it is not vacuous: a threshold at half the distance is recognised almost always
