"""The inequality behind flat search's exactness proof (svsx_flat.hip stage 4), restated in numpy and checked for SOUNDNESS on the
CPU: for every row whose float16 key lies above a threshold tau, the reference's float32 distance (the oracle's) must be no better than the
bound the proof derives from tau -- over ordinary, tiny, huge, clustered and scalar-quantized data.  (That the HIP kernel evaluates this same
inequality is what tests/test_gpu_flat_exact.py pins: its adversarial sets fail the proof exactly where they should and stay bit-exact.)"""
import numpy as np
import pytest

from conftest import deep_like

L2, IP, COS = 0, 1, 2
U, TWO25 = 2.0**-11, 2.0**-25


def shadow_keys(metric, q, x):
    """float16 shadow, float32 accumulation: the keys of svsx_flat.hip (L2: |x_h|^2 - 2 q_h.x_h, IP: -q_h.x_h, cosine: -q_h.x_h / |x_h|)."""
    qh = q.astype(np.float16).astype(np.float32)
    xh = x.astype(np.float16).astype(np.float32)
    n2 = (xh * xh).sum(1, dtype=np.float32)
    sc = qh @ xh.T
    if metric == L2:
        return n2[None, :] - np.float32(2.0) * sc, n2
    if metric == IP:
        return -sc, n2
    return -sc / np.sqrt(np.where(n2 > 0, n2, 1.0)).astype(np.float32)[None, :], n2


def proof_bound(metric, q, tau, xmax2, xmin2, d, q_star=None):
    """Best reference distance a row with key >= tau can have (L2: a lower bound of the squared distance; IP / cosine: an upper bound of the
    similarity), or None where the kernel refuses to prove anything.  float64, same constants as flat_certify_kernel."""
    dpad = (d + 127) & ~127 if 128 < d <= 768 else (d + 15) & ~15  # the shadow's row padding (svsx_api.hip)
    nq = float(np.sqrt((q.astype(np.float64) ** 2).sum()))
    dqs = 0.0 if q_star is None else float(np.sqrt(((q_star.astype(np.float64) - q.astype(np.float64)) ** 2).sum()))
    ab = np.sqrt(dpad) * TWO25
    xmax, xmin = np.sqrt(float(xmax2)), np.sqrt(float(xmin2))
    e16 = U * nq + ab
    eq = dqs + e16
    ex = 1.001 * (U * xmax + ab)
    qh_hi, qh_lo = nq + e16, max(nq - e16, 0.0)
    gam, e = dpad * 2.0**-22, (d + 8) * 2.0**-22
    if not (nq <= 1e30 and xmax <= 1e30 and np.abs(q).max() <= 65504.0):
        return None
    if metric == L2:
        r2 = tau + qh_lo * qh_lo - gam * (xmax + qh_hi) ** 2
        if not r2 > 0:
            return None
        r = np.sqrt(r2) - eq - ex
        return r * r * (1.0 - e) if r > 0 else None
    if metric == IP:
        return -tau + gam * qh_hi * xmax + eq * xmax + nq * ex + 1.001 * e * nq * xmax
    if not (xmin > 0 and nq > 0):
        return None
    return (-tau + abs(tau) * gam + gam * qh_hi + eq + qh_hi * 2.002 * (U + ab / xmin)) / nq + 2.0 * e


def data_sets():
    rng = np.random.default_rng(3)
    yield "deep-like, unit norm", deep_like(4000, 96, 12, seed=1), deep_like(24, 96, 12, seed=2)
    yield "gaussian d = 17", rng.standard_normal((3000, 17)).astype(np.float32), rng.standard_normal((16, 17)).astype(np.float32)
    yield "tiny values (float16 subnormals)", (1e-6 * rng.standard_normal((2000, 32))).astype(np.float32), (1e-6 * rng.standard_normal((8, 32))).astype(np.float32)
    yield "large values", (3e3 * rng.standard_normal((2000, 64))).astype(np.float32), (2e3 * rng.standard_normal((8, 64))).astype(np.float32)
    x = rng.standard_normal((3000, 48)).astype(np.float32)
    x[100:700] = x[0] + np.float32(1e-5) * rng.standard_normal((600, 48)).astype(np.float32)
    yield "cluster below float16 resolution", x, (x[:8] + np.float32(1e-3) * rng.standard_normal((8, 48))).astype(np.float32)
    x = rng.standard_normal((2000, 40)).astype(np.float32) * np.exp(rng.uniform(-6, 6, (2000, 1))).astype(np.float32)
    yield "row norms over five decades", x, rng.standard_normal((8, 40)).astype(np.float32)
    yield "200 dimensions (shadow padded to 256)", rng.standard_normal((2500, 200)).astype(np.float32), rng.standard_normal((8, 200)).astype(np.float32)
    yield "int8-valued rows and queries", rng.integers(-128, 128, (3000, 100)).astype(np.float32), rng.integers(-128, 128, (8, 100)).astype(np.float32)


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_bound_never_excludes_a_row_that_could_be_in_the_answer(oracle, metric):
    checked = 0
    for name, x, q in data_sets():
        keys, n2 = shadow_keys(metric, q, x)
        ref = oracle.distance_batch(metric, q, x)  # the reference's own float32 numbers
        for kk in (1, 40, 256):
            for i in range(q.shape[0]):
                order = np.sort(keys[i])
                tau = float(order[min(kk, len(order)) - 1])
                outside = keys[i] >= np.float32(tau)  # includes the rows AT the threshold: the bound must hold for them too
                bound = proof_bound(metric, q[i], tau, n2.max(), n2.min(), x.shape[1])
                if bound is None or not outside.any():
                    continue
                checked += 1
                if metric == L2:
                    worst = float(ref[i][outside].min())
                    assert worst >= bound, (name, metric, kk, i, worst, bound)
                else:
                    best = float(ref[i][outside].max())
                    assert best <= bound, (name, metric, kk, i, best, bound)
    assert checked > 200


def test_bound_is_tight_enough_to_prove_ordinary_queries(oracle):
    """on deep-like data the bound from the 40th float16 key must leave the 10th exact neighbour provable for (nearly) every query -- otherwise every
    search would fall back to the exhaustive pass."""
    x, q = deep_like(20000, 96, 12, seed=5), deep_like(64, 96, 12, seed=6)
    for metric in (L2, IP, COS):
        keys, n2 = shadow_keys(metric, q, x)
        ref = oracle.distance_batch(metric, q, x)
        proven = 0
        for i in range(q.shape[0]):
            tau = float(np.sort(keys[i])[39])
            bound = proof_bound(metric, q[i], tau, n2.max(), n2.min(), 96)
            kth = np.sort(ref[i])[9] if metric == L2 else -np.sort(-ref[i])[9]
            proven += bound is not None and ((bound > kth) if metric == L2 else (bound < kth))
        assert proven >= 60, (metric, proven)


def test_compressed_query_offset(oracle):
    """EuclideanCompressed (scalar.h:88-94) measures from the query's decompressed code q*: the bound with ||q* - q|| added stays sound against the
    oracle's SQ distances."""
    xs, qs = deep_like(3000, 64, 16, seed=8, normalize=False), deep_like(12, 64, 16, seed=9, normalize=False)
    codes, scale, bias = oracle.sq_compress(xs, 2)
    xdec = (np.float32(scale) * codes.astype(np.float32) + np.float32(bias)).astype(np.float32)
    ids, dists = oracle.flat_search(L2, codes, qs, xs.shape[0], sq=(float(scale), float(bias)))
    ref = np.empty((qs.shape[0], xs.shape[0]), np.float32)
    for i in range(qs.shape[0]):
        ref[i, ids[i]] = dists[i]
    qcode = np.clip(np.round((qs - np.float32(bias)) / np.float32(scale)), -128, 127)
    qstar = (np.float32(scale) * qcode + np.float32(bias)).astype(np.float32)
    keys, n2 = shadow_keys(L2, qs, xdec)
    for i in range(qs.shape[0]):
        tau = float(np.sort(keys[i])[39])
        bound = proof_bound(L2, qs[i], tau, n2.max(), n2.min(), 64, q_star=qstar[i])
        outside = keys[i] >= np.float32(tau)
        assert bound is None or float(ref[i][outside].min()) >= bound
