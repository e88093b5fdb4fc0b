"""Where the reference's bits depend on whether the compiler contracts a multiply-add -- and which way this repository goes.

Three pieces of reference arithmetic sit in translation units whose flags decide between `fma(a, b, c)` and `round(round(a * b) + c)`, and cannot
be compiled here (their headers need toml++), so the oracle restates them from source with the contraction the reference's own build implies:
    * scalar-quantization decompress  `scale * float(v) + bias`      quantization/scalar/scalar.h:44-46   (used per element by
      CosineSimilarityCompressed, scalar.h:175-184)                                                        -> NOT contracted
    * cosine query norm               `accum += x * x`               core/distance/distance_core.h:45-52 (inlined into the caller's TU, built
      with -march=x86-64: no FMA available, cmake default)                                                 -> NOT contracted
    * SQ inner product epilogue       `scale * ip + bias * sum(q)`   scalar.h:134-141                     -> NOT contracted
These tests construct inputs on which the two readings give DIFFERENT float32 bits, and check that the oracle follows the un-contracted
reading operation for operation (numpy float32 arithmetic never contracts).  The GPU path is held to the oracle bit for bit on the same
inputs in tests/test_gpu_parity.py::test_contraction_sensitive_inputs.  If a future reference build enables contraction there, these fixtures
say exactly which results move."""
import numpy as np
import pytest

from oracle.oracle import COS, I8, IP, L2


def _fma32(a, b, c):
    """float32 fma through float64: a * b is exact in float64 (24 + 24 bits); the one rounding of the sum to 53 bits before the rounding to 24 can
    only matter on a measure-zero set, irrelevant for FINDING inputs where the contracted and un-contracted forms differ"""
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


def contraction_sensitive_sq(seed=3):
    """(scale, bias, codes): int8 codes for which fma(scale, v, bias) != fl(fl(scale * v) + bias) on many elements"""
    rng = np.random.default_rng(seed)
    while True:
        scale = np.float32(rng.uniform(0.001, 0.02))
        bias = np.float32(rng.uniform(-1.0, 1.0))
        v = np.arange(-128, 128, dtype=np.float32)
        plain = (scale * v).astype(np.float32) + bias
        fused = np.array([_fma32(scale, x, bias) for x in v], np.float32)
        differ = np.flatnonzero(plain.view(np.uint32) != fused.view(np.uint32))
        if len(differ) >= 40:
            return scale, bias, (differ - 128).astype(np.int8)


def test_sq_decompress_and_cosine_follow_the_uncontracted_reading(oracle):
    scale, bias, codes = contraction_sensitive_sq()
    d = 64
    rng = np.random.default_rng(1)
    x = rng.choice(codes, size=(30, d)).astype(np.int8)     # rows made only of contraction-sensitive codes
    q = rng.standard_normal((4, d)).astype(np.float32)
    g = np.zeros((30, 2), np.uint32)
    e = np.zeros(1, np.uint32)
    ids, dists = oracle.flat_search(COS, x, q, 30, sq=(float(scale), float(bias)))
    plain = (scale * x.astype(np.float32)).astype(np.float32) + bias        # numpy float32: multiply and add rounded separately
    fused = _fma32(scale, x.astype(np.float32), bias)
    assert (plain.view(np.uint32) != fused.view(np.uint32)).mean() > 0.9
    want_plain = oracle.flat_search(COS, plain, q, 30)                        # CosineSimilarityCompressed == plain cosine on the decompressed rows
    want_fused = oracle.flat_search(COS, fused, q, 30)
    assert (ids == want_plain[0]).all() and (dists.view(np.uint32) == want_plain[1].view(np.uint32)).all()
    assert (dists.view(np.uint32) != want_fused[1].view(np.uint32)).any()     # the contracted reading gives other bits on these inputs
    # the inner-product epilogue scale * ip + bias * sum(q) (scalar.h:134-141): two roundings for the product terms, one for the sum
    ids_ip, d_ip = oracle.flat_search(IP, x, q, 30, sq=(float(scale), float(bias)))
    raw = oracle.distance_batch(IP, q, x)                                      # IP(q_f32, codes) by the reference-order kernel
    off = np.array([bias * np.float32(oracle_sum) for oracle_sum in _reduce4(q)], np.float32)
    want = ((scale * raw).astype(np.float32) + off[:, None]).astype(np.float32)
    for i in range(len(q)):
        assert (d_ip[i].view(np.uint32) == want[i][ids_ip[i]].view(np.uint32)).all()


def _reduce4(q):
    """std::reduce of libstdc++ 11 (<numeric>:299-309) as InnerProductCompressed::fix_argument uses it: groups of four, (q0 + q1) + (q2 + q3)
    added to the running sum, then the tail"""
    out = []
    for row in q.astype(np.float32):
        init = np.float32(0.0)
        i = 0
        while len(row) - i >= 4:
            init = np.float32(init + np.float32(np.float32(row[i] + row[i + 1]) + np.float32(row[i + 2] + row[i + 3])))
            i += 4
        for v in row[i:]:
            init = np.float32(init + v)
        out.append(init)
    return out


def test_cosine_query_norm_follows_the_uncontracted_reading(oracle):
    """norm(): accum += x * x in element order (distance_core.h:45-66); with FMA the partial sums round once instead of twice"""
    rng = np.random.default_rng(7)
    found = 0
    for _ in range(200):
        q = rng.standard_normal((1, 96)).astype(np.float32)
        plain, fused = np.float32(0.0), np.float32(0.0)
        for v in q[0]:
            plain = np.float32(plain + np.float32(v * v))
            fused = _fma32(v, v, fused)
        if plain.view(np.uint32) == fused.view(np.uint32):
            continue
        found += 1
        x = rng.standard_normal((5, 96)).astype(np.float32)
        got = oracle.distance_batch(COS, q, x)[0]
        dot = oracle.distance_batch(IP, q, x)[0]
        nx = np.array([oracle.distance_batch(IP, x[j : j + 1], x[j : j + 1])[0, 0] for j in range(5)], np.float32)  # sum x^2 by the same 16-lane tree
        want = dot / (np.sqrt(nx).astype(np.float32) * np.sqrt(plain).astype(np.float32))
        assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert found >= 20  # a fifth of random queries are sensitive: the choice is not academic
