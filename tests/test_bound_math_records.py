"""The arithmetic of the graph-inline bound records (mode 2, the default product path: csrc/svsx_codes.hip pca_encode_kernel +
build_fat_rows in csrc/svsx_api.hip on the data side, csrc/svsx_search_kernel.h "graph-inline records" on the query side),
restated in numpy -- float64 where the library uses float64, float32 operation by operation where the kernel uses float32 --
and checked against the oracle's exact distances: with the threshold `worst` set to a candidate's OWN exact distance the
bound must never reject it (SearchBuffer::can_skip, index/vamana/search_buffer.h:338-340, drops a candidate only when
worst is strictly better than its distance).  An unsound bound is silent on the GPU -- it changes ids only on rare
inputs -- so the adversarial sets live here: far offset band, near-zero remainder norms, clamped codes, zero rows and
queries, d in {32, 96, 768, 4096}, record sizes 32 / 64 / 128 bytes, L2 / inner product / cosine, scalar-quantized rows
(quantization/scalar/scalar.h:44-46,107-195).  CPU-only; the GPU parity tests check the kernel itself."""
import numpy as np
import pytest

from conftest import deep_like
from oracle.oracle import COS, IP, L2

f32 = np.float32
f64 = np.float64
EPS24 = f32(5.9604645e-8)


def f16_up(v):
    """smallest float16 >= v (v >= 0), as bits (svsx_codes.hip half_round_up); +inf when it does not fit"""
    v = np.asarray(v, f64)
    with np.errstate(over="ignore"):
        h = v.astype(f32).astype(np.float16)
    bits = h.view(np.uint16).copy()
    low = h.astype(f64) < v
    bits[low] += np.uint16(1)
    bits[~(v == v)] = np.uint16(0x7C00)
    return bits


def f16_val(bits):
    return np.asarray(bits, np.uint16).view(np.float16).astype(f32)


class Records:
    """Data side: build_fat_rows (svsx_api.hip) + pca_stats_kernel / pca_encode_kernel (svsx_codes.hip).  The directions come
    from numpy's SVD instead of the library's subspace iteration: ANY orthonormal set is valid for the bound."""

    def __init__(self, z_rows, metric, rb, mean_from_sample=True):
        z = np.asarray(z_rows, f32).astype(f64)  # the vector the metric sees, exactly widened
        n, d = z.shape
        self.metric, self.rb, self.d = metric, rb, d
        l2, cosine = metric == L2, metric == COS
        meta = 4 if l2 else 6
        self.dc = rb - meta
        dcu = min(self.dc, d)
        if cosine:
            nrm = np.sqrt((z * z).sum(1))
            scale = np.where(nrm > 0, 1.0 / np.where(nrm > 0, nrm, 1.0), 0.0)
            zs = z * scale[:, None]
        else:
            zs = z
        # round 6: every metric is centred (cosine: the unit vectors; coordinates() normalises, then subtracts)
        self.mean = np.zeros(d, f32)
        if mean_from_sample:
            self.mean = (zs.sum(0) / n).astype(f32)
        zc = zs - self.mean.astype(f64)
        mean_norm = np.sqrt((self.mean.astype(f64) ** 2).sum())
        self.mean_norm = f32(0.0) if l2 else f32(mean_norm * 1.000001 + 1e-30)
        degenerate = (nrm == 0) if cosine else np.zeros(n, bool)
        # similarity metrics with a mean that matters: the unit mean direction is coordinate 0 of the basis and the query keeps its coordinate along
        # it in float32 (svsx_records.hip / SearchArgs::pca_mean_dir)
        trace = (zc * zc).sum() / n
        self.mean_dir = bool((not l2) and dcu >= 2 and mean_norm * mean_norm > 0.01 * trace)
        _, sv, vt = np.linalg.svd(zc - 0.0, full_matrices=False)
        q64 = np.zeros((dcu, d))
        q64[: min(dcu, vt.shape[0])] = vt[:dcu]
        if vt.shape[0] < dcu:  # fewer samples than directions: complete with any orthonormal rows
            rng = np.random.default_rng(1)
            q64[vt.shape[0] :] = rng.standard_normal((dcu - vt.shape[0], d))
        if self.mean_dir:
            q64 = np.concatenate([(self.mean.astype(f64) / mean_norm)[None, :], q64[: dcu - 1]])
        q64, _ = np.linalg.qr(q64.T)
        q64 = q64.T.copy()
        if self.mean_dir and (q64[0] @ self.mean.astype(f64)) < 0:
            q64[0] = -q64[0]  # (QR may flip the sign; either is an orthonormal basis, keep the direction of the mean)
        sigma = np.zeros(dcu)
        sigma[: min(dcu, sv.shape[0])] = sv[:dcu] / np.sqrt(n)
        self.P = np.zeros((self.dc, d), f32)
        self.P[:dcu] = q64.astype(f32)
        self.delta = f32(np.sqrt(((self.P[:dcu].astype(f64) - q64) ** 2).sum(1)).max() * 1.01 + 1e-9)
        y = zc @ q64.T  # [n, dcu] float64 coordinates
        n2 = (zc * zc).sum(1)
        ny2 = (y * y).sum(1)
        tail = np.sqrt(np.maximum(n2 - ny2, 0.0))
        lo = y.astype(f32).min(0).astype(f64)
        hi = y.astype(f32).max(0).astype(f64)
        pos = sigma > 0
        lo = np.where(pos, np.maximum(lo, -8.0 * sigma), lo)  # outliers beyond 8 sigma clamp (their coding error is in their own ex)
        hi = np.where(pos, np.minimum(hi, 8.0 * sigma), hi)
        hi = np.where(hi > lo, hi, lo)
        self.bias = np.zeros(self.dc, f32)
        if l2:
            self.bias[:dcu] = lo.astype(f32)
            rng_ = (hi - self.bias[:dcu].astype(f64)).max() if dcu else 0.0
            rng_ = max(rng_, (hi - lo).max())
            bmax = np.abs(self.bias.astype(f64)).max()
        else:
            rng_ = max(np.abs(lo).max(), np.abs(hi).max())
            bmax = 0.0
        step = f32(rng_ / (255.0 if l2 else 127.0))
        if not (step > 0) or not (step < 3.0e38):
            step = f32(1.0)
        self.step = step
        tmax = f32(f32(tail.max()) * f32(1.0000002))
        if not (tmax > 0) or not (tmax < 3.0e38):
            tmax = f32(1.0)
        self.tail_unit = f32(tmax / f32(60000.0))
        s, inv = f64(step), 1.0 / f64(step)
        if l2:
            b = self.bias[:dcu].astype(f64)
            c = np.clip(np.rint((y - b) * inv), 0.0, 255.0)
            yh = b + s * c
        else:
            c = np.clip(np.rint(y * inv), -127.0, 127.0)
            yh = s * c
        self.codes = np.zeros((n, self.dc), np.int64)
        self.codes[:, :dcu] = c.astype(np.int64)
        e2 = ((y - yh) ** 2).sum(1)
        self.ex = f16_up(np.sqrt(e2) * inv * (1.0 + 1e-12))
        if cosine:
            self.ex[degenerate] = np.uint16(0x7C00)
        if l2:
            with np.errstate(over="ignore"):
                self.tx = (tail / f64(self.tail_unit)).astype(f32).astype(np.float16).view(np.uint16)  # nearest
            self.ny = None
        else:
            self.ny = f16_up(np.sqrt((yh * yh).sum(1)) * inv * (1.0 + 1e-12))
            self.tx = f16_up(tail / f64(self.tail_unit) * (1.0 + 1e-12))
        self.eq_abs = f32(np.sqrt(f64(dcu)) * 1.2e-7 * (bmax + 255.0 * f64(step)) * 1.01)
        self.ref_extra = f32(0.0) if l2 else f32(mean_norm * 1.01)  # inner product: ||x|| <= ||x - mean|| + ||mean||


def fma32(a, b, c):
    """float32 fused multiply-add: the product of two float32 is exact in float64; one extra rounding is within the slack the
    kernel carries for ANY summation order"""
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


class Query:
    """Query side of search_kernel (svsx_search_kernel.h, `if constexpr (FAT)` block), float32 operation by operation: two half-range
    fma chains per coordinate, their sum, coding, residual / coordinate norms, bracketed remainder norm."""

    def __init__(self, rec: Records, q):
        metric, d = rec.metric, rec.d
        l2, cosine = metric == L2, metric == COS
        dpad = (d + 15) & ~15
        hlen = dpad >> 1
        qp = np.zeros(dpad, f32)
        qp[:d] = np.asarray(q, f32)
        mp = np.zeros(dpad, f32)
        mp[:d] = rec.mean
        Pp = np.zeros((rec.dc, dpad), f32)
        Pp[:, :d] = rec.P
        v = (qp - mp).astype(f32) if l2 else qp
        up = np.zeros((2, rec.dc), f32)
        vvp = np.zeros(2, f32)
        qmp = np.zeros(2, f32)
        for half in range(2):
            for j in range(half * hlen, (half + 1) * hlen):
                up[half] = fma32(Pp[:, j], np.broadcast_to(v[j], (rec.dc,)), up[half])
                vvp[half] = fma32(v[j : j + 1], v[j : j + 1], vvp[half : half + 1])[0]
                if not l2:
                    qmp[half] = fma32(v[j : j + 1], mp[j : j + 1], qmp[half : half + 1])[0]
        with np.errstate(all="ignore"):
            u = (up[0] + up[1]).astype(f32)
            vv = f32(vvp[0] + vvp[1])
            vv_raw = vv
            qm = f32(qmp[0] + qmp[1])
            if cosine:
                u = (u * f32(f32(1.0) / np.sqrt(vv, dtype=f32))).astype(f32)
            if l2:
                c = np.rint(((u - rec.bias).astype(f32) / rec.step).astype(f32))
                c = np.clip(c, 0.0, 255.0)
                c[~(c == c)] = 0.0
                uh = (rec.bias + (rec.step * c.astype(f32)).astype(f32)).astype(f32)
            else:
                c = np.rint((u / rec.step).astype(f32))
                c = np.clip(c, -127.0, 127.0)
                c[~(c == c)] = 0.0
                uh = (rec.step * c.astype(f32)).astype(f32)
                if rec.mean_dir:  # coordinate 0 stays in float32: no code, no residual
                    c[0] = 0.0
                    uh[0] = u[0]
            r = (u - uh).astype(f32)
            r2 = f32(0.0)
            un2 = f32(0.0)
            for i in range(rec.dc):
                r2 = fma32(r[i : i + 1], r[i : i + 1], np.array([r2], f32))[0]
                un2 = fma32(u[i : i + 1], u[i : i + 1], np.array([un2], f32))[0]
            self.codes = c.astype(np.int64)
            qq = int((self.codes**2).sum())
            if cosine:
                vv = f32(1.0) if vv > 0 else vv
            vn = f32(np.sqrt(vv, dtype=f32) * f32(1.00001))
            extra = (d // 2 + 6) if cosine else 0
            urel = f32(f32(f32(f32(d + 3 + extra) * EPS24) * f32(1.01)) + rec.delta)
            erot = f32(f32(np.sqrt(f32(rec.dc), dtype=f32) * urel) * vn)
            nu = np.sqrt(un2, dtype=f32)
            eq = f32(f32(f32(np.sqrt(r2, dtype=f32) * f32(1.0001)) + erot) + rec.eq_abs)
            drel = f32(f32(f32(d + 4) * EPS24) * f32(1.01))
            alo = f32(f32(nu * f32(0.99999)) - erot)
            ahi = f32(f32(nu * f32(1.00001)) + erot)
            alo2 = f32(alo * alo) if alo > 0 else f32(0.0)
            thi2 = f32(f32(vv * f32(f32(1.0) + drel)) - alo2)
            tlo2 = f32(f32(vv * f32(f32(1.0) - drel)) - f32(ahi * ahi))
            tq_hi = f32(np.sqrt(thi2 if thi2 > 0 else f32(0.0), dtype=f32) * f32(1.00001))
            tq_lo = f32(np.sqrt(tlo2 if tlo2 > 0 else f32(0.0), dtype=f32) * f32(0.99999))
            if not (eq < 3.0e38) or not (tq_hi < 3.0e38):
                eq = f32(np.inf)
                tq_hi = f32(np.inf)
            # upper bound of q . mean (cosine: of the unit query), as the kernel forms it
            qmu = qm
            if cosine:
                qmu = f32(qmu * f32(f32(1.0) / np.sqrt(vv_raw, dtype=f32)))
            qslack = f32(f32(f32(f32(f32(d // 2 + 6) * EPS24) * f32(1.01)) * vn) * rec.mean_norm)
            self.qmu = f32(0.0) if l2 else f32(f32(qmu + qslack) + f32(np.abs(qmu) * f32(1.2e-7)))
        self.qq, self.eq, self.tq_hi, self.tq_lo = qq, eq, tq_hi, tq_lo
        self.s2m = f32(f32(rec.step * rec.step) * (f32(0.99999) if l2 else f32(1.0)))
        self.nuh = f32(f32(rec.step * np.sqrt(f32(qq), dtype=f32)) * f32(1.00001))
        self.u0s = f32(0.0)
        if (not l2) and rec.mean_dir:
            with np.errstate(all="ignore"):
                self.nuh = f32(np.sqrt(f32(f32(self.nuh * self.nuh) + f32(u[0] * u[0])), dtype=f32) * f32(1.00001))
                self.u0s = f32(u[0] * rec.step)
        self.nq = vn
        self.rec = rec


def rejected(qs: Query, worst, wunit=None):
    """the lane-parallel bound of the main loop (svsx_search_kernel.h, `if (full)` block) for every record against ONE threshold per record.
    wunit: scalar-quantized L2 only -- fl(1 / fl(scale * scale)), the factor that takes a buffer distance (scale^2 * I, scalar.h:88-94) to the
    code space the records of such an index live in"""
    rec = qs.rec
    worst = np.asarray(worst, f32)
    with np.errstate(all="ignore"):
        if rec.metric == L2:
            if wunit is not None:
                worst = (worst * f32(wunit)).astype(f32)
            wp = (worst * f32(1.0001)).astype(f32)
            # v_sqrt_f32 is accurate to 1 ulp: take the correctly rounded root and the two neighbours, the bound must hold for each
            out = np.zeros(worst.shape, bool)
            root = np.sqrt(wp, dtype=f32)
            for nudge in (0, 1, -1):
                rt = (root.view(np.int32) + np.int32(nudge)).view(f32) if nudge else root
                r2 = ((f32(2.0) * rt).astype(f32) * f32(1.00001)).astype(f32)
                fi = (qs.qq + (rec.codes**2).sum(1) - 2 * (rec.codes * qs.codes[None, :]).sum(1)).astype(f32)
                ex = (f16_val(rec.ex) * rec.step).astype(f32)
                tx = (f16_val(rec.tx) * rec.tail_unit).astype(f32)
                e = (qs.eq + ex).astype(f32)
                txe = ((tx * f32(4.94e-4)).astype(f32) + f32(rec.tail_unit * f32(6.0e-8))).astype(f32)
                t1 = (qs.tq_lo - (tx + txe).astype(f32)).astype(f32)
                t2 = ((tx - txe).astype(f32) - qs.tq_hi).astype(f32)
                t = np.where(t1 > t2, t1, t2)
                bb = (np.where(t > 0, t, f32(0.0)) * f32(0.99999)).astype(f32)
                rhs = ((wp - (bb * bb).astype(f32)).astype(f32) + ((e * (r2 + e).astype(f32)).astype(f32) * f32(1.00001)).astype(f32)).astype(f32)
                out |= (fi * qs.s2m).astype(f32) > rhs
            return out
        dot = (rec.codes * qs.codes[None, :]).sum(1)
        main = (qs.s2m * dot.astype(f32)).astype(f32)
        if rec.mean_dir:
            main = fma32(np.broadcast_to(qs.u0s, main.shape).astype(f32), rec.codes[:, 0].astype(f32), main)
        ex = (f16_val(rec.ex) * rec.step).astype(f32)
        ny = (f16_val(rec.ny) * rec.step).astype(f32)
        tx = (f16_val(rec.tx) * rec.tail_unit).astype(f32)
        pos = ((qs.nuh * ex).astype(f32) + (qs.eq * (ny + ex).astype(f32)).astype(f32)).astype(f32)
        pos = (pos + (qs.tq_hi * tx).astype(f32)).astype(f32)
        ub = ((main + pos).astype(f32) + ((np.abs(main) + pos).astype(f32) * f32(4.0e-6)).astype(f32)).astype(f32)
        ub = ((ub + qs.qmu).astype(f32) + ((np.abs(ub) + np.abs(qs.qmu)).astype(f32) * f32(1.2e-7)).astype(f32)).astype(f32)  # + q . mean
        if rec.metric == COS:
            ub = (ub + f32(f32(f32(rec.d + 8) * f32(6.3e-8)) + f32(1.0e-7))).astype(f32)
        else:
            w = ((ny + ex).astype(f32) + tx).astype(f32)
            w = (w + rec.ref_extra).astype(f32)
            ub = (ub + (f32(f32(f32(rec.d + 2) * f32(6.1e-8)) * qs.nq) * w).astype(f32)).astype(f32)
        return ub < worst


# ---- adversarial sets -------------------------------------------------------------------------------------------------


def _sets(d):
    rng = np.random.default_rng(d)
    lat = min(24, d // 2)
    x = deep_like(300, d, lat, seed=1)
    yield "deep-like", x, deep_like(24, d, lat, seed=2)
    xo = (deep_like(300, d, 8, seed=5, normalize=False) * f32(1e3) + f32(5e3)).astype(f32)
    yield "large offset", xo, (xo[:16] + rng.standard_normal((16, d)).astype(f32)).astype(f32)
    xb = (deep_like(300, d, 8, seed=7, normalize=False) * f32(1e-2) + f32(1e3)).astype(f32)  # |x| / spread = 10^5: the far offset band
    yield "far offset band", xb, (xb[:16] + f32(1e-2) * rng.standard_normal((16, d)).astype(f32)).astype(f32)
    xs = deep_like(300, d, 8, seed=6, normalize=False) * f32(1e-3)
    xs[::50] *= f32(100.0)  # outliers stretch the range: clamped codes
    yield "clamped codes", xs.astype(f32), (xs[:16] * f32(30.0)).astype(f32)  # queries far outside the coded range
    # near-zero remainder norms: the data lies (to rounding) inside the coded subspace
    k = 8
    basis, _ = np.linalg.qr(rng.standard_normal((d, k)))
    xz = (rng.standard_normal((300, k)) @ basis.T).astype(f32)
    yield "zero tails", xz, np.concatenate([xz[:8] + f32(1e-6), rng.standard_normal((8, d)).astype(f32)])
    xu = rng.standard_normal((300, d)).astype(f32)
    xu /= np.linalg.norm(xu, axis=1, keepdims=True)
    yield "equidistant", xu, np.zeros((4, d), f32) + f32(1e-4)
    yield "duplicates", np.concatenate([xu[:100], xu[:100]]), (xu[:12] + f32(1e-3)).astype(f32)
    xzr = xu.copy()
    xzr[::7] = 0.0  # zero rows: NaN under cosine (never bounded), ordinary rows otherwise
    yield "zero rows and a zero query", xzr, np.concatenate([np.zeros((1, d), f32), xu[:8]])


def _check(oracle, metric, x, q, rb, name, sq=None):
    if sq is None:
        z = x
        exact = oracle.distance_batch(metric, q, x)  # the reference's float32 values, bit for bit
        rec = Records(z, metric, rb)
    else:
        codes, scale, bias = sq
        z = (f32(scale) * codes.astype(f32)).astype(f32) + f32(bias)  # scalar.h:44-46, multiply and add rounded separately
        rec = Records(z, metric, rb)
        rec.ref_extra = f32((f64(rec.ref_extra) / 1.01 + 2.0 * abs(f64(bias)) * np.sqrt(f64(x.shape[1]))) * 1.01)
        exact = x  # the caller passes the oracle's compressed distances [nq, n]
    fired = 0
    for qi in range(q.shape[0]):
        qs = Query(rec, q[qi])
        own = rejected(qs, exact[qi])
        assert not own.any(), (name, metric, rb, qi, np.nonzero(own)[0][:5])
        with np.errstate(all="ignore"):
            if metric == L2:
                far = rejected(qs, (exact[qi] * f32(0.25)).astype(f32))
            else:  # similarities: a threshold clearly better than the candidate
                far = rejected(qs, (exact[qi] + np.abs(exact[qi]) + f32(0.5) * np.abs(exact[qi]).max()).astype(f32))
        fired += int(far.sum())
    return fired, q.shape[0] * x.shape[0]


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("d,rb", [(32, 32), (96, 28), (96, 32), (96, 128), (128, 28), (768, 64), (768, 128)])
def test_record_bound_never_rejects_a_candidate_against_its_own_distance(oracle, metric, d, rb):
    total_fired = 0
    for name, x, q in _sets(d):
        fired, total = _check(oracle, metric, x, q, rb, name)
        if name == "deep-like":
            total_fired = fired
            assert fired > 0.5 * total, (fired, total)  # and it is not vacuous on the bench distribution
    assert total_fired > 0


@pytest.mark.parametrize("metric", [L2, IP, COS])
def test_record_bound_d4096(oracle, metric):
    d = 4096
    x = deep_like(64, d, 16, seed=11)
    q = deep_like(3, d, 16, seed=12)
    fired, total = _check(oracle, metric, x, q, 128, "d4096")
    assert fired > 0
    xo = (x * f32(10.0) + f32(100.0)).astype(f32)
    _check(oracle, metric, xo, (xo[:2] + f32(0.01)).astype(f32), 128, "d4096 offset")


@pytest.mark.parametrize("metric", [IP, COS])
@pytest.mark.parametrize("rb", [32, 128])
def test_record_bound_scalar_quantized_rows(oracle, metric, rb):
    """config 4's row type: int8 codes with one global (scale, bias); the records describe the DECOMPRESSED vectors and the exact
    distances are the reference's compressed distances (scalar.h:107-195) from the oracle's search-free path"""
    d = 768
    xf = deep_like(200, d, 32, seed=21, normalize=False)
    q = deep_like(6, d, 32, seed=22, normalize=False)
    codes, sc, bias = oracle.sq_compress(xf)
    # exact compressed distances of every (query, row): flat search with k = n returns all of them, by id
    ids, dists = oracle.flat_search(metric, codes, q, xf.shape[0], sq=(sc, bias))
    exact = np.empty((q.shape[0], xf.shape[0]), f32)
    for qi in range(q.shape[0]):
        exact[qi, ids[qi].astype(np.int64)] = dists[qi]
    fired, total = _check(oracle, metric, exact, q, rb, "sq", sq=(codes, sc, bias))
    assert fired > 0


# ---- round 6: records over integer rows (raw int8 / uint8 rows searched with queries of the same type; the scalar-quantized L2 distance) -----
def _int_sets(d, dt):
    rng = np.random.default_rng(d + (7 if dt == np.int8 else 0))
    lo, hi = (-128, 127) if dt == np.int8 else (0, 255)
    off = 0 if dt == np.int8 else 128

    def quant(a, gain):
        return np.clip(np.round(a * gain + off), lo, hi).astype(dt)

    xf = deep_like(300, d, min(24, d // 2), seed=31, normalize=False)
    g = 100.0 / float(np.abs(xf).max())
    yield "deep-like", quant(xf, g), quant(deep_like(16, d, min(24, d // 2), seed=32, normalize=False), g)
    yield "small values", quant(xf, g * 0.03), quant(deep_like(8, d, min(24, d // 2), seed=33, normalize=False), g * 0.03)
    xs = quant(xf, g * 4.0)  # most elements saturate
    yield "saturated", xs, np.concatenate([xs[:6], np.full((1, d), hi, dt), np.full((1, d), lo, dt)])
    xc = quant(xf, g)
    xc[::9] = 7      # constant rows
    xc[::13] = 0     # zero rows: NaN under cosine (never bounded)
    xc[5] = xc[4]    # duplicates
    yield "constant, zero and duplicated rows", xc, np.concatenate([np.zeros((1, d), dt), np.full((1, d), 3, dt), xc[:6]])
    yield "uniform noise", rng.integers(lo, hi + 1, (300, d)).astype(dt), rng.integers(lo, hi + 1, (8, d)).astype(dt)


@pytest.mark.parametrize("metric", [L2, IP, COS])
@pytest.mark.parametrize("dt", [np.int8, np.uint8])
@pytest.mark.parametrize("d,rb", [(96, 32), (768, 64), (768, 128)])
def test_record_bound_integer_rows(oracle, metric, dt, d, rb):
    """int8 x int8 / uint8 x uint8: the reference's values are exact int32 sums narrowed to float32 (euclidean.h:265-313, inner_product.h:222-267)
    and, under cosine, float(sum ab) / (||a|| sqrt(float(sum bb))) (cosine.h:261-326); the records are built over float(row element) and the
    query's image is float(query element) -- both exact"""
    fired_deep = 0
    for name, x, q in _int_sets(d, dt):
        exact = oracle.distance_batch(metric, q, x)
        rec = Records(x.astype(f32), metric, rb)
        fired = 0
        for qi in range(q.shape[0]):
            qs = Query(rec, q[qi].astype(f32))
            own = rejected(qs, exact[qi])
            assert not own.any(), (name, metric, rb, qi, np.nonzero(own)[0][:5])
            with np.errstate(all="ignore"):
                if metric == L2:
                    far = rejected(qs, (exact[qi] * f32(0.25)).astype(f32))
                else:
                    far = rejected(qs, (exact[qi] + np.abs(exact[qi]) + f32(0.5) * np.abs(exact[qi]).max()).astype(f32))
            fired += int(far.sum())
        if name == "deep-like":
            fired_deep = fired
            assert fired > 0.5 * q.shape[0] * x.shape[0], (fired, q.shape[0] * x.shape[0])
        # float32 queries over the same rows (pairs (float, int8) / (float, uint8), euclidean.h:318-358): the same records, the float trees
        qf = (q[:6].astype(f32) + f32(0.37) * np.arange(1, 7, dtype=f32)[:, None]).astype(f32)
        exact_f = oracle.distance_batch(metric, qf, x)
        for qi in range(qf.shape[0]):
            own = rejected(Query(rec, qf[qi]), exact_f[qi])
            assert not own.any(), (name, "float queries", metric, rb, qi, np.nonzero(own)[0][:5])
    assert fired_deep > 0


def _sq_compress_query(q, scale, bias, dt):
    """detail::compress (scalar.h:37-42): clamp(std::round((v - bias) / scale)) in float32, half away from zero"""
    t = ((np.asarray(q, f32) - f32(bias)).astype(f32) / f32(scale)).astype(f32).astype(f64)
    r = np.trunc(t + np.copysign(0.5, t))
    lo, hi = (-128, 127) if dt == np.int8 else (0, 255)
    return np.clip(r, lo, hi).astype(dt)


@pytest.mark.parametrize("rb", [32, 64, 128])
@pytest.mark.parametrize("code", [2, 3])
def test_record_bound_scalar_quantized_l2(oracle, rb, code):
    """EuclideanCompressed (scalar.h:58-105): distance = fl(scale * scale) * float(sum (cq - cx)^2) between the CODED query and the coded row; the
    records live in code space and the kernel takes the buffer's worst distance there with one multiplication (fat_wunit)"""
    d = 768
    dt = np.int8 if code == 2 else np.uint8
    for seed, gain, shift in ((41, 1.0, 0.0), (43, 1e-3, 5.0), (45, 300.0, -1e4)):
        xf = (deep_like(200, d, 32, seed=seed, normalize=False) * f32(gain) + f32(shift)).astype(f32)
        q = (deep_like(6, d, 32, seed=seed + 1, normalize=False) * f32(gain) + f32(shift)).astype(f32)
        q[0] = xf[3]
        q[1] = q[1] * f32(50.0)  # far outside the coded range: clamped query codes
        codes, sc, bias = oracle.sq_compress(xf, code)
        assert codes.dtype == dt
        ids, dists = oracle.flat_search(L2, codes, q, xf.shape[0], sq=(sc, bias))
        exact = np.empty((q.shape[0], xf.shape[0]), f32)
        for qi in range(q.shape[0]):
            exact[qi, ids[qi].astype(np.int64)] = dists[qi]
        rec = Records(codes.astype(f32), L2, rb)
        wunit = f32(1.0) / f32(f32(sc) * f32(sc))
        fired = 0
        for qi in range(q.shape[0]):
            qs = Query(rec, _sq_compress_query(q[qi], sc, bias, dt).astype(f32))
            own = rejected(qs, exact[qi], wunit)
            assert not own.any(), (seed, rb, qi, np.nonzero(own)[0][:5])
            fired += int(rejected(qs, (exact[qi] * f32(0.25)).astype(f32), wunit).sum())
        assert fired > 0
