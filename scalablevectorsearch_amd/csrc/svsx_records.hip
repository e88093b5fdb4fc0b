// svsx_records.hip -- svsx_index_build_bound_codes: the host side of the result-neutral bound schemes (include/svsx.h).  Graph-inline records:
// sample -> leading principal directions (float64, host) -> per-vector records and node blocks (device kernels: svsx_codes.hip), the record
// size chosen from the spectrum and from the free HBM; 128-byte code rows (svsx_distance.hip).  No reference counterpart.
#include "svsx_host.h"

namespace {

// cyclic Jacobi eigen-decomposition of a symmetric d x d matrix (row-major, f64): A -> diagonal, V columns = eigenvectors
void jacobi_eigh(std::vector<double>& A, int d, std::vector<double>& V) {
    V.assign(static_cast<size_t>(d) * d, 0.0);
    for (int i = 0; i < d; ++i) V[static_cast<size_t>(i) * d + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) (i == j ? diag : off) += A[static_cast<size_t>(i) * d + j] * A[static_cast<size_t>(i) * d + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < d - 1; ++p) {
            for (int q = p + 1; q < d; ++q) {
                const double apq = A[static_cast<size_t>(p) * d + q];
                if (apq == 0.0) continue;
                const double theta = (A[static_cast<size_t>(q) * d + q] - A[static_cast<size_t>(p) * d + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < d; ++k) { // columns p, q
                    const double akp = A[static_cast<size_t>(k) * d + p], akq = A[static_cast<size_t>(k) * d + q];
                    A[static_cast<size_t>(k) * d + p] = c * akp - sn * akq;
                    A[static_cast<size_t>(k) * d + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) { // rows p, q
                    const double apk = A[static_cast<size_t>(p) * d + k], aqk = A[static_cast<size_t>(q) * d + k];
                    A[static_cast<size_t>(p) * d + k] = c * apk - sn * aqk;
                    A[static_cast<size_t>(q) * d + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {
                    const double vkp = V[static_cast<size_t>(k) * d + p], vkq = V[static_cast<size_t>(k) * d + q];
                    V[static_cast<size_t>(k) * d + p] = c * vkp - sn * vkq;
                    V[static_cast<size_t>(k) * d + q] = sn * vkp + c * vkq;
                }
            }
        }
    }
}

float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t u;
    if (ex == 0) {
        if (man == 0) u = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            u = sign | ((127 - 15 - e) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (ex == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((ex + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}


// modified Gram-Schmidt on the m columns of V (d x m, column j at V[i * m + j]); twice for orthogonality to rounding
void orthonormalize_columns(std::vector<double>& V, uint32_t d, uint32_t m) {
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t j = 0; j < m; ++j) {
            for (uint32_t k = 0; k < j; ++k) {
                double dot = 0.0;
                for (uint32_t i = 0; i < d; ++i) dot += V[static_cast<size_t>(i) * m + j] * V[static_cast<size_t>(i) * m + k];
                for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] -= dot * V[static_cast<size_t>(i) * m + k];
            }
            double nrm = 0.0;
            for (uint32_t i = 0; i < d; ++i) nrm += V[static_cast<size_t>(i) * m + j] * V[static_cast<size_t>(i) * m + j];
            nrm = std::sqrt(nrm);
            if (!(nrm > 1e-300)) { // degenerate direction: replace by a unit vector and let the next repetition clean it up
                for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] = (i == j % d) ? 1.0 : 0.0;
                nrm = 1.0;
            }
            for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] /= nrm;
        }
}

constexpr int kRecordsDoNotFit = -1000; // build_fat_rows: not even 32-byte records fit the free HBM (the caller falls back)

// graph-inline records: sample -> leading principal directions (host, f64) -> per-vector records (device) -> node rows (device).
// *asked_rb / *built_rb: the record size the spectrum rule (or the caller) wanted and the one that fitted the free HBM.
constexpr int kRecordsDoNotPay = -1001; // build_fat_rows(only_if_worth): the size rule expects no saving over the plain kernel on this data
int build_fat_rows(svsx_index* ix, uint32_t forced_rb, uint32_t* asked_rb, uint32_t* built_rb, bool only_if_worth) {
    const uint32_t d = static_cast<uint32_t>(ix->d), n = static_cast<uint32_t>(ix->n);
    const size_t es = dtype_size(ix->data_dtype);
    const bool l2 = ix->metric == M_L2, cosine = ix->metric == M_COS;
    // the vector the metric sees: scalar-quantized rows decompressed under inner product / cosine (scalar.h:107-195), but NOT under L2, whose
    // compressed distance is scale^2 x the integer distance between CODED query and coded row (scalar.h:58-105): those records live in code
    // space, like the records of raw int8 / uint8 rows (float(element): the VNNI pairs and the float32-query pairs see the same vector)
    const bool decomp = ix->sq_enabled && !l2;
    const uint32_t meta = l2 ? 4u : 6u; // bytes of float16 fields at the end of a record
    // 1. sample (evenly strided) -> host, the vector the metric sees (decompressed / normalised), f64
    const size_t want = d <= 128 ? 65536 : (d <= 256 ? 32768 : 8192);
    const size_t ns = n < want ? n : want;
    const size_t step = n / ns;
    std::vector<unsigned char> raw(ns * d * es);
    HIP_TRY(hipMemcpy2D(raw.data(), d * es, ix->d_data, step * ix->row_bytes, d * es, ns, hipMemcpyDeviceToHost));
    std::vector<double> xs(ns * d);
    for (size_t i = 0; i < ns * d; ++i) {
        float f;
        switch (ix->data_dtype) {
            case DT_F32: f = reinterpret_cast<const float*>(raw.data())[i]; break;
            case DT_F16: f = half_bits_to_float(reinterpret_cast<const uint16_t*>(raw.data())[i]); break;
            case DT_I8: f = static_cast<float>(reinterpret_cast<const int8_t*>(raw.data())[i]); break;
            default: f = static_cast<float>(raw[i]); break;
        }
        if (decomp) {
            const float t = ix->sq_scale * f;
            f = t + ix->sq_bias;
        }
        xs[i] = f;
        if (!(xs[i] - xs[i] == 0.0)) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    }
    if (cosine)
        for (size_t r = 0; r < ns; ++r) {
            double nrm = 0.0;
            for (uint32_t j = 0; j < d; ++j) nrm += xs[r * d + j] * xs[r * d + j];
            nrm = nrm > 0.0 ? 1.0 / std::sqrt(nrm) : 0.0;
            for (uint32_t j = 0; j < d; ++j) xs[r * d + j] *= nrm;
        }
    const uint32_t pd = (d + 15u) & ~15u;
    std::vector<float> mean_f(pd, 0.0f);
    {   // every metric is centred (round 6; svsx_codes.hip has the geometry): under cosine the mean of the UNIT vectors
        std::vector<double> mean(d, 0.0);
        for (size_t r = 0; r < ns; ++r)
            for (uint32_t j = 0; j < d; ++j) mean[j] += xs[r * d + j];
        for (uint32_t j = 0; j < d; ++j) mean_f[j] = static_cast<float>(mean[j] / static_cast<double>(ns));
        for (size_t r = 0; r < ns; ++r)
            for (uint32_t j = 0; j < d; ++j) xs[r * d + j] -= static_cast<double>(mean_f[j]);
    }
    double mean_norm = 0.0;
    for (uint32_t j = 0; j < d; ++j) mean_norm += static_cast<double>(mean_f[j]) * static_cast<double>(mean_f[j]);
    mean_norm = std::sqrt(mean_norm);
    // 2. second-moment matrix (upper triangle, then mirrored)
    std::vector<double> cov(static_cast<size_t>(d) * d, 0.0);
    for (size_t r = 0; r < ns; ++r) {
        const double* row = &xs[r * d];
        for (uint32_t i = 0; i < d; ++i) {
            const double ri = row[i];
            double* c = &cov[static_cast<size_t>(i) * d];
            for (uint32_t j = i; j < d; ++j) c[j] += ri * row[j];
        }
    }
    for (uint32_t i = 0; i < d; ++i)
        for (uint32_t j = i; j < d; ++j) {
            const double v = cov[static_cast<size_t>(i) * d + j] / static_cast<double>(ns);
            cov[static_cast<size_t>(i) * d + j] = v;
            cov[static_cast<size_t>(j) * d + i] = v;
        }
    // 3. leading directions by subspace iteration + Rayleigh-Ritz (ANY orthonormal set is valid for the bound; a good one
    //    makes it tight).  m >= dc columns for convergence margin.
    // candidate record sizes: a record larger than a third of the row it protects cannot pay for itself
    uint32_t rb_max = 32;
    if (ix->row_bytes >= 3 * 64) rb_max = 64;
    if (ix->row_bytes >= 3 * 128) rb_max = 128;
    const uint32_t m = d < rb_max - meta + 12u ? d : rb_max - meta + 12u;
    std::vector<double> V(static_cast<size_t>(d) * m), W(static_cast<size_t>(d) * m);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (auto& v : V) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        v = static_cast<double>(static_cast<int64_t>(lcg >> 11) % 2000001 - 1000000) * 1e-6;
    }
    orthonormalize_columns(V, d, m);
    for (int it = 0; it < 48; ++it) {
        std::fill(W.begin(), W.end(), 0.0);
        for (uint32_t i = 0; i < d; ++i) {
            double* w = &W[static_cast<size_t>(i) * m];
            const double* c = &cov[static_cast<size_t>(i) * d];
            for (uint32_t k = 0; k < d; ++k) {
                const double cik = c[k];
                const double* v = &V[static_cast<size_t>(k) * m];
                for (uint32_t j = 0; j < m; ++j) w[j] += cik * v[j];
            }
        }
        V.swap(W);
        orthonormalize_columns(V, d, m);
    }
    // Rayleigh-Ritz: T = V^T C V (m x m), eigenvectors Z, V <- V Z ordered by decreasing eigenvalue
    std::fill(W.begin(), W.end(), 0.0);
    for (uint32_t i = 0; i < d; ++i)
        for (uint32_t k = 0; k < d; ++k) {
            const double cik = cov[static_cast<size_t>(i) * d + k];
            for (uint32_t j = 0; j < m; ++j) W[static_cast<size_t>(i) * m + j] += cik * V[static_cast<size_t>(k) * m + j];
        }
    std::vector<double> T(static_cast<size_t>(m) * m, 0.0), Z;
    for (uint32_t a = 0; a < m; ++a)
        for (uint32_t b = 0; b < m; ++b) {
            double t = 0.0;
            for (uint32_t i = 0; i < d; ++i) t += V[static_cast<size_t>(i) * m + a] * W[static_cast<size_t>(i) * m + b];
            T[static_cast<size_t>(a) * m + b] = t;
        }
    for (uint32_t a = 0; a < m; ++a)
        for (uint32_t b = a + 1; b < m; ++b) {
            const double t = 0.5 * (T[static_cast<size_t>(a) * m + b] + T[static_cast<size_t>(b) * m + a]);
            T[static_cast<size_t>(a) * m + b] = T[static_cast<size_t>(b) * m + a] = t;
        }
    jacobi_eigh(T, static_cast<int>(m), Z);
    std::vector<uint32_t> order(m);
    for (uint32_t i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return T[static_cast<size_t>(x) * m + x] > T[static_cast<size_t>(y) * m + y]; });
    // Record size by expected lines per hop: block lines (4 + RB bytes per slot) + the rows of the candidates the bound cannot
    // exclude.  How many it excludes follows the share f of the sample's second moment the coded directions carry -- the remainder
    // only enters the bound through a product of norms, which is loose: measured (profiles/r02_notes.md) the bound is useless
    // below f ~ 0.6 and rejects ~88 % of the candidates at f ~ 0.9 -- modelled as rej = 0.88 clamp((f - 0.6) / 0.3, 0, 1) -- and a little more
    // towards f = 1 (+ 0.05 clamp((f - 0.9) / 0.1, 0, 1): 0.86 -> 0.926 measured between 64- and 128-byte records on config 4's rows,
    // profiles/r05_notes.md).  A row of scalar-quantized codes costs far more than its lines: every element is decompressed and multiplied
    // (2.5 vector instructions, scalar.h:155-195) on a kernel whose few resident waves are bound by their instruction count, not by lines: a
    // 768-byte row is ~34 wave instructions, doubling the records of a 128-slot row ~50 -- weighted 8x (round 5: 128-byte records beat 64-byte
    // ones by 18 % on 768-byte rows although they double the block).  A larger record must win by 10 % (the 32-byte form has
    // static-dimension kernels and the fewest registers).
    double trace = 0.0;
    for (uint32_t i = 0; i < d; ++i) trace += cov[static_cast<size_t>(i) * d + i];
    const double row_lines = static_cast<double>((ix->row_bytes + 127) / 128) * (decomp ? 8.0 : 1.0); // (integer paths: v_dot4, no decompression)
    const double cand = 0.9 * static_cast<double>(ix->gstride);
    // Round 6: float32 / float16 rows also have a 28-byte form (a 16- and a 12-byte plane: a 64-slot block is 16 lines instead of 18, four codes
    // fewer); it competes with the 32-byte form on the estimate alone (same kernel family, same registers), larger records must win by 10 %.
    const bool has28 = (ix->data_dtype == DT_F32 || ix->data_dtype == DT_F16) && !ix->sq_enabled;
    uint32_t rb = 32;
    double best_cost = 0.0;
    for (uint32_t r = has28 ? 28u : 32u; r <= rb_max; r = r == 28u ? 32u : r * 2u) {
        double got = 0.0;
        for (uint32_t c = 0; c < r - meta && c < m; ++c) got += std::max(T[static_cast<size_t>(order[c]) * m + order[c]], 0.0);
        const double f = trace > 0.0 ? got / trace : 1.0;
        const double rej = 0.88 * std::min(std::max((f - 0.6) / 0.3, 0.0), 1.0) + 0.05 * std::min(std::max((f - 0.9) / 0.1, 0.0), 1.0);
        const double cost = static_cast<double>(ix->gstride) * (4.0 + r) / 128.0 + cand * (1.0 - rej) * row_lines;
        if (r == (has28 ? 28u : 32u) || (r == 32u && rb == 28u && cost < best_cost) || cost < 0.9 * best_cost) {
            best_cost = cost;
            rb = r;
        }
    }
    // mode 3: the same estimate for the plain kernel (ids + every neighbour's row); records that are not expected to save a tenth of the lines are
    // not built -- on full-rank data under a weak bound they cost more than they save (1 M x 128 sift-like uint8, L2: 0.07 % rejected, 0.80 x the
    // plain kernel's rate; profiles/r06_notes.md)
    if (only_if_worth && forced_rb == 0) {
        const double plain_cost = static_cast<double>(ix->gstride) * 4.0 / 128.0 + cand * static_cast<double>((ix->row_bytes + 127) / 128);
        if (best_cost > 0.9 * plain_cost) return kRecordsDoNotPay;
    }
    if (forced_rb != 0) { // the caller asked for this record size (svsx_index_build_bound_codes mode 32 / 64 / 128)
        if (forced_rb == 28 && !has28) return fail(SVSX_NOT_IMPLEMENTED, "28-byte records exist for float32 / float16 rows");
        if (forced_rb > rb_max) return fail(SVSX_INVALID_ARGUMENT, "records of %u bytes exceed a third of the %zu-byte rows they protect", forced_rb, ix->row_bytes);
        rb = forced_rb;
    }
    // Memory: n x (4 + rb) x slots for the blocks (+ n x rb for the records they are assembled from, + the degree classes) -- 2.3 GB for
    // 1 M x R 64 with 32-byte records, 169 GB for 10 M x R 128 with 128-byte ones.  A size that does not fit what is free shrinks (a 32-byte
    // record still rejects most candidates of a low-rank set); when not even 32 bytes fit the caller falls back to another scheme.
    *asked_rb = rb;
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        auto need = [&](uint32_t r) { return static_cast<double>(n) * (static_cast<double>((ix->gstride * (4 + r) + 127) & ~size_t(127)) + r + 1.0) + (64 << 20); };
        while (rb > 32 && need(rb) > 0.98 * static_cast<double>(free_b)) rb /= 2;
        if (rb == 32 && has28 && need(rb) > 0.98 * static_cast<double>(free_b)) rb = 28;
        if (need(rb) > 0.98 * static_cast<double>(free_b)) return kRecordsDoNotFit;
    }
    *built_rb = rb;
    const uint32_t dc = rb - meta;
    const uint32_t dcu = dc < m ? dc : m; // directions really used (d < dc cannot happen: d >= 32 and m >= dc unless d is smaller)
    std::vector<double> Q(static_cast<size_t>(d) * dcu), sigma(dcu);
    for (uint32_t c = 0; c < dcu; ++c) {
        for (uint32_t i = 0; i < d; ++i) {
            double v = 0.0;
            for (uint32_t j = 0; j < m; ++j) v += V[static_cast<size_t>(i) * m + j] * Z[static_cast<size_t>(j) * m + order[c]];
            Q[static_cast<size_t>(i) * dcu + c] = v;
        }
        sigma[c] = std::sqrt(std::max(T[static_cast<size_t>(order[c]) * m + order[c]], 0.0));
    }
    // Similarity metrics on data with a common offset (round 6).  The records describe x - mean but the QUERY is not centred: its component along
    // the mean direction -- large whenever the data has a large mean -- would sit in its uncoded remainder tq and blow up the tq tx term of the
    // bound (Cauchy-Schwarz), and as a CODED coordinate it would lie far outside the 8-bit range of the centred rows' coordinates (its code clamps,
    // the coding error eq swallows the bound: measured, profiles/r06_notes.md section 8).  So when the mean is not negligible against the spread,
    // the unit mean direction becomes coordinate 0 of the basis (in place of the weakest coded direction; Gram-Schmidt keeps it and cleans the
    // others) and the kernel keeps the QUERY's coordinate 0 in float32 instead of coding it (SearchArgs::pca_mean_dir): u0 * s * c0(row) next to the
    // integer dot product of the other coordinates.  Any orthonormal set is valid for the bound.
    bool mean_dir = false;
    if (!l2 && dcu >= 2 && mean_norm * mean_norm > 0.01 * trace) {
        mean_dir = true;
        for (uint32_t i = 0; i < d; ++i) {
            for (uint32_t c = dcu - 1; c >= 1; --c) Q[static_cast<size_t>(i) * dcu + c] = Q[static_cast<size_t>(i) * dcu + c - 1];
            Q[static_cast<size_t>(i) * dcu] = static_cast<double>(mean_f[i]) / mean_norm;
        }
        for (uint32_t c = dcu - 1; c >= 1; --c) sigma[c] = sigma[c - 1];
        double var0 = 0.0; // spread of the centred sample along the mean direction: m^T C m
        for (uint32_t i = 0; i < d; ++i) {
            double ci = 0.0;
            for (uint32_t j = 0; j < d; ++j) ci += cov[static_cast<size_t>(i) * d + j] * static_cast<double>(mean_f[j]) / mean_norm;
            var0 += ci * static_cast<double>(mean_f[i]) / mean_norm;
        }
        sigma[0] = std::sqrt(std::max(var0, 0.0));
    }
    orthonormalize_columns(Q, d, dcu);
    // device copies: float32 rows for the query side (+ delta = how far they are from the f64 directions), f64 rows for the data side
    std::vector<float> host(static_cast<size_t>(kPcaRows) * pd + pd + 128, 0.0f); // P | mean | bias
    std::vector<double> q64(static_cast<size_t>(dcu) * d);
    double delta = 0.0;
    for (uint32_t c = 0; c < dcu; ++c) {
        double e2 = 0.0;
        for (uint32_t j = 0; j < d; ++j) {
            const double q = Q[static_cast<size_t>(j) * dcu + c];
            const float f = static_cast<float>(q);
            host[static_cast<size_t>(c) * pd + j] = f;
            q64[static_cast<size_t>(c) * d + j] = q;
            e2 += (static_cast<double>(f) - q) * (static_cast<double>(f) - q);
        }
        delta = std::max(delta, std::sqrt(e2));
    }
    for (uint32_t j = 0; j < d; ++j) host[static_cast<size_t>(kPcaRows) * pd + j] = mean_f[j];
    if (ix->d_pca) { (void)hipFree(ix->d_pca); ix->d_pca = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ix->d_pca), host.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(ix->d_pca, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    DeviceBuffer dq;
    HIP_TRY(dq.alloc(q64.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(dq.p, q64.data(), q64.size() * sizeof(double), hipMemcpyHostToDevice));
    // 4. per-coordinate range of y over ALL vectors (device), clipped to +-8 sigma; step, biases, tail unit
    DeviceBuffer st;
    HIP_TRY(st.alloc((2 * 128 + 2) * 4));
    std::vector<uint32_t> init(2 * 128 + 2, 0u);
    for (uint32_t i = 0; i < 128; ++i) init[i] = 0xFFFFFFFFu;
    HIP_TRY(hipMemcpy(st.p, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    PcaArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.data = ix->d_data; pa.row_bytes = ix->row_bytes; pa.dtype = ix->data_dtype; pa.n = n; pa.d = d; pa.dc = dcu; pa.rb = rb;
    pa.decomp = decomp ? 1 : 0; pa.sq_scale = ix->sq_scale; pa.sq_bias = ix->sq_bias; pa.normalize = cosine ? 1 : 0; pa.signed_codes = l2 ? 0 : 1;
    pa.q64 = static_cast<const double*>(dq.p);
    pa.mean = ix->d_pca + static_cast<size_t>(kPcaRows) * pd; pa.bias = pa.mean + pd;
    pa.stat_min = static_cast<uint32_t*>(st.p); pa.stat_max = pa.stat_min + 128; pa.stat_tail = pa.stat_max + 128; pa.stat_bad = pa.stat_tail + 1;
    HIP_TRY(launch_pca_stats(pa, nullptr));
    std::vector<uint32_t> stats(2 * 128 + 2);
    HIP_TRY(hipMemcpy(stats.data(), st.p, stats.size() * 4, hipMemcpyDeviceToHost));
    if (stats[2 * 128 + 1] != 0) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    double range = 0.0, bmax = 0.0;
    std::vector<float> bias(128, 0.0f);
    for (uint32_t i = 0; i < dcu; ++i) {
        double lo = pca_key_to_float(stats[i]), hi = pca_key_to_float(stats[128 + i]);
        if (sigma[i] > 0.0) { // outliers beyond 8 sigma clamp (their coding error is in their own ex)
            lo = std::max(lo, -8.0 * sigma[i]);
            hi = std::min(hi, 8.0 * sigma[i]);
        }
        if (!(hi > lo)) hi = lo;
        if (l2) {
            bias[i] = static_cast<float>(lo);
            range = std::max(range, hi - lo);
            bmax = std::max(bmax, std::fabs(static_cast<double>(bias[i])));
        } else {
            range = std::max(range, std::max(std::fabs(lo), std::fabs(hi)));
        }
    }
    float stepf = static_cast<float>(range / (l2 ? 255.0 : 127.0));
    if (!(stepf > 0.0f) || !(stepf < 3.0e38f)) stepf = 1.0f;
    float tail_max = pca_key_to_float(stats[2 * 128]);
    if (!(tail_max > 0.0f) || !(tail_max < 3.0e38f)) tail_max = 1.0f;
    const float tail_unit = tail_max / 60000.0f;
    HIP_TRY(hipMemcpy(ix->d_pca + static_cast<size_t>(kPcaRows) * pd + pd, bias.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    // 5. records, then node rows
    DeviceBuffer rec;
    HIP_TRY(rec.alloc(static_cast<size_t>(n) * rb + 128));
    pa.step = stepf; pa.tail_unit = tail_unit; pa.records = static_cast<uint32_t*>(rec.p);
    HIP_TRY(launch_pca_encode(pa, nullptr));
    const size_t fat_stride = (ix->gstride * (4 + rb) + 127) & ~size_t(127);
    void* fat = nullptr;
    if (hipMalloc(&fat, static_cast<size_t>(n) * fat_stride + 128) != hipSuccess) {
        (void)hipGetLastError(); // (another allocator of the process may hold what hipMemGetInfo reported free)
        return kRecordsDoNotFit;
    }
    // degree-aware block loads (svsx_search_kernel.h): ids below 2^26 leave bits [26, 31) of a buffer word for ceil(degree / 8)
    constexpr uint32_t kDegShift = 26;
    const bool deg_aware = n <= (1u << kDegShift) && ix->gstride <= 248; // class = ceil(degree / 8) <= 31: five bits
    DeviceBuffer cls;
    if (deg_aware && cls.alloc(n) != hipSuccess) {
        (void)hipFree(fat);
        return fail(SVSX_RUNTIME_ERROR, "hipMalloc(%u) for the degree classes failed", n);
    }
    hipError_t e = launch_fat_build(ix->d_graph, static_cast<uint32_t>(ix->gstride), n, rec.p, rb, fat, fat_stride, deg_aware ? static_cast<uint8_t*>(cls.p) : nullptr,
                                    kDegShift, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(fat);
        return fail(SVSX_RUNTIME_ERROR, "graph-inline record kernels failed: %s", hipGetErrorString(e));
    }
    ix->d_fat = fat;
    ix->fat_stride = fat_stride;
    ix->fat_rb = rb;
    ix->fat_degshift = deg_aware ? kDegShift : 0u;
    ix->pca_step = stepf;
    ix->pca_tail_unit = tail_unit;
    ix->pca_delta = static_cast<float>(delta * 1.01 + 1e-9);
    // rounding of fl(b + s c) (L2) / fl(s c) (signed codes) in the query's residual: <= 2^-23 (|b| + 255 s) per coordinate
    ix->pca_eq_abs = static_cast<float>(std::sqrt(static_cast<double>(dcu)) * 1.2e-7 * (bmax + 255.0 * stepf) * 1.01);
    // inner product: ||x|| <= ||x - mean|| + ||mean|| enters the slack for the reference's own float32 rounding; scalar-quantized rows add 2 |bias| sqrt(d)
    ix->pca_ref_extra = static_cast<float>(((l2 ? 0.0 : mean_norm) + (decomp ? 2.0 * std::fabs(static_cast<double>(ix->sq_bias)) * std::sqrt(static_cast<double>(d)) : 0.0)) * 1.01);
    ix->pca_mean_norm = l2 ? 0.0f : static_cast<float>(mean_norm * 1.000001 + 1e-30);
    ix->pca_mean_dir = mean_dir ? 1u : 0u;
    return SVSX_SUCCESS;
}

} // namespace

namespace {

// Mode 3's second opinion: the size rule is a model (share of the second moment -> expected rejects), and it can be wrong -- under the similarity
// metrics it does not see how much of a QUERY lies outside the coded directions.  So the records it decided to build are timed against the plain
// kernel on a sample of the index's own rows as queries (2 048 rows, window 64, the row's own element type; float32 for scalar-quantized rows),
// and kept only if they are faster by 5 %.  Small indexes are launch-latency bound and are not timed.  Returns 1 keep, 0 drop; errors keep.
int records_pay_on_a_sample(svsx_index* ix) {
    if (ix->n < 100000 || !ix->d_fat) return 1;
    const size_t nq = 2048, k = 10, d = ix->d;
    const int q_dtype = ix->sq_enabled ? DT_F32 : ix->data_dtype;
    const size_t qes = dtype_size(q_dtype), es = dtype_size(ix->data_dtype), step = ix->n / nq;
    DeviceBuffer dq, di, dd;
    if (dq.alloc(nq * d * qes) != hipSuccess || di.alloc(nq * k * 8) != hipSuccess || dd.alloc(nq * k * 4) != hipSuccess) { (void)hipGetLastError(); return 1; }
    if (!ix->sq_enabled) {
        if (hipMemcpy2D(dq.p, d * es, ix->d_data, step * ix->row_bytes, d * es, nq, hipMemcpyDeviceToDevice) != hipSuccess) return 1;
    } else { // the queries of a scalar-quantized index are float vectors: the rows' decompressed values (scalar.h:44-46)
        std::vector<unsigned char> raw(nq * d);
        std::vector<float> qf(nq * d);
        if (hipMemcpy2D(raw.data(), d, ix->d_data, step * ix->row_bytes, d, nq, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (size_t i = 0; i < nq * d; ++i) {
            const float c = ix->data_dtype == DT_I8 ? static_cast<float>(static_cast<int8_t>(raw[i])) : static_cast<float>(raw[i]);
            qf[i] = ix->sq_scale * c + ix->sq_bias;
        }
        if (hipMemcpy(dq.p, qf.data(), qf.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    svsx_search_params sp;
    memset(&sp, 0, sizeof(sp));
    sp.search_window_size = 64;
    sp.search_buffer_capacity = 64;
    sp.visited_set = 1;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 1;
    auto best_of = [&](float& ms) -> bool {
        ms = 1e30f;
        for (int rep = 0; rep < 5; ++rep) { // two warm-ups, three timed
            if (hipEventRecord(e0, nullptr) != hipSuccess) return false;
            if (svsx_search_device(ix, dq.p, q_dtype, nq, k, &sp, static_cast<uint64_t*>(di.p), nullptr, static_cast<float*>(dd.p), nullptr, nullptr) != SVSX_SUCCESS) return false;
            if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) return false;
            float t = 0.0f;
            if (hipEventElapsedTime(&t, e0, e1) != hipSuccess) return false;
            if (rep >= 2 && t < ms) ms = t;
        }
        return true;
    };
    float t_rec = 0.0f, t_plain = 0.0f;
    bool ok = best_of(t_rec);
    void* fat = ix->d_fat;
    ix->d_fat = nullptr; // the same index without its records = the plain kernel
    ok = best_of(t_plain) && ok;
    ix->d_fat = fat;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!ok) return 1;
    ix->calib_ms_records = t_rec;
    ix->calib_ms_plain = t_plain;
    return t_rec < 0.95f * t_plain ? 1 : 0;
}

} // namespace

extern "C" int svsx_index_build_bound_codes(svsx_index* ix, int mode) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    uint32_t forced_rb = 0;
    if (mode == 28 || mode == 32 || mode == 64 || mode == 128) { // graph-inline records of exactly this size
        forced_rb = static_cast<uint32_t>(mode);
        mode = 2;
    }
    bool only_if_worth = false;
    if (mode == 3) { // graph-inline records where the size rule expects them to pay, nothing where it does not
        only_if_worth = true;
        mode = 2;
    }
    if (mode < 0 || mode > 2) return fail(SVSX_INVALID_ARGUMENT, "bound-code mode must be 0 (off), 1 (128-B rows), 2 (graph-inline records), 3 (records where they are expected to pay) or 28 / 32 / 64 / 128 (inline records of that size)");
    bool fell_from_records = false;
    HIP_TRY(hipSetDevice(ix->device));
    std::lock_guard<std::mutex> lock(ix->shadow_mutex);
    HIP_TRY(hipDeviceSynchronize());
    // one scheme at a time: drop what the new mode does not use
    if (mode != 1 && ix->d_codes) { (void)hipFree(ix->d_codes); ix->d_codes = nullptr; }
    if (mode != 2 && ix->d_fat) { (void)hipFree(ix->d_fat); ix->d_fat = nullptr; }
    if (mode == 0) return SVSX_SUCCESS;
    if (mode == 2) {
        if (ix->d_fat && (forced_rb == 0 || forced_rb == ix->fat_rb)) return SVSX_SUCCESS;
        if (ix->d_fat) { (void)hipFree(ix->d_fat); ix->d_fat = nullptr; }
        // float32 / float16 rows, scalar-quantized rows (any metric; L2 in code space) and raw int8 / uint8 rows (round 6).  Which query types
        // then take the record kernels is the dispatch's business (svsx_search_kernel.h launch_dim); the others run the plain kernels.
        if (ix->d < 32 || ix->d > 4096 || !ix->d_graph) return fail(SVSX_NOT_IMPLEMENTED, "graph-inline records cover graph indexes of dimension 32..4096");
        uint32_t asked = 0, built = 0;
        const int rc = build_fat_rows(ix, forced_rb, &asked, &built, only_if_worth);
        if (rc == kRecordsDoNotPay) { // not a failure and not a fallback: the answer of mode 3 on this data is "no scheme" (svsx_index_memory()[8] == 0)
            (void)fail(SVSX_SUCCESS, "mode 3: the spectrum of this data promises no saving from inline records: none built");
            return SVSX_SUCCESS;
        }
        if (rc == SVSX_SUCCESS && only_if_worth && records_pay_on_a_sample(ix) == 0) { // the model said yes, the clock says no
            (void)hipFree(ix->d_fat);
            ix->d_fat = nullptr;
            (void)fail(SVSX_SUCCESS, "mode 3: inline records of %u bytes ran at %.3f ms against %.3f ms for the plain kernel on a sample of the index's own rows: not kept",
                       built, ix->calib_ms_records, ix->calib_ms_plain);
            return SVSX_SUCCESS;
        }
        if (rc == SVSX_SUCCESS && built == asked) return SVSX_SUCCESS;
        if (rc == SVSX_SUCCESS) // smaller records than the spectrum rule / the caller wanted: still inline records, still result-neutral
            return fail(SVSX_FELL_BACK, "graph-inline records of %u bytes do not fit the free HBM: built %u-byte records instead", asked, built);
        if (rc != kRecordsDoNotFit) return rc;
        // not even 32-byte records fit: 128-byte code rows where the row type has them, else no bound scheme at all (searches run the plain kernels)
        const bool rows_ok = ix->metric == M_L2 && (ix->data_dtype == DT_F32 || ix->data_dtype == DT_F16) && !ix->sq_enabled && (ix->d == 96 || ix->d == 128);
        if (!rows_ok) return fail(SVSX_FELL_BACK, "graph-inline records do not fit the free HBM and this row type has no 128-byte code rows: no bound scheme in effect");
        fell_from_records = true;
        mode = 1;
    }
    if (ix->metric != M_L2 || (ix->data_dtype != DT_F32 && ix->data_dtype != DT_F16) || ix->sq_enabled)
        return fail(SVSX_NOT_IMPLEMENTED, "128-byte bound-code rows cover float32 / float16 L2 indexes");
    if (ix->d_codes) return SVSX_SUCCESS;
    if (ix->d != 96 && ix->d != 128) return fail(SVSX_NOT_IMPLEMENTED, "128-byte bound-code rows cover dimension 96 or 128");
    const uint32_t dc = static_cast<uint32_t>(ix->d < 120 ? ix->d : 120);
    DeviceBuffer mm;
    HIP_TRY(mm.alloc(12));
    HIP_TRY(launch_codes_minmax(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), dc, static_cast<uint32_t*>(mm.p), nullptr));
    uint32_t h[3];
    HIP_TRY(hipMemcpy(h, mm.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h[2] != 0) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    const float lo = codes_key_to_float(h[0]), hi = codes_key_to_float(h[1]);
    float scale = (hi - lo) / 255.0f;
    if (!(scale > 0.0f) || !(scale < 3.0e38f)) scale = 1.0f;
    void* codes = nullptr;
    if (hipMalloc(&codes, ix->n * 128 + 128) != hipSuccess) {
        (void)hipGetLastError();
        if (fell_from_records) return fail(SVSX_FELL_BACK, "neither graph-inline records nor 128-byte code rows fit the free HBM: no bound scheme in effect");
        return fail(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for the 128-byte code rows failed", ix->n * 128 + 128);
    }
    hipError_t e = launch_codes_build(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), dc, scale, lo, codes, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(codes);
        return fail(SVSX_RUNTIME_ERROR, "bound code kernel failed: %s", hipGetErrorString(e));
    }
    ix->code_scale = scale;
    ix->code_bias = lo;
    ix->code_eq_abs = static_cast<float>(std::sqrt(static_cast<double>(dc)) * 2.4e-7 * (std::fabs(static_cast<double>(lo)) + 255.0 * scale) * 1.01);
    ix->d_codes = codes;
    if (fell_from_records) return fail(SVSX_FELL_BACK, "graph-inline records do not fit the free HBM: built 128-byte code rows (mode 1) instead");
    return SVSX_SUCCESS;
}
