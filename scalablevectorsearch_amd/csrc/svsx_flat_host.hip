// svsx_flat_host.hip -- svsx_flat_search*: the host side of the exhaustive search (FlatIndex::search, index/flat/flat.h:322-460): float16 shadow,
// threshold levels / shortlist kernels (svsx_flat.hip), exact re-rank, per-query proof, exhaustive float32 pass for unproven queries.
#include "svsx_host.h"

extern "C" {

// rank at which the threshold levels take their next threshold: k + 4, at least 12, at most the shortlist length (round 4: k + 8, at least 16;
// with the cheaper appends of round 5 the ~16 kt appended rows per query weigh more: k + 4 is worth 4-8 % of a search, k + 2 one more per cent,
// and no query of the bench sets needs the exhaustive pass at either; profiles/r05_notes.md section 5)
#ifndef SVSX_FLAT_KT
#define SVSX_FLAT_KT(k, kk) ((k) + 4 > 12 ? ((k) + 4 < (kk) ? (k) + 4 : (kk)) : (12 < (kk) ? size_t(12) : (kk)))
#endif
int svsx_flat_search_device(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, void* hip_stream
) {
    return svsx_flat_search_device_counted(ix, d_queries, q_dtype, nq, k, d_ids64, d_dists, nullptr, hip_stream);
}

// host pointers in, host results out (svs::Flat::search through the C ABI): staging in stream order, one synchronisation at the end
int svsx_flat_search(const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, uint64_t* ids64, float* dists) {
    if (int rc = check_search_args(ix, queries, q_dtype, nq, k, false)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    if (!ids64) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    CallStream cs;
    HIP_TRY(cs.create());
    StreamBuffer dq, di, dd;
    const size_t q_bytes = nq * ix->d * dtype_size(q_dtype);
    HIP_TRY(dq.alloc(q_bytes, cs.s));
    HIP_TRY(di.alloc(nq * k * 8, cs.s));
    HIP_TRY(dd.alloc(nq * k * 4, cs.s));
    HIP_TRY(hipMemcpyAsync(dq.p, queries, q_bytes, hipMemcpyHostToDevice, cs.s));
    if (int rc = svsx_flat_search_device_counted(ix, dq.p, q_dtype, nq, k, static_cast<uint64_t*>(di.p), static_cast<float*>(dd.p), nullptr, cs.s)) {
        (void)hipStreamSynchronize(cs.s);
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(ids64, di.p, nq * k * 8, hipMemcpyDeviceToHost, cs.s));
    if (dists) HIP_TRY(hipMemcpyAsync(dists, dd.p, nq * k * 4, hipMemcpyDeviceToHost, cs.s));
    HIP_TRY(hipStreamSynchronize(cs.s));
    return SVSX_SUCCESS;
}

int svsx_flat_search_device_counted(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, uint32_t* d_recomputed,
    void* hip_stream
) {
    if (int rc = check_search_args(ix, d_queries, q_dtype, nq, k, false)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    if (!d_ids64) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    const int qmode = select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled);
    if (qmode < 0) return fail(SVSX_INVALID_ARGUMENT, "unsupported (query dtype %d, data dtype %d) pair", q_dtype, ix->data_dtype);
    if (qmode == QM_SCALAR)
        return fail(SVSX_NOT_IMPLEMENTED, "flat search serves the (query, data) type pairs with a SIMD kernel in the reference; (%d, %d) has the scalar loop only: graph search and svsx_distance_batch cover it", q_dtype, ix->data_dtype);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // FlatIndex::search (index/flat/flat.h:322-460) bounds neither k nor the dimension.  The float16 pipeline below keeps k + 32 <= 256 shortlist
    // entries per query (and, beyond d = 128, the 64 queries of a wave) in LDS; what does not fit it -- k > 224, or a dimension whose 64
    // float16 queries plus k + 8 entries exceed 160 KB (d >= ~1 200) -- is answered by the exhaustive float32 pass alone: every query
    // against every row with the reference-order routines and a k-entry list (stage 5 of the pipeline, exact by construction; one wave per
    // query, so it is the slow path: 10 000 x 1 M x 96 takes ~1 s instead of 4 ms).
    auto exhaustive_only = [&]() -> int {
        const size_t lds = ((ix->d + 15) & ~size_t(15)) * 4 + k * 8;
        if (lds > 160 * 1024) return fail(SVSX_INVALID_ARGUMENT, "flat search: k = %zu at d = %zu needs %zu bytes of LDS for one query's list (160 KB available)", k, ix->d, lds);
        StreamBuffer redo;
        HIP_TRY(redo.alloc(nq * 4, stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(redo.p), 1, nq, stream));
        FlatExactArgs xa;
        memset(&xa, 0, sizeof(xa));
        xa.r.data = ix->d_data; xa.r.row_bytes = ix->row_bytes; xa.r.d = static_cast<uint32_t>(ix->d); xa.r.sq_scale = ix->sq_scale; xa.r.sq_bias = ix->sq_bias;
        xa.r.id_offset = ix->id_offset; xa.r.queries = d_queries; xa.r.q_row_bytes = ix->d * dtype_size(q_dtype); xa.r.q_dtype = q_dtype;
        xa.r.nq = static_cast<uint32_t>(nq); xa.r.kk = static_cast<uint32_t>(k); xa.r.k = static_cast<uint32_t>(k);
        xa.r.ids64 = d_ids64; xa.r.dists = d_dists;
        xa.redo = static_cast<const uint32_t*>(redo.p); xa.n = static_cast<uint32_t>(ix->n);
        hipError_t e = launch_flat_exact(ix->data_dtype, ix->metric, qmode, xa, stream);
        if (e == hipSuccess && d_recomputed != nullptr) e = launch_add_word(d_recomputed, static_cast<uint32_t>(nq), stream);
        if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "flat search launch failed: %s", hipGetErrorString(e));
        return SVSX_SUCCESS;
    };
    if (k > 224) return exhaustive_only();
    // rows of the float16 shadow: padded to 16 elements, between 129 and 768 dimensions to 128 (the threshold kernel for those takes the
    // dimension in chunks of 128)
    const uint32_t dpad = static_cast<uint32_t>(ix->d > 128 && ix->d <= 768 ? (ix->d + 127) & ~size_t(127) : (ix->d + 15) & ~size_t(15));
    // shortlist length: max(4k, k + 32) up to 128 entries; beyond k = 96 (the reference's own ground-truth files hold 100 neighbours per
    // query, data/test_dataset/groundtruth_*.ivecs) k + 32, the general insert path of the shortlist kernel, up to 256 entries
    size_t kk = 4 * k > k + 32 ? 4 * k : k + 32;
    if (kk > 128) kk = k + 32 > 128 ? k + 32 : 128;
    if (kk > ix->n) kk = ix->n < k ? k : ix->n;
    {   // the shortlist kernel keeps 64 shortlists (and, beyond d = 128, the 64 queries) of a wave in LDS; where that does not fit the
        // shortlist shrinks towards k + 8 -- the answer stays exact, a shorter shortlist only makes the proof of stage 4 fail more often
        const size_t fixed = (dpad > 128 ? 64 * (static_cast<size_t>(dpad) * 2 + 16) : 0) + 256;
        while (fixed + 64 * kk * 8 > 160 * 1024 && kk > k + 8) kk -= 4;
        if (kk < k) kk = k;
        const size_t need = fixed + 64 * kk * 8;
        // (decided BEFORE the float16 shadow is built: the exhaustive pass never reads it, and at 1 M x 1 536 it would be 3 GB of HBM kept for the index's life -- ADVICE r5)
        if (need > 160 * 1024) return exhaustive_only(); // the 64 float16 queries of a wave + its shortlists do not fit (d >= ~1 200, or a long list at a large d)
    }
    {   // float16 shadow of the database + squared norms, built once
        std::lock_guard<std::mutex> lock(ix->shadow_mutex);
        if (!ix->d_half) {
            void* h = nullptr;
            float* nrm = nullptr;
            HIP_TRY(hipMalloc(&h, (ix->n + kShadowPadRows) * dpad * 2 + 64)); // padding: the threshold kernels load whole 32-row tiles / 128-row steps
            if (hipMalloc(reinterpret_cast<void**>(&nrm), ix->n * 4) != hipSuccess) { (void)hipFree(h); return fail(SVSX_RUNTIME_ERROR, "hipMalloc failed"); }
            // the padding tile is read (masked) by the last tile of the threshold kernel: zeros, not whatever the allocation held
            hipError_t e = hipMemsetAsync(static_cast<unsigned char*>(h) + ix->n * static_cast<size_t>(dpad) * 2, 0, kShadowPadRows * static_cast<size_t>(dpad) * 2 + 64, stream);
            if (e == hipSuccess)
                e = launch_to_half(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), static_cast<uint32_t>(ix->d), dpad,
                                   ix->sq_enabled, ix->sq_scale, ix->sq_bias, h, nrm, stream);
            uint32_t* mm = nullptr; // {max, min} bits of the squared norms
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&mm), 8);
            uint32_t range[2] = {0u, 0xFFFFFFFFu};
            if (e == hipSuccess) e = hipMemcpyAsync(mm, range, 8, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = launch_norm_range(nrm, static_cast<uint32_t>(ix->n), mm, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(range, mm, 8, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (mm) (void)hipFree(mm);
            if (e != hipSuccess) { (void)hipFree(h); (void)hipFree(nrm); return fail(SVSX_RUNTIME_ERROR, "float16 shadow conversion failed: %s", hipGetErrorString(e)); }
            memcpy(&ix->shadow_max2, &range[0], 4);
            memcpy(&ix->shadow_min2, &range[1], 4);
            ix->d_half = h;
            ix->d_norm2 = nrm;
        }
    }
    const size_t nqb = (nq + 63) / 64;
    // one round of resident waves: LDS (64 shortlists of kk entries) bounds the waves per CU; a second round would
    // double the time, more slices than needed only add shortlist inserts (measured: 14.1 ms at 80 % of the slots)
    const size_t flat_lds = 64 * kk * 8 + 256 + (dpad > 128 ? 64 * (static_cast<size_t>(dpad) * 2 + 16) : 0); // shortlists + (beyond d = 128) the wave's 64 queries
    size_t waves_per_cu = (160 * 1024) / flat_lds;
    if (waves_per_cu > 8) waves_per_cu = 8;
    if (waves_per_cu < 1) waves_per_cu = 1;
    size_t target_waves = 256 * waves_per_cu * 4 / 5;
    size_t slices = (target_waves + nqb - 1) / nqb; // thresholds are shared across the slices of a query block
    if (slices > 64) slices = 64;
    const size_t max_slices = (ix->n + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const size_t q_stride = ix->d * dtype_size(q_dtype);
    StreamBuffer bqh, bsk, bsi, bmk, bmi, bredo, bdist;
    HIP_TRY(bredo.alloc(nq * 4, stream));
    if (!d_dists) { // the proof needs the k-th exact distance
        HIP_TRY(bdist.alloc(nq * k * 4, stream));
        d_dists = static_cast<float*>(bdist.p);
    }
    HIP_TRY(bqh.alloc(nq * dpad * 2 + 64, stream));
    HIP_TRY(bsk.alloc(slices * nq * kk * 4, stream));
    HIP_TRY(bsi.alloc(slices * nq * kk * 8, stream));
    HIP_TRY(bmk.alloc(nq * kk * 4, stream));
    HIP_TRY(bmi.alloc(nq * kk * 8, stream));
    void *qh = bqh.p, *sk = bsk.p, *si = bsi.p, *mk = bmk.p, *mi = bmi.p;
    hipError_t e = launch_to_half(q_dtype, d_queries, q_stride, static_cast<uint32_t>(nq), static_cast<uint32_t>(ix->d), dpad, 0, 1.0f, 0.0f, qh, nullptr, stream);
    FlatArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.xh = ix->d_half; fa.xnorm2 = ix->d_norm2; fa.n = static_cast<uint32_t>(ix->n); fa.dpad = dpad; fa.qh = qh;
    fa.nq = static_cast<uint32_t>(nq); fa.kk = static_cast<uint32_t>(kk); fa.n_slices = static_cast<uint32_t>(slices);
    fa.out_keys = static_cast<float*>(sk); fa.out_ids = static_cast<uint64_t*>(si);
    fa.tile_step = 1;
    StreamBuffer bgt;
    HIP_TRY(bgt.alloc(nq * 4, stream));
    fa.gtau = static_cast<uint32_t*>(bgt.p);
    MergeArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.dists = static_cast<const float*>(sk); ma.ids = static_cast<const uint64_t*>(si);
    ma.out_dists = static_cast<float*>(mk); ma.out_ids = static_cast<uint64_t*>(mi);
    ma.n_shards = static_cast<uint32_t>(slices); ma.nq = static_cast<uint32_t>(nq); ma.k = static_cast<uint32_t>(kk);
    RerankArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.data = ix->d_data; ra.row_bytes = ix->row_bytes; ra.d = static_cast<uint32_t>(ix->d); ra.sq_scale = ix->sq_scale; ra.sq_bias = ix->sq_bias;
    ra.id_offset = ix->id_offset; ra.queries = d_queries; ra.q_row_bytes = q_stride; ra.q_dtype = q_dtype;
    ra.nq = static_cast<uint32_t>(nq); ra.kk = static_cast<uint32_t>(kk); ra.k = static_cast<uint32_t>(k);
    ra.cand_ids = static_cast<const uint64_t*>(mi); ra.ids64 = d_ids64; ra.dists = d_dists;
    // Large sets with d <= 128 take the threshold pipeline (svsx_flat.hip stage 1b): nested strided samples of the 32-row
    // tiles, every 16^j-th tile for j = L .. 1, each ~16x larger than the one before.  The smallest (>= 256 rows) is
    // taken whole (threshold +inf); at every level the threshold kernel appends the rows at or below the current
    // threshold (about 16 kk per query) and the select kernel takes the kk-th best key of that list as the next, tighter
    // threshold -- which at least kk rows of the next level satisfy, because the samples are nested.  The last level is
    // the full set; its lists (>= kk rows, ~16 kk expected) are re-ranked exactly.  Queries whose lists overflow
    // (unrepresentative samples) are recomputed by the full shortlist kernel, gated per 64-query block.
    FlatProofArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.metric = ix->metric; pa.nq = ra.nq; pa.k = ra.k; pa.kk = ra.kk; pa.d = ra.d; pa.dpad = dpad;
    pa.queries = d_queries; pa.q_row_bytes = q_stride; pa.q_dtype = q_dtype;
    pa.sq_l2 = qmode == QM_SQ_L2 ? 1 : 0; pa.data_dtype = ix->data_dtype; pa.sq_scale = ix->sq_scale; pa.sq_bias = ix->sq_bias;
    pa.out_ids = d_ids64; pa.out_dists = d_dists;
    pa.short_keys = static_cast<const float*>(mk); pa.short_ids = static_cast<const uint64_t*>(mi);
    pa.xmax2 = ix->shadow_max2; pa.xmin2 = ix->shadow_min2;
    pa.redo = static_cast<uint32_t*>(bredo.p); pa.redo_count = d_recomputed;
    bool thresh_path = flat_thresh_supports(dpad) && ix->n >= 65536 && nq <= 65536;
#ifdef SVSX_DIAG
    if (getenv("SVSX_FLAT_SHORTLIST_ONLY")) thresh_path = false; // A/B builds only
#endif
    StreamBuffer bcand, bkeys, bcnt, bgate, btau_a, btau_b;
    if (thresh_path) {
        // sampling unit: a 32-row tile (d > 128) or a 64- / 128-row step of the four-wave kernel (d <= 128)
        const size_t unit = flat_thresh_unit_rows(dpad);
        const size_t units = (ix->n + unit - 1) / unit;
        uint32_t step = 1;
        while ((units / (static_cast<size_t>(step) * 16)) * unit >= 128) step *= 16; // coarsest level: 128 ... 2 047 rows
        size_t s2 = flat_thresh_slices(ix->metric, dpad, nq); // one generation of resident workgroups
        if (s2 > ix->n / 4096) s2 = ix->n / 4096;
        if (s2 < 1) s2 = 1;
        const size_t lps = flat_thresh_lists_per_slice(dpad); // candidate lists per (query, slice)
        if (s2 * lps > 1024) s2 = 1024 / lps;
        const size_t lists = s2 * lps;
        // rows per list: ~16 kk / lists expected at every level, at most the coarsest level's rows; 4x head room
        size_t cap = 512;
        // The thresholds are taken at rank kt < kk: with the proof of stage 4 behind it, the threshold pipeline needs no float16 safety margin
        // in its candidate lists -- ~16 kt rows per query (224 for k = 10) are appended and re-ranked instead of ~16 kk (640), and whether
        // that was enough is decided per query by the proof, not by the margin.  (The shortlist kernel keeps kk: its proof needs the gap
        // between the k-th and the kk-th key.)
        const size_t kt = SVSX_FLAT_KT(k, kk);
        while (cap < 4 * (16 * kt / lists + 1)) cap *= 2;
        const size_t coarse_rows = (units / step + 1) * unit; // all of them are appended at the first level
        while (cap < 2 * coarse_rows / s2 && cap < 8192) cap *= 2;
        HIP_TRY(bcand.alloc(nq * lists * cap * 4, stream));
        HIP_TRY(bkeys.alloc(nq * lists * cap * 4, stream));
        HIP_TRY(bcnt.alloc(nq * lists * 4, stream));
        HIP_TRY(bgate.alloc(nqb * 4, stream));
        HIP_TRY(btau_a.alloc(nq * 4, stream));
        HIP_TRY(btau_b.alloc(nq * 4, stream));
        float* tau_cur = static_cast<float*>(btau_a.p);
        float* tau_nxt = static_cast<float*>(btau_b.p);
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tau_cur), 0x7F800000, nq, stream); // +inf
        FlatThreshArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.xh = ix->d_half; ta.xnorm2 = ix->d_norm2; ta.n = static_cast<uint32_t>(ix->n); ta.dpad = dpad; ta.qh = qh;
        ta.nq = static_cast<uint32_t>(nq);
        ta.n_slices = static_cast<uint32_t>(s2);
        ta.tau_stride = 1;
        ta.cand = static_cast<uint32_t*>(bcand.p);
        ta.cnt = static_cast<uint32_t*>(bcnt.p); ta.cap = static_cast<uint32_t>(cap);
        for (; step > 1; step /= 16) {
            ta.tile_step = step;
            ta.tau = tau_cur;
            ta.cand_keys = static_cast<float*>(bkeys.p);
            if (e == hipSuccess) e = launch_flat_thresh(ix->metric, ta, stream);
            if (e == hipSuccess) e = launch_flat_select(ta.cand_keys, ta.cnt, ta.nq, static_cast<uint32_t>(lists), ta.cap, static_cast<uint32_t>(kt), tau_cur, 1, tau_nxt, stream);
            float* t = tau_cur; tau_cur = tau_nxt; tau_nxt = t;
        }
        ta.tile_step = 1;
        ta.tau = tau_cur;
        ta.cand_keys = nullptr;
        if (e == hipSuccess) e = launch_flat_thresh(ix->metric, ta, stream);
        if (e == hipSuccess) e = launch_flat_gate_kk(ta.cnt, ta.nq, static_cast<uint32_t>(lists), ta.cap, static_cast<uint32_t>(kk), static_cast<uint32_t*>(bgate.p), stream);
        // gated fallback: full shortlist pass and merge only for the query blocks that overflowed
        fa.tile_step = 1;
        fa.block_gate = static_cast<const uint32_t*>(bgate.p);
        ma.gate64 = fa.block_gate;
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bgt.p), 0xFF800000, nq, stream);
        if (e == hipSuccess) e = launch_flat_mfma(ix->metric, fa, stream);
        if (e == hipSuccess) e = launch_merge(M_L2, ma, stream);
        ra.cand32 = ta.cand; ra.cnt = ta.cnt; ra.cap = ta.cap; ra.n_lists = static_cast<uint32_t>(lists);
        if (e == hipSuccess) e = launch_rerank2(ix->data_dtype, ix->metric, qmode, ra, stream);
        pa.tau = tau_cur; pa.cnt = ta.cnt; pa.cap = ta.cap; pa.n_lists = static_cast<uint32_t>(lists);
        if (e == hipSuccess) e = launch_flat_certify(pa, stream);
    } else {
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bgt.p), 0xFF800000, nq, stream); // +inf, order-encoded
        if (e == hipSuccess) e = launch_flat_mfma(ix->metric, fa, stream);
        if (e == hipSuccess) e = launch_merge(M_L2, ma, stream); // keys: smaller is better for every metric
        if (e == hipSuccess) e = launch_rerank(ix->data_dtype, ix->metric, qmode, ra, stream);
        if (e == hipSuccess) e = launch_flat_certify(pa, stream);
    }
    // the queries whose shortlist is not PROVEN to hold the exact answer are recomputed exhaustively in float32 (svsx_flat.hip stages 4, 5)
    FlatExactArgs xa;
    xa.r = ra; xa.redo = pa.redo; xa.n = static_cast<uint32_t>(ix->n);
    if (e == hipSuccess) e = launch_flat_exact(ix->data_dtype, ix->metric, qmode, xa, stream);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "flat search launch failed: %s", hipGetErrorString(e));
    return SVSX_SUCCESS;
}

} // extern "C"
