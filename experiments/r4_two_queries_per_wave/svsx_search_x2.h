// svsx_search_x2.h -- batched Vamana greedy search, two queries per wavefront, INTERLEAVED (gfx950 / CDNA4; round 4).
//
// The one-query kernel (svsx_search_kernel.h) keeps 6 waves per SIMD = 6 144 queries resident, so a 10 000-query launch runs 1.63
// generations; its second generation works at 3.8 waves per SIMD, latency-bound, and costs ~45 % of the launch for 39 % of the queries
// (profiles/r04_notes.md).  Here a wave owns TWO queries and does one expansion of each per loop iteration, each with all 64 lanes and the
// one-query kernel's own wave-uniform code (scalar-unit control flow, the same device routines: wave_next, wave_merge_insert,
// wave_insert, the reference-order row arithmetic), with the phases of the two expansions interleaved so that each query's memory
// round trips run under the other query's instructions:
//
//     X: bound, filter, compaction, rows, distances, merge insert, next(), REQUEST X's next block | Y: the same | X: ... | Y: ...
//
// -- a query's block is requested as soon as its expansion ends and arrives while the wave works on the other query.
//
// Per-query state is what it was (sorted buffer, query, visited filter in LDS; size / best_unvisited / worst in scalar registers), times
// two; what a wave adds is the second query's loads in flight: 96 vector registers = 5 waves per SIMD = 20 per CU = 40 queries per CU =
// 10 240 resident queries: the whole 10 000-query batch is one generation.  LDS per query <= 3 840 bytes (d = 96: 1 024 exact filter tags).
// Instructions per query are the one-query kernel's -- unlike the SIMT form (svsx_search_2q.h: one query per 32-lane half, measured
// 0.68x: per-half control flow costs more vector and scalar instructions than the shared distance pass saves).
//
// Scope: 32-byte inline records, float32 / float16 rows of d = 96 / 128 under L2 / inner product, buffers of <= 128 entries, one adjacency
// chunk (max degree <= 64), <= 16 entry points, no history / cancel / iterator side channel.  NaN distances, duplicates and ties take the
// one-query kernel's sequential insert path (the same code), so there is nothing to hand back.
#pragma once

// (included by svsx_search_kernel.h after search_one; nothing else includes it)

namespace svsx {

// LDS of ONE query: [buffer 129 entries, padded][query DS f32][compaction scratch 64 ids][record codes + bound constants][filter]
__host__ __device__ constexpr uint32_t x2_fixed_bytes(uint32_t ds) { return 1040u + ds * 4u + 256u + 64u; }
__host__ __device__ constexpr uint32_t x2_query_bytes(uint32_t ds, uint32_t filter_entries) { return x2_fixed_bytes(ds) + filter_entries * 2u; }

#ifndef SVSX_X2_WAVES
#define SVSX_X2_WAVES 5
#endif

template <int DT, int METRIC, int DS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SVSX_X2_WAVES))) void search_kernel_x2(const SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(METRIC == M_L2 || METRIC == M_IP, "interleaved kernel: L2 and inner product");
    static_assert(DT == DT_F32 || DT == DT_F16, "interleaved kernel: float rows");
    static_assert(DS > 0 && (DS & 15) == 0, "interleaved kernel: static dimension, whole 16-blocks");
    constexpr bool GREATER = METRIC != M_L2;
    constexpr int FDC = 32 - (METRIC == M_L2 ? 4 : 6); // coded coordinates of a 32-byte record
    constexpr int G = 4;                               // lanes per candidate row (slot-cyclic f32, block-cyclic f16)
    constexpr uint32_t RPP = 64 / G;                   // rows per pass
    constexpr int d = DS, dpad = DS;

    const uint32_t lane = threadIdx.x;
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    const uint32_t fmask = a.filter_entries - 1u;
    const uint32_t fshift = static_cast<uint32_t>(__builtin_ctz(a.filter_entries | 0x80000000u));
    const uint32_t filt_bytes = a.filter_entries * 2u;
    const uint32_t qbytes = x2_query_bytes(DS, a.filter_entries);
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    const float fat_s2m = a.pca_step * a.pca_step * (METRIC == M_L2 ? 0.99999f : 1.0f);

    // one query's state: LDS regions + wave-uniform scalars (the one-query kernel's, per query)
    struct Ctx {
        WaveBuffer buf;
        float* qf;
        uint32_t* compact;   // 64 survivor ids; the first 128 bytes double as the query's code bytes during the set-up
        uint32_t* qconst;    // words 0..6 record codes of the query, 7 zero, 8 qq, 9 eq, 10 tq_hi, 11 tq_lo, 12 ||uhat||, 13 ||v||
        uint16_t* filt;
        uint32_t qi;
        bool alive, full, nan_seen;
        float worst;
        uint32_t n_exp, n_slots, n_eval, n_ins, n_code_rej;
    };
    auto make_ctx = [&](uint32_t which) {
        Ctx c;
        unsigned char* base = smem + which * qbytes;
        c.buf.e = reinterpret_cast<uint2*>(base);
        c.buf.hbm = false;
        c.buf.size = 0;
        c.buf.best_unvisited = 0;
        c.buf.window = a.window;
        c.buf.capacity = a.capacity;
        c.qf = reinterpret_cast<float*>(base + 1040);
        c.compact = reinterpret_cast<uint32_t*>(base + 1040 + DS * 4);
        c.qconst = c.compact + 64;
        c.filt = reinterpret_cast<uint16_t*>(c.qconst + 16);
        c.qi = blockIdx.x * 2u + which;
        c.alive = c.qi < a.nq;  // an odd batch leaves the last wave with one query
        c.full = false;
        c.nan_seen = false;
        c.worst = 0.0f;
        c.n_exp = c.n_slots = c.n_eval = c.n_ins = c.n_code_rej = 0u;
        return c;
    };
    Ctx X = make_ctx(0u), Y = make_ctx(1u);

    // distance of the row each lane group is pointed at, in one piece (entry points)
    auto group_distance = [&](const Ctx& c, uint32_t cid, bool active) -> float {
        RowData<DT, DS> r;
        row_issue<DT, DS>(data + static_cast<size_t>(active ? cid : 0u) * a.row_bytes, sub, r);
        return row_finish<DT, METRIC, DS>(r, c.qf, sub);
    };

    // ---- set-up of one query: filter, query (maybe_fix_argument, greedy_search.h:140: nothing to fix for L2 / IP), its own record
    // (search_one's inline-record prologue, operation for operation), EntryPointInitializer (greedy_search.h:62-94) ---------------------
    auto setup = [&](Ctx& c) {
        for (uint32_t i = lane; i < filt_bytes / 4u; i += 64) reinterpret_cast<uint32_t*>(c.filt)[i] = 0xFFFFFFFFu;
        const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(c.qi) * a.q_row_bytes;
        (void)prepare_query<DT, METRIC, QM_FLOAT>(qptr, a.q_dtype, d, dpad, c.qf, reinterpret_cast<uint32_t*>(c.qf), 0.0f, 0.0f, lane);
        uint32_t* qcodes = c.compact;
        {
            const uint32_t ci = lane & 31u, half = lane >> 5;
            constexpr int hlen = dpad >> 1;
            const float* qh = c.qf + half * hlen;
            const float* mh = a.pca_mean + half * hlen;
            float r2 = 0.0f, un2 = 0.0f, vv = 0.0f;
            int qq = 0;
            if (lane < 32u) qcodes[lane] = 0u;
            if (lane < 16u) c.qconst[lane] = 0u;
            lds_order();
            {
                const float4* prow = reinterpret_cast<const float4*>(a.pca_t + static_cast<size_t>(ci) * a.pca_pd + half * hlen);
                float up = 0.0f, vvp = 0.0f;
#pragma unroll 1
                for (int j0 = 0; j0 < hlen; j0 += 8) {
                    const float4 p0 = prow[j0 >> 2], p1 = prow[(j0 >> 2) + 1];
                    float4 v0 = *reinterpret_cast<const float4*>(qh + j0), v1 = *reinterpret_cast<const float4*>(qh + j0 + 4);
                    if constexpr (METRIC == M_L2) {
                        const float4 m0 = *reinterpret_cast<const float4*>(mh + j0), m1 = *reinterpret_cast<const float4*>(mh + j0 + 4);
                        v0.x -= m0.x; v0.y -= m0.y; v0.z -= m0.z; v0.w -= m0.w;
                        v1.x -= m1.x; v1.y -= m1.y; v1.z -= m1.z; v1.w -= m1.w;
                    }
                    up = __builtin_fmaf(p0.x, v0.x, up); up = __builtin_fmaf(p0.y, v0.y, up); up = __builtin_fmaf(p0.z, v0.z, up); up = __builtin_fmaf(p0.w, v0.w, up);
                    up = __builtin_fmaf(p1.x, v1.x, up); up = __builtin_fmaf(p1.y, v1.y, up); up = __builtin_fmaf(p1.z, v1.z, up); up = __builtin_fmaf(p1.w, v1.w, up);
                    vvp = __builtin_fmaf(v0.x, v0.x, vvp); vvp = __builtin_fmaf(v0.y, v0.y, vvp); vvp = __builtin_fmaf(v0.z, v0.z, vvp); vvp = __builtin_fmaf(v0.w, v0.w, vvp);
                    vvp = __builtin_fmaf(v1.x, v1.x, vvp); vvp = __builtin_fmaf(v1.y, v1.y, vvp); vvp = __builtin_fmaf(v1.z, v1.z, vvp); vvp = __builtin_fmaf(v1.w, v1.w, vvp);
                }
                const float u = up + __shfl_xor(up, 32);
                vv = vvp + __shfl_xor(vvp, 32);
                if (half == 0u && ci < static_cast<uint32_t>(FDC)) {
                    float cc, uh;
                    if constexpr (METRIC == M_L2) {
                        const float b = a.pca_bias[ci];
                        cc = __builtin_rintf((u - b) / a.pca_step);
                        cc = cc < 0.0f ? 0.0f : (cc > 255.0f ? 255.0f : cc);
                        if (!(cc == cc)) cc = 0.0f;
                        uh = b + a.pca_step * cc;
                    } else {
                        cc = __builtin_rintf(u / a.pca_step);
                        cc = cc < -127.0f ? -127.0f : (cc > 127.0f ? 127.0f : cc);
                        if (!(cc == cc)) cc = 0.0f;
                        uh = a.pca_step * cc;
                    }
                    const int cint = static_cast<int>(cc);
                    const float r = u - uh;
                    r2 = __builtin_fmaf(r, r, r2);
                    un2 = __builtin_fmaf(u, u, un2);
                    qq += cint * cint;
                    reinterpret_cast<uint8_t*>(qcodes)[ci] = static_cast<uint8_t>(cint & 0xFF);
                }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                r2 += __shfl_xor(r2, off);
                un2 += __shfl_xor(un2, off);
                qq += __shfl_xor(qq, off);
            }
            lds_order();
            const float vn = __builtin_sqrtf(vv) * 1.00001f;
            const float urel = (static_cast<float>(d + 3) * 5.9604645e-8f) * 1.01f + a.pca_delta;
            const float erot = __builtin_sqrtf(static_cast<float>(FDC)) * urel * vn;
            const float nu = __builtin_sqrtf(un2);
            float eq = __builtin_sqrtf(r2) * 1.0001f + erot + a.pca_eq_abs;
            const float drel = static_cast<float>(d + 4) * 5.9604645e-8f * 1.01f;
            const float alo = nu * 0.99999f - erot, ahi = nu * 1.00001f + erot;
            const float alo2 = alo > 0.0f ? alo * alo : 0.0f;
            const float thi2 = vv * (1.0f + drel) - alo2, tlo2 = vv * (1.0f - drel) - ahi * ahi;
            float tq_hi = __builtin_sqrtf(thi2 > 0.0f ? thi2 : 0.0f) * 1.00001f;
            const float tq_lo = __builtin_sqrtf(tlo2 > 0.0f ? tlo2 : 0.0f) * 0.99999f;
            if (!(eq < 3.0e38f) || !(tq_hi < 3.0e38f)) { // non-finite query: the bound never fires
                eq = __builtin_inff();
                tq_hi = __builtin_inff();
            }
            // the query's side of the bound lives in LDS and is read back (broadcast) once per expansion, under the block load
            if (lane < 7u) c.qconst[lane] = qcodes[lane];
            if (lane == 0u) {
                c.qconst[8] = static_cast<uint32_t>(qq);
                c.qconst[9] = __float_as_uint(eq);
                c.qconst[10] = __float_as_uint(tq_hi);
                c.qconst[11] = __float_as_uint(tq_lo);
                c.qconst[12] = __float_as_uint(a.pca_step * __builtin_sqrtf(static_cast<float>(qq)) * 1.00001f);
                c.qconst[13] = __float_as_uint(vn);
            }
            lds_order();
        }
        // EntryPointInitializer: push_back + sort == ordered insert without the duplicate check for <= 16 entry points (search_one)
        for (uint32_t base = 0; base < a.n_entry; base += RPP) {
            const uint32_t ci = base + grp;
            const bool act = ci < a.n_entry;
            const uint32_t cid = act ? a.entry[ci] : 0u;
            const float dist = group_distance(c, cid, act);
            unsigned long long m = __ballot(act && sub == 0);
            while (m != 0ull) {
                const int bl = __builtin_ctzll(m);
                m &= m - 1ull;
                const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), bl));
                const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cid), bl));
                ++c.n_eval;
                c.nan_seen = c.nan_seen || dd != dd;
                if (c.buf.size < c.buf.capacity) {
                    if (a.filter_entries != 0u && lane == 0) c.filt[id & fmask] = static_cast<uint16_t>(id >> fshift);
                    uint32_t pos = 0u;
                    for (uint32_t i0 = 0; i0 < c.buf.size; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        bool before = false;
                        if (i < c.buf.size) before = !cmpf<GREATER>(dd, __uint_as_float(c.buf.e[i].x));
                        pos += __popcll(__ballot(before));
                    }
                    lds_order();
                    for (int i = static_cast<int>(c.buf.size) - 1; i >= static_cast<int>(pos); --i) {
                        if (lane == 0) c.buf.e[i + 1] = c.buf.e[i];
                    }
                    if (lane == 0) c.buf.e[pos] = make_uint2(__float_as_uint(dd), id);
                    lds_order();
                    c.buf.size++;
                }
            }
        }
        c.full = c.buf.size == c.buf.capacity;
        c.worst = (c.full && c.buf.capacity > 0u) ? __uint_as_float(c.buf.e[c.buf.capacity - 1].x) : 0.0f;
        c.alive = !wave_done(c.buf);
    };
    if (X.alive) setup(X);
    if (Y.alive) setup(Y);

    struct BlockRegs {
        uint32_t nid;
        uint4 rec[2];
    };
    // phase 1: SearchBuffer::next, then request the node's block [ids | plane 0 | plane 1]: lane c <- neighbour c and its record
    auto request_block = [&](Ctx& c, BlockRegs& b) {
        float node_dist;
        const uint32_t node = wave_next(c.buf, lane, node_dist);
        ++c.n_exp;
        const uint32_t gs = a.gstride;
        const uint32_t* fids = reinterpret_cast<const uint32_t*>(a.fat + static_cast<size_t>(node) * a.fat_stride);
        const uint4* planes = reinterpret_cast<const uint4*>(fids + gs);
        b.nid = kInvalidId;
        b.rec[0] = b.rec[1] = make_uint4(0u, 0u, 0u, 0u);
        if (lane < gs) {
            b.nid = fids[lane];
            b.rec[0] = planes[lane];
            b.rec[1] = planes[gs + lane];
        }
    };
    // phase 2: visited filter, bound of every neighbour from its record, compaction of the survivors (adjacency order) -> their count
    auto screen = [&](Ctx& c, const BlockRegs& b) -> uint32_t {
        const uint32_t nid = b.nid;
        const uint32_t deg_here = __popcll(__ballot(nid != kInvalidId));
        c.n_slots += deg_here;
        bool keep = nid != kInvalidId;
        const uint32_t fh = nid & fmask, ftag = nid >> fshift;
        const bool use_filter = a.filter_entries != 0u && !c.nan_seen; // with a NaN in the buffer a repeated insert is no longer a no-op
        if (use_filter && keep) keep = static_cast<uint32_t>(c.filt[fh]) != ftag;
        if (c.full) {
            const uint4 qa = reinterpret_cast<const uint4*>(c.qconst)[0], qb = reinterpret_cast<const uint4*>(c.qconst)[1];
            const uint4 qc = reinterpret_cast<const uint4*>(c.qconst)[2], qd = reinterpret_cast<const uint4*>(c.qconst)[3];
            const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
            const uint32_t rw[8] = {b.rec[0].x, b.rec[0].y, b.rec[0].z, b.rec[0].w, b.rec[1].x, b.rec[1].y, b.rec[1].z, b.rec[1].w};
            const uint32_t last = rw[7], prev = rw[6];
            bool rej;
            if constexpr (METRIC == M_L2) {
                const int b_qq = static_cast<int>(qc.x);
                const float b_eq = __uint_as_float(qc.y), b_tq_hi = __uint_as_float(qc.z), b_tq_lo = __uint_as_float(qc.w);
                const float wp = c.worst * 1.0001f;
                const float r2 = 2.0f * __builtin_amdgcn_sqrtf(wp) * 1.00001f;
                uint32_t dqx = 0u, dxx = 0u;
#pragma unroll
                for (int w = 0; w < 7; ++w) {
                    dqx = __builtin_amdgcn_udot4(rw[w], qw[w], dqx, false);
                    dxx = __builtin_amdgcn_udot4(rw[w], rw[w], dxx, false);
                }
                const float fi = static_cast<float>(b_qq + static_cast<int>(dxx) - 2 * static_cast<int>(dqx)); // sum (cq - cx)^2 < 2^24: exact
                const float ex = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last & 0xFFFFu))) * a.pca_step;
                const float tx = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last >> 16))) * a.pca_tail_unit;
                const float ee = b_eq + ex;
                const float txe = tx * 4.94e-4f + a.pca_tail_unit * 6.0e-8f;
                const float t1 = b_tq_lo - (tx + txe), t2 = (tx - txe) - b_tq_hi;
                const float tt = t1 > t2 ? t1 : t2;
                const float bb = (tt > 0.0f ? tt : 0.0f) * 0.99999f;
                const float rhs = wp - bb * bb + ee * (r2 + ee) * 1.00001f;
                rej = fi * fat_s2m > rhs; // false for NaN / inf thresholds
            } else {
                const float b_eq = __uint_as_float(qc.y), b_tq_hi = __uint_as_float(qc.z);
                const float b_nuh = __uint_as_float(qd.x), b_nq = __uint_as_float(qd.y);
                int dot = 0;
#pragma unroll
                for (int w = 0; w < 7; ++w) dot = __builtin_amdgcn_sdot4(static_cast<int>(rw[w]), static_cast<int>(qw[w]), dot, false);
                const float main = fat_s2m * static_cast<float>(dot);
                const float ex = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(prev >> 16))) * a.pca_step;
                const float ny = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last & 0xFFFFu))) * a.pca_step;
                const float tx = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last >> 16))) * a.pca_tail_unit;
                const float pos = b_nuh * ex + b_eq * (ny + ex) + b_tq_hi * tx;
                float ub = main + pos + (__builtin_fabsf(main) + pos) * 4.0e-6f;
                ub += static_cast<float>(d + 2) * 6.1e-8f * b_nq * (ny + ex + tx + a.pca_ref_extra); // the reference's own float32 accumulation
                rej = ub < c.worst; // false whenever anything is NaN / inf
            }
            const uint32_t nrej = __popcll(__ballot(keep && rej));
            c.n_code_rej += nrej;
            c.n_eval += nrej;
            keep = keep && !rej;
        }
        if (use_filter && keep) c.filt[fh] = static_cast<uint16_t>(ftag);
        const unsigned long long km = __ballot(keep);
        const uint32_t cnt = __popcll(km);
        const uint32_t rank = __popcll(km & ((1ull << lane) - 1ull));
        lds_order();
        if (keep) c.compact[rank] = nid;
        lds_order();
        return cnt;
    };
    struct PassRegs {
        RowData<DT, DS> rows;
        uint32_t cid;
        bool act;
    };
    // phase 3: request the rows of the pass's (up to 16) survivors; idle groups read row 0 (cache resident, no predicate)
    auto request_rows = [&](Ctx& c, uint32_t p0, uint32_t cnt, PassRegs& p) {
        const uint32_t ci = p0 + grp;
        p.act = ci < cnt;
        p.cid = 0u;
        if (p.act) p.cid = c.compact[ci];
        row_issue<DT, DS>(data + static_cast<size_t>(p.cid) * a.row_bytes, sub, p.rows);
    };
    // phase 4: distances in the reference's operation order, can_skip, then every accepted candidate into the sorted buffer (search_one's code)
    auto finish_pass = [&](Ctx& c, uint32_t p0, uint32_t cnt, const PassRegs& p) {
        const float dist = row_finish<DT, METRIC, DS>(p.rows, c.qf, sub);
        const bool skip = c.full && cmpf<GREATER>(c.worst, dist); // can_skip (search_buffer.h:338-340)
        const bool cand = p.act && sub == 0 && !skip;
        unsigned long long m = __ballot(cand);
        c.n_eval += (cnt - p0) < RPP ? (cnt - p0) : RPP;
        if (m != 0ull) {
            const bool finite = __ballot(cand && !(__builtin_fabsf(dist) < __builtin_inff())) == 0ull;
            bool merged = false;
            if (!c.nan_seen && finite) merged = wave_merge_insert<GREATER, 2>(c.buf, cand, dist, p.cid, lane, c.n_ins, c.worst);
            if (merged) {
                m = 0ull;
                c.full = c.buf.size == c.buf.capacity; // `worst` came out of the merge
            }
        }
        while (m != 0ull) { // NaN / infinite distances, duplicates and ties: one by one (search_buffer.h:342-403)
            const int bl = __builtin_ctzll(m);
            m &= m - 1ull;
            const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), bl));
            if (c.full && cmpf<GREATER>(c.worst, dd)) continue; // worst may have improved within this pass
            const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(p.cid), bl));
            c.nan_seen = c.nan_seen || dd != dd;
            if (wave_insert<GREATER>(c.buf, dd, id, lane, c.nan_seen)) {
                ++c.n_ins;
                c.full = c.buf.size == c.buf.capacity;
                if (c.full) c.worst = __uint_as_float(c.buf.e[c.buf.capacity - 1].x);
            }
        }
    };

    // ---- main loop (greedy_search.h:153-202): the queries take turns, one expansion each.  A query's NEXT block is requested the moment
    // its expansion ends (next() is known then) and arrives while the wave works on the other query: of the two dependent memory round
    // trips of an expansion, the block's is always hidden; the rows' is covered by the other waves of the SIMD, as in the one-query kernel.
    auto expand = [&](Ctx& c, BlockRegs& b) {
        const uint32_t cnt = screen(c, b);
        PassRegs p;
        for (uint32_t p0 = 0u; p0 < cnt; p0 += RPP) {
            request_rows(c, p0, cnt, p);
            finish_pass(c, p0, cnt, p);
        }
        c.alive = !wave_done(c.buf);
        if (c.alive) request_block(c, b);
    };
    BlockRegs bx, by;
    bx.nid = by.nid = kInvalidId;
    bx.rec[0] = bx.rec[1] = by.rec[0] = by.rec[1] = make_uint4(0u, 0u, 0u, 0u);
    if (X.alive) request_block(X, bx);
    if (Y.alive) request_block(Y, by);
    while (X.alive || Y.alive) {
        if (X.alive) expand(X, bx);
        if (Y.alive) expand(Y, by);
    }

    // ---- copy-back (extensions.h:588-590), padded like vamana_index_impl.h:205-209 ---------------------------------------------------
    auto copy_back = [&](const Ctx& c) {
        for (uint32_t j = lane; j < a.k; j += 64) {
            uint32_t id = kInvalidId;
            float dist = __builtin_inff();
            const bool have = j < c.buf.size;
            if (have) {
                const uint2 en = c.buf.e[j];
                id = en.y & kIdMask;
                dist = __uint_as_float(en.x);
            }
            const size_t o = static_cast<size_t>(c.qi) * a.k + j;
            if (a.ids64 != nullptr) a.ids64[o] = have ? (static_cast<uint64_t>(id) + a.id_offset) : ~0ull;
            if (a.ids32 != nullptr) a.ids32[o] = have ? static_cast<uint32_t>(id + static_cast<uint32_t>(a.id_offset)) : kInvalidId;
            if (a.dists != nullptr) a.dists[o] = dist;
        }
        if (lane == 0 && a.stats != nullptr) {
            uint32_t* st = a.stats + static_cast<size_t>(c.qi) * 6;
            st[0] = c.n_exp;
            st[1] = c.n_slots;
            st[2] = c.n_eval;
            st[3] = c.n_ins;
            st[4] = 0u;
            st[5] = c.n_code_rej;
        }
    };
    if (X.qi < a.nq) copy_back(X);
    if (Y.qi < a.nq) copy_back(Y);
}

} // namespace svsx
