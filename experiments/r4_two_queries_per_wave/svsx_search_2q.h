// svsx_search_2q.h -- batched Vamana greedy search, TWO queries per wavefront (gfx950 / CDNA4; round 4).
//
// Same reference semantics as svsx_search_kernel.h (greedy_search.h:124-203, search_buffer.h:104-497, extensions.h:555-592), same
// arithmetic (the row-distance routines of svsx_device.h are shared), different mapping to the machine:
//
//   * a wave carries two independent queries, one per 32-lane half.  Everything that was wave-uniform in the one-query kernel (buffer
//     size, best_unvisited, worst, the node being expanded ...) is a per-half value held in vector registers; ballots are split into
//     their two 32-bit halves; cross-lane sums stay inside a half (quad / row DPP + one ds_swizzle).  The code below is plain SIMT:
//     every lane of a half computes the same control flow, the two halves may diverge (exec masking by the compiler).
//   * why: the one-query kernel is limited to 6 waves per SIMD = 6 144 resident queries, so a 10 000-query launch runs 1.63
//     generations and its tail is latency-bound (profiles/r03_notes.md section 3).  With two queries per wave and <= 3 840 bytes of LDS
//     per query, 20 waves per CU = 10 240 queries are resident at once: the whole batch is one generation.
//   * per expansion and half: lane c owns adjacency slots c and c + 32 (ids + 32-byte records, all loads issued together); the bound
//     and the visited filter are lane-parallel; survivors are compacted in adjacency order; distance passes of 8 rows per half
//     (4 lanes per row, the reference's accumulation tree); ALL accepted candidates of a pass enter the sorted buffer in one stable
//     merge (= the sequence of SearchBuffer::insert calls, search_buffer.h:342-403) done without the scalar unit: the candidates'
//     distances are broadcast through LDS, every lane counts for its own EPL buffer entries how many candidates beat them, and the
//     insert positions come from byte-packed per-rank counts summed over the half.
//   * scope: graph-inline 32-byte records (svsx_codes.hip), float32 / float16 rows, L2 / inner product, static dimension, buffers of
//     <= 128 entries, <= 8 entry points, one adjacency chunk (max degree <= 64), no history / cancel word / iterator state.  Anything
//     else is served by the one-query kernel.  A query that meets a non-finite distance (NaN / +-inf: the sentinel arithmetic of the
//     merge needs a strict weak order) gives up, and once both halves have finished the whole wave answers it again with search_one
//     (svsx_search_kernel.h) -- results never depend on which routine served a query.
#pragma once

// (included by svsx_search_kernel.h after search_one; nothing else includes it)

namespace svsx {

// the 32 bits of a wave-wide ballot that belong to this lane's half
__device__ __forceinline__ uint32_t half_ballot(bool p, uint32_t half) {
    const unsigned long long b = __ballot(p);
    const uint32_t lo = static_cast<uint32_t>(b), hi = static_cast<uint32_t>(b >> 32);
    return half != 0u ? hi : lo;
}

// sum / or over the 32 lanes of a half, result in every lane: quad xor 1, quad xor 2, row_half_mirror, row_mirror (DPP), lane ^ 16 (ds_swizzle)
__device__ __forceinline__ uint32_t half_sum_u32(uint32_t v) {
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x140, 0xf, 0xf, true));
    v += static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x401F));
    return v;
}
__device__ __forceinline__ uint32_t half_or_u32(uint32_t v) {
    v |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0xB1, 0xf, 0xf, true));
    v |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x4E, 0xf, 0xf, true));
    v |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x141, 0xf, 0xf, true));
    v |= static_cast<uint32_t>(__builtin_amdgcn_mov_dpp(static_cast<int>(v), 0x140, 0xf, 0xf, true));
    v |= static_cast<uint32_t>(__builtin_amdgcn_ds_swizzle(static_cast<int>(v), 0x401F));
    return v;
}

// LDS bytes of ONE query (half of a wave's request); the launcher and the kernel agree through this.  Everything but the visited filter has
// a compile-time offset from the half's base (one address register serves all of it); the filter, whose size is a launch parameter, is last.
//   [buffer 32 EPL entries][query DS f32][survivor ids 32][candidate distances / ids 8][record codes + bound constants 16][counters 8][filter]
__host__ __device__ constexpr uint32_t twoq_fixed_bytes(uint32_t epl, uint32_t ds) { return epl * 256u + ds * 4u + 128u + 32u + 64u + 32u; }
__host__ __device__ constexpr uint32_t twoq_half_bytes(uint32_t epl, uint32_t ds, uint32_t filter_entries) { return twoq_fixed_bytes(epl, ds) + filter_entries * 2u; }

// Residency: 5 waves per SIMD = 20 per CU = 40 queries per CU (LDS allows 21 waves of 7 680 bytes): <= 96 vector registers
#ifndef SVSX_2Q_WAVES
#define SVSX_2Q_WAVES 5
#endif
#if SVSX_2Q_WAVES > 0
#define SVSX_2Q_ATTR __attribute__((amdgpu_waves_per_eu(SVSX_2Q_WAVES)))
#else
#define SVSX_2Q_ATTR
#endif

// EPL: buffer entries per lane (capacity <= 32 EPL)
template <int DT, int METRIC, int DS, int EPL>
__global__ __launch_bounds__(64) SVSX_2Q_ATTR void search_kernel_2q(const SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    static_assert(METRIC == M_L2 || METRIC == M_IP, "two-query kernel: L2 and inner product");
    static_assert(DT == DT_F32 || DT == DT_F16, "two-query kernel: float rows");
    static_assert(DS > 0 && (DS & 15) == 0, "two-query kernel: static dimension, whole 16-blocks");
    constexpr bool GREATER = METRIC != M_L2;
    constexpr int FDC = 32 - (METRIC == M_L2 ? 4 : 6);      // coded coordinates of a 32-byte record
    constexpr bool B8 = use_block_cyclic<DT>();
    static_assert(B8 || Elem<DT>::E == 4, "four lanes per candidate row");
    constexpr uint32_t RPP = 8;                              // rows per pass and half
    constexpr float kPast = GREATER ? -__builtin_inff() : __builtin_inff(); // no candidate compares better than an empty slot
    constexpr int d = DS, dpad = DS;

    const uint32_t lane = threadIdx.x, half = lane >> 5, hl = lane & 31u;
    const uint32_t sub = hl & 3u, grp = hl >> 2;
    const uint32_t qraw = blockIdx.x * 2u + half;
    const bool live = qraw < a.nq;                           // an odd batch leaves the last wave's second half idle
    const uint32_t qi = live ? qraw : a.nq - 1u;

    const uint32_t capacity = a.capacity, window = a.window;
    const uint32_t filt_bytes = a.filter_entries * 2u;
    unsigned char* hb = smem + half * twoq_half_bytes(EPL, DS, a.filter_entries);
    uint2* buf = reinterpret_cast<uint2*>(hb);
    float* qf = reinterpret_cast<float*>(hb + EPL * 256);
    uint32_t* compact = reinterpret_cast<uint32_t*>(hb + EPL * 256 + DS * 4);
    uint32_t* cs = compact + 32;      // distances of the (up to 8) candidates of a pass; their ids when a candidate ties with a buffer entry
    uint32_t* qconst = cs + 8;        // words 0..6 record codes of the query, 7 zero, 8 qq, 9 eq, 10 tq_hi, 11 tq_lo, 12 ||uhat||, 13 ||v||
    uint32_t* counters = qconst + 16; // statistics of this query: 0 expansions, 1 neighbour slots, 2 distance evaluations, 3 inserts, 4 bound rejects
    uint16_t* filt = reinterpret_cast<uint16_t*>(counters + 8);

    const uint32_t fmask = a.filter_entries - 1u;
    const uint32_t fshift = static_cast<uint32_t>(__builtin_ctz(a.filter_entries | 0x80000000u));
    const bool use_filter = a.filter_entries != 0u;
    for (uint32_t i = hl; i < filt_bytes / 4u; i += 32u) reinterpret_cast<uint32_t*>(filt)[i] = 0xFFFFFFFFu;

    // ---- load the query (maybe_fix_argument, greedy_search.h:140: nothing to fix for L2 / IP) ---------------------------------------
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(qi) * a.q_row_bytes;
    for (int i = static_cast<int>(hl); i < dpad; i += 32) {
        float v;
        switch (a.q_dtype) { // exact widening of query element i (prepare_query, svsx_device.h)
            case DT_F32: v = reinterpret_cast<const float*>(qptr)[i]; break;
            case DT_F16: v = static_cast<float>(reinterpret_cast<const _Float16*>(qptr)[i]); break;
            case DT_I8: v = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(qptr)[i])); break;
            default: v = static_cast<float>(static_cast<unsigned>(qptr[i])); break;
        }
        qf[i] = v;
    }
    if (hl < 24u) qconst[hl] = 0u; // the query's record words and the counters behind them
    lds_order();
    QueryFix fx;
    fx.norm = 0.0f;
    fx.offset = 0.0f;
    fx.qq = 0;

    // ---- the query's own record: coordinates in the principal basis, codes, remainder norm (search_kernel's FAT prologue, operation for
    // operation: one lane per coordinate, the element range in the same two halves, the same reduction tree over 32 lanes -- so the
    // bound constants, and with them every rejection, are the ones the one-query kernel computes) -----------------------------------------
    float fat_eq, fat_tq_hi, fat_tq_lo, fat_nuh, fat_nq;
    int fat_qq;
    {
        constexpr int hlen = dpad >> 1;
        float up2[2] = {0.0f, 0.0f}, vv2[2] = {0.0f, 0.0f};
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const float4* prow = reinterpret_cast<const float4*>(a.pca_t + static_cast<size_t>(hl) * a.pca_pd + hh * hlen);
            const float* qh = qf + hh * hlen;
            const float* mh = a.pca_mean + hh * hlen;
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
            up2[hh] = up;
            vv2[hh] = vvp;
        }
        const float u = up2[0] + up2[1];
        const float vv = vv2[0] + vv2[1];
        float r2 = 0.0f, un2 = 0.0f;
        int qq = 0;
        if (hl < static_cast<uint32_t>(FDC)) {
            float c, uh;
            if constexpr (METRIC == M_L2) {
                const float b = a.pca_bias[hl];
                c = __builtin_rintf((u - b) / a.pca_step);
                c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
                if (!(c == c)) c = 0.0f;
                uh = b + a.pca_step * c;
            } else {
                c = __builtin_rintf(u / a.pca_step);
                c = c < -127.0f ? -127.0f : (c > 127.0f ? 127.0f : c);
                if (!(c == c)) c = 0.0f;
                uh = a.pca_step * c;
            }
            const int cint = static_cast<int>(c);
            const float r = u - uh;
            r2 = __builtin_fmaf(r, r, r2);
            un2 = __builtin_fmaf(u, u, un2);
            qq += cint * cint;
            reinterpret_cast<uint8_t*>(qconst)[hl] = static_cast<uint8_t>(cint & 0xFF);
        }
        for (int off = 16; off >= 1; off >>= 1) {
            r2 += __shfl_xor(r2, off);
            un2 += __shfl_xor(un2, off);
            qq += __shfl_xor(qq, off);
        }
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
        fat_qq = qq;
        fat_eq = eq;
        fat_tq_hi = tq_hi;
        fat_tq_lo = tq_lo;
        fat_nuh = a.pca_step * __builtin_sqrtf(static_cast<float>(qq)) * 1.00001f;
        fat_nq = vn;
    }
    lds_order();
    if (hl == 0u) { // per-half constants of the bound: re-read from LDS once per expansion (under the block load) instead of living in registers
        qconst[8] = static_cast<uint32_t>(fat_qq);
        qconst[9] = __float_as_uint(fat_eq);
        qconst[10] = __float_as_uint(fat_tq_hi);
        qconst[11] = __float_as_uint(fat_tq_lo);
        qconst[12] = __float_as_uint(fat_nuh);
        qconst[13] = __float_as_uint(fat_nq);
    }
    lds_order();
    const float fat_s2m = a.pca_step * a.pca_step * (METRIC == M_L2 ? 0.99999f : 1.0f);

    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    auto group_distance = [&](uint32_t cid, bool active) -> float {
        const unsigned char* row = data + static_cast<size_t>(active ? cid : 0u) * a.row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        if constexpr (B8) float_row_distance_bc<DT, METRIC, DS, false>(row, active, qf, d, static_cast<int>(sub), 0.0f, 0.0f, fmain, fnrm);
        else float_row_distance<DT, METRIC, DS, false>(row, active, qf, d, static_cast<int>(sub), 0.0f, 0.0f, fmain, fnrm);
        return finish_distance<METRIC, QM_FLOAT>(fmain, fnrm, 0, 0, fx, 0.0f);
    };

    // ---- per-half search state (vector registers; the same value in the 32 lanes of a half) -------------------------------------------
    uint32_t size = 0u, best = 0u;
    bool full = false, bail = false;
    float worst = 0.0f;
    // the statistics live in LDS (one lane adds): five registers fewer in a kernel whose residency is decided by its vector registers
    auto count = [&](int which, uint32_t v) {
        if (hl == 0u) counters[which] += v;
    };

    // All candidates of a pass (lanes with `cand`: sub == 0 of a row group, groups in adjacency order) into the sorted buffer at once.
    // Equal to inserting them one after the other (search_buffer.h:342-403): a sequence of bounded sorted inserts IS a stable merge
    // truncated to the capacity -- old entries keep their order and stay before new entries of equal distance, new entries of equal
    // distance keep adjacency order, whatever ends at a position >= capacity is dropped, a candidate with the id AND distance of an old
    // entry is not inserted (:377-389), best_unvisited = min(old, position of the best new entry).
    //   c_i   = candidates strictly better than old entry i                 -> old entry i moves to i + c_i
    //   le_r  = old entries that stay before the candidate of rank r = #{i : c_i <= r}  (c is non-decreasing along the sorted buffer)
    //           -> the candidate goes to le_r + r;  all eight le_r at once: byte r of a 64-bit word counts the lanes' entries with c_i <= r
    auto merge_pass = [&](bool cand, float dist, uint32_t cid, bool check_dups) {
        uint32_t cm = half_ballot(cand, half);
        if (cm == 0u) return;
        if (half_ballot(cand && !(__builtin_fabsf(dist) < __builtin_inff()), half) != 0u) { // NaN / +-inf: the one-query kernel answers this query
            bail = true;
            return;
        }
        uint2 e[EPL];
        float dv[EPL];
        uint32_t c[EPL];
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const uint32_t i = hl + 32u * t;
            e[t] = make_uint2(__float_as_uint(kPast), kInvalidId);
            if (i < size) e[t] = buf[i];
            dv[t] = __uint_as_float(e[t].x);
        }
        const uint32_t dbits = __float_as_uint(dist);
        float dj[8];
#pragma unroll 1
        for (int attempt = 0; attempt < 2; ++attempt) {
            lds_order();
            if (sub == 0u) cs[grp] = cand ? dbits : 0x7FC00000u; // an empty slot is a NaN: every comparison with it is false
            lds_order();
            const uint4 ca = reinterpret_cast<const uint4*>(cs)[0], cb = reinterpret_cast<const uint4*>(cs)[1];
            dj[0] = __uint_as_float(ca.x); dj[1] = __uint_as_float(ca.y); dj[2] = __uint_as_float(ca.z); dj[3] = __uint_as_float(ca.w);
            dj[4] = __uint_as_float(cb.x); dj[5] = __uint_as_float(cb.y); dj[6] = __uint_as_float(cb.z); dj[7] = __uint_as_float(cb.w);
            bool tie = false;
#pragma unroll
            for (int t = 0; t < EPL; ++t) c[t] = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < EPL; ++t) {
                    c[t] += cmpf<GREATER>(dj[j], dv[t]) ? 1u : 0u;
                    tie = tie || dj[j] == dv[t]; // empty buffer slots hold +-inf, candidates are finite
                }
            }
            if (!check_dups || attempt == 1 || half_ballot(tie, half) == 0u) break;
            // a candidate's distance equals an old entry's: the reference walks back over the entries of equal distance and drops the
            // candidate if one of them has its id (search_buffer.h:377-389).  Find those candidates, take them out, count again.
            lds_order(); // (the distances are in registers by now: the same eight words carry the ids)
            if (sub == 0u) cs[grp] = cid;
            lds_order();
            const uint4 ia = reinterpret_cast<const uint4*>(cs)[0], ib = reinterpret_cast<const uint4*>(cs)[1];
            const uint32_t idj[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
            uint32_t dm = 0u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int t = 0; t < EPL; ++t)
                    if (dj[j] == dv[t] && (e[t].y & kIdMask) == idj[j]) dm |= 1u << j;
            }
            dm = half_or_u32(dm);
            if (dm == 0u) break; // equal distances, different ids: ordinary inserts behind the old entries
            if (((dm >> grp) & 1u) != 0u) cand = false;
            cm = half_ballot(cand, half);
            if (cm == 0u) return;
        }
        // rank of my candidate among the new ones: better first, equal distances in adjacency order
        uint32_t rank = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) rank += (cmpf<GREATER>(dj[j], dist) || (dj[j] == dist && static_cast<uint32_t>(j) < grp)) ? 1u : 0u;
        // byte r of (pk_hi : pk_lo) = this lane's old entries with c_i <= r; summed over the half = le_r (<= 128: no carry between bytes)
        uint32_t pk_lo = 0u, pk_hi = 0u;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const uint32_t i = hl + 32u * t;
            if (i < size) {
                const uint32_t ct = c[t];
                pk_lo += ct < 4u ? (0x01010101u << ((8u * ct) & 31u)) : 0u;
                pk_hi += ct < 4u ? 0x01010101u : (ct < 8u ? (0x01010101u << ((8u * (ct - 4u)) & 31u)) : 0u);
            }
        }
        pk_lo = half_sum_u32(pk_lo);
        pk_hi = half_sum_u32(pk_hi);
        const uint32_t le = rank < 4u ? ((pk_lo >> ((8u * rank) & 31u)) & 0xFFu) : ((pk_hi >> ((8u * (rank - 4u)) & 31u)) & 0xFFu);
        const uint32_t minpos = pk_lo & 0xFFu; // the best new entry (rank 0) lands at le_0
        lds_order();
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const uint32_t i = hl + 32u * t, p = i + c[t];
            if (i < size && c[t] != 0u && p < capacity) buf[p] = e[t];
        }
        const uint32_t pn = le + rank;
        const bool placed = cand && pn < capacity;
        if (placed) buf[pn] = make_uint2(dbits, cid);
        lds_order();
        count(3, static_cast<uint32_t>(__popc(half_ballot(placed, half))));
        const uint32_t m = static_cast<uint32_t>(__popc(cm));
        size = (size + m < capacity) ? size + m : capacity;
        best = minpos < best ? minpos : best;
        full = size == capacity;
        if (full) worst = __uint_as_float(buf[capacity - 1u].x);
        lds_order();
    };

    // ---- EntryPointInitializer (greedy_search.h:62-94): distances to the entry points, push_back, sort.  The host sends at most 8 entry
    // points and never more than the capacity here, so push_back keeps all of them and std::sort of <= 16 elements is a stable insertion
    // sort: one merge into the empty buffer, WITHOUT the duplicate check (push_back has none) --------------------------------------------
    {
        const bool act = live && grp < a.n_entry;
        const uint32_t cid = act ? a.entry[grp] : 0u;
        const float dist = group_distance(cid, act);
        const bool cand = act && sub == 0u;
        count(2, a.n_entry);
        if (use_filter && cand) filt[cid & fmask] = static_cast<uint16_t>(cid >> fshift); // marking a pushed entry point is result-neutral
        merge_pass(cand, dist, cid, false);
    }

    // ---- main loop (greedy_search.h:153-202) -----------------------------------------------------------------------------------------------
    uint32_t guard = 0u; // (a search of a <= 128-entry buffer takes a few hundred expansions; a runaway loop hands the query back instead of hanging)
    while (live && !bail && best != (size < window ? size : window)) {
        if (++guard > (1u << 22)) {
            bail = true;
            break;
        }
        // SearchBuffer::next (search_buffer.h:294-304): the best unvisited entry, marked visited; best_unvisited advances to the next one
        const uint2 en = buf[best];
        lds_order();
        if (hl == 0u) buf[best].y = en.y | kVisitedBit;
        lds_order();
        const uint32_t node = en.y & kIdMask;
        {
            const uint32_t upper = size < window ? size : window;
            uint32_t nb = upper;
#pragma unroll
            for (int t = EPL - 1; t >= 0; --t) {
                const uint32_t i = hl + 32u * t;
                bool unv = false;
                if (i > best && i < upper) unv = (buf[i].y & kVisitedBit) == 0u;
                const uint32_t um = half_ballot(unv, half);
                nb = um != 0u ? 32u * t + static_cast<uint32_t>(__builtin_ctz(um)) : nb;
            }
            best = nb;
        }
        count(0, 1u);

        // one contiguous block [ids | plane 0 | plane 1]: lane c owns neighbours c and c + 32 with their records; every load issued at once
        const uint32_t gs = a.gstride;
        const uint32_t* fids = reinterpret_cast<const uint32_t*>(a.fat + static_cast<size_t>(node) * a.fat_stride);
        const uint4* planes = reinterpret_cast<const uint4*>(fids + gs);
        uint32_t nid0 = kInvalidId, nid1 = kInvalidId;
        uint4 rec0[2], rec1[2];
        rec0[0] = rec0[1] = rec1[0] = rec1[1] = make_uint4(0u, 0u, 0u, 0u);
        if (hl < gs) {
            nid0 = fids[hl];
            rec0[0] = planes[hl];
            rec0[1] = planes[gs + hl];
        }
        if (hl + 32u < gs) {
            nid1 = fids[hl + 32u];
            rec1[0] = planes[hl + 32u];
            rec1[1] = planes[gs + hl + 32u];
        }
        // the query's side of the bound, from LDS while the block is in flight
        const uint4 qa = reinterpret_cast<const uint4*>(qconst)[0], qb = reinterpret_cast<const uint4*>(qconst)[1];
        const uint4 qc = reinterpret_cast<const uint4*>(qconst)[2], qd = reinterpret_cast<const uint4*>(qconst)[3];
        const uint32_t qw[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
        const int b_qq = static_cast<int>(qc.x);
        const float b_eq = __uint_as_float(qc.y), b_tq_hi = __uint_as_float(qc.z), b_tq_lo = __uint_as_float(qc.w);
        const float b_nuh = __uint_as_float(qd.x), b_nq = __uint_as_float(qd.y);

        count(1, static_cast<uint32_t>(__popc(half_ballot(nid0 != kInvalidId, half)) + __popc(half_ballot(nid1 != kInvalidId, half))));

        // the bound of one candidate from its record (svsx_codes.hip has the geometry; the same arithmetic as search_kernel's):
        // a rejected candidate is one SearchBuffer::can_skip (search_buffer.h:338-340) would drop whatever its exact distance
        auto rejected = [&](const uint4 (&rec)[2]) -> bool {
            const uint32_t rw[8] = {rec[0].x, rec[0].y, rec[0].z, rec[0].w, rec[1].x, rec[1].y, rec[1].z, rec[1].w};
            const uint32_t last = rw[7], prev = rw[6];
            if constexpr (METRIC == M_L2) {
                const float wp = worst * 1.0001f;
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
                return fi * fat_s2m > rhs; // false for NaN / inf thresholds
            } else {
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
                return ub < worst; // false whenever anything is NaN / inf
            }
        };
        // visited filter (greedy_search.h:191-193) holding only ids whose rows were really fetched, then the bound
        bool keep0 = nid0 != kInvalidId, keep1 = nid1 != kInvalidId;
        const uint32_t fh0 = nid0 & fmask, ft0 = nid0 >> fshift, fh1 = nid1 & fmask, ft1 = nid1 >> fshift;
        if (use_filter) {
            if (keep0) keep0 = static_cast<uint32_t>(filt[fh0]) != ft0;
            if (keep1) keep1 = static_cast<uint32_t>(filt[fh1]) != ft1;
        }
        if (full) {
            const bool rj0 = keep0 && rejected(rec0), rj1 = keep1 && rejected(rec1);
            const uint32_t nrej = static_cast<uint32_t>(__popc(half_ballot(rj0, half)) + __popc(half_ballot(rj1, half)));
            count(4, nrej);
            count(2, nrej);
            keep0 = keep0 && !rj0;
            keep1 = keep1 && !rj1;
        }
        if (use_filter) {
            if (keep0) filt[fh0] = static_cast<uint16_t>(ft0);
            if (keep1) filt[fh1] = static_cast<uint16_t>(ft1);
        }
        // survivors in adjacency order: slots 0..31 first, then 32..63
        const uint32_t km0 = half_ballot(keep0, half), km1 = half_ballot(keep1, half);
        const uint32_t cnt = static_cast<uint32_t>(__popc(km0) + __popc(km1));
        for (uint32_t base = 0u; base < cnt; base += 32u) { // 32 survivors at a time: one round almost always
            lds_order();
            {
                const uint32_t below = (1u << hl) - 1u;
                const uint32_t rank0 = static_cast<uint32_t>(__popc(km0 & below)), rank1 = static_cast<uint32_t>(__popc(km0) + __popc(km1 & below));
                if (base != 0u) { // a later round (more than 32 survivors, rare): the ids again, from the block that was just read
                    if (hl < gs) nid0 = fids[hl];
                    if (hl + 32u < gs) nid1 = fids[hl + 32u];
                }
                if (((km0 >> hl) & 1u) != 0u && rank0 - base < 32u) compact[rank0 - base] = nid0;
                if (((km1 >> hl) & 1u) != 0u && rank1 - base < 32u) compact[rank1 - base] = nid1;
            }
            lds_order();
            const uint32_t ncur = (cnt - base) < 32u ? (cnt - base) : 32u;
            for (uint32_t p0 = 0u; p0 < ncur; p0 += RPP) {
                const uint32_t ci = p0 + grp;
                const bool act = ci < ncur;
                uint32_t cid = 0u;
                if (act) cid = compact[ci];
                const float dist = group_distance(cid, act);
                const bool skip = full && cmpf<GREATER>(worst, dist); // can_skip (search_buffer.h:338-340)
                const bool cand = act && sub == 0u && !skip;
                count(2, (ncur - p0) < RPP ? (ncur - p0) : RPP);
                merge_pass(cand, dist, cid, true);
            }
        }
    }

    if (live && !bail) {
        // ---- copy-back (extensions.h:588-590), padded like vamana_index_impl.h:205-209 ------------------------------------------------------
        for (uint32_t j = hl; j < a.k; j += 32u) {
            uint32_t id = kInvalidId;
            float dist = __builtin_inff();
            const bool have = j < size;
            if (have) {
                const uint2 en = buf[j];
                id = en.y & kIdMask;
                dist = __uint_as_float(en.x);
            }
            const size_t o = static_cast<size_t>(qi) * a.k + j;
            if (a.ids64 != nullptr) a.ids64[o] = have ? (static_cast<uint64_t>(id) + a.id_offset) : ~0ull;
            if (a.ids32 != nullptr) a.ids32[o] = have ? static_cast<uint32_t>(id + static_cast<uint32_t>(a.id_offset)) : kInvalidId;
            if (a.dists != nullptr) a.dists[o] = dist;
        }
        if (hl == 0u && a.stats != nullptr) {
            uint32_t* st = a.stats + static_cast<size_t>(qi) * 6;
            st[0] = counters[0];
            st[1] = counters[1];
            st[2] = counters[2];
            st[3] = counters[3];
            st[4] = 0u;
            st[5] = counters[4];
        }
    }
    // ---- a half that met a non-finite distance: the WHOLE wave answers that query again with the one-query routine, in the wave's own LDS
    // (the host checks that its layout fits), after the other half's results have left LDS.  Rare (NaN / infinite distances only), and
    // the answer is the one-query kernel's by construction ------------------------------------------------------------------------------------
    const unsigned long long gm = __ballot(live && bail);
    if (gm != 0ull) {
        lds_order();
#pragma unroll 1
        for (uint32_t h = 0u; h < 2u; ++h) {
#ifndef SVSX_2Q_NOREDO
            if (((gm >> (32u * h)) & 1ull) != 0ull) search_one<DT, METRIC, QM_FLOAT, DS, false, 2, 0>(a, blockIdx.x * 2u + h, smem);
#endif
            lds_order();
        }
    }
}

} // namespace svsx
