// svsx_search_kernel.h -- batched Vamana greedy search, one wavefront per query (gfx950 / CDNA4): the kernel template and
// its launcher.  Included by the per-row-type translation units svsx_search_{f32,f16,i8,u8}.hip (the instantiations compile in
// parallel) -- nothing else includes it.
//
// Restates, wave-parallel but order-exact, the reference's per-thread loop:
//   VamanaIndex::search batch        index/vamana/index.h:564-611
//   per_thread_batch_search          index/vamana/extensions.h:555-592
//   greedy_search                    index/vamana/greedy_search.h:124-203
//   EntryPointInitializer            index/vamana/greedy_search.h:62-94
//   SearchBuffer                     index/vamana/search_buffer.h:104-497
//
// Mapping to the machine:
//   * grid = nq single-wave workgroups; the hardware dispatcher is the work queue (a CU holds up to 32 waves,
//     each wave hides its own gather latency behind the others').
//   * query (f32 or bytes) and the sorted candidate buffer live in the wave's LDS slice.
//   * per expansion: one coalesced load of the padded adjacency row (lane c <- neighbour c), then passes of
//     64/G candidate rows: G adjacent lanes stream one row with 16-B loads (row = 384 B at d=96 f32), reduce
//     with DPP quad exchanges in the reference's exact summation tree, and the surviving candidates
//     (ballot against the current worst) are inserted in adjacency order -> identical buffer contents to the
//     sequential CPU loop.
#pragma once

#include "svsx_device.h"
#include "svsx_kernels.h"

#include <cstdio>
#include <cstdlib>

#ifndef SVSX_CODE_BATCH
// bound-code passes (16 candidates each) whose loads are issued together.  Measured on the bench workload: 1 -> 4.90,
// 2 -> 5.17, 3 -> 5.13, 4 -> 4.55 M QPS (4 needs 105 VGPRs = 4 waves/SIMD); profiles/r01_notes.md
#define SVSX_CODE_BATCH 2
#endif

// SVSX_MERGE_V2 (default 0): 1 = lane-parallel merge insert (wave_merge_insert2) for A/B builds.  Measured (profiles/r03_notes.md): its LDS
// round trips cost what the scalar loop's VALU <-> SALU hand-overs cost (config 2: 1.084 vs 1.070 ms), so the loop form stays.
#ifndef SVSX_MERGE_V2
#define SVSX_MERGE_V2 0
#endif
// waves per SIMD asked of the scalar-quantized cosine kernels of d = 768 (config 4): 5 spills 28-40 bytes at windows > 128 (A/B: profiles/r04_notes.md)
#ifndef SVSX_SQCOS_FLOOR
#define SVSX_SQCOS_FLOOR 4
#endif

namespace svsx {

// Occupancy floors (amdgpu_waves_per_eu) of the inline-record kernel families: the register allocator otherwise settles some instantiations
// just over a VGPR bracket, and a resident wave lost costs more than the few registers buy.  Chosen per family so that no instantiation
// spills (tools/kernel_regs.py); 0 = no floor.  (Round 3's "early block" experiment, for which the floors were first introduced, lives in
// experiments/r3_early_block/.)
template <int DT, int METRIC, int QMODE, int DS, bool GENERAL, int CODES, int BIG> constexpr int occupancy_floor() {
    if (GENERAL || CODES < 2) return 0;
    if (QMODE == QM_FLOAT && (DT == DT_F32 || DT == DT_F16)) {
        // configs 2, 3, 5 (static 96 / 128, 32-byte records): <= 80 VGPRs -- the scalar registers allow 6 waves; float16 rows of 96 need 88
        // (float32 rows of 128 keep 8 x 16 bytes of a row in flight per lane: 6 waves would spill 92 bytes)
        if ((DS == 96 || DS == 128) && (CODES == 2 || CODES == 5) && BIG == 0 && METRIC != M_COS) return (DT == DT_F32 && DS == 128) ? 4 : (DT == DT_F16 && DS == 96) ? 5 : 6;
        // any dimension, 32- / 64-byte records: <= 96
        // (a floor of 6 fits the float32 kernels of 64- / 128-byte records too -- 78-80 registers, 12 bytes of scratch for L2 at 128 -- and is slower:
        // isotropic rows 2.63 -> 3.14 ms, G(48, 0.2) 1.58 -> 1.60; profiles/r06_notes.md)
        if (DS == 0 && BIG == 0 && METRIC != M_COS && (CODES <= 3 || CODES == 5)) return 5;
    }
    // windows 513 .. 1024 with 128-byte records (config 4 at recall >= 0.9): 143 VGPRs instead of 181, no scratch -> 3 waves per SIMD instead of 2
    if (DS == 768 && BIG == 2 && CODES == 4 && (QMODE == QM_SQ_IP || QMODE == QM_SQ_COS)) return 3;
    if (DS == 768 && BIG <= 1) { // config 4 and its inner-product sibling
        if (QMODE == QM_SQ_IP) return 4; // (5 before the passes were pipelined: the rows of the next pass now wait in 24 registers)
        if (QMODE == QM_SQ_COS) return SVSX_SQCOS_FLOOR;
    }
    return 0;
}

#if defined(SVSX_WAVES_PER_EU) && defined(SVSX_NUM_SGPR)
// A/B builds: residency is decided by the scalar registers (gfx950 fits 7 waves per SIMD only up to next_free_sgpr = 90, 8 up to 74;
// tools/ubench/occupancy_probe.hip), so a 7-wave build needs both caps
#define SVSX_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(SVSX_WAVES_PER_EU), amdgpu_num_sgpr(SVSX_NUM_SGPR)))
#elif defined(SVSX_WAVES_PER_EU)
#define SVSX_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(SVSX_WAVES_PER_EU)))
#else
#define SVSX_KERNEL_ATTR __attribute__((amdgpu_waves_per_eu(occupancy_floor<DT, METRIC, QMODE, DS, GENERAL, CODES, BIG>() > 0 ? occupancy_floor<DT, METRIC, QMODE, DS, GENERAL, CODES, BIG>() : 1)))
#endif

// BIG: search buffers of 129 .. 512 (BIG = 1) and 513 .. 1024 entries (BIG = 2) get their own instantiations, so that the 4- / 8- /
// 16-entries-per-lane merge does not raise the register count (= lower the occupancy) of the common window <= 128 kernels
// search_one: the search of ONE query by ONE wavefront, with `smem` the wave's LDS slice (the kernel below calls it for query blockIdx.x;
// the two-queries-per-wave experiments of round 4, experiments/r4_two_queries_per_wave/, called it for queries they handed back).
template <int DT, int METRIC, int QMODE, int DS, bool GENERAL = false, int CODES = 0, int BIG = 0>
__device__ __forceinline__ void search_one(const SearchArgs& a, const uint32_t qi, unsigned char* smem) {
    constexpr bool GREATER = METRIC != M_L2;
    // CODES: 0 none, 1 128-byte code rows (svsx_distance.hip), 2 / 3 / 4 graph-inline records of 32 / 64 / 128 bytes (svsx_codes.hip),
    // 5 (round 6) records of 28 bytes: a 16-byte plane and a 12-byte one, so that a 64-slot block is 64 x (4 + 28) = 2 048 bytes = 16 lines
    // instead of 18 -- four codes fewer per neighbour for two lines fewer per expansion, where the spectrum allows it
    constexpr bool FAT = CODES >= 2;
    constexpr int RW = !FAT ? 2 : CODES == 5 ? 2 : (2 << (CODES - 2)); // planes per record (all 16 bytes wide; the last one of the 28-byte form 12)
    constexpr int RB = CODES == 5 ? 28 : RW * 16;                 // record bytes
    constexpr int FDC = RB - (METRIC == M_L2 ? 4 : 6);            // coded coordinates per record
    constexpr bool INTPATH = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    constexpr bool B8 = !INTPATH && use_block_cyclic<DT>();  // 8-bit rows on the float path: block-cyclic mapping
#ifndef SVSX_BC8
#define SVSX_BC8 1
#endif
    constexpr bool BC8 = SVSX_BC8 != 0 && B8 && !GENERAL && use_bc8<DT, DS>(); // 8-bit rows of a long static dimension: eight lanes per row (float_row_distance_bc8)
    constexpr bool INT8L = SVSX_BC8 != 0 && INTPATH && !GENERAL && use_bc8<DT, DS>(); // the integer paths on 8-bit rows of a long static dimension: eight lanes per row too (int_row_sums8)
    constexpr int G = QMODE == QM_SCALAR ? 1 : (BC8 || INT8L) ? 8 : (INTPATH || B8) ? 4 : 16 / Elem<DT>::E;      // lanes per candidate row
    constexpr int RPP = 64 / G;                        // rows per pass

    const uint32_t lane = threadIdx.x;
    const int d = DS > 0 ? DS : static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;

    WaveBuffer buf;
    buf.e = reinterpret_cast<uint2*>(smem);
    buf.hbm = false;
    buf.size = 0;
    buf.best_unvisited = 0;
    buf.window = a.window;
    buf.capacity = a.capacity;
    const uint32_t buf_bytes = ((a.capacity + 1) * 8u + 15u) & ~15u;
    float* qf = reinterpret_cast<float*>(smem + buf_bytes);          // float path: dpad f32
    uint32_t* qw = reinterpret_cast<uint32_t*>(smem + buf_bytes);    // int path: dpad bytes
    const uint32_t q_bytes = INTPATH ? static_cast<uint32_t>(dpad) : static_cast<uint32_t>(dpad) * 4u;
    // VisitedFilter (index/vamana/filter.h:49-130, search_buffer.h:461-463) with the reference's own encoding: slot =
    // low bits of the id, 16-bit tag = the remaining high bits, 0xFFFF = empty.  The host only enables it when
    // (n >> log2(entries)) < 0xFFFF, so slot+tag identify the id exactly: no false positives -- a hit means the id was
    // evaluated earlier in this query and re-inserting it would be a no-op (duplicate, or not better than the worst,
    // which only improves).  Result-neutral by construction; it only removes vector re-fetches.
    // When (n >> log2 entries) < 255 the tag fits 8 bits (0xFF = empty) and the same LDS holds twice the entries.
    uint16_t* filt = reinterpret_cast<uint16_t*>(smem + buf_bytes + q_bytes);
    uint8_t* filt8 = reinterpret_cast<uint8_t*>(filt);
    const bool tag8 = a.filter_tag8 != 0u;
    const uint32_t filt_bytes = a.filter_entries * (tag8 ? 1u : 2u);
    uint32_t* compact = reinterpret_cast<uint32_t*>(smem + buf_bytes + q_bytes + filt_bytes); // 64 ids; inline-record kernels: one slot per adjacency slot
    const uint32_t compact_bytes = FAT ? ((a.gstride + 63u) & ~63u) * 4u : 256u;
    // FAT: the query's own record codes; 32-byte records keep them in SGPRs afterwards, so the 128 bytes alias the compaction scratch
    [[maybe_unused]] uint32_t* qcodes = reinterpret_cast<uint32_t*>(smem + buf_bytes + q_bytes + filt_bytes + (RW > 2 ? compact_bytes : 0u));
    uint2* sort_stack = reinterpret_cast<uint2*>(smem + buf_bytes + q_bytes + filt_bytes + compact_bytes + (RW > 2 ? 128u : 0u)); // GENERAL only

    // ---- BatchIterator::next prologue (iterator.h:249-262): done() check, increment_buffer, resume or restart ---------
    uint32_t* it_meta = nullptr;
    uint32_t it_bs = 0;
    bool resume = false;
    if constexpr (GENERAL) {
        if (a.it_meta != nullptr) {
            it_meta = a.it_meta + static_cast<size_t>(qi) * 8;
            it_bs = a.it_batch_sizes != nullptr ? a.it_batch_sizes[qi] : a.it_batch;
            const bool skip = it_bs == 0xFFFFFFFFu;
            const bool done = it_meta[4] == a.n || (it_meta[6] & 2u) != 0u;
            if (skip || done) { // results_.clear() (:253-256); an untouched query reports nothing either
                for (uint32_t j = lane; j < a.it_batch; j += 64) {
                    a.it_out_ids[static_cast<size_t>(qi) * a.it_batch + j] = ~0ull;
                    a.it_out_dists[static_cast<size_t>(qi) * a.it_batch + j] = __builtin_inff();
                }
                if (lane == 0) a.it_counts[qi] = 0;
                return;
            }
            if (it_bs > a.it_batch) it_bs = a.it_batch;
            buf.window = it_meta[2] + it_bs;      // SearchBufferConfig::increment (search_buffer.h:70-73)
            buf.capacity = it_meta[3] + it_bs;
            resume = (it_meta[6] & 1u) == 0u;
            if (a.it_spill != 0u) { // the buffers outgrew one wave's LDS: work directly on this query's row of the state table in HBM
                buf.e = a.it_entries + static_cast<size_t>(qi) * a.it_estride;
                buf.hbm = true;
            }
        }
    }
    if constexpr (!GENERAL) {
        if (a.cancel != nullptr && __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return; // extensions.h:575-578
    }
    // inline records: the bits of a buffer word that name the row (below the degree class, see the main loop)
    [[maybe_unused]] const uint32_t idmask = (FAT && a.fat_degshift != 0u) ? ((1u << a.fat_degshift) - 1u) : kIdMask;
    const uint32_t fmask = a.filter_entries - 1u;
    const uint32_t fshift = static_cast<uint32_t>(__builtin_ctz(a.filter_entries | 0x80000000u));
    for (uint32_t i = lane; i < filt_bytes / 4u; i += 64) reinterpret_cast<uint32_t*>(filt)[i] = 0xFFFFFFFFu;

    // ---- load + fix the query (maybe_fix_argument, greedy_search.h:140) -------------------------------------
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(qi) * a.q_row_bytes;
    const QueryFix fx = prepare_query<DT, METRIC, QMODE>(qptr, a.q_dtype, d, dpad, qf, qw, a.sq_scale, a.sq_bias, lane);

    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;

    // ---- bound codes (svsx_index_build_bound_codes): the query's own codes, kept in registers -----------------------
    // Row r has codes cx (u8) of its first DC components under the global (scale, bias): xhat_j = bias + scale * cx_j, and
    // the stored ex >= ||x_C - xhat||.  With the query coded the same way (error eq, computed here),
    //     sqrt(distance_L2) >= ||q_C - x_C|| >= ||qhat - xhat|| - eq - ex = scale * sqrt(sum (cq - cx)^2) - eq - ex,
    // and the integer sum is exact (v_dot4).  A candidate whose bound already exceeds the worst buffer entry would be
    // skipped by can_skip (search_buffer.h:338-340) whatever its exact distance -> its 384-B row is never fetched.
    // Margins: 0.99998 on the code distance, 1.001 on eq plus the absolute term code_eq_abs (round 3: the residual v - fl(bias + fl(scale c))
    // rounds at the magnitude of the OFFSET, not of eq), 1.0001 on the threshold; tests/test_bound_math.py restates them on the CPU.
    uint32_t qc[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    int code_qq = 0;
    float code_eq = 0.0f;
    uint32_t n_code_rej = 0;
#if defined(SVSX_DIAG) && SVSX_DIAG == 2
    // phase profile of an expansion (tools/diag_phases.py): shader-clock stamps at 0 loop top, 1 node known, 2 adjacency block arrived, 3 bound +
    // filter + compaction done, 4 distances of a pass known (row wait + arithmetic), 5 inserts done; stats words 1..5 = cycles between them
    uint64_t ph_last = 0;
    uint32_t ph_acc[6] = {0u, 0u, 0u, 0u, 0u, 0u};
#define SVSX_PHASE_USE(v) asm volatile("" ::"s"(v))
#define SVSX_PHASE_STAMP(i)                                          \
    do {                                                             \
        const uint64_t now_ = __builtin_readcyclecounter();          \
        ph_acc[i] += static_cast<uint32_t>(now_ - ph_last);          \
        ph_last = now_;                                              \
    } while (0)
#else
#define SVSX_PHASE_USE(v)
#define SVSX_PHASE_STAMP(i)
#endif
#ifdef SVSX_DIAG
    // diagnostic builds only (tools/build_variant.sh diag -DSVSX_DIAG=1; tools/timeline.py): stats words 4 / 5 = wall_clock64 (100 MHz) at wave start / end
    const uint32_t diag_t0 = static_cast<uint32_t>(wall_clock64());
#endif
    if constexpr (CODES == 1) {
        static_assert(G == 4 && DS > 0, "bound codes: f32 rows of static dimension");
        constexpr int DC = DS < 120 ? DS : 120;
        const float inv = 1.0f / a.code_scale;
        float e2 = 0.0f;
        int qq = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t packed = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int j = sub * 32 + w * 4 + t;
                if (j < DC) {
                    const float v = qf[j];
                    float c = __builtin_rintf((v - a.code_bias) * inv);
                    c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
                    const int ci = static_cast<int>(c);
                    const float r = v - (a.code_bias + a.code_scale * c);
                    e2 += r * r;
                    qq += ci * ci;
                    packed |= static_cast<uint32_t>(ci) << (8 * t);
                }
            }
            qc[w] = packed;
        }
        e2 += xor1f(e2);
        e2 += xor2f(e2);
        qq += xor1i(qq);
        qq += xor2i(qq);
        code_qq = __builtin_amdgcn_readfirstlane(qq); // wave-uniform values live in SGPRs
        // relative margin for the float32 sum + the explicit absolute term for the rounding of v - fl(bias + fl(scale c)) per coordinate
        // (<= 2^-23 (|v| + |bias| + 255 scale); a.code_eq_abs = sqrt(DC) 2.4e-7 (|bias| + 255 scale) 1.01, svsx_index_build_bound_codes)
        code_eq = __builtin_sqrtf(e2) * 1.001f + 1.0e-7f + a.code_eq_abs;
        if (!(code_eq < 3.0e38f)) code_eq = __builtin_inff(); // non-finite query: the bound never fires
        code_eq = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, code_eq)));
    }

    // ---- graph-inline records (svsx_codes.hip; CODES == 2): coordinates, codes and remainder norm of the query, once --------
    //   v = q - mean (mean = 0 unless L2);  u_i = P_i . v  for the DC leading directions (P = float32 copy of the orthonormal
    //   f64 directions Q the records were built with):  |u_i - Q_i . v*| <= ((d + 2) 2^-24 + delta + 2^-24) ||v||  (fma chain,
    //   rounding of P, rounding of q - mean).  Everything below carries that slack explicitly.
    uint32_t fq[7] = {0u, 0u, 0u, 0u, 0u, 0u, 0u};
    [[maybe_unused]] int fat_qq = 0;
    [[maybe_unused]] float fat_eq = 0.0f, fat_tq_hi = 0.0f, fat_tq_lo = 0.0f, fat_s2m = 0.0f, fat_nuh = 0.0f, fat_nq = 0.0f;
    [[maybe_unused]] const float fat_wunit = 1.0f / (a.sq_scale * a.sq_scale); // QM_SQ_L2: buffer distances -> code-space distances
    [[maybe_unused]] float fat_qmu = 0.0f; // similarity metrics: upper bound of q . mean (records are centred, round 6)
    [[maybe_unused]] float fat_u0s = 0.0f; // similarity metrics, mean direction as coordinate 0: (query's coordinate 0) x step
    if constexpr (FAT) {
        constexpr int DC = FDC;
        const uint32_t ci = lane & 31u, half = lane >> 5; // lane -> coordinate ci (+ 32 per group), half of the element range
        const int hlen = dpad >> 1;                       // dpad is a multiple of 16 -> 8-float steps
        const float* qh = qf + half * hlen;
        const float* mh = a.pca_mean + half * hlen;
        // eight consecutive query elements as float32.  The integer paths (int8 x int8 / uint8 x uint8 rows, and the scalar-quantized L2 distance,
        // which is taken between CODED query and coded rows, scalar.h:58-105) keep the query as bytes: the records of such an index are built over
        // float(row element) -- code space for scalar-quantized L2 -- so the query's image is float(query byte), exactly
        auto q8 = [&](const int j0, float4& v0, float4& v1) {
            if constexpr (INTPATH) {
                const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(qw) + half * hlen + j0);
                auto el = [](uint32_t word, int t) -> float {
                    if constexpr (DT == DT_I8) return static_cast<float>(static_cast<int>(static_cast<int8_t>((word >> (8 * t)) & 0xFFu)));
                    else return static_cast<float>((word >> (8 * t)) & 0xFFu);
                };
                v0 = make_float4(el(w.x, 0), el(w.x, 1), el(w.x, 2), el(w.x, 3));
                v1 = make_float4(el(w.y, 0), el(w.y, 1), el(w.y, 2), el(w.y, 3));
            } else {
                v0 = *reinterpret_cast<const float4*>(qh + j0);
                v1 = *reinterpret_cast<const float4*>(qh + j0 + 4);
            }
        };
        float r2 = 0.0f, un2 = 0.0f, vv = 0.0f;
        [[maybe_unused]] float qm = 0.0f; // similarity metrics: q . mean (the records are centred: q . x = q . (x - mean) + q . mean)
        [[maybe_unused]] float u0 = 0.0f; // similarity metrics with the mean direction as coordinate 0: the query's coordinate along it (lane 0 of the first sweep)
        int qq = 0;
        if (lane < 32u) qcodes[lane] = 0u;
        lds_order();
#pragma unroll 1
        for (int g = 0; g * 32 < DC; ++g) { // 32 coordinates per sweep over the query
            const float4* prow = reinterpret_cast<const float4*>(a.pca_t + static_cast<size_t>(ci + 32u * g) * a.pca_pd + half * hlen);
            float up = 0.0f, vvp = 0.0f;
            [[maybe_unused]] float qmp = 0.0f;
#pragma unroll 1
            for (int j0 = 0; j0 < hlen; j0 += 8) { // rolled: the projection must not raise the kernel's register peak
                const float4 p0 = prow[j0 >> 2], p1 = prow[(j0 >> 2) + 1];
                float4 v0, v1;
                q8(j0, v0, v1);
                const float4 m0 = *reinterpret_cast<const float4*>(mh + j0), m1 = *reinterpret_cast<const float4*>(mh + j0 + 4);
                if constexpr (METRIC == M_L2) {
                    v0.x -= m0.x; v0.y -= m0.y; v0.z -= m0.z; v0.w -= m0.w;
                    v1.x -= m1.x; v1.y -= m1.y; v1.z -= m1.z; v1.w -= m1.w;
                } else { // the query itself is not centred; its product with the centre is a constant of the query
                    qmp = __builtin_fmaf(v0.x, m0.x, qmp); qmp = __builtin_fmaf(v0.y, m0.y, qmp); qmp = __builtin_fmaf(v0.z, m0.z, qmp); qmp = __builtin_fmaf(v0.w, m0.w, qmp);
                    qmp = __builtin_fmaf(v1.x, m1.x, qmp); qmp = __builtin_fmaf(v1.y, m1.y, qmp); qmp = __builtin_fmaf(v1.z, m1.z, qmp); qmp = __builtin_fmaf(v1.w, m1.w, qmp);
                }
                up = __builtin_fmaf(p0.x, v0.x, up); up = __builtin_fmaf(p0.y, v0.y, up); up = __builtin_fmaf(p0.z, v0.z, up); up = __builtin_fmaf(p0.w, v0.w, up);
                up = __builtin_fmaf(p1.x, v1.x, up); up = __builtin_fmaf(p1.y, v1.y, up); up = __builtin_fmaf(p1.z, v1.z, up); up = __builtin_fmaf(p1.w, v1.w, up);
                vvp = __builtin_fmaf(v0.x, v0.x, vvp); vvp = __builtin_fmaf(v0.y, v0.y, vvp); vvp = __builtin_fmaf(v0.z, v0.z, vvp); vvp = __builtin_fmaf(v0.w, v0.w, vvp);
                vvp = __builtin_fmaf(v1.x, v1.x, vvp); vvp = __builtin_fmaf(v1.y, v1.y, vvp); vvp = __builtin_fmaf(v1.z, v1.z, vvp); vvp = __builtin_fmaf(v1.w, v1.w, vvp);
            }
            float u = up + __shfl_xor(up, 32);     // coordinate ci + 32 g (both halves of the wave hold it)
            vv = vvp + __shfl_xor(vvp, 32);        // ||v||^2 (the same in every sweep)
            if constexpr (METRIC != M_L2) qm = qmp + __shfl_xor(qmp, 32);
            if constexpr (METRIC == M_COS) {
                // cosine records hold UNIT vectors: work with the unit query q / ||q|| (two more roundings per coordinate, in `urel`)
                u *= 1.0f / __builtin_sqrtf(vv);
            }
            const uint32_t coord = ci + 32u * static_cast<uint32_t>(g);
            if (half == 0u && coord < static_cast<uint32_t>(DC)) {
                float c, uh;
                if constexpr (METRIC == M_L2) {
                    const float b = a.pca_bias[coord];
                    c = __builtin_rintf((u - b) / a.pca_step);
                    c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
                    if (!(c == c)) c = 0.0f;
                    uh = b + a.pca_step * c;
                } else if (coord == 0u && a.pca_mean_dir != 0u) {
                    // the unit mean direction (svsx_records.hip): the query's coordinate along it is kept in float32, not coded -- it lies far outside
                    // the code range of the centred rows.  No code (the integer dot product skips it), no coding residual.
                    c = 0.0f;
                    uh = u;
                    u0 = u;
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
                reinterpret_cast<uint8_t*>(qcodes)[coord] = static_cast<uint8_t>(cint & 0xFF);
            }
        }
        [[maybe_unused]] const float vv_raw = vv; // ||q||^2 as the sweeps accumulated it
        if constexpr (METRIC == M_COS) vv = vv > 0.0f ? 1.0f : vv; // a zero query gives NaN coordinates: the bound never fires
        for (int off = 32; off >= 1; off >>= 1) {
            r2 += __shfl_xor(r2, off);
            un2 += __shfl_xor(un2, off);
            qq += __shfl_xor(qq, off);
        }
        lds_order();
        if constexpr (RW == 2) { // 32-byte records: the query's code words live in SGPRs
#pragma unroll
            for (int w = 0; w < 7; ++w) fq[w] = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(qcodes[w])));
        }
        const float vn = __builtin_sqrtf(vv) * 1.00001f;
        const float urel = (static_cast<float>(d + 3 + (METRIC == M_COS ? d / 2 + 6 : 0)) * 5.9604645e-8f) * 1.01f + a.pca_delta; // cosine: + rounding of ||q|| and of the scaling
        const float erot = __builtin_sqrtf(static_cast<float>(DC)) * urel * vn;
        const float nu = __builtin_sqrtf(un2);
        float eq = __builtin_sqrtf(r2) * 1.0001f + erot + a.pca_eq_abs;
        // remainder norm tq = sqrt(||v||^2 - ||u||^2): a difference of float32 sums, bracketed from both sides
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
        auto uni = [](float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); };
        fat_qq = __builtin_amdgcn_readfirstlane(qq);
        fat_eq = uni(eq);
        fat_tq_hi = uni(tq_hi);
        fat_tq_lo = uni(tq_lo);
        fat_s2m = a.pca_step * a.pca_step * (METRIC == M_L2 ? 0.99999f : 1.0f);
        fat_nuh = uni(a.pca_step * __builtin_sqrtf(static_cast<float>(qq)) * 1.00001f); // ||uhat|| (signed codes)
        if constexpr (METRIC != M_L2) {
            if (a.pca_mean_dir != 0u) { // uhat = (u0, s c_1, s c_2, ...): the float32 coordinate joins the norm; u0 * s multiplies the row's code 0
                const float u0u = uni(u0); // lane 0 holds it (coordinate ci = 0 of sweep 0, half 0)
                fat_nuh = uni(__builtin_sqrtf(fat_nuh * fat_nuh + u0u * u0u) * 1.00001f);
                fat_u0s = uni(u0u * a.pca_step);
            }
        }
        fat_nq = uni(vn);
        if constexpr (METRIC != M_L2) {
            // an UPPER bound of q . mean (cosine: of (q / ||q||) . mean).  The two half-range fma chains and their sum carry at most
            // (d / 2 + 2) 2^-24 ||q|| ||mean|| of rounding; the unit-query scaling of cosine two more roundings, and `vn` is ||q|| (cosine: 1)
            // rounded up.  A non-finite query gives NaN here and the bound never fires.
            float qmu = qm;
            if constexpr (METRIC == M_COS) qmu *= 1.0f / __builtin_sqrtf(vv_raw);
            const float qslack = (static_cast<float>(d / 2 + 6) * 5.9604645e-8f * 1.01f) * vn * a.pca_mean_norm;
            fat_qmu = uni(qmu + qslack + __builtin_fabsf(qmu) * 1.2e-7f);
        }
    }

    // f32 L2 rows of 65..128 dimensions: stop after the first two cache lines when the partial sum already rules the
    // candidate out (l2_row_distance_two_phase); result-neutral, removes the third line of most rows at d = 96
    constexpr bool TWO_PHASE = QMODE == QM_FLOAT && METRIC == M_L2 && DT == DT_F32 && DS > 64 && DS <= 128 && (DS & 15) == 0;
    uint32_t n_early = 0;
    // distance of the candidate whose id each lane-group holds; every lane of the wave participates.
    // `thr`: +inf while the buffer is not full, else worst * (1 + 4e-5).
    auto group_distance = [&](uint32_t cid, bool active, float thr, auto&& hook) -> float {
        const unsigned char* row = data + static_cast<size_t>(active ? cid : 0u) * static_cast<uint32_t>(a.row_bytes); // one v_mad_u64_u32: a row is shorter than 4 GB
        float fmain = 0.0f, fnrm = 0.0f;
        int ab = 0, bb = 0;
        if constexpr (TWO_PHASE && CODES == 0) { // survivors of the bound codes are close to the worst entry: fetch whole rows
            bool rejected = false;
            const float dd = l2_row_distance_two_phase<DT, (DS > 64 && DS <= 128 && (DS & 15) == 0) ? DS : 96>(row, active, qf, sub, thr, rejected);
            n_early += __popcll(wave_ballot(active && rejected && sub == 0));
            return dd;
        } else if constexpr (QMODE == QM_FLOAT && METRIC == M_L2 && DT == DT_F32 && (DS == 0 || (CODES == 0 && DS > 64))) {
            // runtime dimension, and the plain static kernels of more than one 64-block that the two-phase form does not cover (100, 160, 200,
            // 512, 768: `d` is a compile-time constant there): progressive partial-distance rejection after every 64-element block.  Without
            // it the static forms were 8-13 % slower than round 2's runtime-d kernel on the same graphs (profiles/r03_notes.md section 8)
            bool rejected = false;
            float_row_distance<DT, METRIC, 0, false, true>(row, active, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm, thr, &rejected);
            n_early += __popcll(wave_ballot(active && rejected && sub == 0));
            return fmain;
        } else if constexpr (QMODE == QM_SCALAR) {
            scalar_row_distance<DT, METRIC>(row, qf, d, fmain, fnrm);
        } else if constexpr (INT8L) {
            int_row_sums8<DT, DS>(bc8_load<DS>(row, sub), qw, sub, ab, bb);
        } else if constexpr (INTPATH) {
            int_row_sums<DT>(row, active, qw, d, sub, ab, bb);
        } else if constexpr (B8 && QMODE == QM_SQ_COS) {
            // CosineSimilarityCompressed (scalar.h:155-195): the row's own norm does not depend on the query -- svsx_index_create ran the same
            // accumulator chain once per row and stored its square root (sq_row_norm_kernel), so only the dot product is accumulated here:
            // a sixth fewer vector instructions per element and no correctly rounded sqrt per candidate; same bits.
            if constexpr (GENERAL) {
                // indexes whose row norms were dropped (svsx_index_drop_row_norms) are served by this variant only: the norm chain runs per
                // candidate, as the reference does it (scalar.h:175-184)
                if (a.sq_row_norm == nullptr) {
                    float_row_distance_bc<DT, M_COS, DS, true>(row, active, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
                    return finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
                }
            }
            const float bn = a.sq_row_norm[active ? cid : 0u]; // requested with the row
            if constexpr (BC8) float_row_distance_bc8<DT, M_IP, DS, true>(row, qf, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
            else float_row_distance_bc<DT, M_IP, DS, true>(row, active, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
            return fmain / (bn * fx.norm); // cosine.h:334-335
        } else if constexpr (BC8) {
            float_row_distance_bc8<DT, METRIC, DS, QMODE == QM_SQ_COS>(row, qf, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        } else if constexpr (B8) {
            float_row_distance_bc<DT, METRIC, DS, QMODE == QM_SQ_COS>(row, active, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        } else {
            float_row_distance<DT, METRIC, DS, QMODE == QM_SQ_COS>(row, active, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm, 0.0f, nullptr, hook);
        }
        return finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
    };

    uint32_t n_exp = 0, n_slots = 0, n_eval = 0, n_ins = 0;
    [[maybe_unused]] bool nan_seen = false; // a NaN distance reached the buffer: comparisons stop being an order, inserts go one by one

    // ---- EntryPointInitializer: distances to the entry points, push_back, sort --------------------------------
    // (push_back + sort == ordered insert for <= 16 entry points, where std::sort is a stable insertion sort; the
    //  GENERAL variant appends and runs the std::sort restatement, exact for any number of entry points)
    const bool append_then_sort = GENERAL && a.n_entry > 16u;
    if (!resume) {
        for (uint32_t base = 0; base < a.n_entry; base += RPP) {
            const uint32_t ci = base + grp;
            const bool act = ci < a.n_entry;
            const uint32_t cid = act ? a.entry[ci] : 0u;
            const float dist = group_distance(cid, act, __builtin_inff(), NoHook{});
            unsigned long long m = wave_ballot(act && sub == 0);
            while (m != 0ull) {
                const int bl = __builtin_ctzll(m);
                m &= m - 1ull;
                const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), bl));
                const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cid), bl));
                ++n_eval;
                nan_seen = nan_seen || dd != dd;
                if (buf.size < buf.capacity) {
                    // EntryPointInitializer never touches the visited set (greedy_search.h:69-90); marking an entry point that was
                    // really pushed is result-neutral (re-encountering it would be a duplicate), one that push_back dropped is not
                    if (a.filter_entries != 0u && lane == 0) {
                        if (tag8) filt8[id & fmask] = static_cast<uint8_t>(id >> fshift);
                        else filt[id & fmask] = static_cast<uint16_t>(id >> fshift);
                    }
                    // push_back + (stable) sort == ordered insert without the duplicate check
                    uint32_t pos = append_then_sort ? buf.size : 0u;
                    for (uint32_t i0 = 0; !append_then_sort && i0 < buf.size; i0 += 64) {
                        const uint32_t i = i0 + lane;
                        bool before = false;
                        if (i < buf.size) before = !cmpf<GREATER>(dd, __uint_as_float(buf.e[i].x));
                        pos += __popcll(wave_ballot(before));
                    }
                    buf_order(buf);
                    for (int i = static_cast<int>(buf.size) - 1; i >= static_cast<int>(pos); --i) {
                        if (lane == 0) buf.e[i + 1] = buf.e[i];
                    }
                    if (lane == 0) buf.e[pos] = make_uint2(__float_as_uint(dd), id);
                    buf_order(buf);
                    buf.size++;
                }
            }
        }
    }

    if constexpr (GENERAL) {
        if (resume) { // RestartInitializer (iterator.h:36-63): keep the previous contents, sort()
            buf.size = it_meta[0];
            buf.best_unvisited = it_meta[1];
            const uint2* src = a.it_entries + static_cast<size_t>(qi) * a.it_estride;
            for (uint32_t i = lane; i < buf.size; i += 64) {
                const uint2 en = src[i];
                if (!buf.hbm) buf.e[i] = en;
                nan_seen = nan_seen || __uint_as_float(en.x) != __uint_as_float(en.x);
            }
            nan_seen = wave_ballot(nan_seen) != 0ull; // a NaN left in the buffer by an earlier call
            buf_order(buf);
            wave_std_sort<GREATER>(buf, sort_stack, lane, true);
        } else if (append_then_sort) {
            wave_std_sort<GREATER>(buf, sort_stack, lane, false);
        }
    }

    bool full = buf.size == buf.capacity;
    float worst = (full && buf.capacity > 0u) ? __uint_as_float(buf.e[buf.capacity - 1].x) : 0.0f;

    // ---- main loop (greedy_search.h:153-202) --------------------------------------------------------------------
    int cancel_seen = 0;
#if defined(SVSX_DIAG) && SVSX_DIAG == 3
    const uint32_t diag_tloop = static_cast<uint32_t>(wall_clock64()); // tools/diag_prologue.py: where a query's time goes outside the expansions
#endif
    while (!wave_done(buf)) {
        if (a.cancel != nullptr) { // greedy_search.h:155: `if (cancel()) return;` -- the word read here is consumed one expansion later
            if (cancel_seen != 0) break;
            cancel_seen = __hip_atomic_load(a.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SVSX_PHASE_STAMP(0);
        float node_dist;
        const uint32_t node_word = wave_next(buf, lane, node_dist);
        // Inline-record blocks of at most 2^26 nodes carry, above each neighbour id, ceil(degree of that neighbour / 8) (svsx_codes.hip
        // fat_build_kernel): the id travels through the candidate buffer with it, so when the node is expanded the number of 8-slot groups
        // of its block that hold a neighbour is known BEFORE the block is requested, and the lines behind them are never touched
        // (0 = unknown: entry points; the whole block is read).  Mean degree 62 of 64 slots still leaves 14 % of the block lines empty.
        const uint32_t node = FAT ? (node_word & idmask) : node_word;
        [[maybe_unused]] uint32_t slot_limit = a.gstride;
        if constexpr (FAT) {
            const uint32_t cls = a.fat_degshift != 0u ? (node_word >> a.fat_degshift) : 0u; // bit 31 (visited) is already off
            if (cls != 0u && cls * 8u < slot_limit) slot_limit = cls * 8u;
        }
        SVSX_PHASE_USE(node);
        SVSX_PHASE_STAMP(1);
        if (a.history_ids != nullptr && lane == 0 && n_exp < a.history_cap) {
            a.history_ids[static_cast<size_t>(qi) * a.history_cap + n_exp] = node;
            if (a.history_dists != nullptr) a.history_dists[static_cast<size_t>(qi) * a.history_cap + n_exp] = node_dist;
        }
        ++n_exp;
        const uint32_t* grow = a.graph + static_cast<size_t>(node) * a.gstride;
        [[maybe_unused]] uint32_t fat_total = 0; // inline records: survivors of the chunks so far (they share compact[])
        // One loop for both kernel families: a chunk of the adjacency row (load, bound, filter, compaction), then the distance passes.
        // Inline-record kernels go round the chunk part until the row is exhausted and run their passes ONCE, over compact[0 .. cnt) = the
        // survivors of all chunks in adjacency order.  (A two-chunk row -- R = 128 -- used to run its passes chunk by chunk: two half-empty
        // passes where one would do.  The bound of the second chunk is now taken against a `worst` that the first chunk's inserts have not
        // improved yet, which only makes it reject less.)  Other kernels: passes after every chunk, lane c of `nid` holds candidate c.
        // (Written as one loop, not as a lambda called from two places: with the buffer state captured by reference the uniformity analysis
        // gave up on it and the expansion turned into exec-masked vector code, +30 % vector instructions.)
        uint32_t gbase = 0;
        const uint32_t row_slots = FAT ? slot_limit : a.gstride;
        // one contiguous block per chunk of <= 64 slots: [ids | plane 0 | ... | plane RW-1]; lane c owns neighbour c and its record.  Lanes beyond
        // the node's last group of 8 slots repeat that group's last lane (same lines: no request of their own, no exec mask, no zero fill)
        // and are turned into empty slots.  (A lambda that only reads what it captures: see the note on lambdas above.)
        [[maybe_unused]] uint32_t nid_next = kInvalidId;
        [[maybe_unused]] uint4 rec_next[RW];
        [[maybe_unused]] auto fat_load = [&](const uint32_t base, uint32_t& out_id, uint4 (&out_rec)[RW]) {
            const uint32_t cs = (a.gstride - base) < 64u ? (a.gstride - base) : 64u;
            const uint32_t lim = (slot_limit - base) < cs ? (slot_limit - base) : cs; // >= 1
            const uint32_t* fids = reinterpret_cast<const uint32_t*>(a.fat + static_cast<size_t>(node) * a.fat_stride + static_cast<size_t>(base) * (4u + static_cast<uint32_t>(RB)));
            const uint4* planes = reinterpret_cast<const uint4*>(fids + cs);
            const uint32_t ll = opaque_u(lane); // lane-derived offsets are recomputed here, not kept across the distance pass (opaque_u)
            const uint32_t li = ll < lim ? ll : lim - 1u;
            const uint32_t w = fids[li];
#pragma unroll
            for (int p = 0; p < RW; ++p) {
                if constexpr (CODES == 5) {
                    if (p == RW - 1) { // the 12-byte plane: three words per slot behind the full planes
                        const uint32_t* p12 = reinterpret_cast<const uint32_t*>(planes + static_cast<uint32_t>(RW - 1) * cs) + 3u * li;
                        const uint3 t = *reinterpret_cast<const uint3*>(p12);
                        out_rec[p] = make_uint4(t.x, t.y, t.z, 0u);
                        continue;
                    }
                }
                out_rec[p] = planes[static_cast<uint32_t>(p) * cs + li];
            }
            out_id = ll < lim ? w : kInvalidId;
        };
        // (8-bit rows only: the float32 / float16 kernels of d = 96 / 128 sit at their 80-register bracket and would spill the nine registers)
        constexpr bool NEXT_CHUNK = FAT && (DT == DT_I8 || DT == DT_U8) && RW <= 4; // (128-byte records: 33 registers, spills at the 5-wave floor)
        if constexpr (NEXT_CHUNK) {
            if (row_slots != 0u) fat_load(0u, nid_next, rec_next);
        }
        while (gbase < row_slots) {
            uint32_t cnt;
            bool last_chunk;
            uint32_t nid = kInvalidId;
            [[maybe_unused]] uint4 rec[RW];
            if constexpr (FAT) {
                // this chunk's block was requested one trip earlier (or before the loop); the next chunk's -- the degree class says whether
                // the row has one -- is requested now, so a two-chunk row (R = 128) pays one memory round trip for its blocks, not two
                if constexpr (NEXT_CHUNK) {
                    nid = nid_next;
#pragma unroll
                    for (int p = 0; p < RW; ++p) rec[p] = rec_next[p];
                    if (gbase + 64u < row_slots) fat_load(gbase + 64u, nid_next, rec_next);
                } else {
                    fat_load(gbase, nid, rec);
                }
            } else {
                if (gbase + lane < a.gstride) nid = grow[gbase + lane];
            }
            const unsigned long long valid_m = wave_ballot(nid != kInvalidId);
            const uint32_t deg_here = __popcll(valid_m); // ids are packed from slot 0
            SVSX_PHASE_USE(deg_here);
            SVSX_PHASE_STAMP(2);
            n_slots += deg_here;
            cnt = deg_here;
            if constexpr (FAT) {
                bool keep = nid != kInvalidId;
                // visited filter (greedy_search.h:191-193), here holding only the ids whose rows were really fetched: a candidate the
                // bound rejected is rejected again by the bound (worst only improves) at no memory cost, so it need not be
                // remembered -- the same table then covers the whole query almost without evictions.  A hit = this id went
                // through the exact path earlier in this query, re-inserting it would be a no-op.
                const uint32_t fh = nid & fmask, ftag = (nid & idmask) >> fshift; // 16-bit tags only (the host never asks for 8-bit tags with inline records); the hash bits lie below the degree class
                const bool use_filter = a.filter_entries != 0u && !nan_seen; // with a NaN in the buffer a repeated insert is no longer a no-op
                // every lane reads its slot (an empty adjacency slot hashes to the table's last entry): no exec-masked branch in front of the
                // bound, whose arithmetic then runs under the LDS round trip instead of after it
                // (also when the filter is off: the table then has no entries and slot 0 aliases the compaction scratch -- the value is ignored)
                unsigned long long km;
                const uint32_t fseen = static_cast<uint32_t>(filt[a.filter_entries != 0u ? fh : 0u]);
                __builtin_amdgcn_sched_barrier(0); // the read stays in front of the bound's arithmetic (the scheduler otherwise sinks it to its use)
                {
                    // bound of every candidate of this chunk, lane-parallel (svsx_codes.hip has the geometry).  `worst` can only
                    // improve while this expansion's survivors are inserted, so a bound taken against it stays valid.  Evaluated
                    // whether or not the buffer is full (`rej` counts only when it is): without a branch the filter's LDS read above and
                    // the second record plane are awaited under this arithmetic, not in front of it.
                    bool rej;
                    unsigned long long rej_m; // wave mask of the ONE compare that decides (see wave_ballot: masks are combined in scalar registers)
                    constexpr int NW = RB / 4;  // words per record; the last one holds float16 fields
                    auto recw = [&](int w) -> uint32_t {
                        const uint4& r = rec[w >> 2];
                        return (w & 3) == 0 ? r.x : (w & 3) == 1 ? r.y : (w & 3) == 2 ? r.z : r.w;
                    };
                    uint32_t qw[NW];
                    if constexpr (RW == 2) { // (28-byte records: 24 / 22 codes, the bytes behind them are zero in qcodes and so in fq)
#pragma unroll
                        for (int w = 0; w < 7; ++w) qw[w] = fq[w];
                        if constexpr (NW == 8) qw[7] = 0u;
                    } else { // larger records: the query's code words come from LDS (broadcast reads)
#pragma unroll
                        for (int p = 0; p < RW; ++p) {
                            const uint4 q4 = reinterpret_cast<const uint4*>(qcodes)[p];
                            qw[4 * p] = q4.x; qw[4 * p + 1] = q4.y; qw[4 * p + 2] = q4.z; qw[4 * p + 3] = q4.w;
                        }
                    }
                    const uint32_t last = recw(NW - 1), prev = recw(NW - 2);
                    if constexpr (METRIC == M_L2) {
                        // A = max(0, s sqrt(I) - e), e = eq + ex;  B = max(0, |tq - tx| - slack):  dist >= A^2 + B^2.  Reject iff
                        // A^2 + B^2 > W' := worst * 1.0001, implied by  s^2 I > W' - B^2 + e (2 sqrt(W') + e)  (one square root per hop)
                        // (scalar-quantized L2: the buffer holds scale^2 * I, scalar.h:88-94, the records live in code space: worst / fl(scale^2), two
                        // more roundings under the 1e-4 margin; a scale whose square under- or overflows gives inf or NaN and the bound never fires)
                        const float wp = (QMODE == QM_SQ_L2 ? worst * fat_wunit : worst) * 1.0001f;
                        const float r2 = 2.0f * __builtin_amdgcn_sqrtf(wp) * 1.00001f;
                        uint32_t dqx = 0u, dxx = 0u;
#pragma unroll
                        for (int w = 0; w < NW - 1; ++w) {
                            dqx = __builtin_amdgcn_udot4(recw(w), qw[w], dqx, false);
                            dxx = __builtin_amdgcn_udot4(recw(w), recw(w), dxx, false);
                        }
                        const float fi = static_cast<float>(fat_qq + static_cast<int>(dxx) - 2 * static_cast<int>(dqx)); // sum (cq - cx)^2 < 2^24: exact
                        const float ex = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last & 0xFFFFu))) * a.pca_step;
                        const float tx = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last >> 16))) * a.pca_tail_unit;
                        const float e = fat_eq + ex;
                        const float txe = tx * 4.94e-4f + a.pca_tail_unit * 6.0e-8f; // tx is a nearest float16: 2^-11 relative
                        const float t1 = fat_tq_lo - (tx + txe), t2 = (tx - txe) - fat_tq_hi;
                        const float t = t1 > t2 ? t1 : t2;
                        const float bb = (t > 0.0f ? t : 0.0f) * 0.99999f;
                        const float rhs = wp - bb * bb + e * (r2 + e) * 1.00001f;
                        rej = fi * fat_s2m > rhs; // false for NaN / inf thresholds
                        rej_m = wave_ballot(fi * fat_s2m > rhs);
                    } else {
                        // q . z <= s^2 I + ||uhat|| ex + eq (ny + ex) + tq tx,  I = sum cq cx (exact), z = x - mean;  all record fields are upper bounds
                        int dot = 0;
#pragma unroll
                        for (int w = 0; w < NW - 1; ++w) // word NW-2: codes in its low half, a float16 above -- the query's bytes there are 0
                            dot = __builtin_amdgcn_sdot4(static_cast<int>(recw(w)), static_cast<int>(qw[w]), dot, false);
                        float main = fat_s2m * static_cast<float>(dot);
                        if (a.pca_mean_dir != 0u) // coordinate 0 = the mean direction: the query's side in float32, the row's side its code 0 (the query's code byte 0 is 0)
                            main = __builtin_fmaf(fat_u0s, static_cast<float>(static_cast<int>(static_cast<int8_t>(recw(0) & 0xFFu))), main);
                        const float ex = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(prev >> 16))) * a.pca_step;
                        const float ny = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last & 0xFFFFu))) * a.pca_step;
                        const float tx = static_cast<float>(__builtin_bit_cast(_Float16, static_cast<uint16_t>(last >> 16))) * a.pca_tail_unit;
                        const float pos = fat_nuh * ex + fat_eq * (ny + ex) + fat_tq_hi * tx;
                        float ub = main + pos + (__builtin_fabsf(main) + pos) * 4.0e-6f; // >= q . (x - mean)  (cosine: of the unit vectors)
                        ub = ub + fat_qmu + (__builtin_fabsf(ub) + __builtin_fabsf(fat_qmu)) * 1.2e-7f; // + q . mean: >= q . x
                        if constexpr (METRIC == M_COS) {
                            // records hold the UNIT vector and the query was scaled to unit norm: `ub` bounds the real cosine; the
                            // reference's float32 result is within (d + 8) 2^-24 of it (|cos| <= 1)
                            ub += static_cast<float>(d + 8) * 6.3e-8f + 1.0e-7f;
                        } else {
                            // the reference's own float32 accumulation: |computed - real| <= (d + 2) 2^-24 ||q|| ||z||
                            ub += static_cast<float>(d + 2) * 6.1e-8f * fat_nq * (ny + ex + tx + a.pca_ref_extra);
                        }
                        rej = ub < worst; // false whenever anything is NaN / inf
                        rej_m = wave_ballot(ub < worst);
                    }
                    rej = rej && full;
                    rej_m = full ? rej_m : 0ull;
                    keep = keep && !(use_filter && fseen == ftag);
                    const unsigned long long live_m = valid_m & ~(use_filter ? wave_ballot(fseen == ftag) : 0ull); // what the reference would evaluate
                    if (a.stats != nullptr) { // counters only
                        const uint32_t nrej = __popcll(live_m & rej_m);
                        n_code_rej += nrej;
                        n_eval += nrej;
                    }
                    keep = keep && !rej;
                    km = live_m & ~rej_m; // == the wave mask of `keep`
                }
                if (use_filter && keep) filt[fh] = static_cast<uint16_t>(ftag);
                cnt = __popcll(km);
                const uint32_t rank = prefix_count(km);
                lds_order();
                if (keep) compact[fat_total + rank] = nid;
                lds_order();
                // the passes below read compact[pass base + group] themselves: one LDS round trip per pass instead of a read-back here
                // plus a cross-lane permute per pass
            } else if (a.filter_entries != 0u && !nan_seen) { // with a NaN in the buffer a repeated insert is no longer a no-op: filter off
                // emplace_visited (greedy_search.h:191-193): drop ids already evaluated, keep adjacency order
                bool keep = nid != kInvalidId;
                if (keep) {
                    const uint32_t h = nid & fmask;
                    const uint32_t tag = nid >> fshift;
                    if (tag8) {
                        keep = static_cast<uint32_t>(filt8[h]) != tag;
                        filt8[h] = static_cast<uint8_t>(tag);
                    } else {
                        keep = static_cast<uint32_t>(filt[h]) != tag;
                        filt[h] = static_cast<uint16_t>(tag);
                    }
                }
                const unsigned long long km = wave_ballot(keep);
                cnt = __popcll(km);
                const uint32_t rank = prefix_count(km);
                lds_order();
                if (keep) compact[rank] = nid;
                lds_order();
                nid = lane < cnt ? compact[lane] : kInvalidId;
                lds_order();
            }
            if constexpr (CODES == 1) {
                if (full && cnt > 0u) {
                    // bound pass over all candidates of this expansion; survivors keep their adjacency order.  `worst` can
                    // only improve while they are inserted, so a bound taken against the current worst stays valid.
                    // reject iff  scale * sqrt(I) * 0.99998 - eq - ex > sqrt(worst * 1.0001),  I = sum (cq - cx)^2; with
                    // T = (sqrt(worst * 1.0001) * 1.00001 + eq + ex) / (0.99998 * scale) this is implied by I > T^2 * 1.00001
                    // (one square root per expansion instead of one per candidate)
                    const float sthr = __builtin_sqrtf(worst * 1.0001f) * 1.00001f + code_eq;
                    const float tscale = 1.00002f / a.code_scale * 1.000001f;
                    uint32_t nsurv = 0;
                    // the code rows of CB passes (16 candidates each) are requested together before the first is used:
                    // fewer dependent memory round trips per expansion
                    constexpr int CB = SVSX_CODE_BATCH;
                    for (int pg = 0; pg < 4 && static_cast<uint32_t>(pg) * RPP < cnt; pg += CB) {
                    uint4 c0[CB], c1[CB];
                    uint32_t pcid[CB];
#pragma unroll
                    for (int pp = 0; pp < CB; ++pp) {
                        const int p = pp;
                        const uint32_t ci = static_cast<uint32_t>(pg + pp) * RPP + grp;
                        pcid[p] = static_cast<uint32_t>(__shfl(static_cast<int>(nid), static_cast<int>(ci)));
                        if (static_cast<uint32_t>(pg + pp) * RPP < cnt) {
                            const uint4* crow = reinterpret_cast<const uint4*>(a.codes + static_cast<size_t>(ci < cnt ? pcid[p] : 0u) * 128u) + sub * 2;
                            c0[p] = crow[0];
                            c1[p] = crow[1];
                        }
                    }
#pragma unroll
                    for (int pp = 0; pp < CB; ++pp) {
                        const int p = pp;
                        if (static_cast<uint32_t>(pg + pp) * RPP >= cnt) break;
                        const uint32_t ci = static_cast<uint32_t>(pg + pp) * RPP + grp;
                        const uint32_t cid = pcid[p];
                        const bool act = ci < cnt;
                        uint32_t dot = 0;
                        dot = __builtin_amdgcn_udot4(c0[p].x, qc[0], dot, false);
                        dot = __builtin_amdgcn_udot4(c0[p].y, qc[1], dot, false);
                        dot = __builtin_amdgcn_udot4(c0[p].z, qc[2], dot, false);
                        dot = __builtin_amdgcn_udot4(c0[p].w, qc[3], dot, false);
                        dot = __builtin_amdgcn_udot4(c1[p].x, qc[4], dot, false);
                        dot = __builtin_amdgcn_udot4(c1[p].y, qc[5], dot, false);
                        dot = __builtin_amdgcn_udot4(c1[p].z, qc[6], dot, false); // qc[6], qc[7] are 0 in the lane that holds the row tail
                        dot = __builtin_amdgcn_udot4(c1[p].w, qc[7], dot, false);
                        int part = -2 * static_cast<int>(dot);
                        float ex = 0.0f;
                        if (sub == 3) {
                            ex = __uint_as_float(c1[p].z);
                            part += static_cast<int>(c1[p].w);
                        }
                        part += xor1i(part);
                        part += xor2i(part);
                        ex += xor1f(ex);
                        ex += xor2f(ex);
                        const float isum = static_cast<float>(code_qq + part); // sum (cq - cx)^2 < 2^24: exact
                        const float t = (sthr + ex) * tscale;
                        const bool rej = isum > t * t * 1.00001f; // false for NaN / inf thresholds
                        const bool surv = act && !rej && sub == 0;
                        const unsigned long long sm = wave_ballot(surv);
                        if (surv) compact[nsurv + prefix_count(sm)] = cid;
                        nsurv += __popcll(sm);
                    }
                    }
                    lds_order();
                    nid = lane < nsurv ? compact[lane] : kInvalidId;
                    lds_order();
                    n_code_rej += cnt - nsurv;
                    n_eval += cnt - nsurv;
                    cnt = nsurv;
                }
            }
            last_chunk = deg_here < 64u || gbase + 64u >= row_slots;
            gbase += 64u;
            if constexpr (FAT) {
                fat_total += cnt;
                if (!last_chunk) continue;
                cnt = fat_total;
            }
            SVSX_PHASE_USE(cnt);
            SVSX_PHASE_STAMP(3);
            // Eight-lane 8-bit rows with inline records (config 4): the rows of pass p + 1 are requested before pass p is computed -- the list of
            // candidates is fixed for the expansion (compact[]), so the requests do not depend on the inserts in between -- and a pass waits
            // for memory only where the arithmetic and the merge of the pass before it were shorter than the round trip.
            constexpr bool PIPE = (BC8 && FAT && (QMODE == QM_SQ_COS || QMODE == QM_SQ_IP)) || (INT8L && FAT);
            constexpr int PDS = (BC8 || INT8L) ? DS : 64;
            [[maybe_unused]] Bc8Row<PDS> row_next;
            [[maybe_unused]] uint32_t cid_next = 0u;
            [[maybe_unused]] float bn_next = 0.0f;
            [[maybe_unused]] auto pipe_request = [&](const uint32_t base) { // reads only (see the note on lambdas above)
                const uint32_t cj = base + grp;
                cid_next = compact[cj];
                const uint32_t rj = cj < cnt ? (cid_next & idmask) : 0u;
                row_next = bc8_load<PDS>(data + static_cast<size_t>(rj) * static_cast<uint32_t>(a.row_bytes), sub);
                if constexpr (QMODE == QM_SQ_COS) bn_next = a.sq_row_norm[rj];
            };
            if constexpr (PIPE) {
                if (cnt != 0u) pipe_request(0u);
            }
            for (uint32_t p0 = 0; p0 < cnt; p0 += RPP) {
                const uint32_t ci = p0 + grp; // < round_up(cnt, RPP) <= the slots of the row
                uint32_t cid; // the word that goes into the buffer (inline records: id | degree class)
                if constexpr (PIPE) cid = cid_next;
                else if constexpr (FAT) cid = compact[ci];
                else cid = static_cast<uint32_t>(__shfl(static_cast<int>(nid), static_cast<int>(ci)));
                const bool act = ci < cnt;
                const uint32_t rid = FAT ? (cid & idmask) : cid; // the row it names
                [[maybe_unused]] float pipe_dist = 0.0f;
                [[maybe_unused]] uint2 pre_big[8];
                [[maybe_unused]] bool pre_big_valid = false;
                if constexpr (PIPE) {
                    const Bc8Row<PDS> row_cur = row_next;
                    const float bn_cur = bn_next;
                    if (p0 + RPP < cnt) pipe_request(p0 + RPP);
                    if constexpr (BIG == 1 && !GENERAL) { // the merge's entries (up to 8 per lane: buffers of 129 .. 512) arrive under the arithmetic
                        wave_buffer_prefetch<8>(buf, lane, pre_big);
                        pre_big_valid = true;
                    }
                    float fmain = 0.0f, fnrm = 0.0f;
                    if constexpr (INT8L) {
                        int ab = 0, bb = 0;
                        int_row_sums8<DT, PDS>(row_cur, qw, sub, ab, bb);
                        pipe_dist = finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
                    } else if constexpr (QMODE == QM_SQ_COS) {
                        bc8_compute<DT, M_IP, PDS, true>(row_cur, qf, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
                        pipe_dist = fmain / (bn_cur * fx.norm); // cosine.h:334-335
                    } else {
                        bc8_compute<DT, METRIC, PDS, false>(row_cur, qf, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
                        pipe_dist = finish_distance<METRIC, QMODE>(fmain, fnrm, 0, 0, fx, a.sq_scale);
                    }
                }
                // buffers of up to 128 entries: the two entries per lane the merge insert of this pass starts from are requested inside the
                // distance routine, just before its cross-lane reduction (nothing writes the buffer in between)
                constexpr bool PREFETCH = !GENERAL && BIG == 0 && SVSX_MERGE_V2 == 0;
                [[maybe_unused]] uint2 pre[2];
                [[maybe_unused]] bool pre_valid = false;
                float dist;
                if constexpr (PIPE) {
                    dist = pipe_dist;
                } else {
                    dist = group_distance(rid, act, full ? worst * 1.00004f : __builtin_inff(), [&]() {
                        if constexpr (PREFETCH) {
                            if (buf.capacity <= 128u) {
                                wave_buffer_prefetch<2>(buf, lane, pre);
                                pre_valid = true;
                            }
                        }
                    });
                }
                const bool skip = full && cmpf<GREATER>(worst, dist); // can_skip (search_buffer.h:338-340)
                const bool cand = act && sub == 0 && !skip;
                // the wave mask of `cand`, combined in scalar registers from the masks of its single compares
                constexpr unsigned long long kFirstOfGroup = G == 1 ? ~0ull : G == 2 ? 0x5555555555555555ull : G == 4 ? 0x1111111111111111ull : G == 8 ? 0x0101010101010101ull : 0x0001000100010001ull;
                unsigned long long m = wave_ballot(ci < cnt) & kFirstOfGroup & ~(full ? wave_ballot(cmpf<GREATER>(worst, dist)) : 0ull);
                m = uniform_u64(m); // a wave mask is uniform by construction; said explicitly, the buffer state derived from it stays in scalar registers
                SVSX_PHASE_USE(m);
                SVSX_PHASE_STAMP(4);
                n_eval += (cnt - p0) < static_cast<uint32_t>(RPP) ? (cnt - p0) : static_cast<uint32_t>(RPP);
                if constexpr (!GENERAL) {
                    // all accepted candidates of the pass in ONE merge (wave_merge_insert); NaN distances and duplicates of buffer
                    // entries (possible when the visited filter is off or evicted) take the one-by-one path below
                    if (m != 0ull && buf.capacity <= (BIG == 2 ? 1024u : BIG == 1 ? 512u : 128u)) {
                        // one test for NaN and +-inf candidates: either sends the pass down the sequential path, which keeps `nan_seen`
                        const bool finite = (m & wave_ballot(!(__builtin_fabsf(dist) < __builtin_inff()))) == 0ull;
                        bool merged = false;
                        if (!nan_seen && finite) {
#if SVSX_MERGE_V2
                            if constexpr (BIG == 2) {
                                merged = wave_merge_insert2<GREATER, 16>(buf, cand, dist, cid, lane, n_ins, worst, compact);
                            } else if constexpr (BIG == 1) {
                                if (buf.capacity <= 256u) merged = wave_merge_insert2<GREATER, 4>(buf, cand, dist, cid, lane, n_ins, worst, compact);
                                else merged = wave_merge_insert2<GREATER, 8>(buf, cand, dist, cid, lane, n_ins, worst, compact);
                            } else {
                                merged = wave_merge_insert2<GREATER, 2>(buf, cand, dist, cid, lane, n_ins, worst, compact);
                            }
#else
                            if constexpr (BIG == 2) {
                                merged = wave_merge_insert<GREATER, 16>(buf, m, cand, dist, cid, lane, n_ins, worst);
                            } else if constexpr (BIG == 1) {
                                // entries per lane: the loop over the candidates costs two vector instructions per entry register and candidate
                                if (buf.capacity <= 192u) merged = wave_merge_insert<GREATER, 3>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_big_valid ? pre_big : nullptr);
                                else if (buf.capacity <= 256u) merged = wave_merge_insert<GREATER, 4>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_big_valid ? pre_big : nullptr);
                                else if (buf.capacity <= 320u) merged = wave_merge_insert<GREATER, 5>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_big_valid ? pre_big : nullptr); // config 4's window 285
                                else if (buf.capacity <= 384u) merged = wave_merge_insert<GREATER, 6>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_big_valid ? pre_big : nullptr);
                                else merged = wave_merge_insert<GREATER, 8>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_big_valid ? pre_big : nullptr);
                            } else {
                                merged = wave_merge_insert<GREATER, 2>(buf, m, cand, dist, cid, lane, n_ins, worst, pre_valid ? pre : nullptr);
                            }
#endif
                        }
                        if (merged) m = 0ull;
                        full = buf.size == buf.capacity; // (unchanged if nothing was merged) `worst` came out of the merge
                    }
                }
                while (m != 0ull) {
                    const int bl = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), bl));
                    if (full && cmpf<GREATER>(worst, dd)) continue; // worst may have improved within this pass
                    const uint32_t id = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(cid), bl));
                    nan_seen = nan_seen || dd != dd;
                    if (wave_insert<GREATER>(buf, dd, id, lane, nan_seen, FAT ? idmask : kIdMask)) {
                        ++n_ins;
                        full = buf.size == buf.capacity;
                        if (full) worst = __uint_as_float(buf.e[buf.capacity - 1].x);
                    }
                }
                SVSX_PHASE_USE(buf.best_unvisited);
                SVSX_PHASE_STAMP(5);
            }
            if constexpr (FAT) break; // the passes ran over all chunks: nothing (the next-chunk registers included) lives past this point
            else if (last_chunk) break;
        }
    }

#if defined(SVSX_DIAG) && SVSX_DIAG == 3
    const uint32_t diag_tloop_end = static_cast<uint32_t>(wall_clock64());
#endif
    if constexpr (GENERAL) {
        if (it_meta != nullptr) {
            // ---- save the buffer for the next call; copy_from_scratch (iterator.h:104-124): the best entries not yet
            // handed out, in buffer order, at most batch_size of them -------------------------------------------------------
            uint2* dst = a.it_entries + static_cast<size_t>(qi) * a.it_estride;
            if (!buf.hbm) {
                for (uint32_t i = lane; i < buf.size; i += 64) dst[i] = buf.e[i];
            }
            buf_order(buf);
            const uint32_t n_old = it_meta[4];
            uint32_t* yl = a.it_yielded + static_cast<size_t>(qi) * a.it_ystride;
            uint32_t got = 0;
            for (uint32_t base = 0; base < buf.size && got < it_bs; base += 64) {
                const uint32_t i = base + lane;
                const bool valid = i < buf.size;
                uint2 en = make_uint2(0u, kInvalidId);
                if (valid) en = buf.e[i];
                const uint32_t id = valid ? (en.y & kIdMask) : kInvalidId;
                bool fresh = valid;
                for (uint32_t y0 = 0; y0 < n_old; y0 += 64) {
                    if (wave_ballot(fresh) == 0ull) break;
                    uint32_t yv = kInvalidId;
                    if (y0 + lane < n_old) yv = yl[y0 + lane];
                    for (int t = 0; t < 64; ++t) {
                        const uint32_t v = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(yv), t));
                        fresh = fresh && (v != id);
                    }
                }
                const unsigned long long fm = wave_ballot(fresh);
                const uint32_t rank = got + prefix_count(fm);
                if (fresh && rank < it_bs) {
                    a.it_out_ids[static_cast<size_t>(qi) * a.it_batch + rank] = static_cast<uint64_t>(id) + a.id_offset;
                    a.it_out_dists[static_cast<size_t>(qi) * a.it_batch + rank] = __uint_as_float(en.x);
                    yl[n_old + rank] = id;
                }
                const uint32_t add = __popcll(fm);
                got = (got + add < it_bs) ? got + add : it_bs;
            }
            for (uint32_t j = got + lane; j < a.it_batch; j += 64) {
                a.it_out_ids[static_cast<size_t>(qi) * a.it_batch + j] = ~0ull;
                a.it_out_dists[static_cast<size_t>(qi) * a.it_batch + j] = __builtin_inff();
            }
            if (lane == 0) {
                a.it_counts[qi] = got;
                it_meta[0] = buf.size;
                it_meta[1] = buf.best_unvisited;
                it_meta[2] = buf.window;
                it_meta[3] = buf.capacity;
                it_meta[4] = n_old + got;
                it_meta[5] = it_meta[5] + 1u;
                it_meta[6] = (got == 0u && it_bs > 0u) ? 2u : 0u; // restart_search_ = false; is_exhausted_ (:311-316)
            }
        }
    }

    // ---- copy-back (extensions.h:588-590), padded like vamana_index_impl.h:205-209 -----------------------------
    for (uint32_t j = lane; j < a.k; j += 64) {
        uint32_t id = kInvalidId;
        float dist = __builtin_inff();
        const bool have = j < buf.size;
        if (have) {
            const uint2 en = buf.e[j];
            id = en.y & (FAT ? idmask : kIdMask);
            dist = __uint_as_float(en.x);
        }
        const size_t o = static_cast<size_t>(qi) * a.k + j;
        if (a.ids64 != nullptr) a.ids64[o] = have ? (static_cast<uint64_t>(id) + a.id_offset) : ~0ull;
        if (a.ids32 != nullptr) a.ids32[o] = have ? static_cast<uint32_t>(id + static_cast<uint32_t>(a.id_offset)) : kInvalidId;
        if (a.dists != nullptr) a.dists[o] = dist;
    }
    if (lane == 0) {
        if (a.stats != nullptr) {
            uint32_t* st = a.stats + static_cast<size_t>(qi) * 6;
            st[0] = n_exp;
            st[1] = n_slots;
            st[2] = n_eval;
            st[3] = n_ins;
            st[4] = n_early;
            st[5] = n_code_rej;
#ifdef SVSX_DIAG
            st[4] = diag_t0;
            st[5] = static_cast<uint32_t>(wall_clock64());
#if SVSX_DIAG == 2
            for (int i = 1; i <= 5; ++i) st[i] = ph_acc[i]; // [0] stays the expansion count; ph_acc[0] = between expansions (loop control)
#endif
#if SVSX_DIAG == 3
            st[2] = diag_tloop;      // wall_clock64 (100 MHz) when the main loop starts / ends; [4] / [5] = wave start / end
            st[3] = diag_tloop_end;
            // where the wave ran: HW_ID (wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [14:13] ...) | XCC_ID << 28
            st[1] = (static_cast<uint32_t>(__builtin_amdgcn_s_getreg((31 << 11) | 4)) & 0x0FFFFFFFu) | (static_cast<uint32_t>(__builtin_amdgcn_s_getreg((3 << 11) | 20)) << 28);
#endif
#endif
        }
        if (a.history_len != nullptr) a.history_len[qi] = n_exp;
    }
}

template <int DT, int METRIC, int QMODE, int DS, bool GENERAL = false, int CODES = 0, int BIG = 0>
__global__ __launch_bounds__(64) SVSX_KERNEL_ATTR void search_kernel(const SearchArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    search_one<DT, METRIC, QMODE, DS, GENERAL, CODES, BIG>(a, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------------------------
char* last_search_instantiation_buffer(); // thread-local, 96 bytes (svsx_search.hip)

template <int DT, int METRIC, int QMODE, int DS, bool GENERAL = false, int CODES = 0, int BIG = 0> static hipError_t launch_one(const SearchArgs& a, hipStream_t stream) {
    if constexpr (!GENERAL && BIG == 0) {
        // the big-buffer levels exist for the runtime-d kernel and the dimensions of the benchmark configs; the other static dimensions of the
        // reference's table (distance_core.h:31-32) fall back to the runtime-d kernel there: same arithmetic, same results
        constexpr int DB = (DS == 96 || DS == 128 || DS == 768) ? DS : 0;
        if (a.capacity > 128u && a.capacity <= 512u) return launch_one<DT, METRIC, QMODE, DB, GENERAL, CODES, 1>(a, stream);
        if (a.capacity > 512u && a.capacity <= 1024u) return launch_one<DT, METRIC, QMODE, DB, GENERAL, CODES, 2>(a, stream);
    }
    snprintf(last_search_instantiation_buffer(), 96, "search_kernel<dt=%d,metric=%d,qmode=%d,ds=%d,general=%d,codes=%d>%s", DT, METRIC, QMODE, DS, GENERAL ? 1 : 0, CODES,
             QMODE == QM_SCALAR ? " scalar loop" : "");
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    const bool intpath = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    const size_t buf_bytes = ((static_cast<size_t>(a.capacity) + 1) * 8 + 15) & ~size_t(15);
    const size_t lds = buf_bytes + (intpath ? static_cast<size_t>(dpad) : static_cast<size_t>(dpad) * 4) +
                       (a.filter_entries ? static_cast<size_t>(a.filter_entries) * (a.filter_tag8 ? 1 : 2) : 0) +
                       (CODES >= 2 ? ((static_cast<size_t>(a.gstride) + 63) & ~size_t(63)) * 4 : 64 * 4) /* compaction / merge scratch */ + ((CODES == 3 || CODES == 4) ? 128 : 0) + (GENERAL ? 64 * 8 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
#ifdef SVSX_DIAG
    size_t lds_request = lds;
    if (const char* pad = getenv("SVSX_LDS_REQUEST")) { // residency sweeps: ask for more LDS than the wave uses
        const size_t want = static_cast<size_t>(atol(pad));
        if (want > lds && want <= 160 * 1024) lds_request = want;
    }
#else
    const size_t lds_request = lds;
#endif
    auto kern = search_kernel<DT, METRIC, QMODE, DS, GENERAL, CODES, BIG>;
    if (lds_request > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_request));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.nq), dim3(64), lds_request, stream, a);
    return hipGetLastError();
}

template <int DT, int METRIC, int QMODE> static hipError_t launch_dim(const SearchArgs& a, hipStream_t stream) {
    if (a.general || QMODE == QM_SCALAR) return launch_one<DT, METRIC, QMODE, 0, true>(a, stream); // runtime-d kernel: same arithmetic, same results (the scalar-loop pairs have no other variant)
    if constexpr (QMODE == QM_FLOAT && (DT == DT_F32 || DT == DT_F16)) {
        // static dimensions: the reference's list (distance_core.h:31-32: 64, 96, 100, 128, 160, 200, 512, 768, Dynamic)
        if (a.fat != nullptr && a.fat_rb == 28) { // round 6: 28-byte records, the dimensions of the benchmark configs and runtime d
            switch (a.d) {
                case 96: return launch_one<DT, METRIC, QMODE, 96, false, 5>(a, stream);
                case 128: return launch_one<DT, METRIC, QMODE, 128, false, 5>(a, stream);
                default: return launch_one<DT, METRIC, QMODE, 0, false, 5>(a, stream);
            }
        }
        if (a.fat != nullptr && a.fat_rb == 32) {
            switch (a.d) {
                case 64: return launch_one<DT, METRIC, QMODE, 64, false, 2>(a, stream);
                case 96: return launch_one<DT, METRIC, QMODE, 96, false, 2>(a, stream);
                case 100: return launch_one<DT, METRIC, QMODE, 100, false, 2>(a, stream);
                case 128: return launch_one<DT, METRIC, QMODE, 128, false, 2>(a, stream);
                case 160: return launch_one<DT, METRIC, QMODE, 160, false, 2>(a, stream);
                case 200: return launch_one<DT, METRIC, QMODE, 200, false, 2>(a, stream);
                default: break; // 512 / 768: the cost model never picks 32-byte records for rows of 8+ lines
            }
        }
        if (a.fat != nullptr && a.fat_rb == 32) return launch_one<DT, METRIC, QMODE, 0, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64) return launch_one<DT, METRIC, QMODE, 0, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128) return launch_one<DT, METRIC, QMODE, 0, false, 4>(a, stream);
    }
    if constexpr ((QMODE == QM_SQ_IP || QMODE == QM_SQ_COS) && (DT == DT_I8 || DT == DT_U8)) {
        if (a.fat != nullptr && a.fat_rb == 32 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 32) return launch_one<DT, METRIC, QMODE, 0, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64) return launch_one<DT, METRIC, QMODE, 0, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 4>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128) return launch_one<DT, METRIC, QMODE, 0, false, 4>(a, stream);
    }
    if constexpr (QMODE == QM_FLOAT && (DT == DT_I8 || DT == DT_U8)) {
        // float32 queries over raw int8 / uint8 rows (the pairs (float, int8) / (float, uint8) of euclidean.h:318-358: ConvertToFloat<16> loads,
        // the float accumulate trees): the same records serve them -- the vector the metric sees is float(row element) either way
        if (a.fat != nullptr && a.fat_rb == 32) return launch_one<DT, METRIC, QMODE, 0, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64) return launch_one<DT, METRIC, QMODE, 0, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128) return launch_one<DT, METRIC, QMODE, 0, false, 4>(a, stream);
    }
    if constexpr ((QMODE == QM_INT || QMODE == QM_SQ_L2) && (DT == DT_I8 || DT == DT_U8)) {
        // round 6: the exact integer paths (VNNI pairs euclidean.h:265-313, inner_product.h:222-267, cosine.h:261-326; EuclideanCompressed
        // scalar.h:58-105) behind inline records built over float(row element)
        if (a.fat != nullptr && a.fat_rb == 32 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 32) return launch_one<DT, METRIC, QMODE, 0, false, 2>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 64) return launch_one<DT, METRIC, QMODE, 0, false, 3>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128 && a.d == 768) return launch_one<DT, METRIC, QMODE, 768, false, 4>(a, stream);
        if (a.fat != nullptr && a.fat_rb == 128) return launch_one<DT, METRIC, QMODE, 0, false, 4>(a, stream);
    }
    if constexpr (QMODE == QM_FLOAT && (DT == DT_F32 || DT == DT_F16) && METRIC == M_L2) {
        if (a.codes != nullptr && a.d == 96) return launch_one<DT, METRIC, QMODE, 96, false, 1>(a, stream);
        if (a.codes != nullptr && a.d == 128) return launch_one<DT, METRIC, QMODE, 128, false, 1>(a, stream);
    }
    if constexpr (QMODE == QM_FLOAT && (DT == DT_F32 || DT == DT_F16)) {
        if (a.fat == nullptr) {
            switch (a.d) {
                case 64: return launch_one<DT, METRIC, QMODE, 64>(a, stream);
                case 96: return launch_one<DT, METRIC, QMODE, 96>(a, stream);
                case 100: return launch_one<DT, METRIC, QMODE, 100>(a, stream);
                case 128: return launch_one<DT, METRIC, QMODE, 128>(a, stream);
                case 160: return launch_one<DT, METRIC, QMODE, 160>(a, stream);
                case 200: return launch_one<DT, METRIC, QMODE, 200>(a, stream);
                case 512: return launch_one<DT, METRIC, QMODE, 512>(a, stream);
                case 768: return launch_one<DT, METRIC, QMODE, 768>(a, stream);
                default: break;
            }
        }
    }
    if constexpr ((QMODE == QM_SQ_IP || QMODE == QM_SQ_COS || QMODE == QM_INT || QMODE == QM_SQ_L2) && (DT == DT_I8 || DT == DT_U8)) {
        if (a.d == 768) return launch_one<DT, METRIC, QMODE, 768>(a, stream);
    }
    return launch_one<DT, METRIC, QMODE, 0>(a, stream);
}

template <int DT, int QMODE> static hipError_t launch_metric(int metric, const SearchArgs& a, hipStream_t stream) {
    switch (metric) {
        case M_L2:
            if constexpr (QMODE == QM_SQ_IP || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_dim<DT, M_L2, QMODE>(a, stream);
        case M_IP:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_dim<DT, M_IP, QMODE>(a, stream);
        case M_COS:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_IP) return hipErrorInvalidValue;
            else return launch_dim<DT, M_COS, QMODE>(a, stream);
        default: return hipErrorInvalidValue;
    }
}

// one entry point per row type (its own translation unit): dispatch on the query mode, then metric, then dimension
template <int DT> static hipError_t launch_search_for_rows(int metric, int qmode, const SearchArgs& a, hipStream_t stream) {
    switch (qmode) {
        case QM_FLOAT: return launch_metric<DT, QM_FLOAT>(metric, a, stream);
        case QM_INT:
            if constexpr (DT == DT_I8 || DT == DT_U8) return launch_metric<DT, QM_INT>(metric, a, stream);
            break;
        case QM_SQ_IP:
            if constexpr (DT == DT_I8 || DT == DT_U8) return launch_metric<DT, QM_SQ_IP>(metric, a, stream);
            break;
        case QM_SQ_COS:
            if constexpr (DT == DT_I8 || DT == DT_U8) return launch_metric<DT, QM_SQ_COS>(metric, a, stream);
            break;
        case QM_SQ_L2:
            if constexpr (DT == DT_I8 || DT == DT_U8) return launch_metric<DT, QM_SQ_L2>(metric, a, stream);
            break;
        case QM_SCALAR: return launch_metric<DT, QM_SCALAR>(metric, a, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace svsx
