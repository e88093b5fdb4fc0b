// svsx_device.h -- device-side building blocks of the MI355X (gfx950) Vamana greedy-search path.
//
// Arithmetic contract (what makes neighbour ids AND distance bits equal to the reference CPU path):
//   * float kernels reproduce the accumulation tree of simd::generic_simd_op with 16 float lanes
//     (core/distance/simd_utils.h:204-252): 4 independent accumulators per lane-slot l over 64-element blocks,
//     (s0+s1)+(s2+s3), remaining 16-blocks and the masked tail into s0, then the _mm512_reduce_add_ps tree
//     (l,l+8) -> (l,l+4) -> (0,2),(1,3) -> final.  A group of G lanes owns one candidate row; lane `sub` owns the
//     E = 16/G lane-slots l = E*sub .. E*sub+E-1 of EVERY 16-block, so every fma chain stays inside one lane and
//     the tree needs only quad-local DPP exchanges.
//   * int8/uint8 kernels are exact int32 (v_dot4), like the reference's VNNI kernels
//     (euclidean.h:265-313, inner_product.h:222-267, cosine.h:261-326).
//   * compiled with -ffp-contract=off: every fused op is an explicit fmaf, everything else rounds separately.
//     sqrtf and '/' are the correctly rounded forms (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svsx {

enum : int { DT_F32 = 0, DT_F16 = 1, DT_I8 = 2, DT_U8 = 3 };
enum : int { M_L2 = 0, M_IP = 1, M_COS = 2 };
// How the query is held and what epilogue applies.
enum : int {
    QM_FLOAT = 0,   // query widened to f32 in LDS; data T widened on load (K1-K4)
    QM_INT = 1,     // query and data both int8 or both uint8, exact int32 (K5, K6)
    QM_SQ_IP = 2,   // InnerProductCompressed   (quantization/scalar/scalar.h:107-153): float path + scale*ip + bias*sum(q)
    QM_SQ_COS = 3,  // CosineSimilarityCompressed (scalar.h:155-195): decompress on load, then float cosine
    QM_SQ_L2 = 4,   // EuclideanCompressed<T>   (scalar.h:58-105): query compressed to T, scale^2 * int L2
    QM_SCALAR = 5   // type pairs without an AVX-512 specialisation in the reference: its scalar loops generic_l2 / generic_ip /
                    // generic_cosine_similarity (euclidean.h:206-215, inner_product.h:167-176, cosine.h:185-200), one float accumulator in element order
};

constexpr uint32_t kInvalidId = 0xFFFFFFFFu;
constexpr uint32_t kVisitedBit = 0x80000000u;
constexpr uint32_t kIdMask = 0x7FFFFFFFu;

template <typename To, typename From> __device__ __forceinline__ To bit_cast_(From v) {
    return __builtin_bit_cast(To, v);
}

// The wave mask of a predicate.  HIP's __ballot goes through an integer compare of the zero-extended predicate (v_cndmask 0 / 1 + v_cmp_ne per
// call: two vector instructions for a mask that already sits in a scalar register pair); the builtin is the mask itself.  The search kernel
// takes ~12 ballots per expansion and is bound by the vector ALU's issue slots while the machine is full.
__device__ __forceinline__ unsigned long long wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// tell the compiler that a 64-bit value is the same in every lane (folds away when it already knows)
__device__ __forceinline__ unsigned long long uniform_u64(unsigned long long v) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v))));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v >> 32))));
    return (static_cast<unsigned long long>(hi) << 32) | lo;
}

// quad-local lane exchange through DPP (no LDS traffic): lane ^ 1 and lane ^ 2.
// (every lane of the wave is active wherever these are used, and a quad permutation never reads out of bounds, so the `old` operand is
// never seen: mov_dpp leaves it undefined, which spares the v_mov 0 that update_dpp(0, ...) costs per exchange)
__device__ __forceinline__ float xor1f(float v) {
    return bit_cast_<float>(__builtin_amdgcn_mov_dpp(bit_cast_<int>(v), 0xB1, 0xf, 0xf, true));
}
__device__ __forceinline__ float xor2f(float v) {
    return bit_cast_<float>(__builtin_amdgcn_mov_dpp(bit_cast_<int>(v), 0x4E, 0xf, 0xf, true));
}
__device__ __forceinline__ int xor1i(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ int xor2i(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true); }
// v + (v of lane ^ 1 / ^ 2) as ONE instruction: the opaque barrier keeps the SLP vectorizer from pairing the adds into v_pk_add_f32
// (which has no DPP form: 2 v_mov_dpp + 1 v_pk_add per pair), so each exchange folds into its add (v_add_f32_dpp; the compiler
// inserts the DPP hazard wait states itself).  Same IEEE add, same bits.  The VALU issue port is what bounds the search kernel
// (profiles/r03_notes.md): 8 instead of 22 instructions per 4-slot reduction.
__device__ __forceinline__ float opaque_f(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int opaque_i(int v) {
    asm volatile("" : "+v"(v));
    return v;
}
// A copy of `v` the optimizer knows nothing about.  Used on the lane index inside the search kernel's main loop: address terms derived from
// such a copy (lane * 4, lane * 16 as a 64-bit pair, lane | 64 ...) are recomputed where they are used (one instruction each) instead of being
// hoisted out of the loop and kept in registers across the distance pass -- the kernel sits at its 80-register bracket (6 waves per SIMD) and
// spilled exactly those constants, with the reload in front of the adjacency block's loads.
__device__ __forceinline__ uint32_t opaque_u(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float add_xor1f(float v) { return opaque_f(v + xor1f(v)); }
__device__ __forceinline__ float add_xor2f(float v) { return opaque_f(v + xor2f(v)); }
__device__ __forceinline__ int add_xor1i(int v) { return opaque_i(v + xor1i(v)); }
__device__ __forceinline__ int add_xor2i(int v) { return opaque_i(v + xor2i(v)); }

// ---------------------------------------------------------------------------------------------------------
// element traits: E lane-slots per lane, G = 16/E lanes per candidate row, chunk = E elements per load
template <int DT> struct Elem;
template <> struct Elem<DT_F32> { static constexpr int E = 4, BYTES = 4; };   // 16-B loads, 4 lanes / row
template <> struct Elem<DT_F16> { static constexpr int E = 8, BYTES = 2; };   // 16-B loads, 2 lanes / row
template <> struct Elem<DT_I8> { static constexpr int E = 8, BYTES = 1; };    //  8-B loads, 2 lanes / row
template <> struct Elem<DT_U8> { static constexpr int E = 8, BYTES = 1; };

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// Load E consecutive elements and widen to f32 exactly as ConvertToFloat<16>::load does
// (simd_utils.h:262-301: loadu / cvtph_ps / cvtep[iu]8_epi32 + cvtepi32_ps).
template <int DT> __device__ __forceinline__ void load_chunk(const unsigned char* p, float (&out)[Elem<DT>::E]) {
    if constexpr (DT == DT_F32) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
    } else if constexpr (DT == DT_F16) {
        const half8_t v = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) out[e] = static_cast<float>(v[e]);
    } else {
        const uint2 v = *reinterpret_cast<const uint2*>(p);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t w = e < 4 ? v.x : v.y;
            const uint32_t byte = (w >> (8 * (e & 3))) & 0xFFu;
            if constexpr (DT == DT_I8) out[e] = static_cast<float>(static_cast<int>(static_cast<int8_t>(byte)));
            else out[e] = static_cast<float>(byte);
        }
    }
}

template <int E> __device__ __forceinline__ void zero_chunk(float (&x)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = 0.0f;
}

// One accumulate step of L2FloatOp / IPFloatOp / CosineFloatOp (euclidean.h:246-249, inner_product.h:205-207,
// cosine.h:238-241).
template <int METRIC> __device__ __forceinline__ void acc_step(float& s, float& n, float q, float x) {
    if constexpr (METRIC == M_L2) {
        const float c = q - x;
        s = __builtin_fmaf(c, c, s);
    } else {
        s = __builtin_fmaf(q, x, s);
        if constexpr (METRIC == M_COS) n = __builtin_fmaf(x, x, n);
    }
}

// _mm512_reduce_add_ps over the 16 lane-slots of one row, distributed E per lane over G = 16/E adjacent lanes.
template <int E> __device__ __forceinline__ float reduce_slots(float (&v)[E]) {
    if constexpr (E == 4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = add_xor2f(v[e]); // t[j] = v[j+8] + v[j]
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = add_xor1f(v[e]); // u[j] = t[j+4] + t[j]
        const float w0 = v[0] + v[2];
        const float w1 = v[1] + v[3];
        return w0 + w1;
    } else {
        static_assert(E == 8, "E must be 4 or 8");
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = add_xor1f(v[e]); // t[j] = v[j+8] + v[j]
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = v[j + 4] + v[j];     // u[j] = t[j+4] + t[j]
        const float w0 = u[0] + u[2];
        const float w1 = u[1] + u[3];
        return w0 + w1;
    }
}

// float32 rows of a static dimension that is a multiple of 16, slot-cyclic mapping, written on explicit register PAIRS (round 5).  The same
// operations in the same order as float_row_distance below -- lane-slot e of block i goes through acc_step into s[i & 3][e], then
// (s0 + s1) + (s2 + s3), the remaining 16-blocks into s0, reduce_slots -- but two adjacent lane-slots share one v_pk_add_f32 / v_pk_fma_f32
// (each half is the same IEEE operation), and a pair is two CONSECUTIVE registers of the 16-byte row load and of the 16-byte query read.
// Left to itself the SLP vectorizer paired lane-slot e of block i with lane-slot e of block i + 2 and spent 26 v_mov per 16-row pass
// gathering the operands (87 vector instructions per pass at d = 96; 49 in this form).
typedef float f2_t __attribute__((ext_vector_type(2)));

template <int METRIC> __device__ __forceinline__ void acc_step2(f2_t& s, f2_t& n, f2_t q, f2_t x) {
    if constexpr (METRIC == M_L2) {
        const f2_t c = q - x;
        s = __builtin_elementwise_fma(c, c, s);
    } else {
        s = __builtin_elementwise_fma(q, x, s);
        if constexpr (METRIC == M_COS) n = __builtin_elementwise_fma(x, x, n);
    }
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `before_reduce` runs between the last accumulate step and the cross-lane reduction: the search kernel requests the buffer entries of the
// coming merge insert there (an LDS round trip under ~20 instructions that do not depend on it).
template <int METRIC, int DS, typename HOOK = NoHook>
__device__ __forceinline__ void float_row_distance_f32_pairs(const unsigned char* __restrict__ row, const float* __restrict__ qf, int sub, float& out_main, float& out_nrm,
                                                             HOOK&& before_reduce = NoHook{}) {
    static_assert(DS >= 16 && (DS & 15) == 0, "whole 16-blocks");
    constexpr int NB = DS >> 4, NB64 = (DS >> 6) * 4;
    const float4* lane_row = reinterpret_cast<const float4*>(row) + sub; // block i = 4 float4; lane `sub` owns lane-slots 4 sub .. 4 sub + 3
    const float4* lane_q = reinterpret_cast<const float4*>(qf) + sub;
    float4 x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = lane_row[4 * i]; // every load of the row before the first use; inactive groups read row 0
    // the query's blocks come from LDS four ahead of their use, two at a time: at most 16 registers of query are live while the row is in
    // flight, and no read waits on the arithmetic of the block before it (the scheduler, left alone, issued them one by one, each
    // followed by its own wait: five dependent LDS round trips per pass)
#ifndef SVSX_QA
#define SVSX_QA 4
#endif
    constexpr int QA = NB < SVSX_QA ? NB : SVSX_QA;
    float4 q[NB];
#pragma unroll
    for (int i = 0; i < QA; ++i) q[i] = lane_q[4 * i];
    __builtin_amdgcn_sched_barrier(0);
    f2_t s[4][2], n[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[j][0] = f2_t{0.0f, 0.0f}; s[j][1] = f2_t{0.0f, 0.0f};
        n[j][0] = f2_t{0.0f, 0.0f}; n[j][1] = f2_t{0.0f, 0.0f};
    }
#pragma unroll
    for (int i0 = 0; i0 < NB; i0 += 2) {
#pragma unroll
        for (int i = i0; i < i0 + 2 && i < NB; ++i) {
            if (NB64 > 0 && i == NB64) { // the 64-blocks are done: (s0 + s1) + (s2 + s3), the rest goes into s0
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s[0][h] = (s[0][h] + s[1][h]) + (s[2][h] + s[3][h]);
                    if constexpr (METRIC == M_COS) n[0][h] = (n[0][h] + n[1][h]) + (n[2][h] + n[3][h]);
                }
            }
            const int j = i < NB64 ? (i & 3) : 0;
            acc_step2<METRIC>(s[j][0], n[j][0], f2_t{q[i].x, q[i].y}, f2_t{x[i].x, x[i].y});
            acc_step2<METRIC>(s[j][1], n[j][1], f2_t{q[i].z, q[i].w}, f2_t{x[i].z, x[i].w});
        }
#pragma unroll
        for (int i = i0 + QA; i < i0 + QA + 2 && i < NB; ++i) q[i] = lane_q[4 * i];
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NB64 == NB && NB64 > 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            s[0][h] = (s[0][h] + s[1][h]) + (s[2][h] + s[3][h]);
            if constexpr (METRIC == M_COS) n[0][h] = (n[0][h] + n[1][h]) + (n[2][h] + n[3][h]);
        }
    }
    before_reduce();
    float v[4] = {s[0][0].x, s[0][0].y, s[0][1].x, s[0][1].y};
    out_main = reduce_slots<4>(v);
    if constexpr (METRIC == M_COS) {
        float w[4] = {n[0][0].x, n[0][0].y, n[0][1].x, n[0][1].y};
        out_nrm = reduce_slots<4>(w);
    } else {
        out_nrm = 0.0f;
    }
}

// Float-path distance of one candidate row against the query held (as f32, zero padded to 16) in LDS.
// All 64 lanes execute this in lock-step; `active` only predicates the global loads.
//   DS > 0: compile-time dimension (fully unrolled, all row loads issued up front); DS == 0: runtime d.
//   DECOMP: apply the SQ decompression scale*float(v)+bias to every loaded element (scalar.h:44-46,175-184).
// PRUNE (squared L2, runtime dimension only): after every 64-element block the partial sum is a lower bound of the
// distance; groups whose bound already exceeds `thr` (= worst * (1 + 4e-5), see l2_row_distance_two_phase) stop
// fetching, and the loop ends once every group of the wave has stopped.  out_main = +inf for a rejected group.
template <int DT, int METRIC, int DS, bool DECOMP, bool PRUNE = false, typename HOOK = NoHook>
__device__ __forceinline__ void float_row_distance(
    const unsigned char* __restrict__ row, bool active, const float* __restrict__ qf, int d_runtime, int sub,
    float sq_scale, float sq_bias, float& out_main, float& out_nrm, float thr = 0.0f, bool* rejected_out = nullptr, HOOK&& before_reduce = NoHook{}
) {
    static_assert(!PRUNE || (METRIC == M_L2 && DS == 0 && !DECOMP), "partial-distance rejection: runtime-d squared L2 only");
#ifndef SVSX_F32_PAIRS
#define SVSX_F32_PAIRS 1
#endif
    if constexpr (SVSX_F32_PAIRS != 0 && DT == DT_F32 && DS >= 16 && DS <= 256 && (DS & 15) == 0 && !DECOMP && !PRUNE) {
        float_row_distance_f32_pairs<METRIC, DS>(row, qf, sub, out_main, out_nrm, before_reduce);
        return;
    }
    constexpr int E = Elem<DT>::E;
    constexpr int CB = E * Elem<DT>::BYTES; // bytes per chunk
    const int d = DS > 0 ? DS : d_runtime;
    const int nb = d >> 4;
    const int n64 = nb >> 2;
    const unsigned char* lane_row = row + sub * CB;
    const float* lane_q = qf + sub * E;

    float s[4][E], n[4][E];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        zero_chunk<E>(s[j]);
        zero_chunk<E>(n[j]);
    }
    bool rejected = false;

    auto fetch = [&](int block, float (&x)[E]) {
        // callers point inactive groups at row 0 (always mapped, cache resident), so the load needs no predicate -- an exec-masked
        // load costs three scalar instructions, and the scalar unit is this kernel's scarce issue port; only the PRUNE form
        // really stops fetching
        if constexpr (PRUNE) {
            zero_chunk<E>(x);
            if (active) load_chunk<DT>(lane_row + block * (16 * Elem<DT>::BYTES), x);
        } else {
            load_chunk<DT>(lane_row + block * (16 * Elem<DT>::BYTES), x);
        }
        if constexpr (DECOMP) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float t = sq_scale * x[e];
                x[e] = t + sq_bias;
            }
        }
    };
    auto fetch_q = [&](int block, float (&q)[E]) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(lane_q + block * 16 + e);
            q[e] = v.x; q[e + 1] = v.y; q[e + 2] = v.z; q[e + 3] = v.w;
        }
    };

    int b = 0;
    // static dimension: EVERY load of the row is issued before the first use -- the 64-blocks, the remaining 16-blocks and the ragged
    // block.  (Round 2 issued the loads of the epilogue blocks after the 64-block arithmetic: at d = 96 the third cache line of each
    // row was a dependent HBM round trip of its own inside every distance pass.)
    constexpr int SB64 = DS > 0 ? (DS >> 6) * 4 : 0, SB1 = DS > 0 ? (DS >> 4) : 0, SRAG = (DS > 0 && (DS & 15) != 0) ? 1 : 0;
    [[maybe_unused]] float xs[(SB1 + SRAG) > 0 ? (SB1 + SRAG) : 1][E];
    if constexpr (DS > 0) {
#pragma unroll
        for (int i = 0; i < SB1 + SRAG; ++i) fetch(i, xs[i]);
    }
    if (n64 > 0) {
        if constexpr (DS > 0) {
#pragma unroll
            for (int i = 0; i < SB64; ++i) {
                float q[E];
                fetch_q(i, q);
#pragma unroll
                for (int e = 0; e < E; ++e) acc_step<METRIC>(s[i & 3][e], n[i & 3][e], q[e], xs[i][e]);
            }
            b = SB64;
        } else {
            for (int t = 0; t < n64; ++t) {
                float x[4][E];
#pragma unroll
                for (int j = 0; j < 4; ++j) fetch(4 * t + j, x[j]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float q[E];
                    fetch_q(4 * t + j, q);
#pragma unroll
                    for (int e = 0; e < E; ++e) acc_step<METRIC>(s[j][e], n[j][e], q[e], x[j][e]);
                }
                if constexpr (PRUNE) {
                    if (16 * (4 * t + 4) < d) { // something is left to skip
                        float p[E];
#pragma unroll
                        for (int e = 0; e < E; ++e) p[e] = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
                        const float partial = reduce_slots<E>(p);
                        if (partial > thr) {
                            rejected = true;
                            active = false; // no further loads for this group
                        }
                        if (wave_ballot(active) == 0ull) break;
                    }
                }
            }
            b = 4 * n64;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            s[0][e] = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
            if constexpr (METRIC == M_COS) n[0][e] = (n[0][e] + n[1][e]) + (n[2][e] + n[3][e]);
        }
    }
    // full-width epilogue: remaining 16-blocks into s0
    if constexpr (DS > 0) {
#pragma unroll
        for (int i = SB64; i < SB1; ++i) {
            float q[E];
            fetch_q(i, q);
#pragma unroll
            for (int e = 0; e < E; ++e) acc_step<METRIC>(s[0][e], n[0][e], q[e], xs[i][e]);
        }
        b = SB1;
    } else {
        for (; b < nb; ++b) {
            float x[E], q[E];
            fetch(b, x);
            fetch_q(b, q);
#pragma unroll
            for (int e = 0; e < E; ++e) acc_step<METRIC>(s[0][e], n[0][e], q[e], x[e]);
        }
    }
    // ragged epilogue: masked lane-slots keep their accumulator (mask3_fmadd)
    if ((d & 15) != 0) {
        float x[E], q[E];
        if constexpr (DS > 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) x[e] = xs[SB1][e];
        } else {
            fetch(b, x);
        }
        fetch_q(b, q);
        const int first = 16 * b + sub * E;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float ts = s[0][e], tn = n[0][e];
            acc_step<METRIC>(ts, tn, q[e], x[e]);
            const bool in = (first + e) < d;
            s[0][e] = in ? ts : s[0][e];
            if constexpr (METRIC == M_COS) n[0][e] = in ? tn : n[0][e];
        }
    }
    before_reduce();
    out_main = reduce_slots<E>(s[0]);
    if constexpr (METRIC == M_COS) out_nrm = reduce_slots<E>(n[0]);
    else out_nrm = 0.0f;
    if constexpr (PRUNE) {
        if (rejected) out_main = __builtin_inff();
        if (rejected_out != nullptr) *rejected_out = rejected;
    }
}

// Two-phase squared-L2 for the slot-cyclic mapping with a compile-time dimension DS in (64, 128]: after the first
// 64-element block (two cache lines of an f32 row) the partial sum  reduce((s0+s1)+(s2+s3))  is a lower bound of the
// full distance (every later term is a square).  If it already exceeds `thr` = worst * (1 + 4e-5) the candidate cannot
// enter the full buffer -- SearchBuffer::can_skip (search_buffer.h:338-340) would drop it -- and the rest of the row is
// never fetched.  The margin covers the rounding of both summations: sums of non-negative terms through per-slot fma
// chains of d/64 links plus a 7-level tree have relative error <= (d/64 + 8) * 2^-24 (1.2e-6 at d = 768), so a
// candidate the reference would insert is never rejected; survivors
// continue with exactly the reference's operation order, so their distance bits are unchanged.
template <int DT, int DS>
__device__ __forceinline__ float l2_row_distance_two_phase(
    const unsigned char* __restrict__ row, bool active, const float* __restrict__ qf, int sub, float thr, bool& rejected
) {
    static_assert(DS > 64 && DS <= 128 && (DS & 15) == 0, "one leading 64-block plus full 16-blocks");
    constexpr int E = Elem<DT>::E;
    constexpr int CB = E * Elem<DT>::BYTES;
    constexpr int NB64 = (DS >> 6) * 4, B1 = DS >> 4;
    const unsigned char* lane_row = row + sub * CB;
    const float* lane_q = qf + sub * E;
    auto fetch = [&](int block, bool pred, float (&x)[E]) {
        zero_chunk<E>(x);
        if (pred) load_chunk<DT>(lane_row + block * (16 * Elem<DT>::BYTES), x);
    };
    auto fetch_q = [&](int block, float (&q)[E]) {
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            const float4 v = *reinterpret_cast<const float4*>(lane_q + block * 16 + e);
            q[e] = v.x; q[e + 1] = v.y; q[e + 2] = v.z; q[e + 3] = v.w;
        }
    };
    float s[4][E];
    float dummy = 0.0f;
    {
        float x[4][E];
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch(j, active, x[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q[E];
            fetch_q(j, q);
            zero_chunk<E>(s[j]);
#pragma unroll
            for (int e = 0; e < E; ++e) acc_step<M_L2>(s[j][e], dummy, q[e], x[j][e]);
        }
    }
    float p[E];
#pragma unroll
    for (int e = 0; e < E; ++e) p[e] = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
    const float partial = reduce_slots<E>(p); // identical on all lanes of the group
    rejected = partial > thr;
    const bool need = active && !rejected;
    if (wave_ballot(need) == 0ull) return __builtin_inff();
    if constexpr (NB64 > 4) { // d = 128: the second 64-block continues the four accumulators before the combine
        float x[4][E];
#pragma unroll
        for (int j = 0; j < 4; ++j) fetch(4 + j, need, x[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float q[E];
            fetch_q(4 + j, q);
#pragma unroll
            for (int e = 0; e < E; ++e) acc_step<M_L2>(s[j][e], dummy, q[e], x[j][e]);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) s[0][e] = (s[0][e] + s[1][e]) + (s[2][e] + s[3][e]);
    if constexpr (B1 > NB64) {
        float x[B1 - NB64][E];
#pragma unroll
        for (int i = NB64; i < B1; ++i) fetch(i, need, x[i - NB64]);
#pragma unroll
        for (int i = NB64; i < B1; ++i) {
            float q[E];
            fetch_q(i, q);
#pragma unroll
            for (int e = 0; e < E; ++e) acc_step<M_L2>(s[0][e], dummy, q[e], x[i - NB64][e]);
        }
    }
    const float full = reduce_slots<E>(s[0]);
    return need ? full : __builtin_inff();
}

// Block-cyclic float-path distance (alternative lane mapping; same arithmetic, operation for operation, as
// float_row_distance).  Four adjacent lanes own one row; lane j owns the WHOLE accumulator vector s_j (all 16
// lane-slots) and streams the 16-element blocks 4t+j, so every access is whole aligned dwordx4 loads (int8: 1, f16: 2,
// f32: 4 per block) and the four lanes of a group read 64*sizeof(T) contiguous bytes per step.  (s0+s1)+(s2+s3) is
// formed with two quad DPP exchanges per lane-slot; the remaining 16-blocks and the masked tail go into s0 on all four
// lanes redundantly (same address -> one request), and the _mm512_reduce_add_ps tree is lane-local.
// Used for f16 and 8-bit rows (see use_block_cyclic below).
template <int DT> struct BlockRaw { uint4 v[(16 * Elem<DT>::BYTES) / 16]; };

template <int DT> __device__ __forceinline__ void block_to_float(const BlockRaw<DT>& raw, float (&x)[16]) {
    if constexpr (DT == DT_F32) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            x[4 * i + 0] = __uint_as_float(raw.v[i].x); x[4 * i + 1] = __uint_as_float(raw.v[i].y);
            x[4 * i + 2] = __uint_as_float(raw.v[i].z); x[4 * i + 3] = __uint_as_float(raw.v[i].w);
        }
    } else if constexpr (DT == DT_F16) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const half8_t h = __builtin_bit_cast(half8_t, raw.v[i]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[8 * i + e] = static_cast<float>(h[e]);
        }
    } else {
        const uint32_t w[4] = {raw.v[0].x, raw.v[0].y, raw.v[0].z, raw.v[0].w};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const uint32_t byte = (w[e >> 2] >> (8 * (e & 3))) & 0xFFu;
            if constexpr (DT == DT_I8) x[e] = static_cast<float>(static_cast<int>(static_cast<int8_t>(byte)));
            else x[e] = static_cast<float>(byte);
        }
    }
}

template <int DT, int METRIC, int DS, bool DECOMP>
__device__ __forceinline__ void float_row_distance_bc(
    const unsigned char* __restrict__ row, bool active, const float* __restrict__ qf, int d_runtime, int sub,
    float sq_scale, float sq_bias, float& out_main, float& out_nrm
) {
    constexpr int BB = 16 * Elem<DT>::BYTES; // bytes per 16-element block
    const int d = DS > 0 ? DS : d_runtime;
    const int nb = d >> 4;
    const int n64 = nb >> 2;
    float s[16], n[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { s[e] = 0.0f; n[e] = 0.0f; }

    auto load_block = [&](int c) -> BlockRaw<DT> {
        BlockRaw<DT> raw;
#pragma unroll
        for (int i = 0; i < BB / 16; ++i) raw.v[i] = *reinterpret_cast<const uint4*>(row + BB * c + 16 * i); // inactive groups read row 0
        return raw;
    };
    auto accumulate = [&](int c, const BlockRaw<DT>& raw, int nvalid) {
        float x[16];
        if constexpr (DT != DT_I8 && DT != DT_U8) block_to_float<DT>(raw, x);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 qv = *reinterpret_cast<const float4*>(qf + 16 * c + 4 * g4);
            const float q[4] = {qv.x, qv.y, qv.z, qv.w};
            if constexpr (DT == DT_I8 || DT == DT_U8) { // widen one 4-byte word at a time: short live ranges
                const uint32_t w = g4 == 0 ? raw.v[0].x : g4 == 1 ? raw.v[0].y : g4 == 2 ? raw.v[0].z : raw.v[0].w;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint32_t byte = (w >> (8 * b)) & 0xFFu;
                    if constexpr (DT == DT_I8) x[4 * g4 + b] = static_cast<float>(static_cast<int>(static_cast<int8_t>(byte)));
                    else x[4 * g4 + b] = static_cast<float>(byte);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int e = 4 * g4 + b;
                float xv = x[e];
                if constexpr (DECOMP) {
                    const float t = sq_scale * xv;
                    xv = t + sq_bias;
                }
                float ts = s[e], tn = n[e];
                acc_step<METRIC>(ts, tn, q[b], xv);
                const bool in = e < nvalid;
                s[e] = in ? ts : s[e];
                if constexpr (METRIC == M_COS) n[e] = in ? tn : n[e];
            }
        }
    };

    if (n64 > 0) {
        if constexpr (DS > 0) {
            constexpr int N64 = DS >> 6;
#ifndef SVSX_BC_CH_COS
#define SVSX_BC_CH_COS 4
#endif
#ifndef SVSX_BC_CH_OTHER
#define SVSX_BC_CH_OTHER 4
#endif
            constexpr int MAXQ = METRIC == M_COS ? SVSX_BC_CH_COS : SVSX_BC_CH_OTHER; // dwordx4 loads in flight per lane
            constexpr int MAXCH = (MAXQ / (BB / 16)) > 0 ? MAXQ / (BB / 16) : 1;
            constexpr int CH = N64 < MAXCH ? (N64 > 0 ? N64 : 1) : MAXCH;
#pragma unroll 1
            for (int t0 = 0; t0 < N64; t0 += CH) { // rolled: CH blocks in flight per lane, registers bounded
                BlockRaw<DT> raw[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) if (t0 + i < N64) raw[i] = load_block(4 * (t0 + i) + sub);
#pragma unroll
                for (int i = 0; i < CH; ++i) if (t0 + i < N64) accumulate(4 * (t0 + i) + sub, raw[i], 16);
            }
        } else {
            constexpr int CH = 4 / (BB / 16) > 0 ? 4 / (BB / 16) : 1;
            int t = 0;
            for (; t + CH <= n64; t += CH) {
                BlockRaw<DT> raw[CH];
#pragma unroll
                for (int i = 0; i < CH; ++i) raw[i] = load_block(4 * (t + i) + sub);
#pragma unroll
                for (int i = 0; i < CH; ++i) accumulate(4 * (t + i) + sub, raw[i], 16);
            }
            for (; t < n64; ++t) {
                const BlockRaw<DT> raw = load_block(4 * t + sub);
                accumulate(4 * t + sub, raw, 16);
            }
        }
        // s0 = (s0 + s1) + (s2 + s3) per lane-slot; afterwards every lane of the group holds the combined vector
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] = add_xor2f(add_xor1f(s[e]));
            if constexpr (METRIC == M_COS) n[e] = add_xor2f(add_xor1f(n[e]));
        }
    }
    for (int b = 4 * n64; b < nb; ++b) {
        const BlockRaw<DT> raw = load_block(b);
        accumulate(b, raw, 16);
    }
    if ((d & 15) != 0) {
        const BlockRaw<DT> raw = load_block(nb);
        accumulate(nb, raw, d & 15);
    }
    auto tree = [](float (&v)[16]) -> float {
        float t[8], u[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[j + 8] + v[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = t[j + 4] + t[j];
        const float w0 = u[0] + u[2];
        const float w1 = u[1] + u[3];
        return w0 + w1;
    };
    out_main = tree(s);
    if constexpr (METRIC == M_COS) out_nrm = tree(n);
    else out_nrm = 0.0f;
}

// Block-cyclic mapping on EIGHT lanes per row (round 5; 8-bit rows of a static dimension that is a multiple of 64: the 768-byte rows of config 4).
// Lane (j = sub & 3, h = sub >> 2) owns lane-slots 8h .. 8h + 7 of the accumulator vector s_j and streams the 16-element blocks 4t + j: eight
// bytes per block, so ALL loads of a row (DS / 64 x 8 bytes per lane: 24 registers at d = 768) are in flight before the first use -- the
// four-lane form keeps four 16-byte loads in flight and pays three dependent HBM round trips per 768-byte row, the largest term of config 4's
// phase profile (profiles/r05_notes.md) -- and a pass of 8 rows costs half the vector instructions of a pass of 16, which matters when an
// expansion leaves ~9 candidates per adjacency chunk.  Arithmetic: every lane-slot is its own fma chain over t (acc_step, the reference's
// operation order), (s0 + s1) + (s2 + s3) across j by the same two quad exchanges, then the reduce tree t[e] = v[e + 8] + v[e] (the partner
// half sits four lanes up: one row shift), u, w, sum lane-local -- operation for operation float_row_distance_bc, same bits.
template <int CTRL> __device__ __forceinline__ float add_dpp(float v) {
    return opaque_f(v + bit_cast_<float>(__builtin_amdgcn_mov_dpp(bit_cast_<int>(v), CTRL, 0xf, 0xf, true)));
}

// the row's loads (all of them) and the arithmetic are separate calls: the search kernel requests the rows of pass p + 1 before it computes
// pass p (Bc8Row<DS> = the 8-byte pieces of one lane)
template <int DS> struct Bc8Row { uint2 raw[DS >> 6]; };

template <int DS> __device__ __forceinline__ Bc8Row<DS> bc8_load(const unsigned char* __restrict__ row, int sub) {
    static_assert(DS >= 64 && (DS & 63) == 0, "whole 64-element blocks");
    const unsigned char* lane_row = row + 16 * (sub & 3) + 8 * (sub >> 2);
    Bc8Row<DS> r;
#pragma unroll
    for (int t = 0; t < (DS >> 6); ++t) r.raw[t] = *reinterpret_cast<const uint2*>(lane_row + 64 * t); // inactive groups read row 0
    return r;
}

template <int DT, int METRIC, int DS, bool DECOMP>
__device__ __forceinline__ void bc8_compute(const Bc8Row<DS>& r, const float* __restrict__ qf, int sub, float sq_scale, float sq_bias, float& out_main, float& out_nrm) {
    static_assert(DT == DT_I8 || DT == DT_U8, "8-bit rows");
    constexpr int N64 = DS >> 6;
    const float* lane_q = qf + 16 * (sub & 3) + 8 * (sub >> 2);
    float s[8], n[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.0f; n[e] = 0.0f; }
#pragma unroll
    for (int t = 0; t < N64; ++t) {
#pragma unroll
        for (int g4 = 0; g4 < 2; ++g4) {
            const float4 qv = *reinterpret_cast<const float4*>(lane_q + 64 * t + 4 * g4);
            const float q[4] = {qv.x, qv.y, qv.z, qv.w};
            const uint32_t w = g4 == 0 ? r.raw[t].x : r.raw[t].y;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t byte = (w >> (8 * b)) & 0xFFu;
                float xv;
                if constexpr (DT == DT_I8) xv = static_cast<float>(static_cast<int>(static_cast<int8_t>(byte)));
                else xv = static_cast<float>(byte);
                if constexpr (DECOMP) {
                    const float tt = sq_scale * xv;
                    xv = tt + sq_bias;
                }
                acc_step<METRIC>(s[4 * g4 + b], n[4 * g4 + b], q[b], xv);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { // (s0 + s1) + (s2 + s3): j = lane & 3, two quad exchanges
        s[e] = add_xor2f(add_xor1f(s[e]));
        if constexpr (METRIC == M_COS) n[e] = add_xor2f(add_xor1f(n[e]));
    }
    auto tree = [](float (&v)[8]) -> float {
        float t[8], u[4];
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = add_dpp<0x104>(v[e]); // row_shl:4 -- lane i takes lane i + 4: v[e + 8] + v[e] in the h = 0 lanes (the others are not used)
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = t[e + 4] + t[e];
        const float w0 = u[0] + u[2];
        const float w1 = u[1] + u[3];
        return w0 + w1;
    };
    out_main = tree(s);
    if constexpr (METRIC == M_COS) out_nrm = tree(n);
    else out_nrm = 0.0f;
}

template <int DT, int METRIC, int DS, bool DECOMP>
__device__ __forceinline__ void float_row_distance_bc8(
    const unsigned char* __restrict__ row, const float* __restrict__ qf, int sub, float sq_scale, float sq_bias, float& out_main, float& out_nrm
) {
    const Bc8Row<DS> r = bc8_load<DS>(row, sub);
    bc8_compute<DT, METRIC, DS, DECOMP>(r, qf, sub, sq_scale, sq_bias, out_main, out_nrm);
}

// Which 8-bit float-path kernels take the eight-lane mapping
template <int DT, int DS> constexpr bool use_bc8() { return (DT == DT_I8 || DT == DT_U8) && DS >= 256 && (DS & 63) == 0; }

#ifndef SVSX_F16_BLOCK_CYCLIC
#define SVSX_F16_BLOCK_CYCLIC 1
#endif
// Which mapping each row type uses on the float path (measured on MI355X, round 1):
//   f32      slot-cyclic (float_row_distance, G=4): 6 dwordx4 per lane at d=96, 61 VGPRs, 8 waves/SIMD
//   f16      block-cyclic: 58-60 VGPRs / 8 waves per SIMD at d=128 vs 92 / 5 for the G=2 slot-cyclic mapping;
//            1M x 128 f16 IP: 3.33 M QPS vs 2.85 M (A/B with -DSVSX_F16_BLOCK_CYCLIC=0)
//   int8/u8  block-cyclic: 16-B instead of 8-B loads; 500k x 768 SQ-int8 cosine: 754 k QPS vs 377 k
template <int DT> constexpr bool use_block_cyclic() {
    return DT == DT_I8 || DT == DT_U8 || (DT == DT_F16 && SVSX_F16_BLOCK_CYCLIC != 0);
}

// Exact int32 sums of one row against the query bytes in LDS: ab = sum(q*x), bb = sum(x*x).
// Rows and the LDS query are zero padded to 16 elements, 4 lanes share a row (16-B chunks, any order).
template <int DT>
__device__ __forceinline__ void int_row_sums(
    const unsigned char* __restrict__ row, bool active, const uint32_t* __restrict__ qw, int d, int sub4, int& ab, int& bb
) {
    const int nch = (d + 15) >> 4;
    int sab = 0, sbb = 0;
    auto dots = [&](const uint4& x, const uint4& q) {
        if constexpr (DT == DT_I8) {
            sab = __builtin_amdgcn_sdot4(q.x, x.x, sab, false);
            sab = __builtin_amdgcn_sdot4(q.y, x.y, sab, false);
            sab = __builtin_amdgcn_sdot4(q.z, x.z, sab, false);
            sab = __builtin_amdgcn_sdot4(q.w, x.w, sab, false);
            sbb = __builtin_amdgcn_sdot4(x.x, x.x, sbb, false);
            sbb = __builtin_amdgcn_sdot4(x.y, x.y, sbb, false);
            sbb = __builtin_amdgcn_sdot4(x.z, x.z, sbb, false);
            sbb = __builtin_amdgcn_sdot4(x.w, x.w, sbb, false);
        } else {
            sab = (int)__builtin_amdgcn_udot4(q.x, x.x, (unsigned)sab, false);
            sab = (int)__builtin_amdgcn_udot4(q.y, x.y, (unsigned)sab, false);
            sab = (int)__builtin_amdgcn_udot4(q.z, x.z, (unsigned)sab, false);
            sab = (int)__builtin_amdgcn_udot4(q.w, x.w, (unsigned)sab, false);
            sbb = (int)__builtin_amdgcn_udot4(x.x, x.x, (unsigned)sbb, false);
            sbb = (int)__builtin_amdgcn_udot4(x.y, x.y, (unsigned)sbb, false);
            sbb = (int)__builtin_amdgcn_udot4(x.z, x.z, (unsigned)sbb, false);
            sbb = (int)__builtin_amdgcn_udot4(x.w, x.w, (unsigned)sbb, false);
        }
    };
    int m = sub4;
    for (; m + 12 < nch; m += 16) { // four 16-B loads in flight per lane
        uint4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = *reinterpret_cast<const uint4*>(row + 16 * (m + 4 * i)); // inactive groups read row 0
#pragma unroll
        for (int i = 0; i < 4; ++i) dots(x[i], *reinterpret_cast<const uint4*>(qw + 4 * (m + 4 * i)));
    }
    for (; m < nch; m += 4) {
        const uint4 x = *reinterpret_cast<const uint4*>(row + 16 * m);
        dots(x, *reinterpret_cast<const uint4*>(qw + 4 * m));
    }
    sab = add_xor2i(add_xor1i(sab));
    sbb = add_xor2i(add_xor1i(sbb));
    ab = sab;
    bb = sbb;
}

// The same sums on EIGHT lanes per row (round 6; 8-bit rows of a static dimension that is a multiple of 64: the 768-byte rows of SURVEY 8d
// C4(b)): every lane holds 8 bytes of each 64-byte block (Bc8Row, the mapping of the float path's eight-lane form -- for exact integer sums any
// mapping gives the same value), so ALL loads of a row are in flight before the first use and the search kernel can request the rows of the next
// pass before it computes this one.  The four-lane form above pays three dependent round trips per 768-byte row.  Results in the lanes sub < 4.
template <int DT, int DS>
__device__ __forceinline__ void int_row_sums8(const Bc8Row<DS>& r, const uint32_t* __restrict__ qw, int sub, int& ab, int& bb) {
    static_assert(DT == DT_I8 || DT == DT_U8, "8-bit rows");
    const unsigned char* lane_q = reinterpret_cast<const unsigned char*>(qw) + 16 * (sub & 3) + 8 * (sub >> 2);
    int sab = 0, sbb = 0;
#pragma unroll
    for (int t = 0; t < (DS >> 6); ++t) {
        const uint2 q = *reinterpret_cast<const uint2*>(lane_q + 64 * t);
        const uint2 x = r.raw[t];
        if constexpr (DT == DT_I8) {
            sab = __builtin_amdgcn_sdot4(q.x, x.x, sab, false);
            sab = __builtin_amdgcn_sdot4(q.y, x.y, sab, false);
            sbb = __builtin_amdgcn_sdot4(x.x, x.x, sbb, false);
            sbb = __builtin_amdgcn_sdot4(x.y, x.y, sbb, false);
        } else {
            sab = (int)__builtin_amdgcn_udot4(q.x, x.x, (unsigned)sab, false);
            sab = (int)__builtin_amdgcn_udot4(q.y, x.y, (unsigned)sab, false);
            sbb = (int)__builtin_amdgcn_udot4(x.x, x.x, (unsigned)sbb, false);
            sbb = (int)__builtin_amdgcn_udot4(x.y, x.y, (unsigned)sbb, false);
        }
    }
    sab = add_xor2i(add_xor1i(sab));
    sbb = add_xor2i(add_xor1i(sbb));
    ab = opaque_i(sab + __builtin_amdgcn_mov_dpp(sab, 0x104, 0xf, 0xf, true)); // row_shl:4 -- lane i takes lane i + 4: the other half of the row
    bb = opaque_i(sbb + __builtin_amdgcn_mov_dpp(sbb, 0x104, 0xf, 0xf, true));
}

// SQ compress (scalar.h:37-42): clamp(round((v - bias) / scale), MIN, MAX); std::round = half away from zero.
template <int DT> __device__ __forceinline__ int sq_compress(float v, float scale, float bias) {
    float r = roundf((v - bias) / scale);
    constexpr float lo = DT == DT_I8 ? -128.0f : 0.0f;
    constexpr float hi = DT == DT_I8 ? 127.0f : 255.0f;
    r = r < lo ? lo : r;
    r = r > hi ? hi : r;
    return static_cast<int>(r);
}

// Per-query constants produced by fix_argument (concepts/distance.h:88-92).
struct QueryFix {
    float norm;    // cosine: ||q||                                (cosine.h:117-119, distance_core.h:45-66)
    float offset;  // SQ IP: bias * sum(q)                         (scalar.h:122-130)
    int qq;        // int L2: sum(q*q)
};

// Load query `qptr` into the wave's LDS slice and run fix_argument (greedy_search.h:140):
//   float paths: widened exactly to f32 (zero padded to 16) in qf;  int paths: bytes (zero padded) in qw.
// Float16::operator float (lib/float16.h:45-52,103): the reference's SCALAR float16 -> float32 conversion, "without norm/denorm": denormals
// become (signed) zero and the inf / NaN exponent yields finite values.  Its SIMD kernels convert exactly (vcvtph2ps) -- see prepare_query.
__device__ __forceinline__ float half_to_float_scalar(uint16_t x) {
    const uint32_t e = (static_cast<uint32_t>(x) & 0x7C00u) >> 10;
    const uint32_t m = (static_cast<uint32_t>(x) & 0x03FFu) << 13;
    return __uint_as_float(((static_cast<uint32_t>(x) & 0x8000u) << 16) | (e != 0u ? (((e + 112u) << 23) | m) : 0u));
}

template <int DT, int METRIC, int QMODE>
__device__ __forceinline__ QueryFix prepare_query(
    const unsigned char* __restrict__ qptr, int q_dtype, int d, int dpad, float* qf, uint32_t* qw, float sq_scale,
    float sq_bias, uint32_t lane
) {
    constexpr bool INTPATH = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    QueryFix fx;
    fx.norm = 0.0f;
    fx.offset = 0.0f;
    fx.qq = 0;
    // A float16 query element is widened in one of two ways by the reference.  Its SIMD kernels (QM_FLOAT) load it with vcvtph2ps: exact.
    // Everything SCALAR goes through Float16::operator float (lib/float16.h:45-52; half_to_float_scalar above): norm() of the query
    // (distance_core.h:45-58), the generic_* loops (QM_SCALAR), and the scalar-quantized distances, whose fix_argument compresses / copies the
    // query element by element (scalar.h:75-82,123-125 -> detail::compress / set_datum -> relaxed_narrow).  Pinned against the reference's
    // own class.
    constexpr bool SCALAR_WIDEN = QMODE == QM_SCALAR || QMODE == QM_SQ_L2 || QMODE == QM_SQ_IP || QMODE == QM_SQ_COS;
    auto query_elem = [&](int i) -> float { // widening of query element i
        switch (q_dtype) {
            case DT_F32: return reinterpret_cast<const float*>(qptr)[i];
            case DT_F16:
                if constexpr (SCALAR_WIDEN) return half_to_float_scalar(reinterpret_cast<const uint16_t*>(qptr)[i]);
                else return static_cast<float>(reinterpret_cast<const _Float16*>(qptr)[i]);
            case DT_I8: return static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(qptr)[i]));
            default: return static_cast<float>(static_cast<unsigned>(qptr[i]));
        }
    };
    if constexpr (!INTPATH) {
        for (int i = lane; i < dpad; i += 64) qf[i] = i < d ? query_elem(i) : 0.0f;
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if constexpr (METRIC == M_COS) {
            // norm() (distance_core.h:45-66): accum += x*x sequentially in f32, multiply and add rounded separately
            float accum = 0.0f;
            if (!SCALAR_WIDEN && q_dtype == DT_F16) { // the norm of a float16 query takes the scalar conversion even where the kernel's loads do not
                for (int i = 0; i < d; ++i) {
                    const float v = half_to_float_scalar(reinterpret_cast<const uint16_t*>(qptr)[i]);
                    const float p = v * v;
                    accum = accum + p;
                }
            } else {
                for (int i = 0; i < d; ++i) {
                    const float v = qf[i];
                    const float p = v * v;
                    accum = accum + p;
                }
            }
            fx.norm = sqrtf(accum);
        }
        if constexpr (QMODE == QM_SQ_IP) {
            // std::reduce (libstdc++ 11 <numeric>:299-309): groups of four, (q0+q1)+(q2+q3) added to the running sum
            float init = 0.0f;
            int i = 0;
            for (; d - i >= 4; i += 4) {
                const float v1 = qf[i] + qf[i + 1];
                const float v2 = qf[i + 2] + qf[i + 3];
                const float v3 = v1 + v2;
                init = init + v3;
            }
            for (; i < d; ++i) init = init + qf[i];
            fx.offset = sq_bias * init;
        }
    } else {
        unsigned char* qb = reinterpret_cast<unsigned char*>(qw);
        for (int i = lane; i < dpad; i += 64) {
            int v = 0;
            if (i < d) {
                if constexpr (QMODE == QM_SQ_L2) v = sq_compress<DT>(query_elem(i), sq_scale, sq_bias);
                else v = (DT == DT_I8) ? static_cast<int>(reinterpret_cast<const int8_t*>(qptr)[i]) : static_cast<int>(qptr[i]);
            }
            qb[i] = static_cast<unsigned char>(v);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        // sum(q*q) (exact) and, for cosine, the sequential f32 norm of the integer query
        int part = 0;
        for (int m = lane; m < dpad / 4; m += 64) {
            const uint32_t w = qw[m];
            if constexpr (DT == DT_I8) part = __builtin_amdgcn_sdot4(w, w, part, false);
            else part = (int)__builtin_amdgcn_udot4(w, w, (unsigned)part, false);
        }
        for (int off = 32; off >= 1; off >>= 1) part += __shfl_xor(part, off);
        fx.qq = part;
        if constexpr (METRIC == M_COS) {
            float accum = 0.0f;
            for (int i = 0; i < d; ++i) {
                const int v = (DT == DT_I8) ? static_cast<int>(static_cast<int8_t>(qb[i])) : static_cast<int>(qb[i]);
                accum = accum + static_cast<float>(v * v);
            }
            fx.norm = sqrtf(accum);
        }
    }
    return fx;
}

// The reference's scalar fallback loops (QM_SCALAR): ONE lane owns one row -- a single float accumulator that takes the elements in order
// cannot be split across lanes -- so a pass evaluates 64 rows.  `result += temp * temp` is one fma: the loops are instantiated in the kernel
// translation unit, built -march=cascadelake with GCC's default -ffp-contract=fast (DESIGN.md section 2 has the argument; round 6: pinned by a
// run of the loop bodies compiled with that unit's flags over the reference's own Float16 class).  Rows and the LDS query are zero padded to 16 elements; the padding adds fma(0, 0, x) = x.
template <int DT, int METRIC>
__device__ __forceinline__ void scalar_row_distance(const unsigned char* __restrict__ row, const float* __restrict__ qf, int d, float& out_main, float& out_nrm) {
    float result = 0.0f, accum = 0.0f;
    const int nb = (d + 15) >> 4;
    for (int b = 0; b < nb; ++b) {
        float x[16];
        BlockRaw<DT> raw;
#pragma unroll
        for (int i = 0; i < (16 * Elem<DT>::BYTES) / 16; ++i) raw.v[i] = *reinterpret_cast<const uint4*>(row + 16 * Elem<DT>::BYTES * b + 16 * i); // inactive lanes read row 0
        if constexpr (DT == DT_F16) { // static_cast<float>(b[i]) of the scalar loops = Float16::operator float, not the exact conversion
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const uint32_t w[4] = {raw.v[i].x, raw.v[i].y, raw.v[i].z, raw.v[i].w};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    x[8 * i + 2 * t] = half_to_float_scalar(static_cast<uint16_t>(w[t] & 0xFFFFu));
                    x[8 * i + 2 * t + 1] = half_to_float_scalar(static_cast<uint16_t>(w[t] >> 16));
                }
            }
        } else {
            block_to_float<DT>(raw, x);
        }
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 qv = *reinterpret_cast<const float4*>(qf + 16 * b + 4 * g4); // the same address in every lane: LDS broadcast
            const float q[4] = {qv.x, qv.y, qv.z, qv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float xi = x[4 * g4 + e];
                if constexpr (METRIC == M_L2) {
                    const float t = q[e] - xi;
                    result = __builtin_fmaf(t, t, result);
                } else {
                    if constexpr (METRIC == M_COS) accum = __builtin_fmaf(xi, xi, accum);
                    result = __builtin_fmaf(q[e], xi, result);
                }
            }
        }
    }
    out_main = result;
    out_nrm = accum;
}

// Final distance value for one candidate from the reduced sums.
template <int METRIC, int QMODE>
__device__ __forceinline__ float finish_distance(float fmain, float fnrm, int ab, int bb, const QueryFix& fx, float sq_scale) {
    if constexpr (QMODE == QM_INT || QMODE == QM_SQ_L2) {
        if constexpr (METRIC == M_L2) {
            // sum((q-x)^2) == qq - 2ab + bb exactly (two's complement ring), then narrow_cast<float>
            const int l2 = fx.qq - 2 * ab + bb;
            const float v = static_cast<float>(l2);
            if constexpr (QMODE == QM_SQ_L2) return (sq_scale * sq_scale) * v; // scalar.h:88-94
            return v;
        } else if constexpr (METRIC == M_IP) {
            return static_cast<float>(ab);
        } else {
            const float b_norm = sqrtf(static_cast<float>(bb)); // cosine.h:284-287
            return static_cast<float>(ab) / (fx.norm * b_norm);
        }
    } else {
        if constexpr (METRIC == M_COS) {
            return fmain / (sqrtf(fnrm) * fx.norm); // cosine.h:334-335
        } else if constexpr (QMODE == QM_SQ_IP) {
            const float t = sq_scale * fmain; // scalar.h:134-141
            return t + fx.offset;
        } else {
            return fmain;
        }
    }
}

template <bool GREATER> __device__ __forceinline__ bool cmpf(float a, float b) { return GREATER ? (a > b) : (a < b); }

// ---------------------------------------------------------------------------------------------------------
// Sorted candidate buffer of one query, resident in LDS, owned by ONE wavefront (search_buffer.h:104-497).
// entry = {dist bits, id | visited<<31}; capacity+1 entries like the reference (spare slot at [capacity]).
// LDS operations of a single wave execute in program order, so no barrier is needed; signal fences stop the
// compiler from reordering the cross-lane dependent accesses.
struct WaveBuffer {
    uint2* e;
    uint32_t size;
    uint32_t best_unvisited;
    uint32_t window;
    uint32_t capacity;
    bool hbm; // entries live in HBM (batched BatchIterator whose buffer outgrew one wave's LDS); constant false everywhere else
};

__device__ __forceinline__ void lds_order() { __atomic_signal_fence(__ATOMIC_SEQ_CST); }

// number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi.  (Written as popcount(mask & ((1 << lane) - 1)) the compiler
// keeps the 64-bit lane mask in a register pair for the whole kernel -- and spills it around the distance pass.)
__device__ __forceinline__ uint32_t prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// Ordering of the cross-lane dependent accesses to the buffer.  In LDS a wave's operations execute in program order and a compiler
// fence is all it takes; for a buffer in HBM the wave's own stores must have completed before other lanes reload them: a
// workgroup-scope fence (s_waitcnt vmcnt(0); the wave's lanes share one L1, so no invalidate is needed).
__device__ __forceinline__ void buf_order(const WaveBuffer& b) {
    if (b.hbm) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    else __atomic_signal_fence(__ATOMIC_SEQ_CST);
}

// SearchBuffer::insert_inner (search_buffer.h:353-403).  d / id are wave-uniform.  Returns true if inserted.
// `unordered`: a NaN distance has reached the buffer, so it is no longer partitioned by the comparison and "the insert position"
// is whatever std::lower_bound's probe sequence finds (libstdc++ bits/stl_algobase.h __lower_bound: half = len >> 1, probe
// first + half) -- replayed here step by step, followed by the reference's backward duplicate walk (search_buffer.h:377-389).
template <bool GREATER>
// `idmask`: the bits of an entry's second word that identify the row (the inline-record kernels keep a degree class above them; a row
// reached as an entry point carries none, so ids are compared below it).
__device__ __forceinline__ bool wave_insert(WaveBuffer& b, float d, uint32_t id, uint32_t lane, bool unordered = false, uint32_t idmask = kIdMask) {
    if (unordered) {
        uint32_t first = 0, len = b.size;
        while (len > 0u) {
            const uint32_t half = len >> 1, mid = first + half;
            const float ed = __uint_as_float(b.e[mid].x); // wave-uniform address: LDS broadcast
            if (!cmpf<GREATER>(d, ed)) {
                first = mid + 1u;
                len = len - half - 1u;
            } else {
                len = half;
            }
        }
        const uint32_t pos = first;
        if (pos != 0u) {
            uint32_t back = pos;
            do {
                --back;
                const uint2 en = b.e[back];
                if (cmpf<GREATER>(__uint_as_float(en.x), d)) break;
                if (((en.y ^ id) & idmask) == 0u) return false;
            } while (back != 0u);
        }
        buf_order(b);
        const uint32_t end = b.size;
        if (end > pos) {
            const uint32_t nchunk = (end - pos + 63u) >> 6;
            for (int c = static_cast<int>(nchunk) - 1; c >= 0; --c) {
                const uint32_t i = pos + (static_cast<uint32_t>(c) << 6) + lane;
                uint2 en = make_uint2(0, 0);
                if (i < end) en = b.e[i];
                buf_order(b);
                if (i < end) b.e[i + 1] = en;
                buf_order(b);
            }
        }
        if (lane == 0) b.e[pos] = make_uint2(__float_as_uint(d), id);
        buf_order(b);
        b.size = (b.size + 1 < b.capacity) ? b.size + 1 : b.capacity;
        b.best_unvisited = pos < b.best_unvisited ? pos : b.best_unvisited;
        return true;
    }
    if (b.size <= 128u) {
        // at most two entries per lane: one LDS read serves the position search, the duplicate scan AND the shift
        // (old[i] -> e[i + 1] straight from the registers: every read precedes every write, no chunk ordering needed)
        const uint32_t i0 = lane, i1 = lane + 64u;
        const bool v0 = i0 < b.size, v1 = i1 < b.size;
        uint2 e0 = make_uint2(0, 0), e1 = make_uint2(0, 0);
        if (v0) e0 = b.e[i0];
        if (v1) e1 = b.e[i1];
        const float d0 = __uint_as_float(e0.x), d1 = __uint_as_float(e1.x);
        const bool bf0 = v0 && !cmpf<GREATER>(d, d0);
        const bool bf1 = v1 && !cmpf<GREATER>(d, d1);
        // duplicate scan (search_buffer.h:377-389): walks back from the insert position over the entries that are not
        // strictly better than d, comparing ids only -- for ordinary values those are the entries equal to d, for a NaN
        // distance (every comparison false) all of them.  (Written as `== d || d is NaN`: the form with two more
        // compares per entry cost 5 % of the bench kernel.)
        const bool dnan = d != d;
        const bool same = (v0 && (d0 == d || dnan) && ((e0.y ^ id) & idmask) == 0u) || (v1 && (d1 == d || dnan) && ((e1.y ^ id) & idmask) == 0u);
        const unsigned long long m0 = wave_ballot(bf0), m1 = wave_ballot(bf1);
        if (wave_ballot(same) != 0ull) return false;
        const uint32_t pos = static_cast<uint32_t>(__popcll(m0) + __popcll(m1)); // sorted: the "before" entries are a prefix
        buf_order(b);
        if (v0 && i0 >= pos) b.e[i0 + 1u] = e0;
        if (v1 && i1 >= pos) b.e[i1 + 1u] = e1;
        if (lane == 0) b.e[pos] = make_uint2(__float_as_uint(d), id);
        buf_order(b);
        b.size = (b.size + 1 < b.capacity) ? b.size + 1 : b.capacity;
        b.best_unvisited = pos < b.best_unvisited ? pos : b.best_unvisited;
        return true;
    }
    // position after all elements that do not compare greater (lower_bound with !cmp(d, other)) + duplicate scan
    uint32_t pos = 0;
    bool dup = false;
    for (uint32_t base = 0; base < b.size; base += 64) {
        const uint32_t i = base + lane;
        const bool valid = i < b.size;
        uint2 en = make_uint2(0, 0);
        if (valid) en = b.e[i];
        const float ed = __uint_as_float(en.x);
        const bool before = valid && !cmpf<GREATER>(d, ed);
        const unsigned long long m = wave_ballot(before);
        const bool same = valid && (ed == d || d != d) && (((en.y ^ id) & idmask) == 0u); // ids only among the entries equal to d (all, if d is NaN)
        dup = dup || (wave_ballot(same) != 0ull);
        const uint32_t cnt = __popcll(m);
        pos += cnt;
        const uint32_t nvalid = (b.size - base) < 64u ? (b.size - base) : 64u;
        if (cnt < nvalid) break; // sorted: nothing further can be "before"
    }
    if (dup) return false;
    buf_order(b);
    // unsafe_insert: copy_backward [pos, size) one slot to the right, highest chunk first
    const uint32_t end = b.size;
    if (end > pos) {
        const uint32_t nchunk = (end - pos + 63u) >> 6;
        for (int c = static_cast<int>(nchunk) - 1; c >= 0; --c) {
            const uint32_t i = pos + (static_cast<uint32_t>(c) << 6) + lane;
            uint2 en = make_uint2(0, 0);
            if (i < end) en = b.e[i];
            buf_order(b);
            if (i < end) b.e[i + 1] = en;
            buf_order(b);
        }
    }
    if (lane == 0) b.e[pos] = make_uint2(__float_as_uint(d), id);
    buf_order(b);
    b.size = (b.size + 1 < b.capacity) ? b.size + 1 : b.capacity;
    b.best_unvisited = pos < b.best_unvisited ? pos : b.best_unvisited;
    return true;
}

// Batched form of insert_inner for one distance pass: every lane with `cand` holds one candidate (dist, id) -- lanes in
// ascending order = adjacency order -- and all of them are merged into the sorted buffer at once.  Equal to inserting them one
// after the other (search_buffer.h:342-403), because a sequence of bounded sorted inserts IS a stable merge truncated to the
// capacity: old entries keep their order and come before new entries of equal distance (upper-bound position), new entries of
// equal distance keep adjacency order, whatever ends at position >= capacity is dropped (a candidate that loses against a
// `worst` that improved inside the pass would have been dropped, or pushed out, anyway), and best_unvisited = min(old, position
// of the best new entry), which is the smallest position any of the sequential inserts reports.
// Preconditions: capacity <= 64 EPL (EPL entries per lane), no NaN distance in the buffer, candidate distances finite (empty slots
// are modelled as +-inf sentinels; comparisons must be a strict weak order).  Returns false -- with LDS untouched -- if a candidate duplicates a buffer entry (same id at equal
// distance, search_buffer.h:377-389); the caller then takes the sequential path for this pass.
// `pre` (optional): the lane's EPL entries as wave_buffer_prefetch read them earlier in the pass -- nothing writes the buffer between the two.
template <int EPL> __device__ __forceinline__ void wave_buffer_prefetch(const WaveBuffer& b, uint32_t lane, uint2 (&pre)[EPL]) {
    lane = opaque_u(lane);
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const uint32_t i = lane + 64u * t;
        pre[t] = b.e[i < b.capacity ? i : b.capacity]; // unconditional (slots beyond `size` are masked by the merge; [capacity] is the spare slot)
    }
}

template <bool GREATER, int EPL>
__device__ __forceinline__ bool wave_merge_insert(WaveBuffer& b, unsigned long long cand_mask, bool cand, float dist, uint32_t id, uint32_t lane, uint32_t& n_inserted,
                                                  float& worst_out, const uint2* pre = nullptr) { // cand_mask: the wave mask of `cand`
    // EPL buffer entries per lane (capacity <= 64 EPL).  Written to stay off the scalar unit (one SALU issue slot per CU per
    // cycle is the scarce resource of this kernel): no boolean mask algebra -- validity is folded into sentinel values, counters
    // advance with carry-in adds, the tie rule of the candidate ranks is one 64-bit compare of (order key, lane).
    constexpr float kPast = GREATER ? -__builtin_inff() : __builtin_inff(); // no candidate compares better than an empty slot
    lane = opaque_u(lane);
    uint2 e[EPL];
    float dv[EPL];
    uint32_t c[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const uint32_t i = lane + 64u * t;
        e[t] = make_uint2(__float_as_uint(kPast), kInvalidId);
        if (pre != nullptr) {
            const bool in = i < b.size;
            e[t].x = in ? pre[t].x : e[t].x;
            e[t].y = in ? pre[t].y : e[t].y;
        } else if (i < b.size) {
            e[t] = b.e[i];
        }
        dv[t] = __uint_as_float(e[t].x);
        c[t] = 0u;
    }
    // total-order key of my own distance (+0.0 first: -0 and +0 compare equal as floats), ascending = better first
    const uint32_t fb = __float_as_uint(dist + 0.0f);
    uint32_t key = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
    if (GREATER) key = ~key;
    const unsigned long long mykey = (static_cast<unsigned long long>(key) << 32) | lane;
    uint32_t rank = 0, minpos = 0xFFFFFFFFu;
    int oldle = 0;
    unsigned long long m = cand_mask;
    const uint32_t ncand = static_cast<uint32_t>(__popcll(m));
    while (m != 0ull) { // 9 vector instructions per candidate (the VALU issue port bounds this kernel)
        const int bl = __builtin_ctzll(m);
        m &= ~(1ull << bl);
        const float dj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), bl));
        const uint32_t kj = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(key), bl));
        uint32_t after = 0; // old entries (and empty slots) candidate j goes before: a suffix of the sorted buffer
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const bool lt = cmpf<GREATER>(dj, dv[t]);
            c[t] += lt ? 1u : 0u;
            after += static_cast<uint32_t>(__popcll(wave_ballot(lt)));
        }
        const uint32_t le = 64u * EPL - after; // old entries that stay before j (empty slots always count as "after")
        const unsigned long long kjl = (static_cast<unsigned long long>(kj) << 32) | static_cast<uint32_t>(bl);
        rank += kjl < mykey ? 1u : 0u; // new entries before mine: better, or equal and earlier in adjacency order
        // lane bl keeps its own count: a select under the one-bit mask the loop has in scalar registers anyway (round 4 used v_writelane,
        // whose lane select has to go through m0 on gfx9: save, set, wait states, write, restore -- five instructions)
        {
            uint32_t tmp;
            asm volatile("v_mov_b32 %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, %3" : "+v"(oldle), "=&v"(tmp) : "s"(le), "s"(1ull << bl));
        }
        minpos = le < minpos ? le : minpos;
    }
    // duplicates (search_buffer.h:377-389: same id at equal distance is not inserted): the same id gives the same distance bits, so a
    // duplicate needs an old entry of EQUAL distance, and the last old entry that stays before candidate j -- index le_j - 1 -- is one if
    // any is.  One lane-parallel LDS read per pass instead of an id compare per candidate and entry; any tie sends the pass down the
    // sequential path, which does the exact id walk.
    {   // every lane reads (lanes without a candidate: entry 0); the three single compares meet as masks
        const float prev = __uint_as_float(b.e[oldle > 0 ? oldle - 1 : 0].x);
        if ((wave_ballot(prev == dist) & wave_ballot(oldle > 0) & cand_mask) != 0ull) return false;
    }
    buf_order(b);
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const uint32_t i = lane + 64u * t, p = i + c[t];
        if (i < b.size && c[t] != 0u && p < b.capacity) b.e[p] = e[t];
    }
    const uint32_t pn = static_cast<uint32_t>(oldle) + rank;
    const bool placed = cand && pn < b.capacity;
    if (placed) b.e[pn] = make_uint2(__float_as_uint(dist), id);
    buf_order(b);
    n_inserted += static_cast<uint32_t>(__popcll(cand_mask & wave_ballot(pn < b.capacity)));
    b.size = (b.size + ncand < b.capacity) ? b.size + ncand : b.capacity;
    b.best_unvisited = minpos < b.best_unvisited ? minpos : b.best_unvisited;
    // the entry that ended at position capacity - 1: read back after the writes above (LDS is in order within a wave); the caller uses it
    // an adjacency block or a row wait later, so the round trip is never waited for (round 4 took it out of the merge's registers:
    // 17 instructions on the critical path of a lone wave)
    if (b.size == b.capacity) worst_out = __uint_as_float(b.e[b.capacity - 1u].x);
    return true;
}

// The same merge without a scalar loop over the candidates (round 3).  The loop form above walks the candidates one by one through the
// scalar unit (s_ff1 -> v_readlane -> v_cmp -> s_bcnt -> ...): ~150 cycles of dependent VALU <-> SALU hand-overs per candidate, ~2 000 cycles
// per expansion in the phase profile (profiles/r03_notes.md) -- the longest link of the per-expansion chain.  Here everything is lane-parallel:
//   1. the m candidates of the pass are compacted into LDS scratch in lane (= adjacency) order: {distance bits, order key}
//   2. every lane reads them back (broadcast reads, m <= 64 / G) and counts, for each of its EPL old entries, the candidates strictly
//      better (c), the bit-equal distances (tie -> sequential path, see below) and -- for its own candidate -- the candidates ranked before it
//   3. c is non-decreasing along the sorted buffer, so "old entries that stay before the r-th best candidate" le[r] = #{i : c_i <= r} is
//      the index of the first old entry with c_i > r: the lanes whose c steps up write it to LDS (c of the predecessor through one DPP shift)
//   4. candidate lanes read le[rank]; old entries move to i + c_i, new ones go to le[rank] + rank
// `scratch`: 48 words of LDS (16 x {dist, key} + 16 x le).  Same preconditions and the same result as wave_merge_insert; a candidate whose
// distance bits equal an old entry's sends the pass down the sequential path (a duplicate id has identical distance bits, so this is a
// superset of search_buffer.h:377-389's "same id at equal distance").
template <bool GREATER, int EPL>
__device__ __forceinline__ bool wave_merge_insert2(WaveBuffer& b, bool cand, float dist, uint32_t id, uint32_t lane, uint32_t& n_inserted, float& worst_out,
                                                   uint32_t* scratch) {
    constexpr float kPast = GREATER ? -__builtin_inff() : __builtin_inff(); // no candidate compares better than an empty slot
    uint2* cs = reinterpret_cast<uint2*>(scratch);
    uint32_t* le_s = scratch + 32;
    const unsigned long long cm = wave_ballot(cand);
    const uint32_t m = static_cast<uint32_t>(__popcll(cm));
    if (m > 16u) return false; // more candidates than scratch slots (cannot happen: 64 / G <= 16 rows per pass)
    const uint32_t slot = prefix_count(cm);
    // total-order key of my own distance (+0.0 first: -0 and +0 compare equal as floats), ascending = better first
    const uint32_t fb = __float_as_uint(dist + 0.0f);
    uint32_t key = (fb & 0x80000000u) ? ~fb : (fb | 0x80000000u);
    if (GREATER) key = ~key;
    if (cand) cs[slot] = make_uint2(__float_as_uint(dist), key);
    if (lane < 16u) le_s[lane] = b.size; // default: every old entry stays before the candidate
    uint2 e[EPL];
    float dv[EPL];
    uint32_t c[EPL];
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const uint32_t i = lane + 64u * t;
        e[t] = make_uint2(__float_as_uint(kPast), kInvalidId);
        if (i < b.size) e[t] = b.e[i];
        dv[t] = __uint_as_float(e[t].x);
        c[t] = 0u;
    }
    lds_order();
    const unsigned long long mykey = (static_cast<unsigned long long>(key) << 32) | slot;
    uint32_t rank = 0, tieacc = 0xFFFFFFFFu;
    for (uint32_t j = 0; j < m; ++j) { // uniform trip count; the reads are broadcasts
        const uint2 cj = cs[j];
        const float dj = __uint_as_float(cj.x);
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            c[t] += cmpf<GREATER>(dj, dv[t]) ? 1u : 0u;
            const uint32_t x = cj.x ^ e[t].x;
            tieacc = x < tieacc ? x : tieacc;
        }
        const unsigned long long kjl = (static_cast<unsigned long long>(cj.y) << 32) | j;
        rank += kjl < mykey ? 1u : 0u; // candidates before mine: better, or equal and earlier in adjacency order
    }
    if (wave_ballot(tieacc == 0u) != 0ull) return false;
    // le[r] for r in [c_{i-1}, c_i) is i: written by the lane that owns entry i (usually one r, at most m)
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const int carry = t == 0 ? 0 : __builtin_amdgcn_readlane(static_cast<int>(c[t - 1]), 63); // entry 64 t - 1 lives in lane 63
        uint32_t r = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(carry, static_cast<int>(c[t]), 0x138, 0xf, 0xf, false)); // wave_shr:1
        const uint32_t i = lane + 64u * t;
        while (wave_ballot(r < c[t]) != 0ull) {
            if (r < c[t]) le_s[r] = i;
            ++r;
        }
    }
    lds_order();
    uint32_t pn = 0xFFFFFFFFu;
    if (cand) pn = le_s[rank] + rank;
    const uint32_t minpos = m != 0u ? le_s[0] : 0xFFFFFFFFu; // the best candidate (rank 0) lands at le[0]
    buf_order(b);
    const uint32_t old_size = b.size;
#pragma unroll
    for (int t = 0; t < EPL; ++t) {
        const uint32_t i = lane + 64u * t, p = i + c[t];
        if (i < old_size && c[t] != 0u && p < b.capacity) b.e[p] = e[t];
    }
    const bool placed = cand && pn < b.capacity;
    if (placed) b.e[pn] = make_uint2(__float_as_uint(dist), id);
    buf_order(b);
    n_inserted += static_cast<uint32_t>(__popcll(wave_ballot(placed)));
    b.size = (old_size + m < b.capacity) ? old_size + m : b.capacity;
    b.best_unvisited = minpos < b.best_unvisited ? minpos : b.best_unvisited;
    if (b.size == b.capacity) {
        // the entry that ends at position capacity - 1 is in some lane's registers: the caller's `worst` without an LDS round trip
        float wv = 0.0f;
        bool hit = false;
#pragma unroll
        for (int t = 0; t < EPL; ++t) {
            const uint32_t i = lane + 64u * t;
            if (i < old_size && i + c[t] + 1u == b.capacity) {
                wv = dv[t];
                hit = true;
            }
        }
        if (placed && pn + 1u == b.capacity) {
            wv = dist;
            hit = true;
        }
        const unsigned long long hm = wave_ballot(hit);
        if (hm != 0ull) worst_out = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), __builtin_ctzll(hm)));
        else worst_out = __uint_as_float(b.e[b.capacity - 1].x); // not reached: exactly one entry ends there
    }
    return true;
}

// SearchBuffer::next (search_buffer.h:294-304): returns the node id, marks it visited, advances best_unvisited.
__device__ __forceinline__ uint32_t wave_next(WaveBuffer& b, uint32_t lane, float& node_dist) {
    // the entry itself and the first 64 slots of the scan for the next unvisited one are requested together (round 5: one LDS round trip
    // instead of two dependent ones); slots beyond `upper` are read (clamped to the spare slot at [capacity]) and masked
    lane = opaque_u(lane);
    const uint32_t upper = b.size < b.window ? b.size : b.window;
    uint32_t nb = b.best_unvisited + 1;
    const uint2 en = b.e[b.best_unvisited];
    uint32_t flag = kVisitedBit;
    {
        const uint32_t i = nb + lane;
        if (!b.hbm) flag = b.e[i < b.capacity ? i : b.capacity].y;
        else if (i < upper) flag = b.e[i].y;
    }
    buf_order(b);
    if (lane == 0) b.e[b.best_unvisited].y = en.y | kVisitedBit;
    buf_order(b);
    node_dist = __uint_as_float(en.x);
    {
        const unsigned long long m = wave_ballot(nb + lane < upper) & wave_ballot(static_cast<int>(flag) >= 0); // bit 31 = visited
        if (m != 0ull) nb += static_cast<uint32_t>(__builtin_ctzll(m));
        else nb += 64;
        if (m == 0ull) {
            while (nb < upper) {
                const uint32_t i = nb + lane;
                bool unvisited = false;
                if (i < upper) unvisited = (b.e[i].y & kVisitedBit) == 0u;
                const unsigned long long m2 = wave_ballot(unvisited);
                if (m2 != 0ull) {
                    nb += static_cast<uint32_t>(__builtin_ctzll(m2));
                    break;
                }
                nb += 64;
            }
        }
    }
    b.best_unvisited = nb < upper ? nb : upper;
    // every lane read the same entry: say so (the block address, the chunk loop's bound and everything after them then stay in scalar registers;
    // left to the uniformity analysis, an LDS read is a per-lane value and the whole expansion turns into exec-masked vector code)
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(en.y & kIdMask)));
}

__device__ __forceinline__ bool wave_done(const WaveBuffer& b) {
    const uint32_t upper = b.size < b.window ? b.size : b.window;
    return b.best_unvisited == upper;
}

// SearchBuffer::sort (search_buffer.h:408) = std::sort(begin(), end(), compare_).  std::sort is not stable, and
// BatchIterator re-sorts an already sorted buffer on every resumed search (iterator.h:61), so entries of exactly
// equal distance come out in the order libstdc++'s algorithm leaves them.  To stay id-exact through ties this restates
// that algorithm (introsort: median of {first+1, mid, last-1} swapped to first, unguarded partition, ranges of <= 16
// left to a final insertion sort; the depth-limit heapsort cannot trigger on sorted input because ranges halve).
// Sequential by nature: lane 0 runs it in LDS; `stack` = 64 uint2 of LDS.  `presorted` callers skip it when no two
// adjacent distances are equal (a sorted range without ties is a fixed point).
template <bool GREATER> __device__ __forceinline__ bool entry_less(const uint2 x, const uint2 y) {
    return cmpf<GREATER>(__uint_as_float(x.x), __uint_as_float(y.x));
}

template <bool GREATER> __device__ inline void lane0_linear_insert(uint2* a, uint32_t last) {
    const uint2 val = a[last];
    uint32_t next = last - 1;
    while (entry_less<GREATER>(val, a[next])) {
        a[last] = a[next];
        last = next;
        --next;
    }
    a[last] = val;
}

template <bool GREATER> __device__ inline void lane0_insertion_sort(uint2* a, uint32_t first, uint32_t last) {
    if (first == last) return;
    for (uint32_t i = first + 1; i != last; ++i) {
        if (entry_less<GREATER>(a[i], a[first])) {
            const uint2 val = a[i];
            for (uint32_t j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = val;
        } else {
            lane0_linear_insert<GREATER>(a, i);
        }
    }
}

template <bool GREATER>
__device__ __noinline__ void wave_std_sort(WaveBuffer& b, uint2* stack, uint32_t lane, bool presorted) {
    const uint32_t n = b.size;
    if (n < 2) return;
    if (presorted) {
        bool tie = false;
        for (uint32_t base = 0; base + 1 < n; base += 64) {
            const uint32_t i = base + lane;
            bool t = false;
            if (i + 1 < n) t = b.e[i].x == b.e[i + 1].x || __uint_as_float(b.e[i].x) == __uint_as_float(b.e[i + 1].x);
            tie = tie || (wave_ballot(t) != 0ull);
        }
        if (!tie) return;
    }
    buf_order(b);
    if (lane == 0) {
        uint2* a = b.e;
        uint32_t sp = 0;
        stack[sp++] = make_uint2(0u, n);
        while (sp > 0) {
            const uint2 r = stack[--sp];
            const uint32_t first = r.x;
            uint32_t last = r.y;
            while (last - first > 16u) {
                const uint32_t ix = first + 1, iy = first + (last - first) / 2, iz = last - 1;
                const uint2 x = a[ix], y = a[iy], z = a[iz];
                uint32_t pick; // __move_median_to_first
                if (entry_less<GREATER>(x, y)) {
                    if (entry_less<GREATER>(y, z)) pick = iy;
                    else if (entry_less<GREATER>(x, z)) pick = iz;
                    else pick = ix;
                } else if (entry_less<GREATER>(x, z)) pick = ix;
                else if (entry_less<GREATER>(y, z)) pick = iz;
                else pick = iy;
                const uint2 pivot = a[pick];
                a[pick] = a[first];
                a[first] = pivot;
                uint32_t lo = first + 1, hi = last; // __unguarded_partition(first + 1, last, first)
                for (;;) {
                    while (entry_less<GREATER>(a[lo], pivot)) ++lo;
                    --hi;
                    while (entry_less<GREATER>(pivot, a[hi])) --hi;
                    if (!(lo < hi)) break;
                    const uint2 t = a[lo];
                    a[lo] = a[hi];
                    a[hi] = t;
                    ++lo;
                }
                if (sp < 64u) stack[sp++] = make_uint2(lo, last);
                last = lo;
            }
        }
        if (n > 16u) { // __final_insertion_sort
            lane0_insertion_sort<GREATER>(a, 0u, 16u);
            for (uint32_t i = 16u; i != n; ++i) lane0_linear_insert<GREATER>(a, i);
        } else {
            lane0_insertion_sort<GREATER>(a, 0u, n);
        }
    }
    buf_order(b);
}

} // namespace svsx
