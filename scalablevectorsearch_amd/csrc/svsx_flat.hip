// svsx_flat.hip -- exhaustive (flat) search: the dense, MFMA-shaped case of this code base.
//
// Reference operator: svs::index::flat::FlatIndex::search (index/flat/flat.h:322-460): every query against every
// vector, k best by the metric's comparator.  Here it is a two-stage pipeline:
//   1. flat_mfma_kernel   float16 "batched-query x candidate-block" products on the matrix cores
//                         (v_mfma_f32_32x32x16_f16, f32 accumulate): a wave owns 64 queries (two 32-row A tiles held in
//                         LDS) and streams a slice of the database in 32-row tiles whose B fragments are 16-B loads
//                         straight from the row-major f16 shadow copy; a per-query sorted shortlist of kk >= k
//                         candidates lives in LDS and is updated with ballot + wave-serial inserts (accepts become
//                         rare once the shortlists warm up: ~kk ln(n/kk) per query).
//   2. merge_kernel       (svsx_distance.hip) merges the per-slice shortlists.
//   3. rerank_kernel      recomputes the shortlisted distances with the SAME exact-order routines as the graph search
//                         (bit-identical to the reference's distance kernels) and emits the k best by
//                         (distance, then id).
// The f16 stage is only a filter (kk = min(128, max(4k, k+32)); the host limits k to 96 so that kk >= k + 32), so the final ids/distances do not depend on f16
// rounding unless a true neighbour is pushed out of the shortlist, which needs an f16 error larger than the gap to
// the kk-th candidate.  Values must fit float16 (|x| < 65504).
#include "svsx_device.h"
#include "svsx_kernels.h"
#include "svsx_flat_proof.h"

namespace svsx {

typedef float float16v_t __attribute__((ext_vector_type(16)));

// order-preserving float <-> uint32 map (for atomicMin on thresholds that may be negative)
__device__ __forceinline__ uint32_t enc_key(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float dec_key(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

// ---- shadow conversion: any stored row type (optionally SQ-decompressed) -> f16 rows padded to 16 + squared norms ----
template <int DT>
__global__ __launch_bounds__(64) void to_half_kernel(const void* __restrict__ src, size_t src_row_bytes, uint32_t n, uint32_t d, uint32_t dpad,
                                                     int decomp, float scale, float bias, _Float16* __restrict__ dst, float* __restrict__ norm2) {
    const uint32_t row = blockIdx.x;
    if (row >= n) return;
    const unsigned char* s = static_cast<const unsigned char*>(src) + static_cast<size_t>(row) * src_row_bytes;
    float acc = 0.0f;
    for (uint32_t i = threadIdx.x; i < dpad; i += 64) {
        float x = 0.0f;
        if (i < d) {
            if constexpr (DT == DT_F32) x = reinterpret_cast<const float*>(s)[i];
            else if constexpr (DT == DT_F16) x = static_cast<float>(reinterpret_cast<const _Float16*>(s)[i]);
            else if constexpr (DT == DT_I8) x = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(s)[i]));
            else x = static_cast<float>(static_cast<unsigned>(s[i]));
            if (decomp) x = scale * x + bias;
        }
        const _Float16 h = static_cast<_Float16>(x);
        dst[static_cast<size_t>(row) * dpad + i] = h;
        const float hf = static_cast<float>(h);
        acc = __builtin_fmaf(hf, hf, acc);
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (norm2 != nullptr && threadIdx.x == 0) norm2[row] = acc;
}

hipError_t launch_to_half(int dtype, const void* src, size_t src_row_bytes, uint32_t n, uint32_t d, uint32_t dpad, int decomp, float scale,
                          float bias, void* dst, float* norm2, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    _Float16* o = static_cast<_Float16*>(dst);
    switch (dtype) {
        case DT_F32: hipLaunchKernelGGL(to_half_kernel<DT_F32>, dim3(n), dim3(64), 0, stream, src, src_row_bytes, n, d, dpad, decomp, scale, bias, o, norm2); break;
        case DT_F16: hipLaunchKernelGGL(to_half_kernel<DT_F16>, dim3(n), dim3(64), 0, stream, src, src_row_bytes, n, d, dpad, decomp, scale, bias, o, norm2); break;
        case DT_I8: hipLaunchKernelGGL(to_half_kernel<DT_I8>, dim3(n), dim3(64), 0, stream, src, src_row_bytes, n, d, dpad, decomp, scale, bias, o, norm2); break;
        case DT_U8: hipLaunchKernelGGL(to_half_kernel<DT_U8>, dim3(n), dim3(64), 0, stream, src, src_row_bytes, n, d, dpad, decomp, scale, bias, o, norm2); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- stage 1: MFMA shortlist ---------------------------------------------------------------------------------------------
// key (smaller = better): L2: |x|^2 - 2 q.x   IP: -q.x   cosine: -q.x / |x|      (per-query constants dropped)
// KS > 0: compile-time number of 16-element k-steps (<= 8): the B fragments of the NEXT tile are loaded while the
// current tile's MFMAs run (software pipeline); KS == 0: runtime k loop.
template <int METRIC, int KS>
__global__ __launch_bounds__(64) void flat_mfma_kernel(const FlatArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (a.block_gate != nullptr && a.block_gate[blockIdx.x] == 0u) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t q0 = blockIdx.x * 64u;
    const uint32_t slice = blockIdx.y;
    const uint32_t dpad = a.dpad;
    const uint32_t ks = dpad >> 4;
    const uint32_t xstep = 32u * (a.tile_step ? a.tile_step : 1u);
    const uint32_t astride = dpad * 2u + 16u; // bytes; +16 shifts consecutive rows by 4 banks
    unsigned char* A = smem;                                         // runtime-k path only
    const uint32_t a_bytes = KS > 0 ? 0u : 64u * astride;
    float* lkey = reinterpret_cast<float*>(smem + a_bytes);          // 64 x kk
    uint32_t* lid = reinterpret_cast<uint32_t*>(lkey + 64u * a.kk); // 64 x kk
    float* tau = reinterpret_cast<float*>(lid + 64u * a.kk);        // 64
    const uint32_t kk = a.kk;

    // A tile: 64 query rows (f16, zero padded), coalesced 16-B copies (runtime-k path; the static path keeps its
    // fragments in registers, loaded straight from global memory below)
    if constexpr (KS == 0) {
        const uint32_t chunks_per_row = dpad / 8u;
        for (uint32_t c = lane; c < 64u * chunks_per_row; c += 64) {
            const uint32_t r = c / chunks_per_row, cc = c % chunks_per_row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + r < a.nq) v = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(a.qh) + (static_cast<size_t>(q0 + r) * dpad + cc * 8u) * 2u);
            *reinterpret_cast<uint4*>(A + r * astride + cc * 16u) = v;
        }
    }
    for (uint32_t i = lane; i < 64u * kk; i += 64) {
        lkey[i] = __builtin_inff();
        lid[i] = kInvalidId;
    }
    tau[lane] = __builtin_inff();
    lds_order();

    const uint32_t rows_per_slice = ((a.n + a.n_slices - 1) / a.n_slices + 31u) & ~31u;
    const uint32_t x_lo = slice * rows_per_slice;
    const uint32_t x_hi = (x_lo + rows_per_slice) < a.n ? (x_lo + rows_per_slice) : a.n;
    const uint32_t x_first = ((x_lo + xstep - 1u) / xstep) * xstep; // sampled tiles are absolute: tile index % tile_step == 0
    const uint32_t half = lane >> 5, col = lane & 31u;
    const _Float16* xh = static_cast<const _Float16*>(a.xh);

    float tl[2][16]; // register copies of tau for this lane's 32 (query tile, C/D row) cells
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) tl[qt][r] = __builtin_inff();
    }
    constexpr int NB = KS > 0 ? KS : 1;
    half8_t bcur[NB], bnext[NB];
    half8_t afr0[NB], afr1[NB]; // static KS: the wave's two A tiles stay in registers for the whole slice
    float xn_cur = 0.0f, xn_next = 0.0f;
    auto load_tile = [&](uint32_t xt, half8_t (&dst)[NB], float& xnv) {
        const uint32_t r = xt + col;
        const uint32_t rr = r < x_hi ? r : x_lo;
        const _Float16* p = xh + static_cast<size_t>(rr) * dpad + 8u * half;
#pragma unroll
        for (int s2 = 0; s2 < NB; ++s2) dst[s2] = *reinterpret_cast<const half8_t*>(p + 16 * s2);
        if constexpr (METRIC != M_IP) xnv = a.xnorm2[rr];
    };
    if constexpr (KS > 0) {
        const _Float16* qh = static_cast<const _Float16*>(a.qh);
        const uint32_t r0 = q0 + col, r1 = q0 + 32u + col;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) {
            half8_t z;
#pragma unroll
            for (int e = 0; e < 8; ++e) z[e] = static_cast<_Float16>(0.0f);
            afr0[s2] = r0 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r0) * dpad + 16 * s2 + 8u * half) : z;
            afr1[s2] = r1 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r1) * dpad + 16 * s2 + 8u * half) : z;
        }
        if (x_first < x_hi) load_tile(x_first, bcur, xn_cur);
    }
    uint32_t tile_no = 0;
    for (uint32_t x0 = x_first; x0 < x_hi; x0 += xstep, ++tile_no) {
        if (a.gtau != nullptr && (tile_no & 7u) == 0u) {
            // every slice of the same query block publishes its kk-th best key; the minimum over slices is a valid
            // pruning threshold for all of them (that slice alone holds kk better candidates)
            if (q0 + lane < a.nq) {
                const float g = dec_key(a.gtau[q0 + lane]);
                if (g < tau[lane]) tau[lane] = g;
            }
            lds_order();
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tl[qt][r] = tau[static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>((r & 3) + 8 * (r >> 2)) + 4u * half];
            }
        }
        const uint32_t row = x0 + col;
        const bool valid = row < x_hi;
        float16v_t acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
        if constexpr (KS > 0) {
            if (x0 + xstep < x_hi) load_tile(x0 + xstep, bnext, xn_next); // in flight during this tile's MFMAs
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[s2], bcur[s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[s2], bcur[s2], acc1, 0, 0, 0);
            }
        } else {
            const _Float16* xr = xh + static_cast<size_t>(valid ? row : x_lo) * dpad + 8u * half;
            for (uint32_t s2 = 0; s2 < ks; ++s2) {
                const half8_t b = *reinterpret_cast<const half8_t*>(xr + 16u * s2);
                const half8_t a0 = *reinterpret_cast<const half8_t*>(A + col * astride + (16u * s2 + 8u * half) * 2u);
                const half8_t a1 = *reinterpret_cast<const half8_t*>(A + (32u + col) * astride + (16u * s2 + 8u * half) * 2u);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b, acc1, 0, 0, 0);
            }
        }
        float xn = 0.0f;
        if constexpr (KS > 0) {
            xn = xn_cur;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) bcur[s2] = bnext[s2];
            xn_cur = xn_next;
        } else if constexpr (METRIC != M_IP) {
            xn = valid ? a.xnorm2[row] : 0.0f;
        }
        float rinv = 1.0f;
        if constexpr (METRIC == M_COS) rinv = 1.0f / sqrtf(xn > 0.0f ? xn : 1.0f);

        // keys for this lane's 32 (query, row) cells; accept test against the thresholds cached in registers
        float key[2][16];
        bool any = false;
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sc = qt == 0 ? acc0[r] : acc1[r];
                float kv;
                if constexpr (METRIC == M_L2) kv = xn - 2.0f * sc;
                else if constexpr (METRIC == M_IP) kv = -sc;
                else kv = -sc * rinv;
                if (!valid) kv = __builtin_inff();
                key[qt][r] = kv;
                any = any || (kv < tl[qt][r]);
            }
        }
        if (__ballot(any) == 0ull) continue; // common case once the shortlists are warm: nothing to insert
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                unsigned long long m = __ballot(key[qt][r] < tl[qt][r]); // register-cached threshold: no LDS on this path
                while (m != 0ull) {
                    const int bl = __builtin_ctzll(m);
                    m &= m - 1ull;
                    const float kq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key[qt][r]), bl));
                    const uint32_t bq = static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>((r & 3) + 8 * (r >> 2)) + 4u * static_cast<uint32_t>(bl >> 5);
                    // current threshold of that cell: the register copy held by lane bl (kept up to date below)
                    const float tq = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, tl[qt][r]), bl));
                    if (!(kq < tq)) continue;
                    if (kk > 64u && !(kq < tau[bq])) continue; // the general path keeps its thresholds in LDS only
                    const uint32_t id = x0 + static_cast<uint32_t>(bl & 31);
                    float* kl = lkey + bq * kk;
                    uint32_t* il = lid + bq * kk;
                    if (kk <= 64u) {
                        // one entry per lane: position by ballot, shift by one lane (DPP wave_shr:1), single write-back
                        float ek = __builtin_inff();
                        uint32_t ei = kInvalidId;
                        if (lane < kk) { ek = kl[lane]; ei = il[lane]; }
                        const uint32_t pos = __popcll(__ballot(lane < kk && ek <= kq)); // ties: existing entries first
                        const float pk = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, ek), 0x138, 0xf, 0xf, false));
                        const uint32_t pi = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(ei), 0x138, 0xf, 0xf, false));
                        const float nk = lane < pos ? ek : (lane == pos ? kq : pk);
                        const uint32_t ni = lane < pos ? ei : (lane == pos ? id : pi);
                        lds_order();
                        if (lane < kk && lane >= pos) { kl[lane] = nk; il[lane] = ni; }
                        const float nt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, nk), static_cast<int>(kk - 1u)));
                        // the lanes of that half-wave hold this cell's threshold in tl[qt][r]: update in place (no LDS refresh)
                        if (half == static_cast<uint32_t>(bl >> 5) && nt < tl[qt][r]) tl[qt][r] = nt;
                        if (lane == 0) {
                            if (nt < tau[bq]) tau[bq] = nt;
                            if (a.gtau != nullptr && q0 + bq < a.nq) atomicMin(a.gtau + q0 + bq, enc_key(nt));
                        }
                        lds_order();
                        continue;
                    }
                    // general path (kk > 64): sorted insert through LDS
                    uint32_t pos = 0;
                    for (uint32_t c0 = 0; c0 < kk; c0 += 64) {
                        const uint32_t i = c0 + lane;
                        pos += __popcll(__ballot(i < kk && kl[i] <= kq));
                    }
                    lds_order();
                    for (int c = static_cast<int>((kk - 1u - pos + 63u) >> 6) - 1; c >= 0; --c) {
                        const uint32_t i = pos + (static_cast<uint32_t>(c) << 6) + lane;
                        float kv = 0.0f;
                        uint32_t iv = 0;
                        const bool mv = i + 1u < kk;
                        if (mv) { kv = kl[i]; iv = il[i]; }
                        lds_order();
                        if (mv) { kl[i + 1] = kv; il[i + 1] = iv; }
                        lds_order();
                    }
                    if (lane == 0) {
                        kl[pos] = kq;
                        il[pos] = id;
                    }
                    lds_order();
                    if (lane == 0) {
                        const float nt = kl[kk - 1];
                        if (nt < tau[bq]) tau[bq] = nt;
                        if (a.gtau != nullptr && q0 + bq < a.nq) atomicMin(a.gtau + q0 + bq, enc_key(nt));
                    }
                    lds_order();
                }
            }
        }
        if (kk > 64u) { // the general insert path only updates the LDS thresholds: refresh the register copies
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tl[qt][r] = tau[static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>((r & 3) + 8 * (r >> 2)) + 4u * half];
            }
        }
    }
    // per-slice shortlists out: [slice][q][kk]
    for (uint32_t i = lane; i < 64u * kk; i += 64) {
        const uint32_t r = i / kk, t = i % kk;
        if (q0 + r < a.nq) {
            const size_t o = (static_cast<size_t>(slice) * a.nq + (q0 + r)) * kk + t;
            a.out_keys[o] = lkey[i];
            a.out_ids[o] = lid[i] == kInvalidId ? ~0ull : static_cast<uint64_t>(lid[i]);
        }
    }
}

hipError_t launch_flat_mfma(int metric, const FlatArgs& a, hipStream_t stream) {
    const uint32_t ks0 = a.dpad >> 4;
    const bool static_ks = ks0 >= 1 && ks0 <= 8;
    const size_t lds = (static_ks ? 0 : 64u * (static_cast<size_t>(a.dpad) * 2 + 16)) + 64u * static_cast<size_t>(a.kk) * 8 + 256;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid((a.nq + 63) / 64, a.n_slices);
    auto go = [&](auto kern) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL(kern, grid, dim3(64), lds, stream, a);
        return hipGetLastError();
    };
    const uint32_t ks = a.dpad >> 4;
#define SVSX_FLAT_CASE(M)                                                   \
    if (ks == 6) return go(flat_mfma_kernel<M, 6>);                         \
    if (ks == 8) return go(flat_mfma_kernel<M, 8>);                         \
    if (ks == 4) return go(flat_mfma_kernel<M, 4>);                         \
    if (ks == 7) return go(flat_mfma_kernel<M, 7>);                         \
    if (ks == 5) return go(flat_mfma_kernel<M, 5>);                         \
    if (ks == 3) return go(flat_mfma_kernel<M, 3>);                         \
    if (ks == 2) return go(flat_mfma_kernel<M, 2>);                         \
    if (ks == 1) return go(flat_mfma_kernel<M, 1>);                         \
    return go(flat_mfma_kernel<M, 0>);
    switch (metric) {
        case M_L2: SVSX_FLAT_CASE(M_L2)
        case M_IP: SVSX_FLAT_CASE(M_IP)
        case M_COS: SVSX_FLAT_CASE(M_COS)
    }
#undef SVSX_FLAT_CASE
    return hipErrorInvalidValue;
}

// ---- stage 1b: threshold pass ------------------------------------------------------------------------------------------------
// With a per-query threshold tau known in advance selection needs no ordered shortlist: every (query, row) with
// key <= tau is appended to the (query, slice) candidate list (LDS counter + store, straight from the lane that owns the
// cell).  No shortlists in LDS, three B tiles in flight per wave; the epilogue is one fma + one compare per cell.
// The thresholds come from two cheap refinement steps (svsx_api.hip): the shortlist kernel on a ~4096-row sample, then
// this kernel on every 16th tile with that threshold, then flat_select_kernel -> the kk-th best key of the 1/16 sample,
// which at least kk rows satisfy by construction.
// A threshold taken as the kk-th key of a 1/16 sample admits ~16 kk rows of the 16x larger set (std ~16 sqrt(kk)): lists of up to
// max(2048, 32 kk) candidates are re-ranked / selected from (LDS), longer ones count as overflow.
constexpr uint32_t kFlatSelectMax = 8192; // keys per query the select kernel accepts (kk <= 256)
__host__ __device__ __forceinline__ uint32_t flat_rerank_cap(uint32_t kk) { return 32u * kk > 2048u ? 32u * kk : 2048u; }
__host__ __device__ __forceinline__ uint32_t flat_select_cap(uint32_t kk) {
    // the coarsest level appends all of its up to 64 tiles = 2 048 rows (+ one partial tile); the key buffer in LDS decides how many of the
    // kernel's one-wave workgroups a compute unit holds, and the kernel is a chain of dependent latencies: 9 KB instead of 16 KB halves its time
    const uint32_t c = 32u * kk > 2304u ? 32u * kk : 2304u;
    return c < kFlatSelectMax ? c : kFlatSelectMax;
}

__device__ __forceinline__ bool flat_overflow(const uint32_t* cnt, uint32_t q, uint32_t n_slices, uint32_t cap, uint32_t& total,
                                              uint32_t total_cap) {
    bool over = false;
    total = 0;
    for (uint32_t s2 = 0; s2 < n_slices; ++s2) {
        const uint32_t c = cnt[static_cast<size_t>(q) * n_slices + s2];
        over = over || c > cap;
        total += c;
    }
    return over || total > total_cap;
}

// Offsets of one query's (query, slice) lists in the flattened candidate order: loff[s] = sum of the counts of slices < s
// (loff[n_slices] = total).  n_slices <= 1024.  All lanes then walk the flattened index space in parallel (independent
// loads) and find an element's slice by binary search in LDS.
// (n_slices <= 1024, svsx_api.hip)
__device__ __forceinline__ void flat_list_offsets(const uint32_t* cnt, uint32_t q, uint32_t n_slices, uint32_t* loff, uint32_t lane) {
    uint32_t carry = 0;
    for (uint32_t s0 = 0; s0 < n_slices; s0 += 64) {
        const uint32_t s2 = s0 + lane;
        const uint32_t c = s2 < n_slices ? cnt[static_cast<size_t>(q) * n_slices + s2] : 0u;
        uint32_t incl = c; // inclusive scan across the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = static_cast<uint32_t>(__shfl_up(static_cast<int>(incl), off));
            if (lane >= static_cast<uint32_t>(off)) incl += v;
        }
        if (s2 < n_slices) loff[s2] = carry + incl - c;
        carry += static_cast<uint32_t>(__shfl(static_cast<int>(incl), 63));
    }
    if (lane == 0) loff[n_slices] = carry;
    lds_order();
}
__device__ __forceinline__ uint32_t flat_find_slice(const uint32_t* loff, uint32_t n_slices, uint32_t e) {
    uint32_t lo = 0, hi = n_slices; // largest s with loff[s] <= e
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (loff[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

#ifdef SVSX_FLAT_NO_WG // the one-wave form of the d <= 128 pass (rounds 3-4): A/B builds only, the four-wave kernel below replaced it
template <int METRIC, int KS>
__global__ __launch_bounds__(64, 2) void flat_thresh_kernel(const FlatThreshArgs a) {
    __shared__ uint32_t lcnt[64];
    __shared__ __align__(16) float ltau[64];
    // B tiles go global -> registers (KS fully coalesced 1-KB loads: a 32-row tile of the f16 shadow is contiguous) -> LDS
    // (row pitch + 16 B: conflict-free ds_read_b128 in the MFMA fragment layout) -> fragments.  Reading the fragments
    // straight from global memory touches 32 cache lines per load instruction and is bound by the vector-memory path.
    __shared__ __align__(16) unsigned char btile[2][32 * (KS * 32 + 16)];
    const uint32_t lane = threadIdx.x;
    const uint32_t q0 = blockIdx.x * 64u;
    const uint32_t slice = blockIdx.y;
    const uint32_t dpad = a.dpad;
    const uint32_t rows_per_slice = ((a.n + a.n_slices - 1) / a.n_slices + 31u) & ~31u;
    const uint32_t x_lo = slice * rows_per_slice;
    const uint32_t x_hi = (x_lo + rows_per_slice) < a.n ? (x_lo + rows_per_slice) : a.n;
    const uint32_t xs = 32u * (a.tile_step ? a.tile_step : 1u);
    const uint32_t x_first = ((x_lo + xs - 1u) / xs) * xs;
    lcnt[lane] = 0u;
    lds_order();
    const uint32_t half = lane >> 5, col = lane & 31u;
    const _Float16* xh = static_cast<const _Float16*>(a.xh);
    const _Float16* qh = static_cast<const _Float16*>(a.qh);

    if (x_lo < x_hi && x_first < x_hi) {
        half8_t afr0[KS], afr1[KS];
        {
            const uint32_t r0 = q0 + col, r1 = q0 + 32u + col;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                half8_t z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = static_cast<_Float16>(0.0f);
                afr0[s2] = r0 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r0) * dpad + 16 * s2 + 8u * half) : z;
                afr1[s2] = r1 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r1) * dpad + 16 * s2 + 8u * half) : z;
            }
        }
        // Thresholds in the accumulator's domain, so that the accept test is ONE compare per cell: the accumulators start from the row's
        // constant (L2: -|x|^2 / 2, through the MFMA's C operand) and a cell is accepted iff acc >= thr,
        //   L2: key = |x|^2 - 2 q.x = -2 acc <= tau  <=>  acc >= -tau / 2;   IP: key = -acc <= tau  <=>  acc >= -tau;
        //   cosine: key = -(acc * rinv) <= tau  <=>  acc * rinv >= -tau.
        // The scalings are exact, so the accepted set is exactly {key <= tau} for the key that is stored and selected from.
        // -inf for padding queries' tau (thr = +inf: never accepted).  A lane's 32 thresholds live in registers.
        {
            const float t = q0 + lane < a.nq ? a.tau[static_cast<size_t>(q0 + lane) * a.tau_stride] : -__builtin_inff();
            ltau[lane] = METRIC == M_L2 ? -0.5f * t : -t;
#ifdef SVSX_FLAT_DIAG_NO_HITS   // timing builds only: nothing is ever accepted at the full-set level
            if (a.tile_step == 1) ltau[lane] = __builtin_inff();
#endif
        }
        lds_order();
        float thr[2][16];
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
#pragma unroll
            for (int r = 0; r < 16; ++r) thr[qt][r] = ltau[qt * 32 + 8 * (r >> 2) + 4 * static_cast<int>(half) + (r & 3)];
        constexpr uint32_t ROWB = KS * 32u;       // bytes of one f16 row (dpad * 2)
        constexpr uint32_t PITCH = ROWB + 16u;
        uint32_t lds_off[KS];                      // where this lane's 16-B piece of 1-KB chunk i lands
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const uint32_t g = static_cast<uint32_t>(i) * 1024u + lane * 16u;
            lds_off[i] = (g / ROWB) * PITCH + (g % ROWB);
        }
        const unsigned char* xbytes = reinterpret_cast<const unsigned char*>(xh);
        // a tile is 32 whole rows = KS contiguous KB: wave-uniform base + lane * 16 + i * 1024 (the shadow is padded by one tile, so the
        // last, partial tile reads initialised bytes; its surplus rows are masked when a cell is appended)
        auto load_tile = [&](uint32_t xt, half8_t (&dst)[KS], float& xnv) { // native 16-B vectors stay in registers
#ifdef SVSX_FLAT_DIAG_NO_LOADS  // timing builds only: every tile is the slice's first (cache-resident)
            if (a.tile_step == 1) xt = x_first;
#endif
            const unsigned char* base = xbytes + static_cast<size_t>(xt) * ROWB + lane * 16u;
#pragma unroll
            for (int i = 0; i < KS; ++i) dst[i] = *reinterpret_cast<const half8_t*>(base + static_cast<size_t>(i) * 1024u);
            if constexpr (METRIC != M_IP) {
                const uint32_t r = xt + col;
                xnv = a.xnorm2[r < a.n ? r : a.n - 1u];
            }
        };
        auto stage = [&](uint32_t buf, const half8_t (&src)[KS]) {
#pragma unroll
            for (int i = 0; i < KS; ++i) *reinterpret_cast<half8_t*>(&btile[buf][lds_off[i]]) = src[i];
        };
        auto fragments = [&](uint32_t buf, half8_t (&dst)[KS]) {
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) dst[s2] = *reinterpret_cast<const half8_t*>(&btile[buf][col * PITCH + 32u * s2 + 16u * half]);
        };
        const size_t stride_q = static_cast<size_t>(a.n_slices) * a.cap;                       // list stride between queries
        const size_t lane_base = (static_cast<size_t>(q0 + 4u * half) * a.n_slices + slice) * a.cap; // this lane's query group
        auto consume = [&](uint32_t x0, const half8_t (&b)[KS], float xn) {
            float16v_t acc0, acc1;
            {
                const float c0 = METRIC == M_L2 ? -0.5f * xn : 0.0f;
                float16v_t cin;
#pragma unroll
                for (int i = 0; i < 16; ++i) cin[i] = c0;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[0], b[0], cin, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[0], b[0], cin, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 1; s2 < KS; ++s2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[s2], b[s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[s2], b[s2], acc1, 0, 0, 0);
            }
#ifdef SVSX_FLAT_DIAG_NO_EPILOGUE  // timing builds only: the MFMAs stay alive through one compare
            if (a.tile_step == 1) {
                if (acc0[0] + acc1[15] == 123.456f) lcnt[0] = 1u;
                return;
            }
#endif
            float rinv = 1.0f;
            if constexpr (METRIC == M_COS) rinv = 1.0f / sqrtf(xn > 0.0f ? xn : 1.0f);
            auto score = [&](float v) -> float { // what is compared with thr
                if constexpr (METRIC == M_COS) return v * rinv;
                else return v;
            };
            auto keyof = [&](float v) -> float { // the key that is stored (smaller = better), an exact scaling of the score
                if constexpr (METRIC == M_L2) return -2.0f * v;
                else return -score(v);
            };
            // accept test: one compare per cell, folded into 8 wave-level masks (4 cells each); the append path only
            // revisits the groups in which some lane hit
            unsigned long long gm[2][4];
            unsigned long long anym = 0ull;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // four independent compares OR-ed as masks (no short circuit: `||` compiles to exec-mask branches here)
                    const bool h0 = score(qt == 0 ? acc0[4 * g4 + 0] : acc1[4 * g4 + 0]) >= thr[qt][4 * g4 + 0];
                    const bool h1 = score(qt == 0 ? acc0[4 * g4 + 1] : acc1[4 * g4 + 1]) >= thr[qt][4 * g4 + 1];
                    const bool h2 = score(qt == 0 ? acc0[4 * g4 + 2] : acc1[4 * g4 + 2]) >= thr[qt][4 * g4 + 2];
                    const bool h3 = score(qt == 0 ? acc0[4 * g4 + 3] : acc1[4 * g4 + 3]) >= thr[qt][4 * g4 + 3];
                    gm[qt][g4] = __ballot(h0) | __ballot(h1) | __ballot(h2) | __ballot(h3);
                    anym |= gm[qt][g4];
                }
            }
            if (anym == 0ull) return;
            const uint32_t row = x0 + col;
            const bool valid = row < x_hi; // lanes of a partial last tile hold rows of the next slice / the padding
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (gm[qt][g4] == 0ull) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        const float v = qt == 0 ? acc0[r] : acc1[r];
                        const uint32_t cell = static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>(j + 8 * g4);
                        const bool hit = valid && score(v) >= thr[qt][r];
                        if (hit) {
                            const uint32_t pos = atomicAdd(&lcnt[cell + 4u * half], 1u);
                            if (pos < a.cap) {
                                const size_t o = lane_base + static_cast<size_t>(cell) * stride_q + pos; // cell * stride_q is wave-uniform
                                a.cand[o] = row;
                                if (a.cand_keys != nullptr) a.cand_keys[o] = keyof(v);
                            }
                        }
                    }
                }
            }
        };
        half8_t gnext[KS];
        half8_t bcur[KS];
        float xn_cur = 0.0f, xn_next = 0.0f;
        load_tile(x_first, gnext, xn_next);
        stage(0u, gnext);
        uint32_t buf = 0;
        // Issuing tile t + 1's MFMAs before tile t's epilogue inside the wave (two accumulator sets, 249 VGPRs) was measured and is slower
        // (6.7 ms against 6.0 per 10 000 x 1 M x 96 search): profiles/r04_notes.md section 7.
        for (uint32_t x0 = x_first; x0 < x_hi; x0 += xs, buf ^= 1u) {
            xn_cur = xn_next;
            // next tile in flight during this tile's MFMAs and epilogue (past the end: a harmless re-read of the first)
#ifdef SVSX_FLAT_DIAG_NO_VMEM      // timing builds only: no vector-memory instruction in the full pass's loop (the first tile is reused)
            if (a.tile_step != 1)
#endif
            load_tile(x0 + xs < x_hi ? x0 + xs : x_first, gnext, xn_next);
#ifdef SVSX_FLAT_DIAG_NO_LDS       // timing builds only: the loaded pieces are used as fragments where they are
            if (a.tile_step == 1) {
#pragma unroll
                for (int i = 0; i < KS; ++i) bcur[i] = gnext[i];
                consume(x0, bcur, xn_cur);
                continue;
            }
#endif
            lds_order();
            fragments(buf, bcur);
            consume(x0, bcur, xn_cur);
            stage(buf ^ 1u, gnext);
            lds_order();
        }
    }
    lds_order();
    if (q0 + lane < a.nq) a.cnt[static_cast<size_t>(q0 + lane) * a.n_slices + slice] = lcnt[lane];
}

#endif // SVSX_FLAT_NO_WG

// ---- stage 1b at d <= 128: four waves share the row tiles ------------------------------------------------------------------------------
// The one-wave kernel above stages every 32-row tile once PER WAVE: 6-8 ds_write_b128 per tile and wave (13 cycles each on the store path,
// MI355X_MICROARCH.md section LDS) next to 12-16 MFMAs of 32 cycles -- with all four SIMDs busy the LDS store path of a compute unit is as
// loaded as its matrix pipes, and every tile crosses the L2 once per 64 queries.  Here a workgroup is four waves = 256 queries; a step is
// TR = 64 (128) rows (16-byte pieces per thread, global -> registers -> LDS, double buffered, ONE barrier per step), staged once and read
// as MFMA fragments by all four waves: a quarter of the stores and of the L2 traffic per query.
// The operands are swapped against the kernel above: the ROWS are the MFMA's A operand (from LDS) and the wave's queries its B operand
// (registers), so a lane's 16 accumulator cells of a 32 x 32 block belong to ONE query and 16 rows.  The threshold test of a block is then
// max over the 16 cells (7 v_max3 + 1 v_max) and one compare against the lane's own threshold instead of 16 compares against 16
// thresholds (32 -> 2 threshold registers per lane); the per-row constant of the accumulator's domain (L2: -|x|^2 / 2) arrives through
// the C operand from a TR-float array staged with the step.  And a lane appends to lists of its own: every (query, slice) has TWO lists
// (one per half-wave = per interleaved group of four rows of a block), so the list length is a register and an append is a store under
// the lane's exec bit -- no LDS counter, no atomic, nothing to wait for.  Accept set and keys are those of the kernel above, cell by cell.
// Sampled levels (tile_step > 1) take every tile_step-th STEP of TR rows (absolute: step index % tile_step == 0, so the samples are nested).
__device__ __forceinline__ float fmax3(float x, float y, float z) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    return r;
}
__device__ __forceinline__ float fmax2(float x, float y) { // the plain instruction: no canonicalising self-max in front (NaN cells are never accepted either way)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ void workgroup_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef SVSX_FLAT_WG_OCC
#define SVSX_FLAT_WG_OCC 3 // workgroups per compute unit the kernel is compiled for.  Measured (profiles/r05_notes.md section 5): 2 -> 3 is worth
#endif                     // 6-8 % of a search; forcing a fourth at d = 96 (128 registers, a few spills) loses 6-14 %; d <= 64 gets four or five anyway
constexpr int flat_wg_occ(int) { return SVSX_FLAT_WG_OCC; }
#ifndef SVSX_FLAT_WG_TR
#define SVSX_FLAT_WG_TR 64 // rows per step: 64 (needs an even number of 16-element k-steps; 128 otherwise).  Measured at d = 96: 128-row steps with
#endif                     // 54 KB of LDS per workgroup lose 16 % against 64-row steps with 27 KB (profiles/r05_notes.md section 5)
constexpr uint32_t flat_wg_rows(int ks) { return (SVSX_FLAT_WG_TR == 64 && ks % 2 == 0) ? 64u : 128u; }

template <int METRIC, int KS>
__global__ __launch_bounds__(256, flat_wg_occ(KS)) void flat_thresh_wg_kernel(const FlatThreshArgs a) {
    constexpr uint32_t ROWB = KS * 32u;       // bytes of one f16 row (dpad * 2)
    constexpr uint32_t PITCH = ROWB + 16u;    // conflict-free ds_read_b128 in the MFMA fragment layout
    constexpr uint32_t TR = flat_wg_rows(KS); // rows per step
    constexpr int NP = static_cast<int>(TR * ROWB / 4096u); // 16-byte pieces per thread and step
    static_assert(NP * 4096u == TR * ROWB, "a step is a whole number of 4-KB chunks");
    __shared__ __align__(16) float lrow[2][TR];  // per-row constant of the step's rows: L2 -|x|^2 / 2, cosine 1 / |x|
    __shared__ __align__(16) unsigned char btile[2][TR * PITCH];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t q0 = blockIdx.x * 256u, qw = q0 + wave * 64u;
    const uint32_t slice = blockIdx.y;
    const uint32_t rows_per_slice = (((a.n + a.n_slices - 1) / a.n_slices + TR - 1u) / TR) * TR;
    const uint32_t x_lo = slice * rows_per_slice < a.n ? slice * rows_per_slice : a.n;
    const uint32_t x_hi = (x_lo + rows_per_slice) < a.n ? (x_lo + rows_per_slice) : a.n;
    const uint32_t xs = TR * (a.tile_step ? a.tile_step : 1u);
    const uint32_t x_first = ((x_lo + xs - 1u) / xs) * xs; // sampled steps are absolute: step index % tile_step == 0
    const uint32_t half = lane >> 5, col = lane & 31u;
    uint32_t cnt0 = 0u, cnt1 = 0u; // lengths of this lane's two lists: (query qw + col, half) and (query qw + 32 + col, half)
    if (x_first < x_hi) { // uniform over the workgroup
        const _Float16* qh = static_cast<const _Float16*>(a.qh);
        half8_t qf0[KS], qf1[KS]; // the wave's 2 x 32 queries as B fragments, resident for the whole slice
        float thr0, thr1;         // this lane's two queries' thresholds in the accumulator's domain (flat_thresh_kernel)
        {
            const uint32_t r0 = qw + col, r1 = qw + 32u + col;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                half8_t z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = static_cast<_Float16>(0.0f);
                qf0[s2] = r0 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r0) * a.dpad + 16 * s2 + 8u * half) : z;
                qf1[s2] = r1 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r1) * a.dpad + 16 * s2 + 8u * half) : z;
            }
            const float t0 = r0 < a.nq ? a.tau[static_cast<size_t>(r0) * a.tau_stride] : -__builtin_inff(); // padding queries: never accepted
            const float t1 = r1 < a.nq ? a.tau[static_cast<size_t>(r1) * a.tau_stride] : -__builtin_inff();
            thr0 = METRIC == M_L2 ? -0.5f * t0 : -t0;
            thr1 = METRIC == M_L2 ? -0.5f * t1 : -t1;
#ifdef SVSX_FLAT_DIAG_NO_HITS   // timing builds only: nothing is ever accepted by the full pass
            if (a.tile_step <= 1u) thr0 = thr1 = __builtin_inff();
#endif
        }
        uint32_t lds_off[NP]; // where this thread's 16-B piece of the step's 4-KB chunk i lands
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const uint32_t g = static_cast<uint32_t>(i) * 4096u + tid * 16u;
            lds_off[i] = (g / ROWB) * PITCH + (g % ROWB);
        }
        const unsigned char* xbytes = reinterpret_cast<const unsigned char*>(a.xh);
        const uint32_t piece = tid * 16u;
        // TR whole rows = NP contiguous 4-KB chunks: wave-uniform base + a 32-bit thread offset (the shadow is padded by more than a step,
        // so the last, partial step reads initialised bytes).  The row constant travels raw: nothing here waits for a load.
        auto load_step = [&](uint32_t xt, half8_t (&dst)[NP], float& xn_raw) {
            const unsigned char* base = xbytes + static_cast<size_t>(xt) * ROWB;
#pragma unroll
            for (int i = 0; i < NP; ++i) dst[i] = *reinterpret_cast<const half8_t*>(base + (piece + static_cast<uint32_t>(i) * 4096u));
            if constexpr (METRIC != M_IP) {
                const uint32_t r = xt + (tid & (TR - 1u));
                xn_raw = a.xnorm2[r < a.n ? r : a.n - 1u];
            }
        };
        auto stage = [&](uint32_t buf, const half8_t (&src)[NP], float xn_raw) {
#pragma unroll
            for (int i = 0; i < NP; ++i) *reinterpret_cast<half8_t*>(&btile[buf][lds_off[i]]) = src[i];
            if constexpr (METRIC != M_IP) {
                if (tid < TR) lrow[buf][tid] = METRIC == M_L2 ? -0.5f * xn_raw : 1.0f / sqrtf(xn_raw > 0.0f ? xn_raw : 1.0f);
            }
        };
        // one accepted cell: the lane's own list grows by one (the count keeps running past the capacity: that is the overflow signal)
        // Past the capacity the last slot is overwritten: a list whose count exceeds the capacity is never read (flat_overflow sends the
        // query to the shortlist kernel), so the store needs no branch.
        const uint32_t last_slot = a.cap - 1u;
        auto emit = [&](uint32_t qt, uint32_t& cnt, uint32_t row, float v) {
            const uint32_t pos = cnt < last_slot ? cnt : last_slot;
            const size_t o = ((static_cast<size_t>(qw + qt * 32u + col) * a.n_slices + slice) * 2u + half) * a.cap + pos;
            a.cand[o] = row;
            // the key that is stored (smaller = better), an exact scaling of what was compared
            if (a.cand_keys != nullptr) a.cand_keys[o] = METRIC == M_L2 ? -2.0f * v : -v;
            ++cnt;
        };
        // one 32-row block of the step against the wave's 64 queries: the matrix part ...
        auto mma = [&](uint32_t buf, uint32_t j, float16v_t& acc0, float16v_t& acc1, float16v_t& rs) {
            float16v_t cin;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if constexpr (METRIC != M_IP) v = *reinterpret_cast<const float4*>(&lrow[buf][32u * j + 8u * static_cast<uint32_t>(g) + 4u * half]);
                if constexpr (METRIC == M_L2) { cin[4 * g + 0] = v.x; cin[4 * g + 1] = v.y; cin[4 * g + 2] = v.z; cin[4 * g + 3] = v.w; }
                else { cin[4 * g + 0] = 0.0f; cin[4 * g + 1] = 0.0f; cin[4 * g + 2] = 0.0f; cin[4 * g + 3] = 0.0f; }
                rs[4 * g + 0] = v.x; rs[4 * g + 1] = v.y; rs[4 * g + 2] = v.z; rs[4 * g + 3] = v.w; // cosine only
            }
            const unsigned char* rowp = &btile[buf][(32u * j + col) * PITCH + 16u * half];
            // (Requesting ALL of the block's fragments before its first MFMA -- one LDS round trip per block instead of one per MFMA pair,
            // -DSVSX_FLAT_WG_FRONTLOAD -- was measured: no gain, 16 more registers; the compiler's order stays.)
            half8_t xf[KS];
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) xf[s2] = *reinterpret_cast<const half8_t*>(rowp + 32 * s2);
#ifdef SVSX_FLAT_WG_FRONTLOAD
            __builtin_amdgcn_sched_barrier(0);
#endif
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[0], qf0[0], cin, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[0], qf1[0], cin, 0, 0, 0);
#pragma unroll
            for (int s2 = 1; s2 < KS; ++s2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[s2], qf0[s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[s2], qf1[s2], acc1, 0, 0, 0);
            }
        };
        // ... and the threshold test with its appends
        auto epilogue = [&](uint32_t xt, float16v_t& acc0, float16v_t& acc1, const float16v_t& rs) {
#ifdef SVSX_FLAT_DIAG_NO_EPILOGUE  // timing builds only: the MFMAs stay alive through one compare
            if (a.tile_step <= 1u) {
                if (acc0[0] + acc1[15] == 123.456f) ++cnt0;
                return;
            }
#endif
            if constexpr (METRIC == M_COS) { // what is compared with the threshold: acc / |x|
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] *= rs[r]; acc1[r] *= rs[r]; }
            }
            if (xt + 32u > x_hi) { // the last block of the last slice (every other slice ends on a step boundary): rows of the padding never match
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (xt + static_cast<uint32_t>(8 * (r >> 2) + (r & 3)) + 4u * half >= x_hi) { acc0[r] = -__builtin_inff(); acc1[r] = -__builtin_inff(); }
                }
            }
            // The v_max3 below are inline assembly, and the compiler's hazard recogniser does not count an inline-assembly statement as a
            // vector-ALU read of a matrix-core result (the hardware does not interlock that read: 8-pass MFMA -> VALU read needs 11 wait
            // states on gfx950; the compiler pads its OWN reads with s_nop, see the masking branch above).  Without the padding the test
            // read accumulators still missing their last k-step.  The statement is tied to both accumulators, so it stays behind the MFMAs.
            asm volatile("s_nop 7\n\ts_nop 3" : "+v"(acc0), "+v"(acc1));
            // maxima of the cells in groups of three; the block's test is one compare of their maximum, the groups find an accepted cell
            float g0[5], g1[5];
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                g0[g] = fmax3(acc0[3 * g], acc0[3 * g + 1], acc0[3 * g + 2]);
                g1[g] = fmax3(acc1[3 * g], acc1[3 * g + 1], acc1[3 * g + 2]);
            }
            const float m0 = fmax2(fmax3(g0[0], g0[1], g0[2]), fmax3(g0[3], g0[4], acc0[15]));
            const float m1 = fmax2(fmax3(g1[0], g1[1], g1[2]), fmax3(g1[3], g1[4], acc1[15]));
            const unsigned long long h0 = wave_ballot(m0 >= thr0), h1 = wave_ballot(m1 >= thr1);
            if ((h0 | h1) == 0ull) return;
            // some query of the wave accepts a row of this block: the lanes that own such cells append them
            auto scan = [&](const float16v_t& v, const float (&gm)[5], float thr, uint32_t qt, uint32_t& cnt) {
                const uint32_t rb = xt + 4u * half;
#pragma unroll
                for (int g = 0; g < 5; ++g) {
                    if (gm[g] >= thr) {
#pragma unroll
                        for (int e = 0; e < 3; ++e) {
                            const int r = 3 * g + e;
                            if (v[r] >= thr) emit(qt, cnt, rb + static_cast<uint32_t>(8 * (r >> 2) + (r & 3)), v[r]);
                        }
                    }
                }
                if (v[15] >= thr) emit(qt, cnt, rb + 27u, v[15]);
            };
            if (h0 != 0ull) scan(acc0, g0, thr0, 0u, cnt0);
            if (h1 != 0ull) scan(acc1, g1, thr1, 1u, cnt1);
        };
        auto blocks = [&](uint32_t buf, uint32_t x0) {
#pragma unroll
            for (uint32_t j = 0; j < TR / 32u; ++j) {
                const uint32_t xt = x0 + 32u * j;
                if (xt >= x_hi) break;
                float16v_t acc0, acc1, rs;
                mma(buf, j, acc0, acc1, rs);
                epilogue(xt, acc0, acc1, rs);
            }
        };
        half8_t gnext[NP];
        float xn_next = 0.0f;
        load_step(x_first, gnext, xn_next);
        stage(0u, gnext, xn_next);
        load_step(x_first + xs < x_hi ? x_first + xs : x_first, gnext, xn_next);
        workgroup_lds_barrier();
        uint32_t buf = 0;
        for (uint32_t x0 = x_first; x0 < x_hi; x0 += xs, buf ^= 1u) {
            blocks(buf, x0);
            // step t + 1 was requested a whole step ago (behind the staging of step t); as soon as its pieces have left the registers the
            // request for step t + 2 goes out into the same registers (past the end: a harmless re-read of the first)
#ifdef SVSX_FLAT_DIAG_NO_STAGE  // timing builds only (wrong results): the full pass neither stages nor requests new rows
            if (a.tile_step > 1u)
#endif
            {
                stage(buf ^ 1u, gnext, xn_next);
#ifdef SVSX_FLAT_DIAG_NO_LOADS
                if (a.tile_step > 1u)
#endif
                load_step(x0 + 2u * xs < x_hi ? x0 + 2u * xs : x_first, gnext, xn_next);
            }
#ifdef SVSX_FLAT_DIAG_NO_BARRIER // timing builds only (racy): the waves of a workgroup run free
            if (a.tile_step > 1u)
#endif
            workgroup_lds_barrier(); // buf ^ 1 is complete, and nobody reads buf any more
        }
    }
    {   // [nq][2 n_slices] list lengths
        const uint32_t r0 = qw + col, r1 = qw + 32u + col;
        if (r0 < a.nq) a.cnt[(static_cast<size_t>(r0) * a.n_slices + slice) * 2u + half] = cnt0;
        if (r1 < a.nq) a.cnt[(static_cast<size_t>(r1) * a.n_slices + slice) * 2u + half] = cnt1;
    }
}

// ---- stage 1b for 128 < d <= 768 ---------------------------------------------------------------------------------------------
// The same threshold pass with the dimension cut into KCH chunks of 128 (the shadow's rows are padded to a multiple of 128 there).  All A
// fragments of a wave's 64 queries stay in registers (2 x 8 KCH x 4 = 384 at d = 768: one wave per SIMD owns the whole 512-register file);
// a B tile arrives chunk by chunk -- 32 rows x 256 B through registers into LDS, fragments re-read from there, like the d <= 128 kernel
// with KS = 8 -- and the accumulators run across the chunks, so one epilogue follows 16 KCH MFMAs instead of 12-16.
template <int METRIC, int KCH>
__global__ __launch_bounds__(64, KCH <= 2 ? 2 : 1) void flat_thresh_big_kernel(const FlatThreshArgs a) {
    __shared__ uint32_t lcnt[64];
    __shared__ __align__(16) float ltau[64];
    constexpr uint32_t CHB = 256u;        // bytes of one row's chunk (128 float16)
    constexpr uint32_t PITCH = CHB + 16u; // conflict-free ds_read_b128 in the MFMA fragment layout
    constexpr uint32_t ROWB = KCH * CHB;  // bytes of one float16 row
    __shared__ __align__(16) unsigned char btile[2][32 * PITCH];
    const uint32_t lane = threadIdx.x;
    const uint32_t q0 = blockIdx.x * 64u;
    const uint32_t slice = blockIdx.y;
    const uint32_t dpad = a.dpad;
    const uint32_t rows_per_slice = ((a.n + a.n_slices - 1) / a.n_slices + 31u) & ~31u;
    const uint32_t x_lo = slice * rows_per_slice;
    const uint32_t x_hi = (x_lo + rows_per_slice) < a.n ? (x_lo + rows_per_slice) : a.n;
    const uint32_t xs = 32u * (a.tile_step ? a.tile_step : 1u);
    const uint32_t x_first = ((x_lo + xs - 1u) / xs) * xs;
    lcnt[lane] = 0u;
    lds_order();
    const uint32_t half = lane >> 5, col = lane & 31u;
    const _Float16* qh = static_cast<const _Float16*>(a.qh);
    if (x_lo < x_hi && x_first < x_hi) {
        half8_t afr0[KCH * 8], afr1[KCH * 8];
        {
            const uint32_t r0 = q0 + col, r1 = q0 + 32u + col;
#pragma unroll
            for (int s2 = 0; s2 < KCH * 8; ++s2) {
                half8_t z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = static_cast<_Float16>(0.0f);
                afr0[s2] = r0 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r0) * dpad + 16 * s2 + 8u * half) : z;
                afr1[s2] = r1 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r1) * dpad + 16 * s2 + 8u * half) : z;
            }
        }
        {   // thresholds in the accumulator's domain (see flat_thresh_kernel); read from LDS in the epilogue: the registers hold A
            const float t = q0 + lane < a.nq ? a.tau[static_cast<size_t>(q0 + lane) * a.tau_stride] : -__builtin_inff();
            ltau[lane] = METRIC == M_L2 ? -0.5f * t : -t;
        }
        lds_order();
        // a chunk of a tile: 32 rows x 256 B = 512 pieces of 16 B, 8 per lane; piece i * 64 + lane is piece (lane % 16) of row 4 i + lane / 16
        const uint32_t prow = lane >> 4, pcol = lane & 15u;
        const unsigned char* xbytes = reinterpret_cast<const unsigned char*>(a.xh);
        auto load_chunk = [&](uint32_t xt, uint32_t c, half8_t (&dst)[8]) {
            const unsigned char* base = xbytes + static_cast<size_t>(xt) * ROWB + static_cast<size_t>(prow) * ROWB + c * CHB + pcol * 16u;
#pragma unroll
            for (int i = 0; i < 8; ++i) dst[i] = *reinterpret_cast<const half8_t*>(base + static_cast<size_t>(i) * 4u * ROWB);
        };
        auto stage = [&](uint32_t buf, const half8_t (&src)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<half8_t*>(&btile[buf][(static_cast<uint32_t>(i) * 4u + prow) * PITCH + pcol * 16u]) = src[i];
        };
        auto fragment = [&](uint32_t buf, int s2) -> half8_t {
            return *reinterpret_cast<const half8_t*>(&btile[buf][col * PITCH + 32u * static_cast<uint32_t>(s2) + 16u * half]);
        };
        const size_t stride_q = static_cast<size_t>(a.n_slices) * a.cap;
        const size_t lane_base = (static_cast<size_t>(q0 + 4u * half) * a.n_slices + slice) * a.cap;
        auto epilogue = [&](uint32_t x0, float xn, const float16v_t& acc0, const float16v_t& acc1) {
            float rinv = 1.0f;
            if constexpr (METRIC == M_COS) rinv = 1.0f / sqrtf(xn > 0.0f ? xn : 1.0f);
            auto score = [&](float v) -> float {
                if constexpr (METRIC == M_COS) return v * rinv;
                else return v;
            };
            auto keyof = [&](float v) -> float {
                if constexpr (METRIC == M_L2) return -2.0f * v;
                else return -score(v);
            };
            unsigned long long gm[2][4];
            unsigned long long anym = 0ull;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 t4 = *reinterpret_cast<const float4*>(&ltau[qt * 32 + 8 * g4 + 4 * static_cast<int>(half)]);
                    const bool h0 = score(qt == 0 ? acc0[4 * g4 + 0] : acc1[4 * g4 + 0]) >= t4.x;
                    const bool h1 = score(qt == 0 ? acc0[4 * g4 + 1] : acc1[4 * g4 + 1]) >= t4.y;
                    const bool h2 = score(qt == 0 ? acc0[4 * g4 + 2] : acc1[4 * g4 + 2]) >= t4.z;
                    const bool h3 = score(qt == 0 ? acc0[4 * g4 + 3] : acc1[4 * g4 + 3]) >= t4.w;
                    gm[qt][g4] = __ballot(h0) | __ballot(h1) | __ballot(h2) | __ballot(h3);
                    anym |= gm[qt][g4];
                }
            }
            if (anym == 0ull) return;
            const uint32_t row = x0 + col;
            const bool valid = row < x_hi;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (gm[qt][g4] == 0ull) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        const float v = qt == 0 ? acc0[r] : acc1[r];
                        const uint32_t cell = static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>(j + 8 * g4);
                        const bool hit = valid && score(v) >= ltau[cell + 4u * half];
                        if (hit) {
                            const uint32_t pos = atomicAdd(&lcnt[cell + 4u * half], 1u);
                            if (pos < a.cap) {
                                const size_t o = lane_base + static_cast<size_t>(cell) * stride_q + pos;
                                a.cand[o] = row;
                                if (a.cand_keys != nullptr) a.cand_keys[o] = keyof(v);
                            }
                        }
                    }
                }
            }
        };
        half8_t gnext[8];
        load_chunk(x_first, 0u, gnext);
        stage(0u, gnext);
        uint32_t buf = 0;
        float xn_next = 0.0f;
        if constexpr (METRIC != M_IP) xn_next = a.xnorm2[x_first + col < a.n ? x_first + col : a.n - 1u];
        for (uint32_t x0 = x_first; x0 < x_hi; x0 += xs) {
            const float xn = xn_next;
            const uint32_t x1 = x0 + xs < x_hi ? x0 + xs : x_first; // past the end: a harmless re-read of the first tile
            if constexpr (METRIC != M_IP) xn_next = a.xnorm2[x1 + col < a.n ? x1 + col : a.n - 1u];
            float16v_t acc0, acc1;
            {
                const float c0 = METRIC == M_L2 ? -0.5f * xn : 0.0f;
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc0[i] = c0; acc1[i] = c0; }
            }
#pragma unroll
            for (int c = 0; c < KCH; ++c) {
                // the next chunk (of this tile, or the first of the next) is in flight during this chunk's MFMAs
                if (c + 1 < KCH) load_chunk(x0, static_cast<uint32_t>(c + 1), gnext);
                else load_chunk(x1, 0u, gnext);
                lds_order();
#pragma unroll
                for (int s2 = 0; s2 < 8; ++s2) {
                    const half8_t bs = fragment(buf, s2);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[c * 8 + s2], bs, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[c * 8 + s2], bs, acc1, 0, 0, 0);
                }
                stage(buf ^ 1u, gnext);
                lds_order();
                buf ^= 1u;
            }
            epilogue(x0, xn, acc0, acc1);
        }
    }
    lds_order();
    if (q0 + lane < a.nq) a.cnt[static_cast<size_t>(q0 + lane) * a.n_slices + slice] = lcnt[lane];
}

// what the host needs to size the pass: which padded dimensions the threshold kernels take and how many waves per SIMD they run at
bool flat_thresh_supports(uint32_t dpad) { return dpad <= 128u || (dpad % 128u == 0u && dpad <= 768u); }
uint32_t flat_thresh_waves_per_simd(uint32_t dpad) { return dpad <= 256u ? 2u : 1u; } // two chunks of A fragments still leave room for two waves
// the sampling unit of the threshold levels (rows per sampled tile / step) and the candidate lists a (query, slice) pair owns
uint32_t flat_thresh_unit_rows(uint32_t dpad) {
#ifndef SVSX_FLAT_NO_WG
    if (dpad <= 128u) return flat_wg_rows(static_cast<int>(dpad >> 4));
#endif
    (void)dpad;
    return 32u;
}
uint32_t flat_thresh_lists_per_slice(uint32_t dpad) {
#ifndef SVSX_FLAT_NO_WG
    if (dpad <= 128u) return 2u;
#endif
    (void)dpad;
    return 1u;
}
#ifndef SVSX_FLAT_NO_WG
typedef void (*flat_thresh_fn)(const FlatThreshArgs);
template <int M>
static flat_thresh_fn flat_wg_kernel_of(uint32_t ks) {
    switch (ks) {
        case 1: return flat_thresh_wg_kernel<M, 1>;
        case 2: return flat_thresh_wg_kernel<M, 2>;
        case 3: return flat_thresh_wg_kernel<M, 3>;
        case 4: return flat_thresh_wg_kernel<M, 4>;
        case 5: return flat_thresh_wg_kernel<M, 5>;
        case 6: return flat_thresh_wg_kernel<M, 6>;
        case 7: return flat_thresh_wg_kernel<M, 7>;
        default: return flat_thresh_wg_kernel<M, 8>;
    }
}
static flat_thresh_fn flat_wg_kernel_of(int metric, uint32_t ks) {
    return metric == M_L2 ? flat_wg_kernel_of<M_L2>(ks) : metric == M_IP ? flat_wg_kernel_of<M_IP>(ks) : flat_wg_kernel_of<M_COS>(ks);
}
#endif
// slices of the database per query block so that ONE generation of workgroups fills the machine (a second generation would double the
// pass): d <= 128 runs four-wave workgroups of 256 queries, as many per compute unit as the runtime says that instantiation gets
// (registers and LDS decide: 2-4); beyond, one-wave workgroups of 64 queries
uint32_t flat_thresh_slices(int metric, uint32_t dpad, size_t nq) {
#ifndef SVSX_FLAT_NO_WG
    if (dpad <= 128u) {
        int per_cu = 0, cus = 0, dev = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(flat_wg_kernel_of(metric, dpad >> 4)), 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const size_t s = (static_cast<size_t>(cus) * static_cast<size_t>(per_cu)) / ((nq + 255u) / 256u);
        return static_cast<uint32_t>(s < 1 ? 1 : s);
    }
#endif
    (void)metric;
    const size_t s = (static_cast<size_t>(flat_thresh_waves_per_simd(dpad)) * 1024u) / ((nq + 63u) / 64u);
    return static_cast<uint32_t>(s < 1 ? 1 : s);
}

hipError_t launch_flat_thresh(int metric, const FlatThreshArgs& a, hipStream_t stream) {
    const uint32_t ks = a.dpad >> 4;
    if (ks < 1 || !flat_thresh_supports(a.dpad)) return hipErrorInvalidValue;
    const dim3 grid((a.nq + 63) / 64, a.n_slices);
    auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(64), 0, stream, a);
        return hipGetLastError();
    };
    if (ks > 8) {
#define SVSX_THRESH_BIG(M)                                        \
    switch (a.dpad / 128u) {                                      \
        case 2: return go(flat_thresh_big_kernel<M, 2>);          \
        case 3: return go(flat_thresh_big_kernel<M, 3>);          \
        case 4: return go(flat_thresh_big_kernel<M, 4>);          \
        case 5: return go(flat_thresh_big_kernel<M, 5>);          \
        default: return go(flat_thresh_big_kernel<M, 6>);         \
    }
        switch (metric) {
            case M_L2: SVSX_THRESH_BIG(M_L2)
            case M_IP: SVSX_THRESH_BIG(M_IP)
            case M_COS: SVSX_THRESH_BIG(M_COS)
        }
#undef SVSX_THRESH_BIG
        return hipErrorInvalidValue;
    }
#ifndef SVSX_FLAT_NO_WG
    { // d <= 128: four waves share the row tiles (svsx_api.hip sizes its lists by flat_thresh_lists_per_slice and samples by flat_thresh_unit_rows)
        const dim3 wgrid((a.nq + 255) / 256, a.n_slices);
        hipLaunchKernelGGL(flat_wg_kernel_of(metric, ks), wgrid, dim3(256), 0, stream, a);
        return hipGetLastError();
    }
#else
#define SVSX_THRESH_KERNEL flat_thresh_kernel
#define SVSX_THRESH_CASE(M)                                   \
    switch (ks) {                                             \
        case 1: return go(SVSX_THRESH_KERNEL<M, 1>);          \
        case 2: return go(SVSX_THRESH_KERNEL<M, 2>);          \
        case 3: return go(SVSX_THRESH_KERNEL<M, 3>);          \
        case 4: return go(SVSX_THRESH_KERNEL<M, 4>);          \
        case 5: return go(SVSX_THRESH_KERNEL<M, 5>);          \
        case 6: return go(SVSX_THRESH_KERNEL<M, 6>);          \
        case 7: return go(SVSX_THRESH_KERNEL<M, 7>);          \
        default: return go(SVSX_THRESH_KERNEL<M, 8>);         \
    }
    switch (metric) {
        case M_L2: SVSX_THRESH_CASE(M_L2)
        case M_IP: SVSX_THRESH_CASE(M_IP)
        case M_COS: SVSX_THRESH_CASE(M_COS)
    }
#undef SVSX_THRESH_CASE
    return hipErrorInvalidValue;
#endif
}

// kk-th smallest key of one query's candidate lists (one wave per query): radix descent over the order-encoded keys, most significant
// differing bit first.  Lists of up to 512 keys (the usual case: ~16 kt rows per level) stay in registers, eight per lane -- one shift, one
// compare and one scalar popcount per key and bit; longer ones (the coarsest level of a small shortlist) go through LDS as before.
// Falls back to the coarser threshold when a list overflowed or holds fewer than kk rows.
constexpr int kSelectRegs = 8;
__global__ __launch_bounds__(64) void flat_select_kernel(const float* __restrict__ keys, const uint32_t* __restrict__ cnt, uint32_t nq,
                                                         uint32_t n_slices, uint32_t cap, uint32_t kk, const float* __restrict__ fb,
                                                         uint32_t fb_stride, float* __restrict__ out) {
    // dynamic LDS: flat_select_cap(kk) keys + n_slices + 1 list offsets (a static 8 192-key buffer would hold the kernel to 4 waves per CU)
    extern __shared__ __align__(16) unsigned char select_smem[];
    uint32_t* buf = reinterpret_cast<uint32_t*>(select_smem);
    uint32_t* loff = buf + flat_select_cap(kk);
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    // list lengths: one parallel pass gives the offsets, the total and the overflow flag
    bool lane_over = false;
    for (uint32_t s0 = 0; s0 < n_slices; s0 += 64) {
        const uint32_t s2 = s0 + lane;
        lane_over = lane_over || (s2 < n_slices && cnt[static_cast<size_t>(q) * n_slices + s2] > cap);
    }
    flat_list_offsets(cnt, q, n_slices, loff, lane);
    const uint32_t total = loff[n_slices];
    if (wave_ballot(lane_over) != 0ull || total > flat_select_cap(kk) || total < kk) {
        if (lane == 0) out[q] = fb[static_cast<size_t>(q) * fb_stride];
        return;
    }
    auto key_of = [&](uint32_t e) -> uint32_t {
        const uint32_t s2 = flat_find_slice(loff, n_slices, e);
        return enc_key(keys[(static_cast<size_t>(q) * n_slices + s2) * cap + (e - loff[s2])]);
    };
    if (total <= 64u * kSelectRegs) {
        uint32_t v[kSelectRegs];
#pragma unroll
        for (int r = 0; r < kSelectRegs; ++r) {
            const uint32_t e = lane + 64u * static_cast<uint32_t>(r);
            v[r] = e < total ? key_of(e) : 0xFFFFFFFFu; // padding: never has a zero bit under a matching prefix, so it is never counted
        }
        const uint32_t v0 = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v[0]))); // element 0 exists: total >= kk >= 1
        uint32_t diff = 0;
#pragma unroll
        for (int r = 0; r < kSelectRegs; ++r) {
            if (lane + 64u * static_cast<uint32_t>(r) < total) diff |= v[r] ^ v0;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) diff |= static_cast<uint32_t>(__shfl_xor(static_cast<int>(diff), off));
        diff = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(diff)));
        if (diff == 0u) { // every key is the same
            if (lane == 0) out[q] = dec_key(v0);
            return;
        }
        const int top = 31 - __builtin_clz(diff);
        uint32_t prefix = top == 31 ? 0u : (v0 & ~((2u << top) - 1u)); // the bits all keys share
        uint32_t remaining = kk;
        for (int bit = top; bit >= 0; --bit) {
            const uint32_t want = prefix >> bit; // this bit still zero
            uint32_t zeros = 0;
#pragma unroll
            for (int r = 0; r < kSelectRegs; ++r) zeros += static_cast<uint32_t>(__popcll(wave_ballot((v[r] >> bit) == want)));
            if (remaining > zeros) {
                remaining -= zeros;
                prefix |= 1u << bit;
            }
        }
        if (lane == 0) out[q] = dec_key(prefix);
        return;
    }
    for (uint32_t e = lane; e < total; e += 64) buf[e] = key_of(e);
    lds_order();
    uint32_t prefix = 0, remaining = kk;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t himask = bit == 31 ? 0u : ~((2u << bit) - 1u);
        uint32_t zeros = 0;
        for (uint32_t i0 = 0; i0 < total; i0 += 64) {
            const uint32_t i = i0 + lane;
            bool z = false;
            if (i < total) {
                const uint32_t v = buf[i];
                z = (v & himask) == prefix && ((v >> bit) & 1u) == 0u;
            }
            zeros += static_cast<uint32_t>(__popcll(__ballot(z)));
        }
        if (remaining > zeros) {
            remaining -= zeros;
            prefix |= 1u << bit;
        }
    }
    if (lane == 0) out[q] = dec_key(prefix);
}

hipError_t launch_flat_select(const float* cand_keys, const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t kk,
                              const float* tau_fallback, uint32_t fallback_stride, float* tau_out, hipStream_t stream) {
    const size_t lds = (static_cast<size_t>(flat_select_cap(kk)) + n_slices + 1) * 4;
    hipLaunchKernelGGL(flat_select_kernel, dim3(nq), dim3(64), lds, stream, cand_keys, cnt, nq, n_slices, cap, kk, tau_fallback, fallback_stride, tau_out);
    return hipGetLastError();
}

// one word per 64-query block: non-zero if any of its queries overflowed (that block re-runs the shortlist kernel)
__global__ void flat_gate_kernel(const uint32_t* __restrict__ cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t total_cap,
                                 uint32_t* __restrict__ gate) {
    const uint32_t q = blockIdx.x * 64u + threadIdx.x;
    uint32_t total = 0;
    const bool over = q < nq && flat_overflow(cnt, q, n_slices, cap, total, total_cap);
    const unsigned long long m = __ballot(over);
    if (threadIdx.x == 0) gate[blockIdx.x] = m != 0ull ? 1u : 0u;
}

hipError_t launch_flat_gate_kk(const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t kk, uint32_t* block_gate, hipStream_t stream) {
    hipLaunchKernelGGL(flat_gate_kernel, dim3((nq + 63) / 64), dim3(64), 0, stream, cnt, nq, n_slices, cap, flat_rerank_cap(kk), block_gate);
    return hipGetLastError();
}
hipError_t launch_flat_gate(const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t* block_gate, hipStream_t stream) {
    return launch_flat_gate_kk(cnt, nq, n_slices, cap, 40u, block_gate, stream);
}

// ---- stage 3: exact re-rank of the shortlist --------------------------------------------------------------------------------
template <int DT, int METRIC, int QMODE>
__global__ __launch_bounds__(64) void rerank_kernel(const RerankArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool GREATER = METRIC != M_L2;
    constexpr bool INTPATH = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    constexpr bool BC = !INTPATH && use_block_cyclic<DT>();
    constexpr int G = (INTPATH || BC) ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    float* qf = reinterpret_cast<float*>(smem);
    uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
    const uint32_t q_bytes = INTPATH ? static_cast<uint32_t>(dpad) : static_cast<uint32_t>(dpad) * 4u;
    float* cd = reinterpret_cast<float*>(smem + q_bytes);      // kk exact distances
    uint32_t* ci = reinterpret_cast<uint32_t*>(cd + a.kk);     // kk ids
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(qi) * a.q_row_bytes;
    const QueryFix fx = prepare_query<DT, METRIC, QMODE>(qptr, a.q_dtype, d, dpad, qf, qw, a.sq_scale, a.sq_bias, lane);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    const float worst = GREATER ? -__builtin_inff() : __builtin_inff();
    for (uint32_t c0 = 0; c0 < a.kk; c0 += RPP) {
        const uint32_t c = c0 + grp;
        uint64_t id64 = ~0ull;
        if (c < a.kk) id64 = a.cand_ids[static_cast<size_t>(qi) * a.kk + c];
        const bool act = id64 != ~0ull;
        const uint32_t id = act ? static_cast<uint32_t>(id64) : 0u;
        const unsigned char* row = data + static_cast<size_t>(id) * a.row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        int ab = 0, bb = 0;
        if constexpr (INTPATH) int_row_sums<DT>(row, act, qw, d, sub, ab, bb);
        else if constexpr (BC) float_row_distance_bc<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        else float_row_distance<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        const float dist = finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
        if (c < a.kk && sub == 0) {
            cd[c] = act ? dist : worst;
            ci[c] = act ? id : kInvalidId;
        }
    }
    lds_order();
    // rank by (distance with the metric's comparator, then id); invalid entries last
    for (uint32_t c = lane; c < a.kk; c += 64) {
        const float dc = cd[c];
        const uint32_t ic = ci[c];
        uint32_t rank = 0;
        for (uint32_t o = 0; o < a.kk; ++o) {
            const float dd = cd[o];
            const uint32_t io = ci[o];
            const bool o_valid = io != kInvalidId, c_valid = ic != kInvalidId;
            bool before;
            if (o_valid != c_valid) before = o_valid;
            else before = cmpf<GREATER>(dd, dc) || (!cmpf<GREATER>(dc, dd) && (io < ic || (io == ic && o < c)));
            rank += (o != c && before) ? 1u : 0u;
        }
        if (rank < a.k) {
            const size_t o = static_cast<size_t>(qi) * a.k + rank;
            const bool have = ic != kInvalidId;
            if (a.ids64 != nullptr) a.ids64[o] = have ? static_cast<uint64_t>(ic) + a.id_offset : ~0ull;
            if (a.dists != nullptr) a.dists[o] = have ? dc : __builtin_inff();
        }
    }
}

// Sorted k-best list of one query in LDS (cd = distances, ci = ids, best first, cnt entries), owned by one wave.  One pass offers up to 64
// candidates (lanes with cand set: distance `dist`, id `id`); they are inserted wave-uniformly, one at a time, by (distance, then id).
// NaN compares false both ways and falls through to the id, like rerank_kernel's ranking.
template <bool GREATER>
__device__ __forceinline__ void topk_offer(float* cd, uint32_t* ci, uint32_t& cnt, uint32_t k, float dist, uint32_t id, bool cand, uint32_t lane) {
    auto before = [](float dc, uint32_t ic, float dd, uint32_t idd) { return cmpf<GREATER>(dc, dd) || (!cmpf<GREATER>(dd, dc) && ic < idd); };
    if (cand && cnt == k) cand = before(dist, id, cd[cnt - 1u], ci[cnt - 1u]);
    unsigned long long m = __ballot(cand);
    while (m != 0ull) {
        const int l = __builtin_ctzll(m);
        m &= m - 1ull;
        const float dd = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dist), l));
        const uint32_t did = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(id), l));
        if (cnt == k && !before(dd, did, cd[cnt - 1u], ci[cnt - 1u])) continue; // an earlier insert of this pass moved the k-th
        uint32_t pos = 0;
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
            const uint32_t i = i0 + lane;
            pos += static_cast<uint32_t>(__builtin_popcountll(__ballot(i < cnt && before(cd[i], ci[i], dd, did))));
        }
        const uint32_t top = cnt < k ? cnt : k - 1u; // entries [pos, top) move one place down (the k-th falls off a full list)
        for (int c = static_cast<int>((k + 63u) / 64u) - 1; c >= 0; --c) {
            const uint32_t i = static_cast<uint32_t>(c) * 64u + lane;
            const bool mv = i >= pos && i < top;
            float v = 0.0f;
            uint32_t w = 0;
            if (mv) { v = cd[i]; w = ci[i]; }
            lds_order();
            if (mv) { cd[i + 1u] = v; ci[i + 1u] = w; }
            lds_order();
        }
        if (lane == 0) { cd[pos] = dd; ci[pos] = did; }
        lds_order();
        if (cnt < k) ++cnt;
    }
}

// ---- stage 3b: exact re-rank of a variable-length candidate list (threshold pass) ----------------------------------------------
// Same exact-order distance routines as rerank_kernel.  The candidates stream through: a pass computes the distances of 64 / G of them and
// offers them to the k-best list in LDS (topk_offer), so a wave's LDS is the query, the list offsets and k entries -- not the whole
// candidate list -- and eight waves per SIMD hide the two dependent gathers (candidate id, then its row); the ids of the next pass are
// requested before the current pass's rows are used.  A query whose list overflowed uses the gated shortlist path's merged list instead.
template <int DT, int METRIC, int QMODE>
__global__ __launch_bounds__(64) void rerank2_kernel(const RerankArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool GREATER = METRIC != M_L2;
    constexpr bool INTPATH = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    constexpr bool BC = !INTPATH && use_block_cyclic<DT>();
    constexpr int G = (INTPATH || BC) ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    float* qf = reinterpret_cast<float*>(smem);
    uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
    const uint32_t q_bytes = INTPATH ? static_cast<uint32_t>(dpad) : static_cast<uint32_t>(dpad) * 4u;
    float* cd = reinterpret_cast<float*>(smem + q_bytes);   // k distances, best first
    uint32_t* ci = reinterpret_cast<uint32_t*>(cd + a.k);   // k ids
    uint32_t* loff = ci + a.k;                              // n_lists + 1 offsets of the (query, slice) lists in the flattened order
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(qi) * a.q_row_bytes;
    const QueryFix fx = prepare_query<DT, METRIC, QMODE>(qptr, a.q_dtype, d, dpad, qf, qw, a.sq_scale, a.sq_bias, lane);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    uint32_t total = 0;
    const bool overflow = flat_overflow(a.cnt, qi, a.n_lists, a.cap, total, flat_rerank_cap(a.kk));
    const uint32_t c_n = overflow ? a.kk : total;
    if (!overflow) flat_list_offsets(a.cnt, qi, a.n_lists, loff, lane);
    auto candidate = [&](uint32_t c) -> uint32_t { // id of flattened candidate c (kInvalidId past the end / empty slot)
        if (c >= c_n) return kInvalidId;
        if (overflow) {
            const uint64_t id64 = a.cand_ids[static_cast<size_t>(qi) * a.kk + c];
            return id64 == ~0ull ? kInvalidId : static_cast<uint32_t>(id64);
        }
        const uint32_t l = flat_find_slice(loff, a.n_lists, c);
        return a.cand32[(static_cast<size_t>(qi) * a.n_lists + l) * a.cap + (c - loff[l])];
    };
    uint32_t cnt = 0;
    uint32_t id_next = candidate(grp);
    for (uint32_t c0 = 0; c0 < c_n; c0 += RPP) {
        const uint32_t id = id_next;
        id_next = candidate(c0 + RPP + grp);
        const bool act = id != kInvalidId;
        const unsigned char* row = data + static_cast<size_t>(act ? id : 0u) * a.row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        int ab = 0, bb = 0;
        if constexpr (INTPATH) int_row_sums<DT>(row, act, qw, d, sub, ab, bb);
        else if constexpr (BC) float_row_distance_bc<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        else float_row_distance<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        const float dist = finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
        topk_offer<GREATER>(cd, ci, cnt, a.k, dist, id, act && sub == 0, lane);
    }
    for (uint32_t j = lane; j < a.k; j += 64) {
        const size_t o = static_cast<size_t>(qi) * a.k + j;
        const bool have = j < cnt;
        if (a.ids64 != nullptr) a.ids64[o] = have ? static_cast<uint64_t>(ci[j]) + a.id_offset : ~0ull;
        if (a.dists != nullptr) a.dists[o] = have ? cd[j] : __builtin_inff();
    }
}

template <int DT, int METRIC, int QMODE> static hipError_t launch_rerank2_one(const RerankArgs& a, hipStream_t stream) {
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const bool intpath = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    const size_t lds = (intpath ? static_cast<size_t>(dpad) : static_cast<size_t>(dpad) * 4) + static_cast<size_t>(a.k) * 8 + (static_cast<size_t>(a.n_lists) + 1) * 4;
    auto kern = rerank2_kernel<DT, METRIC, QMODE>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(kern, dim3(a.nq), dim3(64), lds, stream, a);
    return hipGetLastError();
}

template <int DT, int METRIC, int QMODE> static hipError_t launch_rerank_one(const RerankArgs& a, hipStream_t stream) {
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const bool intpath = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    const size_t lds = (intpath ? static_cast<size_t>(dpad) : static_cast<size_t>(dpad) * 4) + static_cast<size_t>(a.kk) * 8;
    hipLaunchKernelGGL((rerank_kernel<DT, METRIC, QMODE>), dim3(a.nq), dim3(64), lds, stream, a);
    return hipGetLastError();
}

template <int DT, int QMODE> static hipError_t launch_rerank_metric(int metric, const RerankArgs& a, hipStream_t stream) {
    switch (metric) {
        case M_L2:
            if constexpr (QMODE == QM_SQ_IP || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_rerank_one<DT, M_L2, QMODE>(a, stream);
        case M_IP:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_rerank_one<DT, M_IP, QMODE>(a, stream);
        case M_COS:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_IP) return hipErrorInvalidValue;
            else return launch_rerank_one<DT, M_COS, QMODE>(a, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_rerank(int data_dtype, int metric, int qmode, const RerankArgs& a, hipStream_t stream) {
    switch (qmode) {
        case QM_FLOAT:
            switch (data_dtype) {
                case DT_F32: return launch_rerank_metric<DT_F32, QM_FLOAT>(metric, a, stream);
                case DT_F16: return launch_rerank_metric<DT_F16, QM_FLOAT>(metric, a, stream);
                case DT_I8: return launch_rerank_metric<DT_I8, QM_FLOAT>(metric, a, stream);
                case DT_U8: return launch_rerank_metric<DT_U8, QM_FLOAT>(metric, a, stream);
            }
            break;
        case QM_INT:
            if (data_dtype == DT_I8) return launch_rerank_metric<DT_I8, QM_INT>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank_metric<DT_U8, QM_INT>(metric, a, stream);
            break;
        case QM_SQ_IP:
            if (data_dtype == DT_I8) return launch_rerank_metric<DT_I8, QM_SQ_IP>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank_metric<DT_U8, QM_SQ_IP>(metric, a, stream);
            break;
        case QM_SQ_COS:
            if (data_dtype == DT_I8) return launch_rerank_metric<DT_I8, QM_SQ_COS>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank_metric<DT_U8, QM_SQ_COS>(metric, a, stream);
            break;
        case QM_SQ_L2:
            if (data_dtype == DT_I8) return launch_rerank_metric<DT_I8, QM_SQ_L2>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank_metric<DT_U8, QM_SQ_L2>(metric, a, stream);
            break;
    }
    return hipErrorInvalidValue;
}

template <int DT, int QMODE> static hipError_t launch_rerank2_metric(int metric, const RerankArgs& a, hipStream_t stream) {
    switch (metric) {
        case M_L2:
            if constexpr (QMODE == QM_SQ_IP || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_rerank2_one<DT, M_L2, QMODE>(a, stream);
        case M_IP:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_rerank2_one<DT, M_IP, QMODE>(a, stream);
        case M_COS:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_IP) return hipErrorInvalidValue;
            else return launch_rerank2_one<DT, M_COS, QMODE>(a, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_rerank2(int data_dtype, int metric, int qmode, const RerankArgs& a, hipStream_t stream) {
    switch (qmode) {
        case QM_FLOAT:
            switch (data_dtype) {
                case DT_F32: return launch_rerank2_metric<DT_F32, QM_FLOAT>(metric, a, stream);
                case DT_F16: return launch_rerank2_metric<DT_F16, QM_FLOAT>(metric, a, stream);
                case DT_I8: return launch_rerank2_metric<DT_I8, QM_FLOAT>(metric, a, stream);
                case DT_U8: return launch_rerank2_metric<DT_U8, QM_FLOAT>(metric, a, stream);
            }
            break;
        case QM_INT:
            if (data_dtype == DT_I8) return launch_rerank2_metric<DT_I8, QM_INT>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank2_metric<DT_U8, QM_INT>(metric, a, stream);
            break;
        case QM_SQ_IP:
            if (data_dtype == DT_I8) return launch_rerank2_metric<DT_I8, QM_SQ_IP>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank2_metric<DT_U8, QM_SQ_IP>(metric, a, stream);
            break;
        case QM_SQ_COS:
            if (data_dtype == DT_I8) return launch_rerank2_metric<DT_I8, QM_SQ_COS>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank2_metric<DT_U8, QM_SQ_COS>(metric, a, stream);
            break;
        case QM_SQ_L2:
            if (data_dtype == DT_I8) return launch_rerank2_metric<DT_I8, QM_SQ_L2>(metric, a, stream);
            if (data_dtype == DT_U8) return launch_rerank2_metric<DT_U8, QM_SQ_L2>(metric, a, stream);
            break;
    }
    return hipErrorInvalidValue;
}

// ---- stage 4: proof that the re-ranked answer is the exhaustive answer ------------------------------------------------------
// The shortlist is chosen by float16 keys, the answer by the reference-order float32 distances.  A row that was not re-ranked has
// key >= tau (tau = the threshold of the last threshold pass, or the kk-th key of the merged shortlist); from tau, the norms of the
// query and of the longest / shortest row and the rounding model below, a bound on the best reference distance such a row can have
// follows.  If that bound is strictly worse than the k-th re-ranked distance the answer is exact (ties included: an outside row
// with an EQUAL distance and a smaller id would belong in the answer, hence strictly).  Otherwise the query is recomputed by the
// exhaustive float32 kernel below.  Rounding model (u = 2^-11 = float16 unit roundoff, a = sqrt(dpad) 2^-25 = subnormal spacing):
//   ||v - fl16(v)|| <= u ||v|| + a;   float32 accumulation of dpad products (MFMA or FMA chain), |error| <= gam ||.|| ||.||, gam = dpad 2^-22;
//   the reference's float32 value is within a relative e = (d + 8) 2^-22 of the real-number distance (L2: of the distance itself, all terms
//   being non-negative; inner product / cosine: of ||q|| ||x||) -- each several times what the error analysis of a d-term sum gives.
// Non-finite norms or distances, a zero row under cosine, fewer than k answers: not proven.  Everything in double.
__global__ __launch_bounds__(64) void flat_certify_kernel(const FlatProofArgs a) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(q) * a.q_row_bytes;
    double s2 = 0.0, ds2 = 0.0;
    float qmax = 0.0f; // a query component beyond the float16 range makes its shadow +-inf and every key NaN / inf
    for (uint32_t i = lane; i < a.d; i += 64) {
        float v;
        switch (a.q_dtype) {
            case DT_F32: v = reinterpret_cast<const float*>(qptr)[i]; break;
            case DT_F16: v = static_cast<float>(reinterpret_cast<const _Float16*>(qptr)[i]); break;
            case DT_I8: v = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(qptr)[i])); break;
            default: v = static_cast<float>(static_cast<unsigned>(qptr[i])); break;
        }
        s2 += static_cast<double>(v) * static_cast<double>(v);
        qmax = fabsf(v) > qmax || v != v ? (v != v ? __builtin_inff() : fabsf(v)) : qmax;
        if (a.sq_l2) { // the reference's query is the decompressed code of the query
            const int c = a.data_dtype == DT_I8 ? sq_compress<DT_I8>(v, a.sq_scale, a.sq_bias) : sq_compress<DT_U8>(v, a.sq_scale, a.sq_bias);
            const double back = static_cast<double>(a.sq_scale) * static_cast<double>(c) + static_cast<double>(a.sq_bias);
            ds2 += (back - static_cast<double>(v)) * (back - static_cast<double>(v));
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        s2 += __shfl_xor(s2, off);
        ds2 += __shfl_xor(ds2, off);
        const float oq = __shfl_xor(qmax, off);
        qmax = oq > qmax ? oq : qmax;
    }
    if (lane != 0) return;
    const double nq = sqrt(s2), dqs = sqrt(ds2);
    const double u = 0x1p-11, ab = sqrt(static_cast<double>(a.dpad)) * 0x1p-25;
    const double xmax = sqrt(static_cast<double>(a.xmax2)), xmin = sqrt(static_cast<double>(a.xmin2));
    const double e16 = u * nq + ab;            // >= ||q - q_h||
    const double eq = dqs + e16;               // >= ||q* - q_h||, q* = the query the reference measures from
    const double ex = 1.001 * (u * xmax + ab); // >= ||x - x_h|| (xmax is the norm of the SHADOW row; 1.001 also covers the f32 decompression)
    const double qh_hi = nq + e16, qh_lo = nq - e16 > 0.0 ? nq - e16 : 0.0;
    const double gam = static_cast<double>(a.dpad) * 0x1p-22, e = static_cast<double>(a.d + 8u) * 0x1p-22;
    uint32_t total = 0;
    const bool from_shortlist = a.tau == nullptr || flat_overflow(a.cnt, q, a.n_lists, a.cap, total, flat_rerank_cap(a.kk));
    bool every_row = false;
    double tau = 0.0;
    if (from_shortlist) {
        const size_t last = static_cast<size_t>(q) * a.kk + a.kk - 1u;
        if (a.short_ids[last] == ~0ull) every_row = true; // the shortlist never filled: every row was re-ranked
        else tau = static_cast<double>(a.short_keys[last]);
    } else tau = static_cast<double>(a.tau[q]);
    const size_t kth = static_cast<size_t>(q) * a.k + a.k - 1u;
    const bool have_k = a.out_ids[kth] != ~0ull;
    const double dk = static_cast<double>(a.out_dists[kth]);
    bool ok = false;
    if (!(nq <= 1.0e30) || !(xmax <= 1.0e30) || !(qmax <= 65504.0f)) ok = false; // NaN / inf anywhere in the query, the data or their shadows
    else if (every_row) ok = true;
    else if (!have_k) ok = false;
    else if (a.metric == M_L2) {
        const double r2 = tau + qh_lo * qh_lo - gam * (xmax + qh_hi) * (xmax + qh_hi);
        if (r2 > 0.0) {
            const double r = sqrt(r2) - eq - ex;
            ok = r > 0.0 && r * r * (1.0 - e) > dk;
        }
    } else if (a.metric == M_IP) {
        const double best = -tau + gam * qh_hi * xmax + eq * xmax + nq * ex + 1.001 * e * nq * xmax;
        ok = best < dk; // dk is the reference's own float32 number: compared as is
    } else {
        if (xmin > 0.0 && nq > 0.0) {
            const double best = (-tau + fabs(tau) * gam + gam * qh_hi + eq + qh_hi * 2.002 * (u + ab / xmin)) / nq + 2.0 * e;
            ok = best < dk;
        }
    }
#ifdef SVSX_FLAT_DIAG_NO_HITS
    ok = true; // timing builds only
#endif
    a.redo[q] = ok ? 0u : 1u;
    if (!ok && a.redo_count != nullptr) atomicAdd(a.redo_count, 1u);
}

hipError_t launch_flat_certify(const FlatProofArgs& a, hipStream_t stream) {
    if (a.nq == 0) return hipSuccess;
    hipLaunchKernelGGL(flat_certify_kernel, dim3(a.nq), dim3(64), 0, stream, a);
    return hipGetLastError();
}

// max / min of the shadow rows' squared norms as raw bits (non-negative floats order like their bits; a NaN sorts above +inf)
__global__ __launch_bounds__(256) void norm_range_kernel(const float* __restrict__ norm2, uint32_t n, uint32_t* __restrict__ mm) {
    uint32_t hi = 0u, lo = 0xFFFFFFFFu;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t b = __float_as_uint(norm2[i]) & 0x7FFFFFFFu;
        hi = b > hi ? b : hi;
        lo = b < lo ? b : lo;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t oh = static_cast<uint32_t>(__shfl_xor(static_cast<int>(hi), off));
        const uint32_t ol = static_cast<uint32_t>(__shfl_xor(static_cast<int>(lo), off));
        hi = oh > hi ? oh : hi;
        lo = ol < lo ? ol : lo;
    }
    if ((threadIdx.x & 63u) == 0u) {
        atomicMax(mm, hi);
        atomicMin(mm + 1, lo);
    }
}

hipError_t launch_norm_range(const float* norm2, uint32_t n, uint32_t* mm, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    uint32_t blocks = (n + 255u) / 256u;
    if (blocks > 2048u) blocks = 2048u;
    hipLaunchKernelGGL(norm_range_kernel, dim3(blocks), dim3(256), 0, stream, norm2, n, mm);
    return hipGetLastError();
}

// ---- stage 5: exhaustive float32 pass for the queries stage 4 could not prove --------------------------------------------------
// One wave per flagged query over ALL rows with the reference-order distance routines; the k best by (distance, then id) live
// sorted in LDS.  A pass of 64 / G rows inserts (wave-uniform, one candidate at a time) only the rows that beat the current k-th.
template <int DT, int METRIC, int QMODE>
__global__ __launch_bounds__(64) void flat_exact_kernel(const FlatExactArgs xa) {
    const RerankArgs& a = xa.r;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool GREATER = METRIC != M_L2;
    constexpr bool INTPATH = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    constexpr bool BC = !INTPATH && use_block_cyclic<DT>();
    constexpr int G = (INTPATH || BC) ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    if (xa.redo[qi] == 0u) return;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    float* qf = reinterpret_cast<float*>(smem);
    uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
    const uint32_t q_bytes = INTPATH ? static_cast<uint32_t>(dpad) : static_cast<uint32_t>(dpad) * 4u;
    float* cd = reinterpret_cast<float*>(smem + q_bytes);  // k distances, best first
    uint32_t* ci = reinterpret_cast<uint32_t*>(cd + a.k);  // k ids
    const unsigned char* qptr = static_cast<const unsigned char*>(a.queries) + static_cast<size_t>(qi) * a.q_row_bytes;
    const QueryFix fx = prepare_query<DT, METRIC, QMODE>(qptr, a.q_dtype, d, dpad, qf, qw, a.sq_scale, a.sq_bias, lane);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    uint32_t cnt = 0;
    for (uint32_t j0 = 0; j0 < xa.n; j0 += RPP) {
        const uint32_t id = j0 + grp;
        const bool act = id < xa.n;
        const unsigned char* row = data + static_cast<size_t>(act ? id : 0u) * a.row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        int ab = 0, bb = 0;
        if constexpr (INTPATH) int_row_sums<DT>(row, act, qw, d, sub, ab, bb);
        else if constexpr (BC) float_row_distance_bc<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        else float_row_distance<DT, METRIC, 0, QMODE == QM_SQ_COS>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        const float dist = finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, a.sq_scale);
        topk_offer<GREATER>(cd, ci, cnt, a.k, dist, id, act && sub == 0, lane);
    }
    for (uint32_t j = lane; j < a.k; j += 64) {
        const size_t o = static_cast<size_t>(qi) * a.k + j;
        const bool have = j < cnt;
        if (a.ids64 != nullptr) a.ids64[o] = have ? static_cast<uint64_t>(ci[j]) + a.id_offset : ~0ull;
        if (a.dists != nullptr) a.dists[o] = have ? cd[j] : __builtin_inff();
    }
}

template <int DT, int METRIC, int QMODE> static hipError_t launch_flat_exact_one(const FlatExactArgs& xa, hipStream_t stream) {
    const RerankArgs& a = xa.r;
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const bool intpath = (QMODE == QM_INT || QMODE == QM_SQ_L2);
    const size_t lds = (intpath ? static_cast<size_t>(dpad) : static_cast<size_t>(dpad) * 4) + static_cast<size_t>(a.k) * 8;
    auto kern = flat_exact_kernel<DT, METRIC, QMODE>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
    hipLaunchKernelGGL(kern, dim3(a.nq), dim3(64), lds, stream, xa);
    return hipGetLastError();
}

template <int DT, int QMODE> static hipError_t launch_flat_exact_metric(int metric, const FlatExactArgs& a, hipStream_t stream) {
    switch (metric) {
        case M_L2:
            if constexpr (QMODE == QM_SQ_IP || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_flat_exact_one<DT, M_L2, QMODE>(a, stream);
        case M_IP:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_COS) return hipErrorInvalidValue;
            else return launch_flat_exact_one<DT, M_IP, QMODE>(a, stream);
        case M_COS:
            if constexpr (QMODE == QM_SQ_L2 || QMODE == QM_SQ_IP) return hipErrorInvalidValue;
            else return launch_flat_exact_one<DT, M_COS, QMODE>(a, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_flat_exact(int data_dtype, int metric, int qmode, const FlatExactArgs& a, hipStream_t stream) {
    if (a.r.nq == 0 || a.redo == nullptr) return hipErrorInvalidValue;
    const bool i8 = data_dtype == DT_I8, u8 = data_dtype == DT_U8;
    switch (qmode) {
        case QM_FLOAT:
            switch (data_dtype) {
                case DT_F32: return launch_flat_exact_metric<DT_F32, QM_FLOAT>(metric, a, stream);
                case DT_F16: return launch_flat_exact_metric<DT_F16, QM_FLOAT>(metric, a, stream);
                case DT_I8: return launch_flat_exact_metric<DT_I8, QM_FLOAT>(metric, a, stream);
                case DT_U8: return launch_flat_exact_metric<DT_U8, QM_FLOAT>(metric, a, stream);
            }
            break;
        case QM_INT:
            if (i8) return launch_flat_exact_metric<DT_I8, QM_INT>(metric, a, stream);
            if (u8) return launch_flat_exact_metric<DT_U8, QM_INT>(metric, a, stream);
            break;
        case QM_SQ_IP:
            if (i8) return launch_flat_exact_metric<DT_I8, QM_SQ_IP>(metric, a, stream);
            if (u8) return launch_flat_exact_metric<DT_U8, QM_SQ_IP>(metric, a, stream);
            break;
        case QM_SQ_COS:
            if (i8) return launch_flat_exact_metric<DT_I8, QM_SQ_COS>(metric, a, stream);
            if (u8) return launch_flat_exact_metric<DT_U8, QM_SQ_COS>(metric, a, stream);
            break;
        case QM_SQ_L2:
            if (i8) return launch_flat_exact_metric<DT_I8, QM_SQ_L2>(metric, a, stream);
            if (u8) return launch_flat_exact_metric<DT_U8, QM_SQ_L2>(metric, a, stream);
            break;
    }
    return hipErrorInvalidValue;
}

} // namespace svsx
