// svsx_build.hip -- neighbour pruning for the GPU-assisted Vamana builder ("next" scope).
//
// prune_kernel: heuristic_prune_neighbors for a batch of vertices (index/vamana/prune.h:100-246):
//   ProgressivePruneStrategy for L2 (prune.h:38-40, 178-246), IterativePruneStrategy for IP / cosine (:41-48,100-176).
// One wavefront per vertex.  The sorted candidate pool (ids, distances to the vertex) and the per-candidate prune
// state live in LDS; the selection loop is serial (as in the reference), and for every selected candidate s the
// distances d(s, t) to the still-available candidates t > s are computed 16 (or 64/G) rows at a time with the same
// row-group distance routines as the search kernel (rows come from L2 after their first touch: a pool of 256 rows of
// 384 B is 96 KB).  Distances here follow the build metric on the stored vectors; like the reference's builder the
// result is one of many valid graphs (the reference build itself is thread-schedule dependent), so this kernel is
// checked for graph quality and, on identical candidate pools, against the CPU checker's prune (tests/).
#include "svsx_device.h"
#include "svsx_kernels.h"

namespace svsx {

template <int DT, int METRIC, bool DECOMP>
__global__ __launch_bounds__(64) void prune_kernel(const PruneArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool GREATER = METRIC != M_L2;
    constexpr bool BC = use_block_cyclic<DT>();
    constexpr int G = BC ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t v = blockIdx.x;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    const uint32_t P = a.pool_size;

    float* qf = reinterpret_cast<float*>(smem);                 // selected candidate's vector, widened to f32
    int32_t* ids = reinterpret_cast<int32_t*>(qf + dpad);       // P
    float* pd = reinterpret_cast<float*>(ids + P);              // P: distance(vertex, candidate)
    float* state = pd + P;                                      // P

    const int32_t self = a.node_ids[v];
    uint32_t np = 0;
    for (uint32_t i = lane; i < P; i += 64) {
        const int32_t id = a.pool_ids[static_cast<size_t>(v) * P + i];
        ids[i] = id;
        pd[i] = a.pool_dists[static_cast<size_t>(v) * P + i];
        state[i] = GREATER ? 0.0f : -3.402823466e+38f;
    }
    lds_order();
    for (uint32_t i0 = 0; i0 < P; i0 += 64) {
        const uint32_t i = i0 + lane;
        np += __popcll(__ballot(i < P && ids[i] >= 0)); // valid ids are packed at the front
    }

    int32_t* out = a.out + static_cast<size_t>(v) * a.out_stride;
    for (uint32_t i = lane; i < a.out_stride; i += 64) out[i] = -1;
    uint32_t count = 0;
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;

    float cur = 1.0f;
    while (true) {
        for (uint32_t s = 0; s < np && count < a.max_out; ++s) {
            const int32_t sid = ids[s];
            const float st = state[s];
            const bool skip = GREATER ? (st != 0.0f) : (cur < st);
            if (skip || sid == self) continue;
            lds_order();
            if (lane == 0) {
                state[s] = GREATER ? 1.0f : 3.402823466e+38f;
                out[count] = sid;
            }
            ++count;
            if (count >= a.max_out) break;
            // the selected candidate becomes the "query" (distance::maybe_fix_argument, prune.h:134-135)
            const unsigned char* srow = data + static_cast<size_t>(sid) * a.row_bytes;
            lds_order();
            for (int i = lane; i < dpad; i += 64) {
                float x = 0.0f;
                if (i < d) {
                    if constexpr (DT == DT_F32) x = reinterpret_cast<const float*>(srow)[i];
                    else if constexpr (DT == DT_F16) x = static_cast<float>(reinterpret_cast<const _Float16*>(srow)[i]);
                    else if constexpr (DT == DT_I8) x = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(srow)[i]));
                    else x = static_cast<float>(static_cast<unsigned>(srow[i]));
                    if constexpr (DECOMP) x = a.sq_scale * x + a.sq_bias;
                }
                qf[i] = x;
            }
            lds_order();
            QueryFix fx;
            fx.norm = 0.0f; fx.offset = 0.0f; fx.qq = 0;
            if constexpr (METRIC == M_COS) {
                float acc = 0.0f;
                for (int i = lane; i < d; i += 64) acc = __builtin_fmaf(qf[i], qf[i], acc);
                for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
                fx.norm = sqrtf(acc);
            }
            for (uint32_t t0 = s + 1; t0 < np; t0 += RPP) {
                const uint32_t t = t0 + grp;
                bool act = t < np;
                int32_t tid = 0;
                float tst = 0.0f;
                if (act) {
                    tid = ids[t];
                    tst = state[t];
                    act = GREATER ? (tst == 0.0f) : !(cur < tst);
                }
                if (__ballot(act) == 0ull) continue;
                const unsigned char* row = data + static_cast<size_t>(act ? tid : 0) * a.row_bytes;
                float fmain = 0.0f, fnrm = 0.0f;
                if constexpr (BC) float_row_distance_bc<DT, METRIC, 0, DECOMP>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
                else float_row_distance<DT, METRIC, 0, DECOMP>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
                const float djk = finish_distance<METRIC, QM_FLOAT>(fmain, fnrm, 0, 0, fx, 1.0f);
                if (act && sub == 0) {
                    if constexpr (GREATER) {
                        if (cur * djk > pd[t]) state[t] = 2.0f;              // prune.h:150-152
                    } else {
                        const float r = pd[t] / djk;                          // prune.h:222-224
                        if (tst < r) state[t] = r;
                    }
                }
            }
            lds_order();
        }
        if (a.alpha == 1.0f || count >= a.max_out) break;
        const float next = cur * a.alpha;
        if (GREATER ? !(a.alpha <= next) : (a.alpha < next)) break; // while (!cmp(alpha, current_alpha))
        if constexpr (GREATER) {
            lds_order();
            for (uint32_t i = lane; i < np; i += 64) if (state[i] == 2.0f) state[i] = 0.0f; // reenable
            lds_order();
        }
        cur = next;
    }
}

template <int DT, int METRIC> static hipError_t launch_prune_m(const PruneArgs& a, bool decomp, hipStream_t stream) {
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const size_t lds = static_cast<size_t>(dpad) * 4 + static_cast<size_t>(a.pool_size) * 12;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto launch = [&](auto kern) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL(kern, dim3(a.n_nodes), dim3(64), lds, stream, a);
        return hipGetLastError();
    };
    if (decomp) return launch(prune_kernel<DT, METRIC, true>);
    return launch(prune_kernel<DT, METRIC, false>);
}

template <int DT> static hipError_t launch_prune_dt(int metric, const PruneArgs& a, bool decomp, hipStream_t stream) {
    switch (metric) {
        case M_L2: return launch_prune_m<DT, M_L2>(a, decomp, stream);
        case M_IP: return launch_prune_m<DT, M_IP>(a, decomp, stream);
        case M_COS: return launch_prune_m<DT, M_COS>(a, decomp, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_prune(int data_dtype, int metric, bool decomp, const PruneArgs& a, hipStream_t stream) {
    switch (data_dtype) {
        case DT_F32: return launch_prune_dt<DT_F32>(metric, a, false, stream);
        case DT_F16: return launch_prune_dt<DT_F16>(metric, a, false, stream);
        case DT_I8: return launch_prune_dt<DT_I8>(metric, a, decomp, stream);
        case DT_U8: return launch_prune_dt<DT_U8>(metric, a, decomp, stream);
    }
    return hipErrorInvalidValue;
}

} // namespace svsx
