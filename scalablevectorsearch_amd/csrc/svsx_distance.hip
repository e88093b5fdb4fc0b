// svsx_distance.hip -- the svs::distance plugin surface as a batched kernel, the shard top-k merge and the
// graph re-layout kernel (gfx950).
//
//   distance_kernel   out[i][j] = distance::compute(f, a_i, b_j) after f.fix_argument(a_i)
//                     (concepts/distance.h:88-128; K1-K6 of SURVEY 8a) -- same device routines, hence same bits,
//                     as the search kernel.
//   merge_kernel      per query S-way merge of sorted (dist,id) lists (no reference counterpart).
//   pack_graph_kernel SVS graph rows [deg, n0..] (core/graph/graph.h:32-48) -> packed rows padded with 0xFFFFFFFF.
#include "svsx_device.h"
#include "svsx_kernels.h"

namespace svsx {

template <int DT, int METRIC, int QMODE>
__global__ __launch_bounds__(64) void distance_kernel(const DistanceArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool INTPATH = (QMODE == QM_INT);
    constexpr bool B8 = !INTPATH && use_block_cyclic<DT>();
    constexpr int G = QMODE == QM_SCALAR ? 1 : (INTPATH || B8) ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    float* qf = reinterpret_cast<float*>(smem);
    uint32_t* qw = reinterpret_cast<uint32_t*>(smem);
    const size_t a_row = static_cast<size_t>(d) * (a.a_dtype == DT_F32 ? 4 : a.a_dtype == DT_F16 ? 2 : 1);
    const unsigned char* qptr = static_cast<const unsigned char*>(a.a) + static_cast<size_t>(qi) * a_row;
    const QueryFix fx = prepare_query<DT, METRIC, QMODE>(qptr, a.a_dtype, d, dpad, qf, qw, 1.0f, 0.0f, lane);
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    const unsigned char* data = static_cast<const unsigned char*>(a.b);
    for (uint32_t j0 = blockIdx.y * RPP; j0 < a.nb; j0 += gridDim.y * RPP) {
        const uint32_t j = j0 + grp;
        const bool act = j < a.nb;
        const unsigned char* row = data + static_cast<size_t>(act ? j : 0u) * a.b_row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        int ab = 0, bb = 0;
        if constexpr (QMODE == QM_SCALAR) scalar_row_distance<DT, METRIC>(row, qf, d, fmain, fnrm);
        else if constexpr (INTPATH) int_row_sums<DT>(row, act, qw, d, sub, ab, bb);
        else if constexpr (B8) float_row_distance_bc<DT, METRIC, 0, false>(row, act, qf, d, sub, 1.0f, 0.0f, fmain, fnrm);
        else float_row_distance<DT, METRIC, 0, false>(row, act, qf, d, sub, 1.0f, 0.0f, fmain, fnrm);
        const float r = finish_distance<METRIC, QMODE>(fmain, fnrm, ab, bb, fx, 1.0f);
        if (act && sub == 0) a.out[static_cast<size_t>(qi) * a.nb + j] = r;
    }
}

template <int DT, int QMODE> static hipError_t launch_distance_metric(int metric, const DistanceArgs& a, hipStream_t stream) {
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const size_t lds = QMODE == QM_INT ? dpad : dpad * 4;
    uint32_t gy = (a.nb + 255) / 256;
    if (gy == 0) gy = 1;
    if (gy > 64) gy = 64;
    dim3 grid(a.na, gy);
    switch (metric) {
        case M_L2: hipLaunchKernelGGL((distance_kernel<DT, M_L2, QMODE>), grid, dim3(64), lds, stream, a); break;
        case M_IP: hipLaunchKernelGGL((distance_kernel<DT, M_IP, QMODE>), grid, dim3(64), lds, stream, a); break;
        case M_COS: hipLaunchKernelGGL((distance_kernel<DT, M_COS, QMODE>), grid, dim3(64), lds, stream, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_distance(int b_dtype, int metric, int qmode, const DistanceArgs& a, hipStream_t stream) {
    if (qmode == QM_FLOAT) {
        switch (b_dtype) {
            case DT_F32: return launch_distance_metric<DT_F32, QM_FLOAT>(metric, a, stream);
            case DT_F16: return launch_distance_metric<DT_F16, QM_FLOAT>(metric, a, stream);
            case DT_I8: return launch_distance_metric<DT_I8, QM_FLOAT>(metric, a, stream);
            case DT_U8: return launch_distance_metric<DT_U8, QM_FLOAT>(metric, a, stream);
        }
    } else if (qmode == QM_INT) {
        if (b_dtype == DT_I8) return launch_distance_metric<DT_I8, QM_INT>(metric, a, stream);
        if (b_dtype == DT_U8) return launch_distance_metric<DT_U8, QM_INT>(metric, a, stream);
    } else if (qmode == QM_SCALAR) {
        switch (b_dtype) {
            case DT_F32: return launch_distance_metric<DT_F32, QM_SCALAR>(metric, a, stream);
            case DT_F16: return launch_distance_metric<DT_F16, QM_SCALAR>(metric, a, stream);
            case DT_I8: return launch_distance_metric<DT_I8, QM_SCALAR>(metric, a, stream);
            case DT_U8: return launch_distance_metric<DT_U8, QM_SCALAR>(metric, a, stream);
        }
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------
// Norm of every scalar-quantized row as CosineSimilarityCompressed sees it (quantization/scalar/scalar.h:175-184 decompresses
// scale * float(v) + bias, cosine.h:224-256 accumulates x * x in the 16-lane tree, :334-335 takes sqrtf): the accumulator chain never sees
// the query, so it is run ONCE per row here -- the same device routine as the search kernel, hence the same bits -- and the search kernel
// reads sqrtf(sum) instead of repeating 2 d operations and a correctly rounded square root per candidate.
template <int DT, int DS> __global__ __launch_bounds__(64) void sq_row_norm_kernel(const void* data, size_t row_bytes, uint32_t n, uint32_t d_runtime, float scale, float bias, float* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x;
    const int d = DS > 0 ? DS : static_cast<int>(d_runtime);
    const int dpad = (d + 15) & ~15;
    float* qf = reinterpret_cast<float*>(smem); // the routine wants a query: zeros (the dot product is discarded)
    for (int i = static_cast<int>(lane); i < dpad; i += 64) qf[i] = 0.0f;
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    const int sub = static_cast<int>(lane) & 3;
    const uint32_t grp = lane / 4;
    for (uint32_t j0 = blockIdx.x * 16u; j0 < n; j0 += gridDim.x * 16u) {
        const uint32_t j = j0 + grp;
        const bool act = j < n;
        const unsigned char* row = static_cast<const unsigned char*>(data) + static_cast<size_t>(act ? j : 0u) * row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        float_row_distance_bc<DT, M_COS, DS, true>(row, act, qf, d, sub, scale, bias, fmain, fnrm);
        if (act && sub == 0) out[j] = sqrtf(fnrm);
    }
}

hipError_t launch_sq_row_norms(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t d, float scale, float bias, float* d_out, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const size_t lds = static_cast<size_t>((d + 15) & ~15u) * 4;
    uint32_t blocks = (n + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    // the instantiation the search kernel would pick for these rows (launch_dim: static 768, runtime dimension otherwise): same arithmetic either way
    if (dtype == DT_I8 && d == 768) hipLaunchKernelGGL((sq_row_norm_kernel<DT_I8, 768>), dim3(blocks), dim3(64), lds, stream, d_data, row_bytes, n, d, scale, bias, d_out);
    else if (dtype == DT_U8 && d == 768) hipLaunchKernelGGL((sq_row_norm_kernel<DT_U8, 768>), dim3(blocks), dim3(64), lds, stream, d_data, row_bytes, n, d, scale, bias, d_out);
    else if (dtype == DT_I8) hipLaunchKernelGGL((sq_row_norm_kernel<DT_I8, 0>), dim3(blocks), dim3(64), lds, stream, d_data, row_bytes, n, d, scale, bias, d_out);
    else if (dtype == DT_U8) hipLaunchKernelGGL((sq_row_norm_kernel<DT_U8, 0>), dim3(blocks), dim3(64), lds, stream, d_data, row_bytes, n, d, scale, bias, d_out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// One wave per query, lane s = cursor of shard s (S <= 64): every step is a wave-wide arg-best over the list heads,
// ties -> lowest shard (then position, the lists being sorted); the winning lane emits its head and advances.
template <bool GREATER> __global__ __launch_bounds__(64) void merge_kernel(const MergeArgs a) {
    const uint32_t q = blockIdx.x, lane = threadIdx.x;
    if (a.gate64 != nullptr && a.gate64[q >> 6] == 0u) return;
    const bool shard = lane < a.n_shards;
    const size_t dense = static_cast<size_t>(a.nq) * a.k;
    const size_t sl = shard ? lane : 0u;
    const size_t dbase = sl * (a.dist_shard_stride ? a.dist_shard_stride : dense) + static_cast<size_t>(q) * a.k;
    const size_t ibase = sl * (a.id_shard_stride ? a.id_shard_stride : dense) + static_cast<size_t>(q) * a.k;
    uint32_t cur = 0;
    float hd = 0.0f;
    uint64_t hid = ~0ull;
    bool live = false;
    auto fetch = [&]() {
        live = shard && cur < a.k;
        if (live) {
            hid = a.ids[ibase + cur];
            if (hid == ~0ull) live = false; // padded tail
            else hd = a.dists[dbase + cur];
        }
    };
    fetch();
    for (uint32_t j = 0; j < a.k; ++j) {
        float bd = hd;
        uint32_t bl = live ? lane : 64u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(bd, off);
            const uint32_t ol = static_cast<uint32_t>(__shfl_xor(static_cast<int>(bl), off));
            const bool take = ol != 64u && (bl == 64u || cmpf<GREATER>(od, bd) || (!cmpf<GREATER>(bd, od) && ol < bl));
            if (take) { bd = od; bl = ol; }
        }
        const size_t oo = static_cast<size_t>(q) * a.k + j;
        if (bl == 64u) {
            if (lane == 0) {
                a.out_ids[oo] = ~0ull;
                a.out_dists[oo] = __builtin_inff();
            }
        } else if (lane == bl) {
            a.out_ids[oo] = hid;
            a.out_dists[oo] = hd;
            ++cur;
            fetch();
        }
    }
}

hipError_t launch_merge(int metric, const MergeArgs& a, hipStream_t stream) {
    if (a.n_shards == 0 || a.n_shards > 64 || a.k > 65535) return hipErrorInvalidValue;
    const dim3 block(64), grid(a.nq);
    if (metric == M_L2) hipLaunchKernelGGL(merge_kernel<false>, grid, block, 0, stream, a);
    else hipLaunchKernelGGL(merge_kernel<true>, grid, block, 0, stream, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------
// bad[0] counts rows whose degree exceeds the row, bad[1] counts neighbour ids >= n (both make the graph unusable)
__global__ void pack_graph_kernel(const uint32_t* __restrict__ in, uint32_t in_stride, uint32_t n, uint32_t* __restrict__ out, uint32_t out_stride,
                                  uint32_t* __restrict__ bad) {
    const uint32_t row = blockIdx.x;
    if (row >= n) return;
    const uint32_t* r = in + static_cast<size_t>(row) * in_stride;
    const uint32_t deg = r[0];
    if (threadIdx.x == 0 && (deg > in_stride - 1 || deg > out_stride)) atomicAdd(bad, 1u);
    for (uint32_t j = threadIdx.x; j < out_stride; j += blockDim.x) {
        uint32_t id = (j < deg && j + 1 < in_stride) ? r[1 + j] : kInvalidId;
        if (id != kInvalidId && id >= n) {
            atomicAdd(bad + 1, 1u);
            id = kInvalidId;
        }
        out[static_cast<size_t>(row) * out_stride + j] = id;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Bound codes (svsx_index_build_bound_codes): one 128-B row per vector = u8 codes of its first min(d, 120) components
// under one global (scale, bias), zero padding, then at byte 120 ||x_C - xhat||_2 rounded UP (f32) and at byte 124
// sum(code^2) (i32).  The search kernel derives a rigorous lower bound of the L2 distance from them (svsx_search.hip).
__device__ __forceinline__ uint32_t order_key(float f) { // monotone float -> u32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T>
__global__ void codes_minmax_kernel(const T* __restrict__ data, size_t row_floats, uint32_t n, uint32_t dc, uint32_t* __restrict__ mm) {
    float lo = __builtin_inff(), hi = -__builtin_inff();
    bool bad = false;
    for (size_t r = blockIdx.x; r < n; r += gridDim.x) {
        for (uint32_t j = threadIdx.x; j < dc; j += blockDim.x) {
            const float v = static_cast<float>(data[r * row_floats + j]); // float16 rows widen exactly, like the reference's loads
            bad = bad || !(__builtin_fabsf(v) <= 3.0e38f);
            lo = v < lo ? v : lo;
            hi = v > hi ? v : hi;
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        const float l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63u) == 0u) atomicAdd(mm + 2, 1u);
    if ((threadIdx.x & 63u) == 0u) {
        atomicMin(mm, order_key(lo));
        atomicMax(mm + 1, order_key(hi));
    }
}

template <typename T>
__global__ void codes_build_kernel(const T* __restrict__ data, size_t row_floats, uint32_t n, uint32_t dc, float scale, float bias,
                                   unsigned char* __restrict__ codes) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const T* x = data + static_cast<size_t>(r) * row_floats;
    uint32_t* out = reinterpret_cast<uint32_t*>(codes + static_cast<size_t>(r) * 128);
    const float inv = 1.0f / scale;
    double err2 = 0.0;
    int xx = 0;
    for (uint32_t w = 0; w < 30; ++w) {
        uint32_t packed = 0;
        for (uint32_t t = 0; t < 4; ++t) {
            const uint32_t j = w * 4 + t;
            if (j < dc) {
                const float v = static_cast<float>(x[j]);
                float c = __builtin_rintf((v - bias) * inv);
                c = c < 0.0f ? 0.0f : (c > 255.0f ? 255.0f : c);
                const int ci = static_cast<int>(c);
                const double e = static_cast<double>(v) - (static_cast<double>(bias) + static_cast<double>(scale) * static_cast<double>(ci));
                err2 += e * e;
                xx += ci * ci;
                packed |= static_cast<uint32_t>(ci) << (8 * t);
            }
        }
        out[w] = packed;
    }
    float ex = static_cast<float>(__builtin_sqrt(err2) * (1.0 + 1e-9));
    ex = __uint_as_float(__float_as_uint(ex) + 2u); // round up (ex >= 0): the stored value bounds the true error from above
    out[30] = __float_as_uint(ex);
    out[31] = static_cast<uint32_t>(xx);
}

hipError_t launch_codes_minmax(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t dc, uint32_t* d_mm3, hipStream_t stream) {
    const uint32_t init[3] = {0xFFFFFFFFu, 0u, 0u};
    hipError_t e = hipMemcpyAsync(d_mm3, init, sizeof(init), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    const uint32_t blocks = n < 4096u ? n : 4096u;
    if (dtype == DT_F32) hipLaunchKernelGGL(codes_minmax_kernel<float>, dim3(blocks), dim3(64), 0, stream, static_cast<const float*>(d_data), row_bytes / 4, n, dc, d_mm3);
    else if (dtype == DT_F16) hipLaunchKernelGGL(codes_minmax_kernel<_Float16>, dim3(blocks), dim3(64), 0, stream, static_cast<const _Float16*>(d_data), row_bytes / 2, n, dc, d_mm3);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

float codes_key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

hipError_t launch_codes_build(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t dc, float scale, float bias, void* d_codes, hipStream_t stream) {
    if (dtype == DT_F32)
        hipLaunchKernelGGL(codes_build_kernel<float>, dim3((n + 127) / 128), dim3(128), 0, stream, static_cast<const float*>(d_data), row_bytes / 4, n, dc, scale,
                           bias, static_cast<unsigned char*>(d_codes));
    else if (dtype == DT_F16)
        hipLaunchKernelGGL(codes_build_kernel<_Float16>, dim3((n + 127) / 128), dim3(128), 0, stream, static_cast<const _Float16*>(d_data), row_bytes / 2, n, dc,
                           scale, bias, static_cast<unsigned char*>(d_codes));
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_pack_graph(const uint32_t* d_in, uint32_t in_stride, uint32_t n, uint32_t* d_out, uint32_t out_stride, uint32_t* d_bad, hipStream_t stream) {
    hipLaunchKernelGGL(pack_graph_kernel, dim3(n), dim3(64), 0, stream, d_in, in_stride, n, d_out, out_stride, d_bad);
    return hipGetLastError();
}

} // namespace svsx
