// svsx_codes.hip -- graph-inline bound codes ("fat" adjacency rows) for the greedy-search kernel (gfx950).
//
// No reference counterpart: a result-neutral traffic reduction.  The search kernel is bound by the number
// of distinct 128-B lines it requests from the fabric (tools/ubench/gather_bench: ~49 G random lines/s whatever the row
// size <= 128 B), and at window 128 most of those lines were per-candidate code rows.  Here every node's adjacency row
// carries, next to the 64 neighbour ids, a 32-byte record PER NEIGHBOUR -- a copy of that neighbour's record -- so one
// expansion reads ONE contiguous 2.3-KB block (18 lines, address known as soon as the node is) instead of 2 + ~50
// scattered lines, and the lower bound of all 64 candidates is evaluated lane-parallel (lane c <-> neighbour c).
//
// Record of vector x (32 B):  28 x u8 codes of the leading components of y = P (x - mean) | f16 ex | f16 tx
//   P      d x d rotation, rows = eigenvectors of the sample covariance by descending eigenvalue (host: Jacobi, f64),
//          stored as float32; eps >= || P P^T - I ||_F is measured in f64 AFTER the rounding to float32
//   codes  c_i = clamp(rint((y_i - b_i) / s), 0, 255), one global step s, per-component bias b_i
//   ex     || y_C - (b + s c) ||  in units of s, rounded UP to float16       (coding error incl. clamping)
//   tx     || y_T ||  (the components the codes do not cover) in units of `tail_unit`, nearest float16
// With u = P (q - mean) for the query, in real arithmetic
//     (1 + eps) ||q - x||^2  >=  ||u_C - y_C||^2 + (||u_T|| - ||y_T||)^2 ,   ||u_C - y_C|| >= s sqrt(I) - eq - ex ,
// I = sum (cq_i - cx_i)^2 an exact integer.  The kernel (svsx_search.hip, CODES == 2) rejects a candidate only if this
// bound, with explicit slack for every float rounding on the query side, exceeds the worst buffer entry -- i.e. only
// candidates SearchBuffer::can_skip (search_buffer.h:338-340) would drop anyway.
#include "svsx_device.h"
#include "svsx_kernels.h"

namespace svsx {

namespace {

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ uint32_t f32_order_key(float f) { // monotone map float -> u32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// smallest float16 >= v (v >= 0); +inf when it does not fit
__device__ __forceinline__ uint16_t half_round_up(double v) {
    _Float16 h = static_cast<_Float16>(static_cast<float>(v));
    uint16_t bits = __builtin_bit_cast(uint16_t, h);
    if (static_cast<double>(h) < v) bits = static_cast<uint16_t>(bits + 1u); // next representable (positive values; 0x7BFF + 1 = inf)
    return bits;
}

template <typename T> __device__ __forceinline__ double load_elem(const unsigned char* row, uint32_t j) {
    return static_cast<double>(static_cast<float>(reinterpret_cast<const T*>(row)[j]));
}

// y = P (x - mean) in f64 for one vector, component i = lane + 64 t held by lane (t < 2: d <= 128).  vs = LDS, d doubles.
template <typename T>
__device__ __forceinline__ void project(const PcaArgs& a, const unsigned char* row, double* vs, uint32_t lane, double (&y)[2]) {
    for (uint32_t j = lane; j < a.d; j += 64) vs[j] = load_elem<T>(row, j) - static_cast<double>(a.mean[j]);
    __syncthreads();
    y[0] = 0.0;
    y[1] = 0.0;
    const bool has0 = lane < a.d, has1 = lane + 64 < a.d;
    const float* r0 = a.pt + static_cast<size_t>(lane) * a.pd;       // row-major rotation, `pd` floats per row
    const float* r1 = a.pt + static_cast<size_t>(lane + 64u) * a.pd;
    for (uint32_t j = 0; j < a.d; ++j) {
        const double v = vs[j];
        if (has0) y[0] = __builtin_fma(static_cast<double>(r0[j]), v, y[0]);
        if (has1) y[1] = __builtin_fma(static_cast<double>(r1[j]), v, y[1]);
    }
    __syncthreads();
}

} // namespace

// pass 1: per-component min / max of y (order keys), max tail norm, count of non-finite vectors
template <typename T> __global__ __launch_bounds__(64) void pca_stats_kernel(const PcaArgs a) {
    __shared__ double vs[128];
    const uint32_t lane = threadIdx.x;
    float lo[2] = {__builtin_inff(), __builtin_inff()}, hi[2] = {-__builtin_inff(), -__builtin_inff()};
    double tmax = 0.0;
    uint32_t bad = 0;
    for (uint32_t r = blockIdx.x; r < a.n; r += gridDim.x) {
        double y[2];
        project<T>(a, static_cast<const unsigned char*>(a.data) + static_cast<size_t>(r) * a.row_bytes, vs, lane, y);
        double t2 = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t i = lane + 64u * t;
            if (i < a.d) {
                const float f = static_cast<float>(y[t]);
                if (!(f - f == 0.0f)) bad = 1;
                lo[t] = f < lo[t] ? f : lo[t];
                hi[t] = f > hi[t] ? f : hi[t];
                if (i >= a.dc) t2 += y[t] * y[t];
            }
        }
        t2 = wave_sum(t2);
        const double tn = sqrt(t2);
        tmax = tn > tmax ? tn : tmax;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t i = lane + 64u * t;
        if (i < a.d && lo[t] <= hi[t]) {
            atomicMin(&a.stat_min[i], f32_order_key(lo[t]));
            atomicMax(&a.stat_max[i], f32_order_key(hi[t]));
        }
    }
    if (lane == 0) {
        atomicMax(a.stat_tail, f32_order_key(static_cast<float>(tmax) * 1.0000002f));
    }
    if (__ballot(bad != 0) != 0ull && lane == 0) atomicAdd(a.stat_bad, 1u);
}

// pass 2: the 32-byte record of every vector
template <typename T> __global__ __launch_bounds__(64) void pca_encode_kernel(const PcaArgs a) {
    __shared__ double vs[128];
    const uint32_t lane = threadIdx.x;
    const double s = static_cast<double>(a.step), inv = 1.0 / s;
    for (uint32_t r = blockIdx.x; r < a.n; r += gridDim.x) {
        double y[2];
        project<T>(a, static_cast<const unsigned char*>(a.data) + static_cast<size_t>(r) * a.row_bytes, vs, lane, y);
        double e2 = 0.0, t2 = 0.0;
        uint32_t code = 0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t i = lane + 64u * t;
            if (i < a.d) {
                if (i < a.dc) {
                    const double b = static_cast<double>(a.bias[i]);
                    double c = rint((y[t] - b) * inv);
                    c = c < 0.0 ? 0.0 : (c > 255.0 ? 255.0 : c);
                    if (!(c == c)) c = 0.0;
                    code = static_cast<uint32_t>(c);
                    const double rr = y[t] - (b + s * c);
                    e2 += rr * rr;
                } else {
                    t2 += y[t] * y[t];
                }
            }
        }
        e2 = wave_sum(e2);
        t2 = wave_sum(t2);
        // bytes 4w .. 4w+3 -> lane w (dc <= 28 -> 7 words)
        uint32_t w = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) w |= (static_cast<uint32_t>(__shfl(static_cast<int>(code), static_cast<int>((lane & 7u) * 4u + t))) & 0xFFu) << (8 * t);
        if (lane == 7) {
            const uint16_t exh = half_round_up(sqrt(e2) * inv * (1.0 + 1e-12));
            const _Float16 txh = static_cast<_Float16>(static_cast<float>(sqrt(t2) / static_cast<double>(a.tail_unit)));
            w = static_cast<uint32_t>(exh) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, txh)) << 16);
        }
        if (lane < 8) a.records[static_cast<size_t>(r) * 8 + lane] = w;
    }
}

// pass 3: node v -> [ids of chunk | plane 0 (first 16 B of each neighbour's record) | plane 1] per chunk of <= 64 slots
__global__ __launch_bounds__(64) void fat_build_kernel(const uint32_t* __restrict__ graph, uint32_t gstride, uint32_t n, const uint4* __restrict__ records,
                                                       unsigned char* __restrict__ fat, size_t fat_stride) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t v = blockIdx.x; v < n; v += gridDim.x) {
        unsigned char* out = fat + static_cast<size_t>(v) * fat_stride;
        for (uint32_t gbase = 0; gbase < gstride; gbase += 64) {
            const uint32_t cs = (gstride - gbase) < 64u ? (gstride - gbase) : 64u;
            uint32_t* ids = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(gbase) * 36u);
            uint4* p0 = reinterpret_cast<uint4*>(ids + cs);
            uint4* p1 = p0 + cs;
            if (lane < cs) {
                const uint32_t nb = graph[static_cast<size_t>(v) * gstride + gbase + lane];
                uint4 r0 = make_uint4(0, 0, 0, 0), r1 = make_uint4(0, 0, 0, 0);
                if (nb != kInvalidId) {
                    r0 = records[static_cast<size_t>(nb) * 2];
                    r1 = records[static_cast<size_t>(nb) * 2 + 1];
                }
                ids[lane] = nb;
                p0[lane] = r0;
                p1[lane] = r1;
            }
        }
    }
}

__global__ void set_word_kernel(int* w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

hipError_t launch_set_word(int* d_word, int value, hipStream_t stream) {
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, stream, d_word, value);
    return hipGetLastError();
}

float pca_key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

static uint32_t grid_for(uint32_t n) { return n < 65536u ? (n ? n : 1u) : 65536u; }

hipError_t launch_pca_stats(const PcaArgs& a, hipStream_t stream) {
    if (a.d > 128) return hipErrorInvalidValue;
    if (a.dtype == DT_F32) hipLaunchKernelGGL(pca_stats_kernel<float>, dim3(grid_for(a.n)), dim3(64), 0, stream, a);
    else if (a.dtype == DT_F16) hipLaunchKernelGGL(pca_stats_kernel<_Float16>, dim3(grid_for(a.n)), dim3(64), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_pca_encode(const PcaArgs& a, hipStream_t stream) {
    if (a.d > 128 || a.dc > 28) return hipErrorInvalidValue;
    if (a.dtype == DT_F32) hipLaunchKernelGGL(pca_encode_kernel<float>, dim3(grid_for(a.n)), dim3(64), 0, stream, a);
    else if (a.dtype == DT_F16) hipLaunchKernelGGL(pca_encode_kernel<_Float16>, dim3(grid_for(a.n)), dim3(64), 0, stream, a);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fat_build(const uint32_t* d_graph, uint32_t gstride, uint32_t n, const void* d_records, void* d_fat, size_t fat_stride, hipStream_t stream) {
    hipLaunchKernelGGL(fat_build_kernel, dim3(grid_for(n)), dim3(64), 0, stream, d_graph, gstride, n, static_cast<const uint4*>(d_records),
                       static_cast<unsigned char*>(d_fat), fat_stride);
    return hipGetLastError();
}

} // namespace svsx
