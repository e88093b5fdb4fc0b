// svsx_codes.hip -- graph-inline bound records ("fat" adjacency rows) for the greedy-search kernel (gfx950).
//
// No reference counterpart: a result-neutral traffic reduction.  The search kernel is bound by the number
// of distinct 128-B lines it requests from the fabric (tools/ubench/gather_bench: ~49 G random lines/s whatever the row
// size <= 128 B).  Here every node's adjacency row carries, next to the neighbour ids, a 32-byte record PER NEIGHBOUR -- a
// copy of that neighbour's record -- so one expansion reads ONE contiguous block of 36 bytes x degree slots (18 lines at
// R = 64, address known as soon as the node is) and a bound on the distance of all 64 candidates is evaluated lane-parallel
// (lane c <-> neighbour c); only candidates the bound cannot exclude have their vector row fetched.
//
// Geometry.  Q = DC orthonormal directions (rows; the leading principal directions of a data sample, host: svsx_records.hip),
// z = the CENTRED vector the metric sees (L2 and inner product: x - mean; cosine: x / ||x|| - mean of the unit vectors; scalar-quantized
// rows: the decompressed values), y = Q z its DC coordinates, r = z - Q^T y the remainder, ||r||^2 = ||z||^2 - ||y||^2 exactly.
// For a query with coordinates u and remainder norm tq (L2: of q - mean; similarities: of q resp. q / ||q||, NOT centred):
//     L2      ||q - x||^2  =  ||u - y||^2 + ||r_q - r_x||^2  >=  ||u - y||^2 + (tq - tx)^2
//     IP/cos  q . x        =  q . z + q . mean  =  u . y + r_q . r_x + q . mean  <=  u . y + tq tx + q . mean      (round 6: centred)
// Record (RB = 32, 64 or 128 bytes -- the host picks the smallest whose directions carry the sample's energy, svsx_api.hip),
// everything rounded in the SAFE direction:
//     L2      (RB - 4) x u8 codes c_i = clamp(rint((y_i - b_i) / s), 0, 255) | f16 ex (up) | f16 tx (nearest)
//     IP/cos  (RB - 6) x i8 codes c_i = clamp(rint(y_i / s), -127, 127)      | f16 ex (up) | f16 ny (up) | f16 tx (up)
// ex >= ||y - yhat|| (coding error incl. clamping, in units of s), ny >= ||yhat|| (units of s), tx = ||r|| (units of
// `tail_unit`).  The kernel (svsx_search.hip, CODES == 2) adds explicit slack for every float rounding on the query side and
// rejects a candidate only if the bound proves SearchBuffer::can_skip (search_buffer.h:338-340) would drop it anyway.
#include "svsx_device.h"
#include "svsx_kernels.h"

namespace svsx {

namespace {

constexpr int kMaxDc = 124;  // 128-byte records
constexpr int kChunk = 31;   // coordinates accumulated per sweep over the vector (registers: 31 doubles per lane)

__device__ __forceinline__ double wave_sum(double v) {
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ uint32_t f32_order_key(float f) { // monotone map float -> u32
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// smallest float16 >= v (v >= 0); +inf when it does not fit (the bound then never fires for this vector)
__device__ __forceinline__ uint16_t half_round_up(double v) {
    if (!(v == v)) return 0x7C00u;
    _Float16 h = static_cast<_Float16>(static_cast<float>(v));
    uint16_t bits = __builtin_bit_cast(uint16_t, h);
    if (static_cast<double>(h) < v) bits = static_cast<uint16_t>(bits + 1u); // next representable (positive values; 0x7BFF + 1 = inf)
    return bits;
}

// element j of a row as the reference's kernels see it: widened exactly; scalar-quantized rows decompressed in float32 with
// separately rounded multiply and add (quantization/scalar/scalar.h:44-46)
template <int DT> __device__ __forceinline__ double load_elem(const PcaArgs& a, const unsigned char* row, uint32_t j) {
    float f;
    if constexpr (DT == DT_F32) f = reinterpret_cast<const float*>(row)[j];
    else if constexpr (DT == DT_F16) f = static_cast<float>(reinterpret_cast<const _Float16*>(row)[j]);
    else if constexpr (DT == DT_I8) f = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(row)[j]));
    else f = static_cast<float>(static_cast<unsigned>(row[j]));
    if (a.decomp) {
        const float t = a.sq_scale * f;
        f = t + a.sq_bias;
    }
    return static_cast<double>(f);
}

// Coordinates of one vector: z into LDS (`vs`, d doubles), then y_i = Q_i . z for i < dc into LDS (`ys`), n2 = ||z||^2.
// Lane l owns the elements j = l, l + 64, ...: coalesced reads of the f64 basis rows; kChunk coordinates per sweep.
// z = x - mean (L2, inner product) or x / ||x|| - mean (cosine: `mean` is then the mean of the UNIT vectors).  Round 6: the similarity
// metrics are centred too -- q . x = q . (x - mean) + q . mean, the second term is a per-query constant the kernel adds to its bound -- so that
// data with a common offset (non-negative rows, uint8 rows, embeddings with a dominant mean direction) spends its 8-bit codes on what
// distinguishes the vectors, not on the offset.  `degenerate`: a zero row under cosine (the reference's distance is NaN: never bounded).
template <int DT>
__device__ __forceinline__ void coordinates(const PcaArgs& a, const unsigned char* row, double* vs, double* ys, uint32_t lane, double& n2, bool& degenerate) {
    double part = 0.0;
    degenerate = false;
    if (a.normalize) {
        for (uint32_t j = lane; j < a.d; j += 64) {
            const double x = load_elem<DT>(a, row, j);
            vs[j] = x;
            part += x * x;
        }
        const double raw2 = wave_sum(part);
        degenerate = !(raw2 > 0.0);
        const double inv = raw2 > 0.0 ? 1.0 / sqrt(raw2) : 0.0;
        part = 0.0;
        for (uint32_t j = lane; j < a.d; j += 64) { // (every lane rewrites its own elements: no barrier needed in between)
            const double z = vs[j] * inv - static_cast<double>(a.mean[j]);
            vs[j] = z;
            part += z * z;
        }
    } else {
        for (uint32_t j = lane; j < a.d; j += 64) {
            const double z = load_elem<DT>(a, row, j) - static_cast<double>(a.mean[j]);
            vs[j] = z;
            part += z * z;
        }
    }
    n2 = wave_sum(part);
    __syncthreads();
    for (uint32_t c0 = 0; c0 < a.dc; c0 += kChunk) {
        double y[kChunk];
#pragma unroll
        for (int i = 0; i < kChunk; ++i) y[i] = 0.0;
        for (uint32_t j = lane; j < a.d; j += 64) {
            const double z = vs[j];
#pragma unroll
            for (int i = 0; i < kChunk; ++i)
                if (c0 + static_cast<uint32_t>(i) < a.dc) y[i] = __builtin_fma(a.q64[static_cast<size_t>(c0 + i) * a.d + j], z, y[i]);
        }
#pragma unroll
        for (int i = 0; i < kChunk; ++i) {
            const double t = wave_sum(y[i]);
            if (lane == 0 && c0 + static_cast<uint32_t>(i) < a.dc) ys[c0 + i] = t;
        }
    }
    __syncthreads();
}

} // namespace

// pass 1: per-coordinate min / max of y (order keys), max remainder norm, count of non-finite vectors
template <int DT> __global__ __launch_bounds__(64) void pca_stats_kernel(const PcaArgs a) {
    extern __shared__ double vs[];
    double* ys = vs + a.d;
    const uint32_t lane = threadIdx.x;
    float lo[2] = {__builtin_inff(), __builtin_inff()}, hi[2] = {-__builtin_inff(), -__builtin_inff()};
    double tmax = 0.0;
    uint32_t bad = 0;
    for (uint32_t r = blockIdx.x; r < a.n; r += gridDim.x) {
        double n2;
        bool degenerate;
        coordinates<DT>(a, static_cast<const unsigned char*>(a.data) + static_cast<size_t>(r) * a.row_bytes, vs, ys, lane, n2, degenerate);
        double ny2 = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t i = lane + 64u * t;
            if (i < a.dc) {
                const double yv = ys[i];
                ny2 += yv * yv;
                const float f = static_cast<float>(yv);
                lo[t] = f < lo[t] ? f : lo[t];
                hi[t] = f > hi[t] ? f : hi[t];
            }
        }
        ny2 = wave_sum(ny2);
        if (!(n2 - n2 == 0.0)) bad = 1;
        const double t2 = n2 - ny2;
        const double tn = t2 > 0.0 ? sqrt(t2) : 0.0;
        tmax = tn > tmax ? tn : tmax;
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const uint32_t i = lane + 64u * t;
        if (i < a.dc && lo[t] <= hi[t]) {
            atomicMin(&a.stat_min[i], f32_order_key(lo[t]));
            atomicMax(&a.stat_max[i], f32_order_key(hi[t]));
        }
    }
    if (lane == 0) {
        atomicMax(a.stat_tail, f32_order_key(static_cast<float>(tmax) * 1.0000002f));
        if (bad) atomicAdd(a.stat_bad, 1u);
    }
}

// pass 2: the record of every vector (a.rb bytes: codes, then the float16 fields in its last 4 / 6 bytes)
template <int DT> __global__ __launch_bounds__(64) void pca_encode_kernel(const PcaArgs a) {
    extern __shared__ double vs[];
    double* ys = vs + a.d;
    __shared__ uint32_t recw[32];
    unsigned char* recb = reinterpret_cast<unsigned char*>(recw);
    const uint32_t lane = threadIdx.x;
    const double s = static_cast<double>(a.step), inv = 1.0 / s;
    for (uint32_t r = blockIdx.x; r < a.n; r += gridDim.x) {
        double n2;
        bool degenerate;
        coordinates<DT>(a, static_cast<const unsigned char*>(a.data) + static_cast<size_t>(r) * a.row_bytes, vs, ys, lane, n2, degenerate);
        if (lane < 32u) recw[lane] = 0u;
        __syncthreads();
        double e2 = 0.0, ny2 = 0.0, nh2 = 0.0;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint32_t i = lane + 64u * t;
            if (i < a.dc) {
                const double yv = ys[i];
                double c, yh;
                if (a.signed_codes) {
                    c = rint(yv * inv);
                    c = c < -127.0 ? -127.0 : (c > 127.0 ? 127.0 : c);
                    if (!(c == c)) c = 0.0;
                    yh = s * c;
                } else {
                    const double b = static_cast<double>(a.bias[i]);
                    c = rint((yv - b) * inv);
                    c = c < 0.0 ? 0.0 : (c > 255.0 ? 255.0 : c);
                    if (!(c == c)) c = 0.0;
                    yh = b + s * c;
                }
                const double rr = yv - yh;
                e2 += rr * rr;
                ny2 += yv * yv;
                nh2 += yh * yh;
                recb[i] = static_cast<unsigned char>(static_cast<int>(c) & 0xFF);
            }
        }
        e2 = wave_sum(e2);
        ny2 = wave_sum(ny2);
        nh2 = wave_sum(nh2);
        __syncthreads();
        const double t2 = n2 - ny2;
        const double tail = t2 > 0.0 ? sqrt(t2) : 0.0;
        uint16_t exh = half_round_up(sqrt(e2) * inv * (1.0 + 1e-12));
        if (degenerate) exh = 0x7C00u; // zero vector under cosine: the reference's distance is NaN, never bound it
        const uint32_t nw = a.rb / 4u;
        if (lane == 0) {
            if (a.signed_codes) { // codes | ex | ny | tx, all upper bounds
                const uint16_t nyh = half_round_up(sqrt(nh2) * inv * (1.0 + 1e-12));
                const uint16_t txh = half_round_up(tail / static_cast<double>(a.tail_unit) * (1.0 + 1e-12));
                recw[nw - 2] = (recw[nw - 2] & 0xFFFFu) | (static_cast<uint32_t>(exh) << 16);
                recw[nw - 1] = static_cast<uint32_t>(nyh) | (static_cast<uint32_t>(txh) << 16);
            } else {              // codes | ex (up) | tx (nearest)
                const _Float16 txn = static_cast<_Float16>(static_cast<float>(tail / static_cast<double>(a.tail_unit)));
                recw[nw - 1] = static_cast<uint32_t>(exh) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, txn)) << 16);
            }
        }
        __syncthreads();
        if (lane < nw) a.records[static_cast<size_t>(r) * nw + lane] = recw[lane];
        __syncthreads();
    }
}

// pass 3a (indexes of at most 2^26 nodes): ceil(used slots / 8) of every node -- the number of leading 8-slot groups of its block that hold a neighbour
__global__ __launch_bounds__(256) void degree_class_kernel(const uint32_t* __restrict__ graph, uint32_t gstride, uint32_t n, uint8_t* __restrict__ cls) {
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const uint32_t* row = graph + static_cast<size_t>(v) * gstride;
    // the class bounds the SLOTS the search kernel reads (slot_limit = class * 8), so it follows the last slot that holds a neighbour, not the
    // number of neighbours: a caller-supplied row (svsx_index_create_device_view) with a hole in it loses nothing behind the hole
    uint32_t used = 0;
    for (uint32_t j = 0; j < gstride; ++j) used = row[j] != kInvalidId ? j + 1u : used;
    cls[v] = static_cast<uint8_t>((used + 7u) >> 3);
}

// pass 3: node v -> per chunk of <= 64 slots [ids | plane 0 | ... | plane RW-1]; plane p = bytes [16p, 16p + 16) of each
// neighbour's record, so that one load instruction of the search kernel reads one contiguous plane.  With `cls` the id word of a neighbour
// carries that neighbour's degree class in bits [degshift, 31): the search kernel finds it in the candidate buffer when the neighbour is
// expanded and requests only the lines of the block that hold neighbours (svsx_search_kernel.h, main loop).
__global__ __launch_bounds__(64) void fat_build_kernel(const uint32_t* __restrict__ graph, uint32_t gstride, uint32_t n, const uint32_t* __restrict__ records,
                                                       uint32_t rb, unsigned char* __restrict__ fat, size_t fat_stride, const uint8_t* __restrict__ cls, uint32_t degshift) {
    // rb = record bytes: 32 / 64 / 128 (whole 16-byte planes) or 28 (round 6: a 16-byte plane and a 12-byte one -- 64 x (4 + 28) bytes = 16 lines)
    const uint32_t lane = threadIdx.x;
    const uint32_t nw = rb / 4u, full = nw / 4u, tail = nw % 4u; // words per record, full planes, words of the last (narrow) plane
    for (uint32_t v = blockIdx.x; v < n; v += gridDim.x) {
        unsigned char* out = fat + static_cast<size_t>(v) * fat_stride;
        for (uint32_t gbase = 0; gbase < gstride; gbase += 64) {
            const uint32_t cs = (gstride - gbase) < 64u ? (gstride - gbase) : 64u;
            uint32_t* ids = reinterpret_cast<uint32_t*>(out + static_cast<size_t>(gbase) * (4u + rb));
            uint4* planes = reinterpret_cast<uint4*>(ids + cs);
            if (lane < cs) {
                const uint32_t nb = graph[static_cast<size_t>(v) * gstride + gbase + lane];
                ids[lane] = (nb != kInvalidId && cls != nullptr) ? (nb | (static_cast<uint32_t>(cls[nb]) << degshift)) : nb;
                const uint32_t* r = records + static_cast<size_t>(nb != kInvalidId ? nb : 0u) * nw;
                for (uint32_t p = 0; p < full; ++p) {
                    uint4 q = make_uint4(0, 0, 0, 0);
                    if (nb != kInvalidId) q = make_uint4(r[4 * p], r[4 * p + 1], r[4 * p + 2], r[4 * p + 3]);
                    planes[static_cast<size_t>(p) * cs + lane] = q;
                }
                if (tail != 0u) {
                    uint32_t* narrow = reinterpret_cast<uint32_t*>(planes + static_cast<size_t>(full) * cs) + tail * lane;
                    for (uint32_t t = 0; t < tail; ++t) narrow[t] = nb != kInvalidId ? r[4 * full + t] : 0u;
                }
            }
        }
    }
}

__global__ void set_word_kernel(int* w, int v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

hipError_t launch_set_word(int* d_word, int value, hipStream_t stream) {
    hipLaunchKernelGGL(set_word_kernel, dim3(1), dim3(1), 0, stream, d_word, value);
    return hipGetLastError();
}

__global__ void add_word_kernel(uint32_t* w, uint32_t v) { atomicAdd(w, v); }

hipError_t launch_add_word(uint32_t* d_word, uint32_t value, hipStream_t stream) {
    hipLaunchKernelGGL(add_word_kernel, dim3(1), dim3(1), 0, stream, d_word, value);
    return hipGetLastError();
}

float pca_key_to_float(uint32_t k) {
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

static uint32_t grid_for(uint32_t n) { return n < 32768u ? (n ? n : 1u) : 32768u; }

template <int DT> static hipError_t launch_both(const PcaArgs& a, bool encode, hipStream_t stream) {
    const size_t lds = (static_cast<size_t>(a.d) + 128) * sizeof(double);
    if (lds > 60 * 1024 || a.dc > static_cast<uint32_t>(kMaxDc) || (a.rb != 28 && a.rb != 32 && a.rb != 64 && a.rb != 128)) return hipErrorInvalidValue;
    if (encode) hipLaunchKernelGGL(pca_encode_kernel<DT>, dim3(grid_for(a.n)), dim3(64), lds, stream, a);
    else hipLaunchKernelGGL(pca_stats_kernel<DT>, dim3(grid_for(a.n)), dim3(64), lds, stream, a);
    return hipGetLastError();
}

static hipError_t launch_by_dtype(const PcaArgs& a, bool encode, hipStream_t stream) {
    switch (a.dtype) {
        case DT_F32: return launch_both<DT_F32>(a, encode, stream);
        case DT_F16: return launch_both<DT_F16>(a, encode, stream);
        case DT_I8: return launch_both<DT_I8>(a, encode, stream);
        case DT_U8: return launch_both<DT_U8>(a, encode, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_pca_stats(const PcaArgs& a, hipStream_t stream) { return launch_by_dtype(a, false, stream); }
hipError_t launch_pca_encode(const PcaArgs& a, hipStream_t stream) { return launch_by_dtype(a, true, stream); }

hipError_t launch_fat_build(const uint32_t* d_graph, uint32_t gstride, uint32_t n, const void* d_records, uint32_t record_bytes, void* d_fat, size_t fat_stride,
                            uint8_t* d_class_scratch, uint32_t degshift, hipStream_t stream) {
    if (d_class_scratch != nullptr) {
        hipLaunchKernelGGL(degree_class_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_graph, gstride, n, d_class_scratch);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fat_build_kernel, dim3(grid_for(n)), dim3(64), 0, stream, d_graph, gstride, n, static_cast<const uint32_t*>(d_records), record_bytes,
                       static_cast<unsigned char*>(d_fat), fat_stride, static_cast<const uint8_t*>(d_class_scratch), degshift);
    return hipGetLastError();
}

} // namespace svsx
