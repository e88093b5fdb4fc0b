// svsx_builder.hip -- batch-synchronous Vamana construction with every step on the device ("next" scope).
//
// STAGED FOR ROUND 3: cross-compiles for gfx950 against the round-2 headers, has never run.  Round 2's builder
// (scalablevectorsearch_amd/build.py) does candidate-pool assembly and reverse-edge bookkeeping with torch tensor ops; this file is
// the same algorithm as kernels + one library sort, so that `build.py` becomes a thin wrapper of one C-ABI call.
//
// Reference structure followed (index/vamana/vamana_build.h:232-599, index.h:426-439):
//   construct(alpha = 1, then alpha = params.alpha), each a sequence of batches of vertices:
//     generate_neighbors  greedy search with the build window, candidates = full search history + current neighbours
//                         (vamana_build.h:54-91, 330-470), sorted by TotalOrder, deduplicated, truncated to max_candidate_pool_size
//                         (:441-446), heuristic_prune_neighbors(max_degree, params.alpha) (:447-460; "we purposely pass params_.alpha", :271)
//     add_reverse_edges   every new edge (v -> u) offers v to u; an adjacency list that would exceed max_degree is re-pruned to
//                         prune_to with the pass's alpha (:484-599)
// The reference updates the graph under per-vertex locks from many threads, so its result depends on the thread schedule; here a batch is
// synchronous (search all, prune all, then reverse edges in (destination, source) order), which makes the graph a deterministic function of
// the input.  Graph QUALITY (recall vs window) is what is compared with the reference (tests/test_gpu_configs.py builder gate).
//
// Steps per batch [lo, hi):
//   1. search_kernel (svsx_search_device) with the vertex's own row as the query, k = 1, history = expanded nodes + exact distances
//   2. pool_kernel        one wave per vertex: history + current neighbours (distances by the row-group routines of svsx_device.h),
//                         dedupe by id, sort by (distance, id) -- two bitonic sorts of 64-bit keys in LDS -- truncate
//   3. prune_kernel       (svsx_build.hip) -> the vertex's new adjacency row, written straight into the graph
//   4. edge_keys_kernel   one thread per new edge: key = dst << 32 | src unless src is already in dst's row
//   5. rocprim radix sort of the keys (the one library call: a plain sort)
//   6. reverse_scatter_kernel  the first edge of every destination appends the group to the row while it fits and records
//                         (dst, first overflow key, count) otherwise
//   7. pool_kernel (row + overflow candidates, all distances computed) + prune_kernel(prune_to, pass alpha) + scatter_rows_kernel
#include "svsx_device.h"
#include "svsx_kernels.h"
#include "../../include/svsx.h"

#include <cstring> // rocprim 7.2 headers use memset without including it
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <string>

namespace svsx {

// ---------------------------------------------------------------------------------------------------------------------------------
// order-preserving keys
// ---------------------------------------------------------------------------------------------------------------------------------
// ascending unsigned order == ascending float order (finite values; -0 is folded into +0 so that equal distances tie on the id)
__host__ __device__ inline uint32_t order_bits(float x) {
    if (x == 0.0f) x = 0.0f;
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float order_bits_inv(uint32_t e) {
    const uint32_t u = (e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e;
    float x;
    __builtin_memcpy(&x, &u, 4);
    return x;
}
// best-first key of a (distance, id) pair: smaller key = better neighbour; ties by lower id (TotalOrder, vamana_build.h:441)
template <bool GREATER> __host__ __device__ inline uint64_t pool_key(float dist, uint32_t id) {
    const uint32_t e = order_bits(dist);
    return (static_cast<uint64_t>(GREATER ? ~e : e) << 32) | id;
}
template <bool GREATER> __host__ __device__ inline float pool_key_dist(uint64_t key) {
    const uint32_t e = static_cast<uint32_t>(key >> 32);
    return order_bits_inv(GREATER ? ~e : e);
}

constexpr uint64_t kNoKey = ~0ull;

// One wave sorts N (power of two) 64-bit keys in LDS ascending: the classic bitonic network, N/2 compare-exchanges per stage spread
// over the 64 lanes.  Stages are separated by lds_order(): a wave's LDS operations execute in program order.
__device__ inline void wave_bitonic_sort(uint64_t* keys, uint32_t n, uint32_t lane) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (n >> 1); t += 64) {
                const uint32_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // t with a zero bit inserted at log2(j)
                const uint32_t l = i | j;
                const uint64_t a = keys[i], b = keys[l];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[l] = a;
                }
            }
            lds_order();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// candidate pools
// ---------------------------------------------------------------------------------------------------------------------------------
struct PoolArgs {
    const void* data;
    size_t row_bytes;
    uint32_t d;
    float sq_scale, sq_bias;
    const uint32_t* node_ids;   // vertex of wave v = node_ids[v * node_stride]
    uint32_t node_stride;
    uint32_t n_nodes;
    // A: (id, distance) lists computed elsewhere -- the search history of step 1 (may be null)
    const uint32_t* a_ids;
    const float* a_dists;
    const uint32_t* a_len;      // per wave; entries beyond a_cap were not recorded
    uint32_t a_cap;
    // B: the vertex's current adjacency row; distances computed here
    const uint32_t* graph;
    uint32_t gstride, max_degree;
    // C: sources of the sorted edge keys [c_start, c_start + c_cnt) of this wave (null for step 2); distances computed here
    const uint64_t* c_keys;
    const uint32_t* c_start;    // c_start[v * node_stride], c_cnt[v * node_stride]
    const uint32_t* c_cnt;
    uint32_t sort_n;            // power of two >= a_cap + max_degree + max c_cnt
    uint32_t pool_size;
    int32_t* pool_ids;          // n_nodes x pool_size, best first, -1 padded
    float* pool_dists;
};

template <int DT, int METRIC, bool DECOMP> __global__ __launch_bounds__(64) void pool_kernel(const PoolArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool GREATER = METRIC != M_L2;
    constexpr bool BC = use_block_cyclic<DT>();
    constexpr int G = BC ? 4 : 16 / Elem<DT>::E;
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x;
    const uint32_t v = blockIdx.x;
    const int d = static_cast<int>(a.d);
    const int dpad = (d + 15) & ~15;
    const uint32_t N = a.sort_n;

    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                 // N
    float* qf = reinterpret_cast<float*>(keys + N);                     // dpad: the vertex's vector, widened to f32
    uint32_t* cid = reinterpret_cast<uint32_t*>(qf + dpad);             // N
    float* cdist = reinterpret_cast<float*>(cid + N);                   // N

    const uint32_t self = a.node_ids[static_cast<size_t>(v) * a.node_stride];
    const unsigned char* data = static_cast<const unsigned char*>(a.data);
    const unsigned char* srow = data + static_cast<size_t>(self) * a.row_bytes;
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

    // ---- gather: A (with distances), then B and C (ids only) -------------------------------------------------------------------
    uint32_t n_a = 0;
    if (a.a_ids != nullptr) {
        n_a = a.a_len[v] < a.a_cap ? a.a_len[v] : a.a_cap;
        for (uint32_t i = lane; i < n_a; i += 64) {
            cid[i] = a.a_ids[static_cast<size_t>(v) * a.a_cap + i];
            cdist[i] = a.a_dists[static_cast<size_t>(v) * a.a_cap + i];
        }
    }
    uint32_t total = n_a;
    const uint32_t* grow = a.graph + static_cast<size_t>(self) * a.gstride;
    for (uint32_t j0 = 0; j0 < a.max_degree; j0 += 64) { // adjacency ids are packed from slot 0
        const uint32_t j = j0 + lane;
        const uint32_t id = j < a.max_degree ? grow[j] : kInvalidId;
        const unsigned long long m = __ballot(id != kInvalidId);
        if (id != kInvalidId) cid[total + __popcll(m & ((1ull << lane) - 1ull))] = id;
        total += __popcll(m);
    }
    if (a.c_keys != nullptr) {
        const uint32_t c0 = a.c_start[static_cast<size_t>(v) * a.node_stride];
        uint32_t cn = a.c_cnt[static_cast<size_t>(v) * a.node_stride];
        if (total + cn > N) cn = N - total;
        for (uint32_t i = lane; i < cn; i += 64) cid[total + i] = static_cast<uint32_t>(a.c_keys[c0 + i]);
        total += cn;
    }
    lds_order();

    // ---- distances of the B and C candidates to the vertex: 64 / G rows per pass, G lanes per row ---------------------------------
    const int sub = static_cast<int>(lane) & (G - 1);
    const uint32_t grp = lane / G;
    for (uint32_t t0 = n_a; t0 < total; t0 += RPP) {
        const uint32_t t = t0 + grp;
        const bool act = t < total;
        const uint32_t tid = act ? cid[t] : 0u;
        const unsigned char* row = data + static_cast<size_t>(tid) * a.row_bytes;
        float fmain = 0.0f, fnrm = 0.0f;
        if constexpr (BC) float_row_distance_bc<DT, METRIC, 0, DECOMP>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        else float_row_distance<DT, METRIC, 0, DECOMP>(row, act, qf, d, sub, a.sq_scale, a.sq_bias, fmain, fnrm);
        const float dist = finish_distance<METRIC, QM_FLOAT>(fmain, fnrm, 0, 0, fx, 1.0f);
        if (act && sub == 0) cdist[t] = dist;
    }
    lds_order();

    // ---- dedupe by id, keeping the first occurrence: sort (id, position) ---------------------------------------------------------
    for (uint32_t i = lane; i < N; i += 64) keys[i] = i < total ? ((static_cast<uint64_t>(cid[i]) << 32) | i) : kNoKey;
    lds_order();
    wave_bitonic_sort(keys, N, lane);
    // chunks from the top down: a position's predecessor (read by lane 0 across the chunk boundary) is still the (id, position) key then
    for (uint32_t c = N / 64; c-- > 0;) {
        const uint32_t i0 = c * 64;
        const uint32_t i = i0 + lane;
        const uint64_t k = keys[i];
        const uint64_t kp = i > 0 ? keys[i - 1] : kNoKey;
        const uint32_t id = static_cast<uint32_t>(k >> 32);
        const bool drop = k == kNoKey || (i > 0 && kp != kNoKey && static_cast<uint32_t>(kp >> 32) == id);
        const uint64_t nk = drop ? kNoKey : pool_key<GREATER>(cdist[static_cast<uint32_t>(k)], id);
        lds_order(); // the 64 reads of this chunk precede its writes
        keys[i] = nk;
        lds_order();
    }
    wave_bitonic_sort(keys, N, lane);

    // ---- best pool_size candidates ------------------------------------------------------------------------------------------------
    for (uint32_t i = lane; i < a.pool_size; i += 64) {
        const uint64_t k = i < N ? keys[i] : kNoKey;
        const bool have = k != kNoKey;
        a.pool_ids[static_cast<size_t>(v) * a.pool_size + i] = have ? static_cast<int32_t>(static_cast<uint32_t>(k)) : -1;
        a.pool_dists[static_cast<size_t>(v) * a.pool_size + i] = have ? pool_key_dist<GREATER>(k) : (GREATER ? -__builtin_inff() : __builtin_inff());
    }
}

template <int DT, int METRIC> static hipError_t launch_pool_m(const PoolArgs& a, bool decomp, hipStream_t stream) {
    const int dpad = (static_cast<int>(a.d) + 15) & ~15;
    const size_t lds = static_cast<size_t>(a.sort_n) * 16 + static_cast<size_t>(dpad) * 4;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    auto launch = [&](auto kern) {
        if (lds > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        hipLaunchKernelGGL(kern, dim3(a.n_nodes), dim3(64), lds, stream, a);
        return hipGetLastError();
    };
    if (decomp) return launch(pool_kernel<DT, METRIC, true>);
    return launch(pool_kernel<DT, METRIC, false>);
}

template <int DT> static hipError_t launch_pool_dt(int metric, const PoolArgs& a, bool decomp, hipStream_t stream) {
    switch (metric) {
        case M_L2: return launch_pool_m<DT, M_L2>(a, decomp, stream);
        case M_IP: return launch_pool_m<DT, M_IP>(a, decomp, stream);
        case M_COS: return launch_pool_m<DT, M_COS>(a, decomp, stream);
    }
    return hipErrorInvalidValue;
}

static hipError_t launch_pool(int data_dtype, int metric, bool decomp, const PoolArgs& a, hipStream_t stream) {
    if (a.n_nodes == 0) return hipSuccess;
    switch (data_dtype) {
        case DT_F32: return launch_pool_dt<DT_F32>(metric, a, false, stream);
        case DT_F16: return launch_pool_dt<DT_F16>(metric, a, false, stream);
        case DT_I8: return launch_pool_dt<DT_I8>(metric, a, decomp, stream);
        case DT_U8: return launch_pool_dt<DT_U8>(metric, a, decomp, stream);
    }
    return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// reverse edges
// ---------------------------------------------------------------------------------------------------------------------------------
// one thread per slot (v, j) of the batch's new rows: key = dst << 32 | src, or kNoKey when the slot is empty or dst already lists src
__global__ __launch_bounds__(256) void edge_keys_kernel(const uint32_t* graph, uint32_t gstride, uint32_t max_degree, uint32_t lo, uint32_t batch, uint64_t* keys) {
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= batch * max_degree) return;
    const uint32_t v = idx / max_degree, j = idx - v * max_degree;
    const uint32_t src = lo + v;
    const uint32_t dst = graph[static_cast<size_t>(src) * gstride + j];
    uint64_t key = kNoKey;
    if (dst != kInvalidId) {
        const uint4* row = reinterpret_cast<const uint4*>(graph + static_cast<size_t>(dst) * gstride);
        bool present = false;
        for (uint32_t t = 0; t < (max_degree + 3u) / 4u; ++t) { // rows are padded to 16 ids with kInvalidId != src
            const uint4 r = row[t];
            present = present || r.x == src || r.y == src || r.z == src || r.w == src;
        }
        if (!present) key = (static_cast<uint64_t>(dst) << 32) | src;
    }
    keys[idx] = key;
}

// keys sorted ascending.  The thread that holds the FIRST key of a destination handles the whole group: nobody else touches that row,
// so the degree it reads is the degree before this batch's reverse edges.
__global__ __launch_bounds__(256) void reverse_scatter_kernel(const uint64_t* keys, uint32_t m, uint32_t* graph, uint32_t gstride, uint32_t max_degree,
                                                               uint32_t over_cap, uint32_t* n_over, uint32_t* over /* [m][3]: dst, first key, count */) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const uint64_t key = keys[i];
    if (key == kNoKey) return;
    const uint32_t dst = static_cast<uint32_t>(key >> 32);
    if (i > 0 && static_cast<uint32_t>(keys[i - 1] >> 32) == dst) return;
    uint32_t cnt = 1;
    while (i + cnt < m && static_cast<uint32_t>(keys[i + cnt] >> 32) == dst && keys[i + cnt] != kNoKey) ++cnt;
    uint32_t* row = graph + static_cast<size_t>(dst) * gstride;
    uint32_t lo = 0, hi = max_degree; // ids are packed from slot 0: degree = first empty slot
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (row[mid] != kInvalidId) lo = mid + 1;
        else hi = mid;
    }
    const uint32_t deg = lo;
    const uint32_t fit = cnt < max_degree - deg ? cnt : max_degree - deg;
    for (uint32_t r = 0; r < fit; ++r) row[deg + r] = static_cast<uint32_t>(keys[i + r]);
    if (cnt > fit) {
        const uint32_t o = atomicAdd(n_over, 1u);
        over[3 * o] = dst;
        over[3 * o + 1] = i + fit;
        over[3 * o + 2] = (cnt - fit) < over_cap ? (cnt - fit) : over_cap;
    }
}

// rows[over[3 i]] = tmp[i]
__global__ __launch_bounds__(64) void scatter_rows_kernel(const int32_t* tmp, const uint32_t* over, uint32_t* graph, uint32_t gstride) {
    const uint32_t dst = over[3 * blockIdx.x];
    for (uint32_t j = threadIdx.x; j < gstride; j += 64) graph[static_cast<size_t>(dst) * gstride + j] = static_cast<uint32_t>(tmp[static_cast<size_t>(blockIdx.x) * gstride + j]);
}

// the overflow groups come out of the atomic counter in arbitrary order: sort the triples by destination so that the pool / prune launches
// and therefore every intermediate buffer are reproducible (the graph itself does not depend on the order: groups are independent)
__global__ __launch_bounds__(256) void over_keys_kernel(const uint32_t* over, uint32_t n, uint64_t* keys) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) keys[i] = (static_cast<uint64_t>(over[3 * i]) << 32) | i;
}
__global__ __launch_bounds__(256) void over_gather_kernel(const uint32_t* over, const uint64_t* keys, uint32_t n, uint32_t* sorted) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = static_cast<uint32_t>(keys[i]);
    sorted[3 * i] = over[3 * s];
    sorted[3 * i + 1] = over[3 * s + 1];
    sorted[3 * i + 2] = over[3 * s + 2];
}

// use_full_search_history = false: the search returned its buffer (ids + distances, rows of `stride`, padded with 0xFFFFFFFF when the buffer held
// fewer): the number of real entries per vertex
__global__ __launch_bounds__(256) void count_valid_kernel(const uint32_t* ids, uint32_t stride, uint32_t n, uint32_t* len) {
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if (v >= n) return;
    uint32_t c = 0;
    while (c < stride && ids[static_cast<size_t>(v) * stride + c] != 0xFFFFFFFFu) ++c;
    len[v] = c;
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* out, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = first + i;
}
__global__ __launch_bounds__(256) void over_destinations_kernel(const uint32_t* over, uint32_t n, uint32_t* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) out[i] = over[3 * i];
}

// ---------------------------------------------------------------------------------------------------------------------------------
// float rows of a batch (queries of the scalar-quantized build: the decompressed vectors, extensions/vamana/scalar.h:46-97)
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decompress_rows_kernel(const void* data, size_t row_bytes, int dtype, uint32_t d, float scale, float bias, uint32_t lo, uint32_t batch, float* out) {
    const size_t idx = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x;
    if (idx >= static_cast<size_t>(batch) * d) return;
    const uint32_t v = static_cast<uint32_t>(idx / d), c = static_cast<uint32_t>(idx - static_cast<size_t>(v) * d);
    const unsigned char* row = static_cast<const unsigned char*>(data) + static_cast<size_t>(lo + v) * row_bytes;
    const float x = dtype == DT_I8 ? static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(row)[c])) : static_cast<float>(static_cast<unsigned>(row[c]));
    out[idx] = scale * x + bias;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// medoid (utils::find_medioid, core/medioid.h:292-353): the vector closest (L2) to the component-wise mean; the mean is accumulated
// in double in a fixed order (per block of rows, then over the blocks), so the entry point is reproducible
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ inline float load_component(const unsigned char* row, int dtype, uint32_t c, int decomp, float scale, float bias) {
    float x;
    if (dtype == DT_F32) x = reinterpret_cast<const float*>(row)[c];
    else if (dtype == DT_F16) x = static_cast<float>(reinterpret_cast<const _Float16*>(row)[c]);
    else if (dtype == DT_I8) x = static_cast<float>(static_cast<int>(reinterpret_cast<const int8_t*>(row)[c]));
    else x = static_cast<float>(static_cast<unsigned>(row[c]));
    return decomp ? scale * x + bias : x;
}

__global__ __launch_bounds__(256) void column_sums_kernel(const void* data, size_t row_bytes, int dtype, uint32_t n, uint32_t d, int decomp, float scale, float bias,
                                                           uint32_t rows_per_block, double* partial /* [gridDim.x][d] */) {
    const uint32_t r0 = blockIdx.x * rows_per_block;
    const uint32_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    for (uint32_t c = threadIdx.x; c < d; c += 256) {
        double acc = 0.0;
        for (uint32_t r = r0; r < r1; ++r) acc += static_cast<double>(load_component(static_cast<const unsigned char*>(data) + static_cast<size_t>(r) * row_bytes, dtype, c, decomp, scale, bias));
        partial[static_cast<size_t>(blockIdx.x) * d + c] = acc;
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const double* partial, uint32_t blocks, uint32_t n, uint32_t d, float* mean) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= d) return;
    double acc = 0.0;
    for (uint32_t b = 0; b < blocks; ++b) acc += partial[static_cast<size_t>(b) * d + c];
    mean[c] = static_cast<float>(acc / static_cast<double>(n));
}

// one wave per row: squared distance to the mean (float32, lanes stride the components, xor-tree), key = (order bits << 32 | row)
__global__ __launch_bounds__(64) void mean_distance_keys_kernel(const void* data, size_t row_bytes, int dtype, uint32_t n, uint32_t d, int decomp, float scale, float bias,
                                                                 const float* mean, uint64_t* keys) {
    const uint32_t r = blockIdx.x;
    const unsigned char* row = static_cast<const unsigned char*>(data) + static_cast<size_t>(r) * row_bytes;
    float acc = 0.0f;
    for (uint32_t c = threadIdx.x; c < d; c += 64) {
        const float t = load_component(row, dtype, c, decomp, scale, bias) - mean[c];
        acc = __builtin_fmaf(t, t, acc);
    }
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if (threadIdx.x == 0) keys[r] = (static_cast<uint64_t>(order_bits(acc)) << 32) | r;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
struct BuildTarget {          // what svsx_api.hip knows about the index (the svsx_index struct is private to it)
    const svsx_index* index;  // device view over the same data / graph / entry point: searched with svsx_search_device, pruned with svsx_prune_device
    const void* d_data;
    size_t row_bytes;
    int data_dtype, metric;
    uint32_t n, d;
    int sq_enabled;
    float sq_scale, sq_bias;
    uint32_t* d_graph;        // n x gstride, every slot kInvalidId on entry
    uint32_t gstride;
};

struct BuildParams {          // index/vamana/build_params.h:28-75
    float alpha;
    uint32_t graph_max_degree, window_size, max_candidate_pool_size, prune_to;
    int use_full_search_history; // 0: the candidates of a vertex are the search buffer's contents (vamana_build.h:407-421), not the expanded nodes
};

namespace {
struct Dev {
    void* p = nullptr;
    ~Dev() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};
uint32_t pow2_at_least(uint32_t x) {
    uint32_t p = 64;
    while (p < x) p <<= 1;
    return p;
}
} // namespace

// radix sort of n 64-bit keys; the temporary storage grows on demand
static hipError_t sort_keys(void*& temp, size_t& temp_cap, const uint64_t* in, uint64_t* out, uint32_t n, hipStream_t stream) {
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, need, in, out, n, 0, 64, stream);
    if (e != hipSuccess) return e;
    if (need > temp_cap) {
        if (temp != nullptr) {
            e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            (void)hipFree(temp);
            temp = nullptr;
        }
        e = hipMalloc(&temp, need);
        if (e != hipSuccess) return e;
        temp_cap = need;
    }
    return rocprim::radix_sort_keys(temp, need, in, out, n, 0, 64, stream);
}

#define BUILD_TRY(expr)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#expr) + " failed: " + hipGetErrorString(e_);                  \
            return SVSX_RUNTIME_ERROR;                                                       \
        }                                                                                    \
    } while (0)

int find_medoid_device(const BuildTarget& t, uint32_t* medoid, hipStream_t stream, std::string& err) {
    const uint32_t blocks = 1024;
    const uint32_t rows_per_block = (t.n + blocks - 1) / blocks;
    Dev partial, mean, keys, keys_out, temp;
    BUILD_TRY(partial.alloc(sizeof(double) * blocks * t.d));
    BUILD_TRY(mean.alloc(sizeof(float) * t.d));
    BUILD_TRY(keys.alloc(sizeof(uint64_t) * t.n));
    BUILD_TRY(keys_out.alloc(sizeof(uint64_t)));
    hipLaunchKernelGGL(column_sums_kernel, dim3(blocks), dim3(256), 0, stream, t.d_data, t.row_bytes, t.data_dtype, t.n, t.d, t.sq_enabled, t.sq_scale, t.sq_bias, rows_per_block, partial.as<double>());
    hipLaunchKernelGGL(mean_kernel, dim3((t.d + 255) / 256), dim3(256), 0, stream, partial.as<double>(), blocks, t.n, t.d, mean.as<float>());
    hipLaunchKernelGGL(mean_distance_keys_kernel, dim3(t.n), dim3(64), 0, stream, t.d_data, t.row_bytes, t.data_dtype, t.n, t.d, t.sq_enabled, t.sq_scale, t.sq_bias, mean.as<float>(), keys.as<uint64_t>());
    BUILD_TRY(hipGetLastError());
    size_t temp_bytes = 0; // min over (distance, row): ties go to the lower row, like std::min_element over the rows in order
    BUILD_TRY(rocprim::reduce(nullptr, temp_bytes, keys.as<uint64_t>(), keys_out.as<uint64_t>(), kNoKey, t.n, rocprim::minimum<uint64_t>(), stream));
    BUILD_TRY(temp.alloc(temp_bytes));
    BUILD_TRY(rocprim::reduce(temp.p, temp_bytes, keys.as<uint64_t>(), keys_out.as<uint64_t>(), kNoKey, t.n, rocprim::minimum<uint64_t>(), stream));
    uint64_t best = 0;
    BUILD_TRY(hipMemcpyAsync(&best, keys_out.p, sizeof(best), hipMemcpyDeviceToHost, stream));
    BUILD_TRY(hipStreamSynchronize(stream));
    *medoid = static_cast<uint32_t>(best);
    return SVSX_SUCCESS;
}

// VamanaIndex constructor with build parameters (index.h:398-439) for data and graph already in HBM.
int build_vamana(const BuildTarget& t, const BuildParams& p, hipStream_t stream, std::string& err) {
    const uint32_t R = p.graph_max_degree, L = p.window_size;
    if (R == 0 || R > t.gstride || L == 0) { err = "graph_max_degree must be in 1..graph stride and window_size positive"; return SVSX_INVALID_ARGUMENT; }
    const uint32_t prune_to = p.prune_to ? p.prune_to : (R > 4 ? R - 4 : 1);
    const uint32_t over_cap = 64;   // overflow candidates considered per destination and batch (vamana_build.h re-prunes row + all; bounded here)
    uint32_t nb = std::max<uint32_t>(40, (t.n + 4095) / 4096);
    nb = std::min(nb, t.n);
    const uint32_t bs = (t.n + nb - 1) / nb;
    const uint32_t H = 2 * L + 32;   // expanded nodes kept per vertex (a search of window L expands little more than L nodes)
    const uint32_t pool_size = std::min(p.max_candidate_pool_size, H + R);
    const uint32_t m = bs * R;       // edge slots per batch
    const bool decomp = t.sq_enabled != 0;
    const int q_dtype = decomp ? SVSX_F32 : t.data_dtype;

    Dev hist_ids, hist_d, hist_len, out_ids, node_ids, pool_ids, pool_d, keys_a, keys_b, okeys_a, okeys_b, sort_temp, n_over, over, over_sorted, tmp_rows, qfloat;
    BUILD_TRY(hist_ids.alloc(sizeof(uint32_t) * bs * H));
    BUILD_TRY(hist_d.alloc(sizeof(float) * bs * H));
    BUILD_TRY(hist_len.alloc(sizeof(uint32_t) * bs));
    BUILD_TRY(out_ids.alloc(sizeof(uint32_t) * bs));
    BUILD_TRY(node_ids.alloc(sizeof(uint32_t) * std::max<uint32_t>(bs, 4096)));
    const size_t pool_rows = std::max<size_t>(bs, 4096);
    const uint32_t over_pool = R + over_cap;
    BUILD_TRY(pool_ids.alloc(sizeof(int32_t) * pool_rows * std::max(pool_size, over_pool)));
    BUILD_TRY(pool_d.alloc(sizeof(float) * pool_rows * std::max(pool_size, over_pool)));
    BUILD_TRY(keys_a.alloc(sizeof(uint64_t) * m));
    BUILD_TRY(keys_b.alloc(sizeof(uint64_t) * m));
    BUILD_TRY(okeys_a.alloc(sizeof(uint64_t) * m));
    BUILD_TRY(okeys_b.alloc(sizeof(uint64_t) * m));
    BUILD_TRY(n_over.alloc(sizeof(uint32_t)));
    BUILD_TRY(over.alloc(sizeof(uint32_t) * 3 * m));
    BUILD_TRY(over_sorted.alloc(sizeof(uint32_t) * 3 * m));
    BUILD_TRY(tmp_rows.alloc(sizeof(int32_t) * 4096 * t.gstride));
    if (decomp) BUILD_TRY(qfloat.alloc(sizeof(float) * bs * t.d));
    size_t sort_bytes = 0; // sort_temp.p is owned by `sort_temp` (freed on every exit path), resized by sort_keys

    svsx_search_params sp{};
    sp.search_window_size = L;
    sp.search_buffer_capacity = L;
    sp.visited_set = 0;
    svsx_search_aux aux{};
    aux.history_ids = hist_ids.as<uint32_t>();
    aux.history_dists = hist_d.as<float>();
    aux.history_len = hist_len.as<uint32_t>();
    aux.history_cap = H;

    const uint32_t sort_n_fwd = pow2_at_least(H + R);
    const uint32_t sort_n_rev = pow2_at_least(R + over_cap);

    for (int pass = 0; pass < 2; ++pass) {
        const float pass_alpha = pass == 0 ? 1.0f : p.alpha;
        for (uint32_t b = 0; b < nb; ++b) {
            const uint32_t lo = std::min(t.n, bs * b), hi = std::min(t.n, bs * (b + 1));
            const uint32_t B = hi - lo;
            if (B == 0) continue;
            // ---- 1. generate_neighbors: search -------------------------------------------------------------------------------------
            const void* queries;
            if (decomp) {
                hipLaunchKernelGGL(decompress_rows_kernel, dim3((static_cast<size_t>(B) * t.d + 255) / 256), dim3(256), 0, stream, t.d_data, t.row_bytes, t.data_dtype, t.d, t.sq_scale, t.sq_bias, lo, B, qfloat.as<float>());
                queries = qfloat.p;
                aux.query_row_stride_bytes = 0;
            } else {
                queries = static_cast<const unsigned char*>(t.d_data) + static_cast<size_t>(lo) * t.row_bytes;
                aux.query_row_stride_bytes = t.row_bytes;
            }
            uint32_t a_cap = H;
            if (p.use_full_search_history) {
                if (int rc = svsx_search_device(t.index, queries, q_dtype, B, 1, &sp, nullptr, out_ids.as<uint32_t>(), nullptr, &aux, stream)) {
                    err = std::string("search of a build batch failed: ") + svsx_last_error();
                    return rc;
                }
            } else { // the buffer itself: the best L entries the search holds when it ends, in rows of L (H >= L: the history arrays hold them)
                svsx_search_aux plain{};
                plain.query_row_stride_bytes = aux.query_row_stride_bytes;
                if (int rc = svsx_search_device(t.index, queries, q_dtype, B, L, &sp, nullptr, hist_ids.as<uint32_t>(), hist_d.as<float>(), &plain, stream)) {
                    err = std::string("search of a build batch failed: ") + svsx_last_error();
                    return rc;
                }
                hipLaunchKernelGGL(count_valid_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, hist_ids.as<uint32_t>(), L, B, hist_len.as<uint32_t>());
                a_cap = L;
            }
            // ---- 2. candidate pools ------------------------------------------------------------------------------------------------
            BUILD_TRY(hipMemsetAsync(n_over.p, 0, sizeof(uint32_t), stream));
            hipLaunchKernelGGL(iota_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, node_ids.as<uint32_t>(), lo, B);
            PoolArgs pa{};
            pa.data = t.d_data; pa.row_bytes = t.row_bytes; pa.d = t.d; pa.sq_scale = t.sq_scale; pa.sq_bias = t.sq_bias;
            pa.node_ids = node_ids.as<uint32_t>(); pa.node_stride = 1; pa.n_nodes = B;
            pa.a_ids = hist_ids.as<uint32_t>(); pa.a_dists = hist_d.as<float>(); pa.a_len = hist_len.as<uint32_t>(); pa.a_cap = a_cap;
            pa.graph = t.d_graph; pa.gstride = t.gstride; pa.max_degree = R;
            pa.sort_n = sort_n_fwd; pa.pool_size = pool_size;
            pa.pool_ids = pool_ids.as<int32_t>(); pa.pool_dists = pool_d.as<float>();
            BUILD_TRY(launch_pool(t.data_dtype, t.metric, decomp, pa, stream));
            // ---- 3. prune into the graph rows of the batch ("we purposely pass params_.alpha", vamana_build.h:271) -----------------------
            if (int rc = svsx_prune_device(t.index, reinterpret_cast<const int32_t*>(node_ids.p), pool_ids.as<int32_t>(), pool_d.as<float>(), B, pool_size, R, p.alpha,
                                           reinterpret_cast<int32_t*>(t.d_graph + static_cast<size_t>(lo) * t.gstride), t.gstride, stream)) {
                err = std::string("prune of a build batch failed: ") + svsx_last_error();
                return rc;
            }
            // ---- 4.-6. add_reverse_edges ----------------------------------------------------------------------------------------------
            const uint32_t mb = B * R;
            hipLaunchKernelGGL(edge_keys_kernel, dim3((mb + 255) / 256), dim3(256), 0, stream, t.d_graph, t.gstride, R, lo, B, keys_a.as<uint64_t>());
            BUILD_TRY(sort_keys(sort_temp.p, sort_bytes, keys_a.as<uint64_t>(), keys_b.as<uint64_t>(), mb, stream));
            hipLaunchKernelGGL(reverse_scatter_kernel, dim3((mb + 255) / 256), dim3(256), 0, stream, keys_b.as<uint64_t>(), mb, t.d_graph, t.gstride, R, over_cap, n_over.as<uint32_t>(), over.as<uint32_t>());
            BUILD_TRY(hipGetLastError());
            uint32_t h_over = 0;
            BUILD_TRY(hipMemcpyAsync(&h_over, n_over.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            BUILD_TRY(hipStreamSynchronize(stream));
            if (h_over == 0) continue;
            // ---- 7. destinations whose list overflowed: row + overflow candidates -> prune_to, pass alpha -----------------------------
            // reproducible order of the groups (they left the atomic counter in arbitrary order)
            hipLaunchKernelGGL(over_keys_kernel, dim3((h_over + 255) / 256), dim3(256), 0, stream, over.as<uint32_t>(), h_over, okeys_a.as<uint64_t>());
            BUILD_TRY(sort_keys(sort_temp.p, sort_bytes, okeys_a.as<uint64_t>(), okeys_b.as<uint64_t>(), h_over, stream));
            hipLaunchKernelGGL(over_gather_kernel, dim3((h_over + 255) / 256), dim3(256), 0, stream, over.as<uint32_t>(), okeys_b.as<uint64_t>(), h_over, over_sorted.as<uint32_t>());
            for (uint32_t c0 = 0; c0 < h_over; c0 += 4096) {
                const uint32_t cn = std::min<uint32_t>(4096, h_over - c0);
                const uint32_t* oc = over_sorted.as<uint32_t>() + 3 * static_cast<size_t>(c0);
                PoolArgs po{};
                po.data = t.d_data; po.row_bytes = t.row_bytes; po.d = t.d; po.sq_scale = t.sq_scale; po.sq_bias = t.sq_bias;
                po.node_ids = oc; po.node_stride = 3; po.n_nodes = cn;
                po.graph = t.d_graph; po.gstride = t.gstride; po.max_degree = R;
                po.c_keys = keys_b.as<uint64_t>(); po.c_start = oc + 1; po.c_cnt = oc + 2;
                po.sort_n = sort_n_rev; po.pool_size = over_pool;
                po.pool_ids = pool_ids.as<int32_t>(); po.pool_dists = pool_d.as<float>();
                BUILD_TRY(launch_pool(t.data_dtype, t.metric, decomp, po, stream));
                hipLaunchKernelGGL(over_destinations_kernel, dim3((cn + 255) / 256), dim3(256), 0, stream, oc, cn, node_ids.as<uint32_t>()); // dense vertex list for the prune
                if (int rc = svsx_prune_device(t.index, reinterpret_cast<const int32_t*>(node_ids.p), pool_ids.as<int32_t>(), pool_d.as<float>(), cn, over_pool, prune_to, pass_alpha,
                                               tmp_rows.as<int32_t>(), t.gstride, stream)) {
                    err = std::string("re-prune of overflowing adjacency lists failed: ") + svsx_last_error();
                    return rc;
                }
                hipLaunchKernelGGL(scatter_rows_kernel, dim3(cn), dim3(64), 0, stream, tmp_rows.as<int32_t>(), oc, t.d_graph, t.gstride);
                BUILD_TRY(hipGetLastError());
            }
        }
    }
    BUILD_TRY(hipStreamSynchronize(stream));
    return SVSX_SUCCESS;
}

} // namespace svsx
