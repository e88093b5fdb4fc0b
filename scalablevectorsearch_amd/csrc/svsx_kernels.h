// svsx_kernels.h -- host-visible kernel argument blocks and launchers (internal to libsvsx.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace svsx {

struct SearchArgs {
    // index (device layout)
    const void* data;        // n rows, row_bytes each (zero padded to 16 elements)
    size_t row_bytes;
    const uint32_t* graph;   // n rows of gstride ids, packed from slot 0, unused = 0xFFFFFFFF
    uint32_t gstride;
    uint32_t n;
    uint32_t d;
    const uint32_t* entry;
    uint32_t n_entry;
    float sq_scale, sq_bias;
    const float* sq_row_norm;  // scalar-quantized rows under cosine: sqrtf(||decompressed row||^2) per row, computed once at index creation
    uint64_t id_offset;
    // queries
    const void* queries;
    size_t q_row_bytes;
    int q_dtype;
    uint32_t nq;
    // parameters
    uint32_t window, capacity, k;
    uint32_t filter_entries;  // 0 = no visited filter; else power of two (exact-tag direct-mapped filter in LDS)
    uint32_t filter_tag8;     // 1: 8-bit tags (host guarantees (n >> log2 entries) < 255), 0: 16-bit tags
    // outputs
    uint64_t* ids64;
    uint32_t* ids32;
    float* dists;
    uint32_t* stats;
    uint32_t* history_ids;
    float* history_dists;
    uint32_t* history_len;
    uint32_t history_cap;
    // optional 128-B bound-code rows (f32 L2, d = 96 or 128): rigorous lower-bound rejection before a row is fetched
    const unsigned char* codes;
    float code_scale, code_bias, code_eq_abs; // code_eq_abs: absolute rounding term of the query's coding residual (svsx_api.hip)
    // graph-inline records (svsx_codes.hip; CODES == 2): node rows [ids | plane 0 | plane 1] per chunk of <= 64 slots
    const unsigned char* fat;
    size_t fat_stride;        // bytes per node
    uint32_t fat_rb;          // record bytes: 32 / 64 / 128
    uint32_t fat_degshift;    // 0, or 26: the id words of the blocks carry ceil(degree(neighbour) / 8) in bits [26, 31) (indexes of <= 2^26 nodes)
    const float* pca_t;       // [128][pca_pd] leading directions, row-major float32 (rows >= dc are zero), pca_pd = d rounded up to 16
    const float* pca_mean;    // [pca_pd] what the records were centred with (cosine: the mean of the unit vectors)
    const float* pca_bias;    // [128] per-coordinate code bias (L2)
    uint32_t pca_pd;
    float pca_step;           // s
    float pca_tail_unit;      // tx = half(tx16) * tail_unit
    float pca_delta;          // >= max_i || float32(Q_i) - Q_i ||: rounding of the directions the query is projected with
    float pca_eq_abs;         // absolute slack of the query coding error (rounding of b + s c)
    float pca_ref_extra;      // inner product: ||mean|| (+ 2 |bias| sqrt(d) for scalar-quantized rows), enters the slack for the reference's own rounding
    float pca_mean_norm;      // similarity metrics: >= ||mean|| (the records are centred; the kernel adds q . mean to its bound); 0 under L2
    uint32_t pca_mean_dir;    // similarity metrics: 1 = coordinate 0 of the basis is the unit mean direction and the QUERY's coordinate 0 stays in float32 (not coded)
    // cancellation (index/vamana/index.h:541-545, greedy_search.h:155): device word polled once per expansion; non-zero = stop
    const int* cancel;
    // general variant of the kernel (std::sort-exact entry initialisation for > 16 entry points; BatchIterator state)
    uint32_t general;
    // BatchIterator (index/vamana/iterator.h): per-query state in HBM.  meta = 8 u32 per query:
    //   [0] size  [1] best_unvisited  [2] window  [3] capacity  [4] yielded  [5] iteration  [6] flags (1 restart, 2 exhausted)
    uint2* it_entries;       // [nq][it_estride] buffer entries {dist bits, id | visited << 31}
    uint32_t it_estride;
    uint32_t it_spill;       // 1: the buffers no longer fit one wave's LDS; the kernel works on it_entries in place
    uint32_t* it_meta;       // nullptr = plain search
    uint32_t* it_yielded;    // [nq][it_ystride] ids already handed out, in yield order
    uint32_t it_ystride;
    const uint32_t* it_batch_sizes; // optional per-query batch size (0xFFFFFFFF = leave this query untouched)
    uint32_t it_batch;       // uniform batch size and row stride of the outputs
    uint64_t* it_out_ids;    // [nq][it_batch]
    float* it_out_dists;     // [nq][it_batch]
    uint32_t* it_counts;     // [nq]
};

hipError_t launch_search(int data_dtype, int metric, int qmode, const SearchArgs& a, hipStream_t stream);

struct DistanceArgs {
    const void* a;      // na rows, dense, a_dtype
    const void* b;      // nb rows, b_row_bytes each, zero padded to 16 elements
    size_t b_row_bytes;
    int a_dtype;
    uint32_t na, nb, d;
    float* out;         // na x nb
};
hipError_t launch_distance(int b_dtype, int metric, int qmode, const DistanceArgs& a, hipStream_t stream);
// sqrtf of the reference-order sum of squares of every DECOMPRESSED scalar-quantized row (what CosineSimilarityCompressed divides by)
hipError_t launch_sq_row_norms(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t d, float scale, float bias, float* d_out, hipStream_t stream);

struct MergeArgs {
    const float* dists;     // [S][nq][k]
    const uint64_t* ids;    // [S][nq][k]
    float* out_dists;       // [nq][k]
    uint64_t* out_ids;      // [nq][k]
    uint32_t n_shards, nq, k;
    size_t dist_shard_stride, id_shard_stride; // elements between consecutive shards; 0 = dense (nq * k)
    const uint32_t* gate64;  // optional: one word per 64 queries; a query is merged only if its word is non-zero
};
hipError_t launch_merge(int metric, const MergeArgs& a, hipStream_t stream);

// builder: heuristic_prune_neighbors for a batch of vertices (index/vamana/prune.h:100-246)
struct PruneArgs {
    const void* data;
    size_t row_bytes;
    uint32_t d;
    float sq_scale, sq_bias;
    const int32_t* node_ids;    // n_nodes
    const int32_t* pool_ids;    // n_nodes x pool_size, sorted best-first, -1 padded at the end
    const float* pool_dists;    // n_nodes x pool_size, distance(vertex, candidate)
    uint32_t n_nodes, pool_size, max_out;
    float alpha;
    int32_t* out;               // n_nodes x out_stride, packed from slot 0, -1 padded
    uint32_t out_stride;
};
hipError_t launch_prune(int data_dtype, int metric, bool decomp, const PruneArgs& a, hipStream_t stream);

// flat (exhaustive) search: f16 shadow conversion, MFMA shortlist, exact re-rank (svsx_flat.hip)
hipError_t launch_to_half(int dtype, const void* src, size_t src_row_bytes, uint32_t n, uint32_t d, uint32_t dpad, int decomp, float scale,
                          float bias, void* dst, float* norm2, hipStream_t stream);
struct FlatArgs {
    const void* xh;         // n x dpad f16 (zero padded rows)
    const float* xnorm2;    // n
    uint32_t n, dpad;
    const void* qh;         // nq x dpad f16
    uint32_t nq, kk, n_slices;
    uint32_t* gtau;         // nq order-encoded thresholds shared by the slices of a query block (init 0xFF800000 = +inf), or NULL
    float* out_keys;        // [n_slices][nq][kk]
    uint64_t* out_ids;      // [n_slices][nq][kk]
    uint32_t tile_step;     // visit every tile_step-th 32-row tile of the slice (1 = all; > 1 = strided sample)
    const uint32_t* block_gate; // optional, one word per 64-query block: the block only runs if its word is non-zero
};
hipError_t launch_flat_mfma(int metric, const FlatArgs& a, hipStream_t stream);
// threshold pass: every (query, row) whose f16 key is <= tau[query] is appended to the (query, slice) candidate list
struct FlatThreshArgs {
    const void* xh;
    const float* xnorm2;
    uint32_t n, dpad;
    const void* qh;
    uint32_t nq, n_slices;
    uint32_t tile_step;     // only 32-row tiles whose global index is a multiple of tile_step (1 = all)
    const float* tau;       // tau[q * tau_stride]
    uint32_t tau_stride;
    uint32_t* cand;         // [nq][n_slices][cap] row ids, unordered
    float* cand_keys;       // optional, same shape: the f16-pass key of each appended row
    uint32_t* cnt;          // [nq][n_slices] rows below the threshold (may exceed cap: overflow)
    uint32_t cap;           // per (query, slice)
};
hipError_t launch_flat_thresh(int metric, const FlatThreshArgs& a, hipStream_t stream);   // dpad <= 128 only
// tau_out[q] = kk-th smallest key of query q's lists; tau_fallback[q * stride] if a list overflowed or fewer than kk rows
hipError_t launch_flat_select(const float* cand_keys, const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t kk,
                              const float* tau_fallback, uint32_t fallback_stride, float* tau_out, hipStream_t stream);
hipError_t launch_flat_gate(const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t* block_gate, hipStream_t stream);
struct RerankArgs {
    const void* data;
    size_t row_bytes;
    uint32_t d;
    float sq_scale, sq_bias;
    uint64_t id_offset;
    const void* queries;
    size_t q_row_bytes;
    int q_dtype;
    uint32_t nq, kk, k;
    const uint64_t* cand_ids;   // [nq][kk], ~0 = empty
    // candidate lists of the threshold pass (rerank2): cand32 [nq][n_lists][cap] with cnt[q][list] entries; a query with
    // an overflowed list (cnt > cap) falls back to cand_ids
    const uint32_t* cand32;
    const uint32_t* cnt;
    uint32_t cap, n_lists;
    uint64_t* ids64;            // [nq][k]
    float* dists;               // [nq][k]
};
hipError_t launch_rerank(int data_dtype, int metric, int qmode, const RerankArgs& a, hipStream_t stream);
hipError_t launch_rerank2(int data_dtype, int metric, int qmode, const RerankArgs& a, hipStream_t stream);

// graph-inline bound codes (svsx_codes.hip)
struct PcaArgs {
    const void* data;
    size_t row_bytes;
    int dtype;
    uint32_t n, d, dc, rb;  // rb = record bytes (32 / 64 / 128); dc = coded coordinates = rb - 4 (unsigned codes) / rb - 6 (signed)
    int decomp;             // rows are scalar-quantized codes: decompress scale * c + bias in float32 first
    float sq_scale, sq_bias;
    int normalize;          // cosine: code the unit vector
    int signed_codes;       // inner product / cosine record layout
    const double* q64;      // [dc][d] orthonormal directions (f64)
    const float* mean;      // [d] the centre (cosine: of the unit vectors)
    const float* bias;      // [dc] per-coordinate code bias (unsigned codes; encode)
    float step, tail_unit;  //                 (encode)
    uint32_t* stat_min;     // [dc] order keys (stats; init 0xFFFFFFFF)
    uint32_t* stat_max;     // [dc] order keys (stats; init 0)
    uint32_t* stat_tail;    // max remainder norm key (init 0)
    uint32_t* stat_bad;     // vectors with a non-finite component (init 0)
    uint32_t* records;      // [n][rb / 4]     (encode)
};
hipError_t launch_pca_stats(const PcaArgs& a, hipStream_t stream);
hipError_t launch_pca_encode(const PcaArgs& a, hipStream_t stream);
// d_class_scratch: n bytes, or nullptr = plain id words (degshift ignored)
hipError_t launch_fat_build(const uint32_t* d_graph, uint32_t gstride, uint32_t n, const void* d_records, uint32_t record_bytes, void* d_fat, size_t fat_stride,
                            uint8_t* d_class_scratch, uint32_t degshift, hipStream_t stream);
float pca_key_to_float(uint32_t key);
hipError_t launch_set_word(int* d_word, int value, hipStream_t stream); // one agent-scope store (cancellation)
hipError_t launch_add_word(uint32_t* d_word, uint32_t value, hipStream_t stream); // *d_word += value (counters)
// name of the search_kernel instantiation launch_search picked last on this thread (tests assert which one ran)
const char* last_search_instantiation();

// layout conversion kernels (index creation)
// bound codes (svsx_index_build_bound_codes): d_mm3 = {min key, max key, non-finite count}
hipError_t launch_codes_minmax(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t dc, uint32_t* d_mm3, hipStream_t stream);
float codes_key_to_float(uint32_t key);
hipError_t launch_codes_build(int dtype, const void* d_data, size_t row_bytes, uint32_t n, uint32_t dc, float scale, float bias, void* d_codes, hipStream_t stream);
hipError_t launch_pack_graph(const uint32_t* d_svs_graph, uint32_t svs_stride, uint32_t n, uint32_t* d_out, uint32_t out_stride, uint32_t* d_bad2,
                             hipStream_t stream);

} // namespace svsx
