/*
 * svsx.h -- C ABI of the MI355X-native batched Vamana greedy-search path.
 *
 * This is the drop-in boundary: a plain `extern "C"` shared library (libsvsx.so) with host- and
 * device-pointer entry points, no C++/torch types in any signature.  Each entry point names the reference
 * interface (intel/ScalableVectorSearch 0.4.0, paths relative to the reference root) it stands behind;
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Semantics are those of the reference CPU path: results (neighbour ids AND distance bits) are identical to
 * `svs::index::vamana::VamanaIndex::search` on the same graph/data/queries/parameters, for the type pairs the
 * reference has AVX-512 kernels for.
 *
 * Threading: an index is immutable after creation; `svsx_search*` may be called concurrently from several
 * host threads on the same index (index/vamana/index.h:541-545 contract).  `svsx_last_error` is thread-local.
 */
#ifndef SVSX_H
#define SVSX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bindings/cpp/include/svs/runtime/api_defs.h:110-117  (enum class ErrorCode) */
typedef enum {
    SVSX_SUCCESS = 0,
    SVSX_UNKNOWN_ERROR = 1,
    SVSX_INVALID_ARGUMENT = 2,
    SVSX_NOT_IMPLEMENTED = 3,
    SVSX_NOT_INITIALIZED = 4,
    SVSX_RUNTIME_ERROR = 5,
    /* not in the reference's enum and NOT a failure: svsx_index_build_bound_codes completed, but with a smaller scheme than asked for because the
     * one asked for does not fit the free HBM (searches stay result-identical; svsx_last_error() says what was built,
     * svsx_index_memory()[8] / [3] which scheme and record size are in effect) */
    SVSX_FELL_BACK = 64
} svsx_status;

/* svs::DistanceType (core/distance.h) / runtime MetricType (api_defs.h:95).  The distance functors these
 * select are DistanceL2 (euclidean.h:142-166), DistanceIP (inner_product.h:103-127),
 * DistanceCosineSimilarity (cosine.h:102-144). */
typedef enum { SVSX_L2 = 0, SVSX_IP = 1, SVSX_COSINE = 2 } svsx_metric;

/* svs::DataType subset with AVX-512 kernels (lib/datatype.h): float32, float16, int8, uint8 */
typedef enum { SVSX_F32 = 0, SVSX_F16 = 1, SVSX_I8 = 2, SVSX_U8 = 3 } svsx_dtype;

typedef struct svsx_index svsx_index;

/* What VamanaIndex is assembled from (index/vamana/index.h:363-377, auto_assemble :1017-1046):
 *   data   SimpleData<T>        row-major n x d                (core/data/simple.h:259-618)
 *   graph  SimpleGraph<uint32>  n rows of graph_row_stride u32, row[0]=degree, row[1..deg]=neighbour ids
 *                               (core/graph/graph.h:32-48)
 *   entry points                (index.h:304,446-452)
 *   SQDataset<T> scale/bias     (quantization/scalar/scalar.h:363-534) when sq_enabled
 * Host buffers are copied (and re-laid-out) into HBM; they may be freed after the call returns. */
typedef struct {
    int metric;                     /* svsx_metric */
    int data_dtype;                 /* svsx_dtype  */
    size_t n, d;
    const void* data;
    size_t data_row_stride_bytes;   /* 0 = dense (d * sizeof(T)) */
    const uint32_t* graph;
    size_t graph_row_stride_u32;    /* = max_degree + 1 in SVS files */
    uint32_t max_degree;
    const uint32_t* entry_points;
    size_t n_entry;
    int sq_enabled;                 /* data holds SQ codes (int8/uint8) with one global scale/bias */
    float sq_scale, sq_bias;
    int device;                     /* HIP device ordinal */
    uint64_t id_offset;             /* added to every returned id (database shard base) */
} svsx_index_desc;

/* Adopt arrays that already live in HBM in the device layout (no copy, not owned):
 *   data   n rows of data_row_stride_bytes (multiple of 16), element j of row i at i*stride + j*sizeof(T)
 *   graph  n rows of graph_row_stride_u32 ids (multiple of 16), unused slots = 0xFFFFFFFF, no degree slot; the ids of a row are PACKED from
 *          slot 0 (core/graph/graph.h:32-48: `[deg, n0 ... n_{deg-1}]` -- an SVS row has no holes): the kernels take the first empty slot of a
 *          64-slot chunk as the end of the row, exactly as get_node's span(row + 1, deg) ends there
 * Used by the GPU graph builder, which rewrites the graph between search batches. */
typedef struct {
    int metric;
    int data_dtype;
    size_t n, d;
    const void* d_data;
    size_t data_row_stride_bytes;
    const uint32_t* d_graph;
    size_t graph_row_stride_u32;
    const uint32_t* d_entry_points;
    size_t n_entry;
    int sq_enabled;
    float sq_scale, sq_bias;
    int device;
    uint64_t id_offset;
} svsx_device_index_desc;

/* VamanaSearchParameters (index/vamana/search_params.h:28-61) /
 * runtime VamanaSearchParameters (bindings/cpp/include/svs/runtime/vamana_index.h:37-44).
 * prefetch_* are accepted for source compatibility; on the GPU latency is hidden by wave occupancy. */
typedef struct {
    size_t search_window_size;      /* SearchBufferConfig window   (search_buffer.h:39-96) */
    size_t search_buffer_capacity;  /* SearchBufferConfig capacity; 0 = same as window     */
    int visited_set;                /* search_buffer_visited_set_: result-neutral on both sides */
    size_t prefetch_lookahead;
    size_t prefetch_step;
} svsx_search_params;

/* Optional side outputs of a search (device pointers for svsx_search_device, host for svsx_search).
 *   stats        nq x 6 u32: {expansions, neighbour slots, distance evaluations started, accepted inserts,
 *                candidates rejected after a partial (lower-bound) distance, candidates rejected by their bound codes}; the first two are
 *                the GreedySearchTracker counts of greedy_search.h:165 / search_tracker.h:29-62
 *   history_*    per query, the expanded nodes in order (id, distance) -- the builder's
 *                use_full_search_history pool (vamana_build.h:54-91); history_len may exceed history_cap. */
/* ABI version of this header = the number in the library's SONAME (libsvsx.so.3).  It changes whenever a struct below changes layout --
 * version 3 is the one in which svsx_search_aux gained its leading struct_size -- so a program built against an older header cannot bind to
 * this library by accident; svsx_abi_version() lets a dlopen / ctypes host check the same thing. */
#define SVSX_ABI_VERSION 3
int svsx_abi_version(void);

typedef struct {
    size_t struct_size;             /* sizeof(svsx_search_aux) as the CALLER compiled it; fields beyond it read as 0, so the struct can grow at
                                       its END without breaking callers built against an older version-3 header.  0 means "the ABI-3 size"
                                       (64 bytes: the fields below), never "whatever the library was built with" */
    uint32_t* stats;
    uint32_t* history_ids;
    float* history_dists;
    uint32_t* history_len;
    size_t history_cap;
    size_t query_row_stride_bytes;  /* 0 = dense */
    /* The `cancel` predicate of VamanaIndex::search (index/vamana/index.h:541-545,564-570), polled once per graph hop like
     * greedy_search.h:155 and once per query like extensions.h:575-578.  A word the caller may set to non-zero from any
     * thread while the call is in flight: svsx_search takes a HOST pointer (the library mirrors it into a device word the
     * kernel polls); svsx_search_device takes a DEVICE pointer (set it with an agent-scope store from another stream).
     * After cancellation the call still returns SVSX_SUCCESS; rows of queries that had not started hold the "unspecified"
     * sentinels (id all-ones, distance +inf; vamana_index_impl.h:115-123) after svsx_search and stay untouched after
     * svsx_search_device; rows of queries in flight hold whatever their buffer contained (the reference leaves both
     * unspecified).  svsx_search with a cancel word runs the batch as one launch on one stream (no chunk pipeline). */
    const volatile int* cancel_flag;
} svsx_search_aux;

/* auto_assemble / VamanaIndex ctor (index/vamana/index.h:363-377,1017-1046);
 * runtime VamanaIndex::build/load -> out-param (runtime/vamana_index.h:111-120) */
int svsx_index_create(const svsx_index_desc* desc, svsx_index** out);
int svsx_index_create_device_view(const svsx_device_index_desc* desc, svsx_index** out);
/* Scalar-quantized rows under cosine similarity (CosineSimilarityCompressed, quantization/scalar/scalar.h:155-195) carry one float per row: the
 * norm of the decompressed row, computed when the index (or view) is created -- by a kernel on the NULL stream, which the creating call waits
 * for.  A view therefore needs its rows final and visible to the NULL stream when it is created (rows written on a hipStreamNonBlocking stream:
 * synchronise that stream first, or refresh).  _refresh recomputes the norms on `hip_stream`, stream-ordered and without waiting (rows that were
 * rewritten, or written on a side stream); searches must follow on the same stream or after the caller has synchronised it.  _drop releases
 * them: such an index is answered by the kernel variant that takes the norm per candidate (identical results, ~14 % slower, no inline
 * records).  Both are no-ops on any other kind of index. */
int svsx_index_refresh_row_norms(svsx_index* index, void* hip_stream);
int svsx_index_drop_row_norms(svsx_index* index);
/* runtime VamanaIndex::destroy (runtime/vamana_index.h:111-120) */
void svsx_index_destroy(svsx_index* index);

/* Batch search, host pointers, synchronous.
 *   VamanaIndex::search(QueryResultView<I>, queries, VamanaSearchParameters, cancel)  index/vamana/index.h:564-611
 *   runtime VamanaIndex::search(n, x, k, distances, labels, params, filter)           runtime/vamana_index.h:75-83
 * ids64 (size_t labels as in QueryResult<size_t>, core/query_result.h:53-172) and/or ids32 (low 32 bits of
 * id + id_offset) may be NULL.
 * Rows with fewer than k results are padded with id = all-ones, dist = +inf (vamana_index_impl.h:205-209).
 * If capacity < k both window and capacity are raised to k (index.h:590-592).
 * Errors: k == 0, window > capacity (search_buffer.h:84-93), unsupported (query, data) type pair
 * -> SVSX_INVALID_ARGUMENT; NULL index -> SVSX_NOT_INITIALIZED (vamana_index_impl.h:115-130). */
int svsx_search(
    const svsx_index* index, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* ids64, uint32_t* ids32, float* dists, const svsx_search_aux* aux
);

/* Same, everything already resident in HBM; enqueued on `hip_stream` (hipStream_t, NULL = default stream),
 * returns without synchronising. */
int svsx_search_device(
    const svsx_index* index, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* d_ids64, uint32_t* d_ids32, float* d_dists, const svsx_search_aux* d_aux, void* hip_stream
);

/* Distance plugin surface: out[i*nb+j] = distance::compute(f, a_i, b_j) after f.fix_argument(a_i)
 * (concepts/distance.h:88-128; euclidean.h:191-200, inner_product.h:152-161, cosine.h:171-181).
 * Host pointers, synchronous. */
int svsx_distance_batch(
    int metric, int a_dtype, int b_dtype, size_t d, const void* a, size_t na, const void* b, size_t nb, float* out,
    int device
);

/* Multi-shard merge (new, no reference counterpart): per query, S sorted lists of k
 * (dist, id) -> best k overall with the metric's comparator; ties broken by lower shard then position.
 * Device pointers: dists/ids laid out [S][nq][k]. */
int svsx_merge_topk_device(
    int metric, size_t n_shards, size_t nq, size_t k, const float* d_dists, const uint64_t* d_ids, float* d_out_dists,
    uint64_t* d_out_ids, void* hip_stream
);

/* Same with explicit shard strides (in elements; 0 = dense nq*k): lets every rank ship its (distances | ids) as ONE
 * packed buffer through ONE all-gather and merge straight out of the gathered buffer. */
int svsx_merge_topk_strided_device(
    int metric, size_t n_shards, size_t nq, size_t k, const float* d_dists, size_t dist_shard_stride, const uint64_t* d_ids,
    size_t id_shard_stride, float* d_out_dists, uint64_t* d_out_ids, void* hip_stream
);

/* Sharded search (new design -- the reference has no sharded mode, the closest anchor is its partition of the queries over
 * threads, index/vamana/index.h:571-574).  One object per process / GPU.  This rank's `n_local` shards are the global shards
 * rank * n_local + s in order (contiguous id ranges: svsx_index_desc.id_offset); every rank passes the same query batch.  Per call:
 * search every local shard, merge them locally, ONE ncclAllGather (RCCL over xGMI) of nq * k * 12 bytes per rank (per chunk), world-way merge --
 * ties -> lower shard, then position, so the answer equals one merge over all shards in shard order and is identical on every rank.
 *   svsx_sharded_unique_id         ncclGetUniqueId (128 bytes): rank 0 makes it, the host program hands it to every rank
 *   svsx_sharded_create            joins the communicator (ncclCommInitRank; collective).  id128 == NULL with world == 1: no exchange step
 *   svsx_sharded_create_with_comm  for hosts that already own an ncclComm_t (not destroyed with the object)
 *   svsx_sharded_search_device     device pointers, enqueued on hip_stream (internal streams join it before the call's work ends); any nq, k.
 *                                  Calls on one object are serialised and chained, whatever streams they are made on.  Batches of >= 8 192
 *                                  queries are exchanged in two chunks: the all-gather + merge of the first runs under the second's search
 *   svsx_sharded_info              info[0] local shards, [1] world, [2] rank, [3] 1 when a communicator is attached
 * RCCL is loaded at run time (librccl.so.1); without it only world == 1 / id128 == NULL objects can be created. */
typedef struct svsx_sharded svsx_sharded;
int svsx_sharded_unique_id(void* id128);
int svsx_sharded_create(const svsx_index* const* local_shards, size_t n_local, const void* id128, int rank, int world, svsx_sharded** out);
int svsx_sharded_create_with_comm(const svsx_index* const* local_shards, size_t n_local, void* nccl_comm, int rank, int world, svsx_sharded** out);
void svsx_sharded_destroy(svsx_sharded* sharded);
int svsx_sharded_info(const svsx_sharded* sharded, uint64_t info[4]);
int svsx_sharded_search_device(
    svsx_sharded* sharded, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params, uint64_t* d_ids64,
    float* d_dists, void* hip_stream
);

/* The pipeline of svsx_sharded_search_device as data: every operation one call enqueues, in order, on its three streams (0 = the caller's,
 * 1 / 2 = internal).  A pure host function (no GPU, no object): the library executes exactly this list, tests replay it through a
 * happens-before model, a host can inspect what will overlap with what.  n_local local shards, exchange != 0 when an all-gather follows
 * (world > 1 or a communicator attached), have_last != 0 when an earlier call's staging must be chained behind.  Writes up to `cap` operations
 * and the total count (ops may be NULL to ask for the count only). */
enum { SVSX_OP_WAIT = 0, SVSX_OP_RECORD = 1, SVSX_OP_SEARCH = 2, SVSX_OP_LOCAL_MERGE = 3, SVSX_OP_ALLGATHER = 4, SVSX_OP_WORLD_MERGE = 5, SVSX_OP_STAMP = 6 };
enum { SVSX_EV_START = 0, SVSX_EV_JOIN0 = 1, SVSX_EV_JOIN1 = 2, SVSX_EV_READY0 = 3, SVSX_EV_READY1 = 4, SVSX_EV_DONE0 = 5, SVSX_EV_DONE1 = 6, SVSX_EV_LAST = 7 };
typedef struct {
    int32_t kind;    /* SVSX_OP_* */
    int32_t stream;  /* 0 caller's stream, 1 / 2 internal streams */
    int32_t chunk;   /* chunk of the batch (-1: the whole call) */
    int32_t arg;     /* WAIT / RECORD: SVSX_EV_*; SEARCH: local shard; STAMP: phase boundary 0..4 (timing events, recorded only when timing is on) */
} svsx_sharded_op;
int svsx_sharded_plan(size_t n_local, int exchange, size_t nq, int have_last, svsx_sharded_op* ops, size_t cap, size_t* n_ops);

/* Phase times of the LAST call on this object, for explaining a scaling curve: enable, search, read.  ms[0] shard searches, [1] local merge,
 * [2] all-gather (includes waiting for the slowest rank), [3] world merge -- summed over the chunks of the call -- and [4] the span from the first
 * search to the last merge (smaller than the sum when the second chunk's searches overlap the first chunk's exchange).  Synchronises with the
 * call.  Timing adds five event records per chunk; leave it off in production. */
int svsx_sharded_set_timing(svsx_sharded* sharded, int enabled);
int svsx_sharded_phase_times(svsx_sharded* sharded, float ms[5]);

/* Exhaustive search: svs::index::flat::FlatIndex::search (index/flat/flat.h:322-460) -- every query
 * against every vector, k best by (distance, then id), EXACT, any k and d.  Up to k = 224 a float16 MFMA pass shortlists min(128, max(4k, k+32)) candidates per query
 * (k + 32 beyond k = 96; always >= k + 32; large sets up to d = 768: every row under a sampled threshold, about 16 (k + 4) per query), which are re-ranked with the exact reference-order distance kernels; then every query's answer is
 * proven against the float16 rounding bounds (every row left out is strictly worse than the k-th answer), and a query that cannot be proven
 * (near-ties below float16 resolution, non-finite values, zero rows under cosine) is recomputed by an exhaustive float32 pass (which alone answers k > 224 and d beyond ~1 200).
 * _counted adds the number of such queries to *d_recomputed (a device word the caller zeroes; NULL = not wanted).
 * Works on any index, including one created with graph == NULL (a flat index).  Device pointers, enqueued on hip_stream; d_dists may be NULL. */
int svsx_flat_search_device(
    const svsx_index* index, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, void* hip_stream
);
int svsx_flat_search_device_counted(
    const svsx_index* index, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, uint32_t* d_recomputed,
    void* hip_stream
);
/* The same search with host pointers in and out (svs::Flat::search, include/svs/orchestrators/exhaustive.h): queries [nq][d] of q_dtype, ids64 [nq][k],
 * dists [nq][k] or NULL.  Blocks until the answer is in the caller's arrays. */
int svsx_flat_search(const svsx_index* index, const void* queries, int q_dtype, size_t nq, size_t k, uint64_t* ids64, float* dists);

/* VamanaBuildParameters (index/vamana/build_params.h:28-75). */
typedef struct {
    float alpha;
    size_t graph_max_degree;
    size_t window_size;
    size_t max_candidate_pool_size;
    size_t prune_to;                /* 0 = graph_max_degree - 4 */
    int use_full_search_history;    /* 1 (the reference's default): a vertex's candidates are the nodes its search expanded; 0: the search buffer's contents (vamana_build.h:407-421) */
} svsx_build_params;

/* Graph construction ("next" scope): the VamanaIndex build constructor (index/vamana/index.h:398-439 ->
 * VamanaBuilder::construct twice, vamana_build.h:232-599) for an index created with svsx_index_create_device_view over the data,
 * ONE entry point and a graph whose every slot is 0xFFFFFFFF.  Fills that graph in place; batch-synchronous and deterministic
 * (the reference's build depends on its thread schedule: graphs are comparable by quality, not byte for byte).
 * svsx_find_medoid_device: utils::find_medioid (core/medioid.h:292-353), the reference's default entry point. */
int svsx_vamana_build_device(svsx_index* index, const svsx_build_params* params, void* hip_stream);
int svsx_find_medoid_device(const svsx_index* index, uint32_t* medoid);

/* Builder support ("next" scope): heuristic_prune_neighbors (index/vamana/prune.h:100-246; progressive
 * strategy for L2, iterative for IP / cosine) for a batch of vertices of `index`.  Device pointers:
 *   pool_ids [n_nodes][pool_size] candidates sorted best-first by (distance, id), -1 padded at the end,
 *   pool_dists the matching distances to the vertex, out [n_nodes][out_stride] pruned adjacency (-1 padded). */
int svsx_prune_device(
    const svsx_index* index, const int32_t* d_node_ids, const int32_t* d_pool_ids, const float* d_pool_dists, size_t n_nodes,
    size_t pool_size, size_t max_out, float alpha, int32_t* d_out, size_t out_stride, void* hip_stream
);

/* Batched BatchIterator: svs::index::vamana::BatchIterator (index/vamana/iterator.h:81-339), one independent iterator
 * per query of the batch, advanced together by one kernel launch per next().  Each next() widens every live query's
 * search window and capacity by its batch size, RESUMES the graph search from the buffer left by the previous call
 * (RestartInitializer, iterator.h:36-63) and hands out the best buffer entries not handed out before
 * (copy_from_scratch, :104-124).  This is what the reference's runtime builds filtered and range search on
 * (bindings/cpp/src/vamana_index_impl.h:140-320).  Per-query state (search buffer, yielded ids) stays in HBM.
 *   extra_search_buffer_capacity  SIZE_MAX = svs::ITERATOR_EXTRA_BUFFER_CAPACITY_DEFAULT (100)  (iterator.h:154-170)
 * Not thread-safe: one caller at a time per iterator (as for the reference's object). */
typedef struct svsx_batch_iterator svsx_batch_iterator;

/* VamanaIndex::make_batch_iterator (index/vamana/index.h:936), one per row of `queries` (host pointer). */
int svsx_iterator_create(
    const svsx_index* index, const void* queries, int q_dtype, size_t nq, size_t extra_search_buffer_capacity,
    svsx_batch_iterator** out
);
void svsx_iterator_destroy(svsx_batch_iterator* it);
/* BatchIterator::update (iterator.h:177-190): new queries (same count and type), all state reset. */
int svsx_iterator_update(svsx_batch_iterator* it, const void* queries);
/* BatchIterator::restart_next_search (iterator.h:233) for every query. */
int svsx_iterator_restart_next_search(svsx_batch_iterator* it);
/* BatchIterator::next (iterator.h:249-317).  Outputs (host pointers): ids64/dists [nq][batch_size] (unused slots
 * all-ones / +inf), counts [nq] = size() of each query's new batch.  batch_sizes (optional, host, nq entries <=
 * batch_size): per-query batch size, 0xFFFFFFFF = leave that query untouched.  A query that is done() yields nothing
 * and keeps its state (iterator.h:253-256). */
int svsx_iterator_next(
    svsx_batch_iterator* it, size_t batch_size, const uint32_t* batch_sizes, uint64_t* ids64, float* dists, uint32_t* counts
);
/* Same with device pointers, enqueued on hip_stream (state growth, if needed, synchronises the device first). */
int svsx_iterator_next_device(
    svsx_batch_iterator* it, size_t batch_size, const uint32_t* d_batch_sizes, uint64_t* d_ids64, float* d_dists,
    uint32_t* d_counts, void* hip_stream
);
/* Per-query state, 8 u32 each (host pointer): [0] buffer size, [1] best_unvisited, [2] search window, [3] capacity
 * (parameters_for_current_iteration, iterator.h:236-245), [4] ids yielded so far, [5] batch_number() (:227),
 * [6] flags: 1 = next search restarts from the entry points, 2 = exhausted; done() (:230) = [4] == n || flags & 2. */
int svsx_iterator_state(const svsx_batch_iterator* it, uint32_t* meta);

/* Batch search with an id filter: the runtime's VamanaIndex::search(n, x, k, distances, labels, params, IDFilter* filter)
 * (bindings/cpp/include/svs/runtime/vamana_index.h:75-83; algorithm bindings/cpp/src/vamana_index_impl.h:140-218 with
 * filter_estimate_batch = false; the runtime's default, true, is svsx_search_filtered_ex below).  id_bitmap: bit i of the (n + 63) / 64 words = IDFilter::is_member(i), local ids.
 * filter_stop: give up on a query (all-padding row) once its observed hit rate falls below it; 0 = never
 * (SearchParams::filter_stop, api_defs.h).  Per query the answer equals the reference's iterator loop: adaptive batches from
 * a BatchIterator until k members are found or the iterator is exhausted.  Host pointers, synchronous. */
int svsx_search_filtered(
    const svsx_index* index, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    const uint64_t* id_bitmap, float filter_stop, uint64_t* ids64, float* dists
);
/* Same with SearchParams::filter_estimate_batch (api_defs.h; the runtime's default is TRUE): before any graph traversal the filter's hit
 * rate is estimated on max(200, 1 / filter_stop) ids drawn by std::mt19937(42) (sample_filter_hits, svs_runtime_utils.h:445-471); a
 * rate below filter_stop answers every query with padding (vamana_index_impl.h:163-166), otherwise it sets the first batch size of every
 * query (predict_further_processing, :167-169).  filter_estimate_batch = 0 is svsx_search_filtered. */
int svsx_search_filtered_ex(
    const svsx_index* index, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    const uint64_t* id_bitmap, float filter_stop, int filter_estimate_batch, uint64_t* ids64, float* dists
);

/* Range search: the runtime's VamanaIndex::range_search(n, x, radius, ResultsAllocator, params, IDFilter* filter)
 * (bindings/cpp/include/svs/runtime/vamana_index.h:85-92; algorithm bindings/cpp/src/vamana_index_impl.h:220-326).  Per query a BatchIterator
 * is advanced in batches of max(search_window_size, 10); its results come in distance order, every one closer than `radius` (svs::distance
 * comparator of the index's metric: < for L2, > for inner product / cosine) and -- with a filter -- a member of it is kept, and the query ends at
 * its first out-of-range neighbour or when the iterator is done.  All queries advance together, one kernel launch per round.
 * When the counts are known `allocate(user, counts, nq, &storage)` is called ONCE (ResultsAllocator::allocate, api_defs.h:194-207) and must hand
 * back storage for sum(counts) labels and distances; they are filled query by query, in the order found (labels are global ids, id_offset included).
 * id_bitmap: NULL = no filter, else bit i of (n + 63) / 64 words = IDFilter::is_member(i), local ids.  Host pointers, synchronous. */
typedef struct {
    uint64_t* labels;
    float* distances;
} svsx_range_storage;
typedef void (*svsx_results_allocator)(void* user, const size_t* result_counts, size_t nq, svsx_range_storage* out); /* fills *out (an out-parameter: callable from ctypes / cgo / JNI) */
int svsx_range_search(
    const svsx_index* index, const void* queries, int q_dtype, size_t nq, float radius, const svsx_search_params* params,
    const uint64_t* id_bitmap, svsx_results_allocator allocate, void* user
);

/* The runtime's VamanaIndex::get_distance(id, query, &distance) (runtime/vamana_index.h:94-96; index/vamana/index.h:915-933 ->
 * extensions::get_distance_ext, extensions.h:643-662): distance::compute(fix_argument(query), data[id]) with the index's own kernels -- the
 * value a search reports for that row, bit for bit (scalar-quantized indexes: the compressed distance).  `id` is global (id_offset included).
 * Host pointers, synchronous. */
int svsx_get_distance(const svsx_index* index, uint64_t id, const void* query, int q_dtype, double* distance);

/* Bound codes (no reference counterpart; result-neutral).  The search kernel turns them into a rigorous BOUND of the distance (lower
 * bound for L2, upper bound for inner product / cosine) and drops candidates that SearchBuffer::can_skip (search_buffer.h:338-340) would
 * drop anyway before their full row is fetched -- ids and distances returned are unchanged, only the number of 128-byte lines requested
 * per hop falls.
 *   mode 2        graph-inline records (the default of bench.py): every adjacency row carries a 32- / 64- / 128-byte record per neighbour
 *                 (8-bit codes of the leading coordinates in a principal basis of the data, coding error, norm of the uncoded remainder),
 *                 so one hop reads ONE contiguous block of (4 + record) * max_degree bytes and bounds all its neighbours at once.  The record
 *                 size (28 / 32 / 64 / 128 bytes) is chosen from the data's spectrum (expected lines per hop).  float32 / float16 rows, scalar-quantized rows (the
 *                 compressed L2 distance with records in code space, scalar.h:58-105) and raw int8 / uint8 rows searched with queries of
 *                 their own type (the VNNI pairs), under any metric; dimension 32..4096.
 *                 MEMORY: + (4 + record) * graph_row_stride bytes of HBM per vector -- 2.3 GB for 1 M vectors at R = 64 with 32-byte records
 *                 (6x the 384 MB of d = 96 float32 rows), 29 GB per 12.5 M-vector shard, 169 GB for 10 M x R = 128 with 128-byte records.
 *                 When that does not fit the free HBM the call does NOT fail: it builds smaller records (64, 32 bytes), then mode 1 where the
 *                 row type has it, then nothing, and returns SVSX_FELL_BACK (svsx_last_error() names what was built).
 *   mode 3        mode 2 where the size rule expects the records to save at least a tenth of the lines per hop, NOTHING where it does not (full-rank
 *                 data whose leading directions carry too little: a bound that rejects nothing costs its block lines for nothing).  Indexes of
 *                 >= 100 000 vectors get a second opinion: the records are timed against the plain kernel on 2 048 of the index's own rows as
 *                 queries and kept only if faster.  SVSX_SUCCESS either way, svsx_index_memory()[8] tells which, svsx_last_error() why.
 *                 The default a host should use when it does not know its data.
 *   mode 28 / 32 / 64 / 128   the same as mode 2 with exactly this record size (refused when larger than a third of the rows it protects; 28:
 *                 float32 / float16 rows only -- a 16- and a 12-byte plane, so that a 64-slot block is 2 048 bytes = 16 lines instead of 18)
 *   mode 1        one 128-byte row per vector: 8-bit codes of its first min(d, 120) components + exact coding error (float32 / float16
 *                 rows, L2, d = 96 or 128; +128 B per vector, one extra line per candidate).  For indexes whose inline rows do not fit.
 *   mode 0        frees either.
 * SVSX_NOT_IMPLEMENTED for other row types.  Call while no search is in flight on this index. */
int svsx_index_build_bound_codes(svsx_index* index, int mode);

/* Index facts: info[0]=n, [1]=d, [2]=data bytes in HBM, [3]=graph bytes in HBM, [4]=graph row stride (u32),
 * [5]=data row stride (bytes). */
int svsx_index_info(const svsx_index* index, uint64_t info[6]);

/* HBM the index holds and the bound scheme in effect (bytes): mem[0] vector rows, [1] graph, [2] graph-inline records, [3] their record size
 * (32 / 64 / 128; 0 = none), [4] 128-byte code rows, [5] float16 shadow + squared norms of the flat search (built by the first flat search),
 * [6] row norms of scalar-quantized cosine rows, [7] total ALLOCATED BY THE LIBRARY (a device view borrows [0] and [1]), [8] bound scheme in
 * effect (0 none, 1 code rows, 2 inline records), [9] 1 when the index owns its rows and graph. */
int svsx_index_memory(const svsx_index* index, uint64_t mem[10]);

/* Status::message() (api_defs.h:119-184): text of the last failure on this thread. */
const char* svsx_last_error(void);

/* Library version string. */
const char* svsx_version(void);

/* Diagnostics: name of the search-kernel instantiation the last svsx_search* / svsx_iterator_next* call on this thread
 * launched (K7 dispatch, SURVEY 8a: which static-dimension variant served the call), e.g.
 * "search_kernel<dt=1,metric=1,qmode=0,ds=128,general=0,codes=0>".  Empty before the first search. */
const char* svsx_last_kernel_info(void);

#ifdef __cplusplus
}
#endif
#endif /* SVSX_H */
