/*
 * svs_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, CPU restatement of the reference's (intel/ScalableVectorSearch 0.4.0) batched Vamana
 * greedy-search hot path.  It exists to CHECK the HIP path (tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py).  Nothing under scalablevectorsearch_amd/ may include, link or call it.
 *
 * Every function cites the reference file:line it restates (paths relative to the reference root).
 *
 * Parity pinning (see DESIGN.md "Oracle"):
 *   - search buffer / visited filter / greedy traversal: pinned against the reference's own
 *     known-answer tests (tests/svs/index/vamana/search_buffer.cpp:382-520, filter.cpp:34-103) and,
 *     bit-for-bit, against the reference's own headers compiled into oracle/_ref (ref_driver.cpp).
 *   - float/int distance accumulation order: pinned bit-for-bit against the reference's
 *     simd::generic_simd_op skeleton compiled into oracle/_ref.
 *   - SQ (quantization/scalar/scalar.h) and the cosine query norm: "parity unpinned" -- those headers
 *     need third-party headers absent from this image; restated from source only.
 */
#ifndef SVS_ORACLE_H
#define SVS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SVSO_F32 = 0, SVSO_F16 = 1, SVSO_I8 = 2, SVSO_U8 = 3 };
enum { SVSO_L2 = 0, SVSO_IP = 1, SVSO_COS = 2 };

size_t svso_dtype_size(int dtype);

/* lib/float16.h -- hardware-exact half->float (vcvtph2ps semantics, denormals preserved). */
float svso_half_to_float(uint16_t h);
float svso_half_to_float_scalar(uint16_t h); /* Float16::operator float (lib/float16.h:45-52): denormals flushed, inf / NaN finite */
/* round-to-nearest-even float->half (used by data generators only). */
uint16_t svso_float_to_half(float f);

/* core/distance/distance_core.h:45-66 -- sqrt of sequential f32 accumulation of x*x. */
float svso_norm(const void* x, int dtype, size_t d);

/* core/distance/{euclidean,inner_product,cosine}.h AVX-512 paths (K1-K6 of SURVEY 8a).
 * Returns NaN and sets *status=-1 for type pairs without an AVX-512 specialisation.           */
float svso_distance(
    int metric, const void* a, int a_dtype, const void* b, int b_dtype, size_t d, float a_norm, int* status
);

/* all-pairs helper: out[i*nb+j] = distance(a_i, b_j); cosine norms computed internally. */
int svso_distance_batch(
    int metric, const void* a, int a_dtype, size_t na, const void* b, int b_dtype, size_t nb, size_t d, float* out
);

/* ---- index/vamana/search_buffer.h ---- */
typedef struct {
    uint32_t id;
    float dist;
    uint8_t visited;
} svso_neighbor;

typedef struct {
    int greater;        /* 0: std::less (L2), 1: std::greater (IP / cosine) */
    size_t size;
    size_t best_unvisited;
    size_t window;
    size_t capacity;
    svso_neighbor* e;   /* capacity + 1 entries */
    uint16_t* filter;   /* VisitedFilter<uint32_t,16> (filter.h:49-130) or NULL */
} svso_buffer;

int svso_buffer_init(svso_buffer* b, size_t window, size_t capacity, int greater, int visited_set);
void svso_buffer_free(svso_buffer* b);
void svso_buffer_clear(svso_buffer* b);
void svso_buffer_push_back(svso_buffer* b, uint32_t id, float dist);
void svso_buffer_sort(svso_buffer* b);   /* libstdc++ std::sort order, ties included */
int svso_sort_depth_exhausted(void);
size_t svso_buffer_insert(svso_buffer* b, uint32_t id, float dist);
int svso_buffer_done(const svso_buffer* b);
svso_neighbor svso_buffer_next(svso_buffer* b);
int svso_filter_emplace(uint16_t* filter, uint32_t key);
int svso_filter_contains(const uint16_t* filter, uint32_t key);
void svso_filter_reset(uint16_t* filter);

/* ---- index description shared by the search entry points ---- */
typedef struct {
    int metric;
    int data_dtype;
    size_t n, d;
    const void* data;           /* row-major n x d */
    const uint32_t* graph;      /* SVS row layout: row[0]=degree, row[1..deg]=ids (core/graph/graph.h:32-48) */
    size_t graph_stride;        /* u32 per row (= max_degree + 1) */
    const uint32_t* entry_points;
    size_t n_entry;
    int sq_enabled;             /* data are SQ codes (quantization/scalar/scalar.h) */
    float sq_scale, sq_bias;
} svso_index;

typedef struct {
    uint64_t expansions;        /* |E|: nodes popped by next()            */
    uint64_t neighbor_slots;    /* sum of deg(v), v in E                  */
    uint64_t distance_evals;    /* distances actually computed            */
    uint64_t in_buffer;         /* neighbour slots whose id was already in the buffer (analysis only) */
    uint64_t small_filter_hits[4]; /* hits of simulated 512/1024/2048/4096-entry exact-tag filters (analysis only) */
    uint64_t partial_rejects;   /* f32 L2, d > 64: first-64-dims partial distance already above the worst (analysis only) */
} svso_stats;

/* index/vamana/greedy_search.h:124-203 on one query; leaves the result in `buf`. */
int svso_greedy_search(
    const svso_index* ix, const void* query, int q_dtype, svso_buffer* buf, svso_stats* stats,
    uint32_t* history, float* history_dist, size_t history_cap, size_t* history_len
);

/* ---- index/vamana/iterator.h:81-339 BatchIterator over one query (static index) ---- */
typedef struct {
    const svso_index* ix;
    void* query;
    int q_dtype;
    size_t extra;           /* extra_search_buffer_capacity_ */
    svso_buffer buf;        /* scratchspace_.buffer: persists between next() calls */
    uint32_t* yielded;      /* yielded_ (ids in yield order) */
    size_t n_yielded, yielded_cap;
    size_t iteration;       /* batch_number() */
    int restart;            /* restart_search_ */
    int exhausted;          /* is_exhausted_ */
} svso_iterator;

int svso_buffer_change_maxsize(svso_buffer* b, size_t window, size_t capacity);
int svso_iterator_init(svso_iterator* it, const svso_index* ix, const void* query, int q_dtype, size_t extra_capacity);
void svso_iterator_free(svso_iterator* it);
int svso_iterator_done(const svso_iterator* it);
void svso_iterator_restart_next_search(svso_iterator* it);
int svso_iterator_update(svso_iterator* it, const void* query);
long svso_iterator_next(svso_iterator* it, size_t batch_size, uint32_t* ids, float* dists);

/* index/vamana/index.h:564-611 + extensions.h:555-592 (single thread): ids32 nq x k, dists nq x k.
 * Rows with fewer than k results are padded with id 0xFFFFFFFF and dist +inf (runtime convention, api_defs.h:58-66). */
int svso_search_batch(
    const svso_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, size_t window, size_t capacity,
    int visited_set, uint32_t* ids, float* dists, svso_stats* per_query_stats
);

/* ---- quantization/scalar/scalar.h ---- */
void svso_sq_params(const float* data, size_t n, size_t d, int code_dtype, float* scale, float* bias);
void svso_sq_compress(const float* src, size_t count, int code_dtype, float scale, float bias, void* dst);

/* ---- core/recall.h:47-58 ---- */
double svso_k_recall_at_n(const uint32_t* gt, size_t gt_stride, const uint32_t* res, size_t res_stride, size_t nq, size_t k, size_t n);

/* index/flat/flat.h exhaustive search (ground truth), total order = (distance, then position). */
int svso_flat_search(const svso_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, uint32_t* ids, float* dists);

/* index/vamana/prune.h:100-246 on one explicit, sorted pool; returns the number of ids written to out. */
size_t svso_prune(int metric, int data_dtype, const void* data, size_t n, size_t d, uint32_t node, const uint32_t* pool_ids,
                  const float* pool_dists, size_t np, size_t max_out, float alpha, uint32_t* out);

/* index/vamana/vamana_build.h restated (single-threaded, deterministic); graph: n x (R+1) u32 SVS rows. */
int svso_build_vamana(
    int metric, int data_dtype, const void* data, size_t n, size_t d, size_t R, size_t L, size_t C, size_t prune_to,
    float alpha, uint32_t* graph, uint32_t* entry_out
);

#ifdef __cplusplus
}
#endif
#endif
