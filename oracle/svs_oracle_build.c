/*
 * svs_oracle_build.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Single-threaded, deterministic restatement of the reference's batch-synchronous Vamana builder, used to
 * make small graphs for parity fixtures and tests.  Restates:
 *   VamanaBuilder::construct / generate_neighbors / add_reverse_edges   index/vamana/vamana_build.h:232-599
 *   heuristic_prune_neighbors (progressive for L2, iterative for IP/cos) index/vamana/prune.h:38-48,100-246
 *   two passes (alpha = 1.0 then alpha)                                  index/vamana/index.h:426-439
 *   medoid entry point                                                   core/medioid.h:292-353
 * The reference build is thread-schedule dependent, so graphs are only comparable by quality;
 * search parity is always "same graph bytes in, same ids out".
 */
#include "svs_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint32_t id;
    float dist;
} cand;

static int g_greater;
static int cand_total_order(const void* pa, const void* pb) { /* TotalOrder: lib/neighbor.h:151-163 */
    const cand* a = (const cand*)pa;
    const cand* b = (const cand*)pb;
    int ab = g_greater ? (a->dist > b->dist) : (a->dist < b->dist);
    int ba = g_greater ? (b->dist > a->dist) : (b->dist < a->dist);
    if (ab) return -1;
    if (ba) return 1;
    return (a->id < b->id) ? -1 : (a->id > b->id);
}

static float pair_distance(const svso_index* ix, uint32_t a, uint32_t b) {
    size_t rs = ix->d * svso_dtype_size(ix->data_dtype);
    const char* pa = (const char*)ix->data + (size_t)a * rs;
    const char* pb = (const char*)ix->data + (size_t)b * rs;
    float an = ix->metric == SVSO_COS ? svso_norm(pa, ix->data_dtype, ix->d) : 0.0f;
    int st;
    return svso_distance(ix->metric, pa, ix->data_dtype, pb, ix->data_dtype, ix->d, an, &st);
}

/* prune.h:100-246.  pool sorted by TotalOrder; writes <= max_out ids. */
static size_t prune(const svso_index* ix, uint32_t node, const cand* pool, size_t np, size_t max_out, float alpha, uint32_t* out) {
    int greater = ix->metric != SVSO_L2;
    size_t nout = 0;
    if (np == 0) return 0;
    float* state = (float*)malloc(np * sizeof(float));
    if (!greater) {
        /* ProgressivePruneStrategy (L2): state[t] = max ratio d(node,t)/d(sel,t); -FLT_MAX = untouched */
        for (size_t i = 0; i < np; ++i) state[i] = -3.402823466e+38f;
        float cur = 1.0f;
        while (nout < max_out && !(alpha < cur)) {
            size_t start = 0;
            while (nout < max_out && start < np) {
                uint32_t id = pool[start].id;
                if (cur < state[start] || id == node) { ++start; continue; }
                state[start] = 3.402823466e+38f;
                out[nout++] = id;
                for (size_t t = start + 1; t < np; ++t) {
                    if (cur < state[t]) continue;
                    float djk = pair_distance(ix, id, pool[t].id);
                    float r = pool[t].dist / djk;
                    if (state[t] < r) state[t] = r;
                }
                ++start;
            }
            if (alpha == 1.0f) break;
            cur *= alpha;
        }
    } else {
        /* IterativePruneStrategy (IP / cosine): 0 available, 1 added, 2 pruned */
        memset(state, 0, np * sizeof(float));
        float cur = 1.0f;
        while (nout < max_out && !(alpha > cur)) {
            size_t start = 0;
            while (nout < max_out && start < np) {
                uint32_t id = pool[start].id;
                if (state[start] != 0.0f || id == node) { ++start; continue; }
                state[start] = 1.0f;
                out[nout++] = id;
                for (size_t t = start + 1; t < np; ++t) {
                    if (state[t] != 0.0f) continue;
                    float djk = pair_distance(ix, id, pool[t].id);
                    if (cur * djk > pool[t].dist) state[t] = 2.0f;
                }
                ++start;
            }
            if (alpha == 1.0f) break;
            for (size_t t = 0; t < np; ++t) if (state[t] == 2.0f) state[t] = 0.0f;
            cur *= alpha;
        }
    }
    free(state);
    return nout;
}

static uint32_t medoid(const svso_index* ix) {
    double* c = (double*)calloc(ix->d, sizeof(double));
    size_t es = svso_dtype_size(ix->data_dtype);
    for (size_t i = 0; i < ix->n; ++i) {
        const char* row = (const char*)ix->data + i * ix->d * es;
        for (size_t j = 0; j < ix->d; ++j) {
            double v;
            switch (ix->data_dtype) {
                case SVSO_F32: v = ((const float*)row)[j]; break;
                case SVSO_F16: v = svso_half_to_float(((const uint16_t*)row)[j]); break;
                case SVSO_I8: v = ((const int8_t*)row)[j]; break;
                default: v = ((const uint8_t*)row)[j]; break;
            }
            c[j] += v;
        }
    }
    for (size_t j = 0; j < ix->d; ++j) c[j] /= (double)ix->n;
    double best = 1e300;
    uint32_t arg = 0;
    for (size_t i = 0; i < ix->n; ++i) {
        const char* row = (const char*)ix->data + i * ix->d * es;
        double s = 0;
        for (size_t j = 0; j < ix->d; ++j) {
            double v;
            switch (ix->data_dtype) {
                case SVSO_F32: v = ((const float*)row)[j]; break;
                case SVSO_F16: v = svso_half_to_float(((const uint16_t*)row)[j]); break;
                case SVSO_I8: v = ((const int8_t*)row)[j]; break;
                default: v = ((const uint8_t*)row)[j]; break;
            }
            s += (v - c[j]) * (v - c[j]);
        }
        if (s < best) { best = s; arg = (uint32_t)i; }
    }
    free(c);
    return arg;
}

static int has_edge(const uint32_t* row, uint32_t id) {
    for (uint32_t j = 0; j < row[0]; ++j) if (row[1 + j] == id) return 1;
    return 0;
}

static int g_full_history = 1; /* VamanaBuildParameters::use_full_search_history (build_params.h:28-75) */

static void one_pass(svso_index* ix, uint32_t* graph, size_t R, size_t L, size_t C, size_t prune_to, float build_alpha, float pass_alpha, uint32_t entry) {
    size_t n = ix->n, stride = R + 1;
    size_t nb = (n + 4095) / 4096;
    if (nb < 40) nb = 40;
    if (nb > n) nb = n;
    size_t bs = (n + nb - 1) / nb;
    int greater = ix->metric != SVSO_L2;
    g_greater = greater;
    svso_buffer buf;
    svso_buffer_init(&buf, L, L, greater, 1);
    size_t hist_cap = 64 * L + 1024;
    uint32_t* hist = (uint32_t*)malloc(hist_cap * sizeof(uint32_t));
    float* hist_d = (float*)malloc(hist_cap * sizeof(float));
    cand* pool = (cand*)malloc((hist_cap + R + 4096) * sizeof(cand));
    uint8_t* mark = (uint8_t*)calloc(n, 1);
    uint32_t* updates = (uint32_t*)malloc(bs * stride * sizeof(uint32_t));
    /* overflow back-edges: (dst, src) pairs */
    size_t ov_cap = bs * R + 16, ov_n;
    uint32_t* ov = (uint32_t*)malloc(ov_cap * 2 * sizeof(uint32_t));
    uint32_t ep[1] = {entry};
    ix->entry_points = ep;
    ix->n_entry = 1;
    size_t es = svso_dtype_size(ix->data_dtype);

    for (size_t b = 0; b < nb; ++b) {
        size_t lo = bs * b < n ? bs * b : n, hi = bs * (b + 1) < n ? bs * (b + 1) : n;
        /* generate_neighbors: search with the CURRENT graph, prune, apply all updates afterwards */
        for (size_t node = lo; node < hi; ++node) {
            size_t hl = 0;
            svso_greedy_search(ix, (const char*)ix->data + node * ix->d * es, ix->data_dtype, &buf, NULL, hist + 1, hist_d + 1, hist_cap - 1, &hl);
            /* full search history = entry point + every expanded node (tracker.visited) */
            size_t np = 0;
            hist[0] = entry;
            hist_d[0] = pair_distance(ix, (uint32_t)node, entry);
            if (hl > hist_cap - 1) hl = hist_cap - 1;
            if (g_full_history) {
                for (size_t i = 0; i <= hl; ++i) {
                    if (mark[hist[i]]) continue;
                    mark[hist[i]] = 1;
                    pool[np].id = hist[i];
                    pool[np].dist = hist_d[i];
                    np++;
                }
            } else { /* vamana_build.h:416-421: "pull results directly out of the search buffer" */
                for (size_t i = 0; i < buf.size; ++i) {
                    uint32_t id = buf.e[i].id & 0x7FFFFFFFu;
                    if (mark[id]) continue;
                    mark[id] = 1;
                    pool[np].id = id;
                    pool[np].dist = buf.e[i].dist;
                    np++;
                }
            }
            const uint32_t* row = graph + node * stride;
            for (uint32_t j = 0; j < row[0]; ++j) {
                uint32_t id = row[1 + j];
                if (mark[id]) continue;
                mark[id] = 1;
                pool[np].id = id;
                pool[np].dist = pair_distance(ix, (uint32_t)node, id);
                np++;
            }
            for (size_t i = 0; i < np; ++i) mark[pool[i].id] = 0;
            qsort(pool, np, sizeof(cand), cand_total_order);
            if (np > C) np = C;
            uint32_t* up = updates + (node - lo) * stride;
            up[0] = (uint32_t)prune(ix, (uint32_t)node, pool, np, R, build_alpha, up + 1);
        }
        for (size_t node = lo; node < hi; ++node) memcpy(graph + node * stride, updates + (node - lo) * stride, stride * sizeof(uint32_t));

        /* add_reverse_edges */
        ov_n = 0;
        for (size_t node = lo; node < hi; ++node) {
            const uint32_t* row = graph + node * stride;
            uint32_t deg = row[0];
            for (uint32_t j = 0; j < deg; ++j) {
                uint32_t other = graph[node * stride + 1 + j];
                uint32_t* orow = graph + (size_t)other * stride;
                if (has_edge(orow, (uint32_t)node)) continue;
                if (orow[0] < R) {
                    orow[1 + orow[0]] = (uint32_t)node;
                    orow[0]++;
                } else if (ov_n < ov_cap) {
                    ov[2 * ov_n] = other;
                    ov[2 * ov_n + 1] = (uint32_t)node;
                    ov_n++;
                }
            }
        }
        /* prune every overflowed vertex on (overflow candidates U old adjacency) */
        for (size_t i = 0; i < ov_n; ++i) {
            uint32_t src = ov[2 * i];
            if (src == 0xFFFFFFFFu) continue;
            size_t np = 0;
            for (size_t k2 = i; k2 < ov_n; ++k2) {
                if (ov[2 * k2] != src) continue;
                uint32_t id = ov[2 * k2 + 1];
                ov[2 * k2] = 0xFFFFFFFFu;
                if (mark[id]) continue;
                mark[id] = 1;
                pool[np].id = id;
                pool[np].dist = pair_distance(ix, src, id);
                np++;
            }
            uint32_t* srow = graph + (size_t)src * stride;
            for (uint32_t j = 0; j < srow[0]; ++j) {
                uint32_t id = srow[1 + j];
                if (mark[id]) continue;
                mark[id] = 1;
                pool[np].id = id;
                pool[np].dist = pair_distance(ix, src, id);
                np++;
            }
            for (size_t t = 0; t < np; ++t) mark[pool[t].id] = 0;
            qsort(pool, np, sizeof(cand), cand_total_order);
            if (np > C) np = C;
            srow[0] = (uint32_t)prune(ix, src, pool, np, prune_to, pass_alpha, srow + 1);
        }
    }
    svso_buffer_free(&buf);
    free(hist); free(hist_d); free(pool); free(mark); free(updates); free(ov);
}

/* heuristic_prune_neighbors on one explicit pool (checker for the HIP prune kernel). */
size_t svso_prune(int metric, int data_dtype, const void* data, size_t n, size_t d, uint32_t node, const uint32_t* pool_ids,
                  const float* pool_dists, size_t np, size_t max_out, float alpha, uint32_t* out) {
    svso_index ix;
    memset(&ix, 0, sizeof(ix));
    ix.metric = metric; ix.data_dtype = data_dtype; ix.n = n; ix.d = d; ix.data = data;
    cand* pool = (cand*)malloc((np + 1) * sizeof(cand));
    for (size_t i = 0; i < np; ++i) { pool[i].id = pool_ids[i]; pool[i].dist = pool_dists[i]; }
    size_t r = prune(&ix, node, pool, np, max_out, alpha, out);
    free(pool);
    return r;
}

int svso_build_vamana(
    int metric, int data_dtype, const void* data, size_t n, size_t d, size_t R, size_t L, size_t C, size_t prune_to,
    float alpha, uint32_t* graph, uint32_t* entry_out
);

/* the same with VamanaBuildParameters::use_full_search_history (0: candidates = the search buffer's contents, vamana_build.h:407-421) */
int svso_build_vamana_ex(
    int metric, int data_dtype, const void* data, size_t n, size_t d, size_t R, size_t L, size_t C, size_t prune_to,
    float alpha, int use_full_search_history, uint32_t* graph, uint32_t* entry_out
) {
    g_full_history = use_full_search_history ? 1 : 0;
    const int rc = svso_build_vamana(metric, data_dtype, data, n, d, R, L, C, prune_to, alpha, graph, entry_out);
    g_full_history = 1;
    return rc;
}

/* Builds an SVS-layout graph (n x (R+1) u32, slot 0 = degree) into `graph`; returns the entry point. */
int svso_build_vamana(
    int metric, int data_dtype, const void* data, size_t n, size_t d, size_t R, size_t L, size_t C, size_t prune_to,
    float alpha, uint32_t* graph, uint32_t* entry_out
) {
    svso_index ix;
    memset(&ix, 0, sizeof(ix));
    ix.metric = metric;
    ix.data_dtype = data_dtype;
    ix.n = n;
    ix.d = d;
    ix.data = data;
    ix.graph = graph;
    ix.graph_stride = R + 1;
    memset(graph, 0, n * (R + 1) * sizeof(uint32_t));
    uint32_t entry = medoid(&ix);
    one_pass(&ix, graph, R, L, C, prune_to, alpha, 1.0f, entry);
    one_pass(&ix, graph, R, L, C, prune_to, alpha, alpha, entry);
    *entry_out = entry;
    return 0;
}
