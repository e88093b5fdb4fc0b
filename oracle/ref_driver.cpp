// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// C-ABI wrapper that compiles the REFERENCE'S OWN HEADERS, included from where they lie under
// /root/reference/include, into oracle/_ref/libsvs_ref.so (see oracle/Makefile).  No reference source is
// copied into this repository and no stand-in for a missing third-party header is used.
//
// What is real reference code here (buildable with only the fmt headers that ship inside the image's torch):
//   * svs::index::vamana::greedy_search          index/vamana/greedy_search.h:124-203
//   * svs::index::vamana::EntryPointInitializer  index/vamana/greedy_search.h:62-94
//   * svs::index::vamana::SearchBuffer           index/vamana/search_buffer.h:104-497
//   * svs::index::vamana::VisitedFilter          index/vamana/filter.h:49-130
//   * svs::simd::generic_simd_op, ConvertToFloat<16>, ConvertForVNNI<int16_t,32>, create_mask
//                                                core/distance/simd_utils.h:204-456
//   * svs::Float16                               lib/float16.h
//   * svs::index::flat::BulkInserter / HeapInserter (what FlatIndex::search keeps its k best in)   index/flat/inserters.h:76-235
//   * svs::io::vecs::get_dims, VecsReader, VecsWriter ([fbi]vecs files)                            core/io/vecs.h:97-229
//
// What is NOT buildable in this image and therefore restated below in this file (third-party toml++ /
// robin-map headers are absent; core/distance/{euclidean,inner_product,cosine}.h include lib/saveload.h ->
// toml++):
//   * the three 16-lane accumulate/combine/reduce policies plugged into generic_simd_op
//     (L2FloatOp<16> euclidean.h:240-259, IPFloatOp<16> inner_product.h:199-216,
//      CosineFloatOp<16> cosine.h:224-256) and the VNNI policies (euclidean.h:265-313,
//      inner_product.h:222-267, cosine.h:261-326).  They are a handful of intrinsics each.
//   * dataset / graph containers (trivial row views satisfying concepts/data.h, concepts/graph.h).
//   * the batch loop of VamanaIndex::search (index/vamana/index.h:564-611): static partition over threads.
//   * the loop of FlatIndex::search (index/flat/flat.h:322-460: data indices ascending, every query's distance inserted into ITS heap,
//     cleanup = sort_heap, copy out) around the real BulkInserter -- flat.h itself includes core/data.h -> toml++.
//
#include "svs/concepts/data.h"
#include "svs/concepts/distance.h"
#include "svs/concepts/graph.h"
#include "svs/core/distance/simd_utils.h"
#include "svs/core/io/vecs.h"
#include "svs/index/flat/inserters.h"
#include "svs/index/vamana/filter.h"
#include "svs/index/vamana/greedy_search.h"
#include "svs/index/vamana/search_buffer.h"
#include "svs/lib/float16.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <limits>
#include <span>
#include <thread>
#include <unordered_set>
#include <vector>

namespace {

using svs::Float16;
namespace vamana = svs::index::vamana;

enum { F32 = 0, F16 = 1, I8 = 2, U8 = 3 };
enum { L2 = 0, IP = 1, COS = 2 };

// ---- accumulate policies for the reference's generic_simd_op -------------------------------------------
struct SqDiffPolicy : svs::simd::ConvertToFloat<16> {
    using mask_t = svs::simd::ConvertToFloat<16>::mask_t;
    static __m512 init() { return _mm512_setzero_ps(); }
    static __m512 accumulate(__m512 acc, __m512 x, __m512 y) {
        __m512 diff = _mm512_sub_ps(x, y);
        return _mm512_fmadd_ps(diff, diff, acc);
    }
    static __m512 accumulate(mask_t m, __m512 acc, __m512 x, __m512 y) {
        __m512 diff = _mm512_maskz_sub_ps(m, x, y);
        return _mm512_mask3_fmadd_ps(diff, diff, acc, m);
    }
    static __m512 combine(__m512 x, __m512 y) { return _mm512_add_ps(x, y); }
    static float reduce(__m512 x) { return _mm512_reduce_add_ps(x); }
};

struct DotPolicy : svs::simd::ConvertToFloat<16> {
    using mask_t = svs::simd::ConvertToFloat<16>::mask_t;
    static __m512 init() { return _mm512_setzero_ps(); }
    static __m512 accumulate(__m512 acc, __m512 x, __m512 y) { return _mm512_fmadd_ps(x, y, acc); }
    static __m512 accumulate(mask_t m, __m512 acc, __m512 x, __m512 y) {
        return _mm512_mask3_fmadd_ps(x, y, acc, m);
    }
    static __m512 combine(__m512 x, __m512 y) { return _mm512_add_ps(x, y); }
    static float reduce(__m512 x) { return _mm512_reduce_add_ps(x); }
};

struct DotNormPolicy : svs::simd::ConvertToFloat<16> {
    using mask_t = svs::simd::ConvertToFloat<16>::mask_t;
    struct Acc {
        __m512 dot;
        __m512 nrm;
    };
    static Acc init() { return {_mm512_setzero_ps(), _mm512_setzero_ps()}; }
    static Acc accumulate(Acc acc, __m512 x, __m512 y) {
        return {_mm512_fmadd_ps(x, y, acc.dot), _mm512_fmadd_ps(y, y, acc.nrm)};
    }
    static Acc accumulate(mask_t m, Acc acc, __m512 x, __m512 y) {
        return {_mm512_mask3_fmadd_ps(x, y, acc.dot, m), _mm512_mask3_fmadd_ps(y, y, acc.nrm, m)};
    }
    static Acc combine(Acc x, Acc y) { return {_mm512_add_ps(x.dot, y.dot), _mm512_add_ps(x.nrm, y.nrm)}; }
    static std::pair<float, float> reduce(Acc x) {
        return {_mm512_reduce_add_ps(x.dot), _mm512_reduce_add_ps(x.nrm)};
    }
};

struct SqDiffVnni : svs::simd::ConvertForVNNI<int16_t, 32> {
    using base = svs::simd::ConvertForVNNI<int16_t, 32>;
    using reg_t = base::reg_t;
    using mask_t = base::mask_t;
    static reg_t init() { return _mm512_setzero_si512(); }
    static reg_t accumulate(reg_t acc, reg_t x, reg_t y) {
        reg_t diff = _mm512_sub_epi16(x, y);
        return _mm512_dpwssd_epi32(acc, diff, diff);
    }
    static reg_t accumulate(mask_t m, reg_t acc, reg_t x, reg_t y) {
        reg_t diff = _mm512_maskz_sub_epi16(m, x, y);
        return _mm512_mask_dpwssd_epi32(acc, m, diff, diff);
    }
    static reg_t combine(reg_t x, reg_t y) { return _mm512_add_epi32(x, y); }
    static float reduce(reg_t x) { return static_cast<float>(_mm512_reduce_add_epi32(x)); }
};

struct DotVnni : svs::simd::ConvertForVNNI<int16_t, 32> {
    using base = svs::simd::ConvertForVNNI<int16_t, 32>;
    using reg_t = base::reg_t;
    using mask_t = base::mask_t;
    static reg_t init() { return _mm512_setzero_si512(); }
    static reg_t accumulate(reg_t acc, reg_t x, reg_t y) { return _mm512_dpwssd_epi32(acc, x, y); }
    static reg_t accumulate(mask_t m, reg_t acc, reg_t x, reg_t y) {
        return _mm512_mask_dpwssd_epi32(acc, m, x, y);
    }
    static reg_t combine(reg_t x, reg_t y) { return _mm512_add_epi32(x, y); }
    static float reduce(reg_t x) { return static_cast<float>(_mm512_reduce_add_epi32(x)); }
};

template <typename T> struct is_small_int : std::false_type {};
template <> struct is_small_int<int8_t> : std::true_type {};
template <> struct is_small_int<uint8_t> : std::true_type {};

// cosine.h:261-326 -- 32 bytes per step, masked last step, two i32 accumulators.
template <typename T> float cosine_small_int(const T* a, const T* b, float a_norm, size_t n) {
    auto length = svs::lib::MaybeStatic(n);
    __m512i dot = _mm512_setzero_epi32();
    __m512i nrm = _mm512_setzero_epi32();
    auto tail = svs::create_mask<32>(length);
    auto full = svs::no_mask<32>();
    for (size_t j = 0; j < n; j += 32) {
        bool last = j + 32 >= n;
        auto ra = _mm256_maskz_loadu_epi8(last ? tail : full, a + j);
        auto rb = _mm256_maskz_loadu_epi8(last ? tail : full, b + j);
        __m512i wa, wb;
        if constexpr (std::is_signed_v<T>) {
            wa = _mm512_cvtepi8_epi16(ra);
            wb = _mm512_cvtepi8_epi16(rb);
        } else {
            wa = _mm512_cvtepu8_epi16(ra);
            wb = _mm512_cvtepu8_epi16(rb);
        }
        nrm = _mm512_dpwssd_epi32(nrm, wb, wb);
        dot = _mm512_dpwssd_epi32(dot, wa, wb);
    }
    float b_norm = std::sqrt(static_cast<float>(_mm512_reduce_add_epi32(nrm)));
    return static_cast<float>(_mm512_reduce_add_epi32(dot)) / (a_norm * b_norm);
}

template <typename A, typename B> constexpr bool float_simd_pair() {
    if constexpr (std::is_same_v<A, float>) return true;
    if constexpr (std::is_same_v<A, Float16> && (std::is_same_v<B, float> || std::is_same_v<B, Float16>)) return true;
    return false;
}

template <typename A, typename B> float kernel(int metric, const A* a, const B* b, size_t n, float a_norm) {
    auto len = svs::lib::MaybeStatic(n);
    if constexpr (std::is_same_v<A, B> && is_small_int<A>::value) {
        if (metric == L2) return svs::simd::generic_simd_op(SqDiffVnni{}, a, b, len);
        if (metric == IP) return svs::simd::generic_simd_op(DotVnni{}, a, b, len);
        return cosine_small_int(a, b, a_norm, n);
    } else if constexpr (float_simd_pair<A, B>()) {
        if (metric == L2) return svs::simd::generic_simd_op(SqDiffPolicy{}, a, b, len);
        if (metric == IP) return svs::simd::generic_simd_op(DotPolicy{}, a, b, len);
        auto [dot, nrm] = svs::simd::generic_simd_op(DotNormPolicy{}, a, b, len);
        return dot / (std::sqrt(nrm) * a_norm);
    } else {
        return std::numeric_limits<float>::quiet_NaN();
    }
}

// distance_core.h:45-66 restated (separately rounded multiply / add; product in the promoted element type).
template <typename T> __attribute__((optimize("fp-contract=off"))) float query_norm(const T* x, size_t n) {
    float accum = 0;
    for (size_t i = 0; i < n; ++i) {
        if constexpr (is_small_int<T>::value) {
            int v = x[i];
            accum += static_cast<float>(v * v);
        } else {
            float v = static_cast<float>(x[i]);
            float p = v * v;
            accum += p;
        }
    }
    return std::sqrt(accum);
}

template <typename F> auto with_type(int dtype, F&& f) {
    switch (dtype) {
        case F32: return f(static_cast<const float*>(nullptr));
        case F16: return f(static_cast<const Float16*>(nullptr));
        case I8: return f(static_cast<const int8_t*>(nullptr));
        default: return f(static_cast<const uint8_t*>(nullptr));
    }
}

// ---- containers satisfying the reference concepts -------------------------------------------------------
template <typename T> struct RowData {
    using value_type = std::span<const T>;
    using const_value_type = std::span<const T>;
    const T* base;
    size_t n, d;
    size_t size() const { return n; }
    size_t dimensions() const { return d; }
    const_value_type get_datum(size_t i) const { return {base + i * d, d}; }
    void prefetch(size_t i) const { svs::lib::prefetch(get_datum(i)); }
};

struct RowGraph {
    using index_type = uint32_t;
    using reference = std::span<const uint32_t>;
    using const_reference = std::span<const uint32_t>;
    const uint32_t* base;
    size_t n, stride;
    size_t max_degree() const { return stride - 1; }
    size_t n_nodes() const { return n; }
    const_reference get_node(uint32_t i) const {
        const uint32_t* row = base + size_t(i) * stride;
        return {row + 1, row[0]};
    }
    size_t get_node_degree(uint32_t i) const { return base[size_t(i) * stride]; }
    void prefetch_node(uint32_t i) const { svs::lib::prefetch_l0(base + size_t(i) * stride); }
};

template <typename Q, typename T, typename Cmp> struct Functor {
    using compare = Cmp;
    static constexpr bool must_fix_argument = true;
    int metric;
    float norm = 0;
    void fix_argument(std::span<const Q> q) {
        if (metric == COS) norm = query_norm(q.data(), q.size());
    }
    float compute(std::span<const Q> q, std::span<const T> y) const {
        return kernel(metric, q.data(), y.data(), q.size(), norm);
    }
};

struct CountingTracker {
    uint64_t expansions = 0, slots = 0;
    template <typename I> void visited(svs::Neighbor<I>, size_t n) {
        // EntryPointInitializer reports n=1 per entry point (greedy_search.h:85); expansions report deg.
        ++calls;
        last = n;
        slots += n;
    }
    uint64_t calls = 0;
    size_t last = 0;
};

template <typename Q, typename T, typename Cmp>
int search_typed(
    int metric, const T* data, size_t n, size_t d, const uint32_t* graph, size_t stride, const uint32_t* entry,
    size_t n_entry, const Q* queries, size_t nq, size_t k, size_t window, size_t capacity, int visited_set,
    int nthreads, uint32_t* ids, float* dists, uint64_t* stats
) {
    RowData<T> ds{data, n, d};
    RowGraph g{graph, n, stride};
    if (nthreads < 1) nthreads = 1;
    size_t nt = std::min<size_t>(size_t(nthreads), nq == 0 ? 1 : nq);
    auto work = [&](size_t lo, size_t hi) {
        auto buffer = vamana::SearchBuffer<uint32_t, Cmp>{vamana::SearchBufferConfig(window, capacity), Cmp{}, visited_set != 0};
        if (buffer.capacity() < k) buffer.change_maxsize(vamana::SearchBufferConfig{k});
        Functor<Q, T, Cmp> fn{metric};
        auto accessor = svs::data::GetDatumAccessor{};
        for (size_t i = lo; i < hi; ++i) {
            std::span<const Q> q{queries + i * d, d};
            CountingTracker tr;
            vamana::greedy_search(
                g, ds, accessor, q, fn, buffer,
                vamana::EntryPointInitializer<uint32_t>{std::span<const uint32_t>(entry, n_entry)},
                vamana::NeighborBuilder{}, tr
            );
            for (size_t j = 0; j < k; ++j) {
                if (j < buffer.size()) {
                    ids[i * k + j] = buffer[j].id();
                    dists[i * k + j] = buffer[j].distance();
                } else {
                    ids[i * k + j] = 0xFFFFFFFFu;
                    dists[i * k + j] = std::numeric_limits<float>::infinity();
                }
            }
            if (stats) {
                stats[2 * i + 0] = tr.calls - n_entry;
                stats[2 * i + 1] = tr.slots - n_entry;
            }
        }
    };
    if (nt <= 1) {
        work(0, nq);
        return 0;
    }
    // StaticPartition (lib/threads/types.h:341-380): contiguous balanced chunks
    std::vector<std::thread> pool;
    size_t base = nq / nt, extra = nq % nt, lo = 0;
    for (size_t t = 0; t < nt; ++t) {
        size_t len = base + (t < extra ? 1 : 0);
        pool.emplace_back(work, lo, lo + len);
        lo += len;
    }
    for (auto& th : pool) th.join();
    return 0;
}

// BatchIterator (index/vamana/iterator.h:81-339) for a static index, on the REAL SearchBuffer / greedy_search /
// EntryPointInitializer.  iterator.h itself pulls in index.h and dynamic_index.h (unbuildable here), so its ~40 lines of
// control flow are restated: RestartInitializer (:36-63), initialize_buffer (:127-131), increment_buffer (:135-139),
// next (:249-317), copy_from_scratch (:104-124), done (:230).
struct RestartInit {
    std::span<const uint32_t> entry_points;
    bool hard_restart;
    template <typename Buffer, typename Computer, typename Graph, typename Builder, typename Tracker>
    void operator()(Buffer& buffer, const Computer& computer, const Graph& graph, const Builder& builder, Tracker& tracker) const {
        if (hard_restart) {
            vamana::EntryPointInitializer<uint32_t>{entry_points}(buffer, computer, graph, builder, tracker);
            return;
        }
        buffer.sort();
    }
};

template <typename Q, typename T, typename Cmp>
int iterate_typed(
    int metric, const T* data, size_t n, size_t d, const uint32_t* graph, size_t stride, const uint32_t* entry,
    size_t n_entry, const Q* query, size_t extra, const size_t* batch_sizes, const uint8_t* restart_before,
    size_t n_batches, uint32_t* ids, float* dists, size_t* counts, uint8_t* done_after
) {
    RowData<T> ds{data, n, d};
    RowGraph g{graph, n, stride};
    if (extra == std::numeric_limits<size_t>::max()) extra = 100;
    auto buffer = vamana::SearchBuffer<uint32_t, Cmp>{vamana::SearchBufferConfig(0, extra), Cmp{}, false};
    buffer.clear();
    std::unordered_set<uint32_t> yielded;
    bool restart = true, exhausted = false;
    Functor<Q, T, Cmp> fn{metric};
    auto accessor = svs::data::GetDatumAccessor{};
    std::span<const Q> q{query, d};
    size_t out = 0;
    for (size_t b = 0; b < n_batches; ++b) {
        const size_t bs = batch_sizes[b];
        if (restart_before && restart_before[b]) restart = true; // restart_next_search()
        size_t got = 0;
        if (!(yielded.size() == n || exhausted)) {
            auto config = buffer.config();
            config.increment(bs);
            buffer.change_maxsize(config);
            bool restart_copy = std::exchange(restart, true);
            vamana::NullTracker tracker{};
            vamana::greedy_search(
                g, ds, accessor, q, fn, buffer, RestartInit{std::span<const uint32_t>(entry, n_entry), restart_copy},
                vamana::NeighborBuilder{}, tracker
            );
            restart = false;
            for (size_t i = 0, imax = buffer.size(); i < imax; ++i) {
                auto neighbor = buffer[i];
                if (yielded.insert(neighbor.id()).second) {
                    ids[out + got] = neighbor.id();
                    dists[out + got] = neighbor.distance();
                    ++got;
                }
                if (got == bs) break;
            }
            if (got == 0 && bs > 0) exhausted = true;
        }
        counts[b] = got;
        out += bs;
        done_after[b] = (yielded.size() == n || exhausted) ? 1 : 0;
    }
    return 0;
}

} // namespace

extern "C" {

// One query driven through a list of next(batch_size) calls.  ids/dists: sum(batch_sizes) slots, batch b's results at
// offset sum(batch_sizes[:b]); counts[b] = results of batch b; done_after[b] = done() after that call.
int svsref_iterate(
    int metric, int data_dtype, const void* data, size_t n, size_t d, const uint32_t* graph, size_t graph_stride,
    const uint32_t* entry, size_t n_entry, const void* query, int q_dtype, size_t extra, const size_t* batch_sizes,
    const uint8_t* restart_before, size_t n_batches, uint32_t* ids, float* dists, size_t* counts, uint8_t* done_after
) {
    try {
        return with_type(q_dtype, [&](auto* tq) {
            using Q = std::remove_cv_t<std::remove_pointer_t<decltype(tq)>>;
            return with_type(data_dtype, [&](auto* tt) {
                using T = std::remove_cv_t<std::remove_pointer_t<decltype(tt)>>;
                constexpr bool ok = (std::is_same_v<Q, T> && is_small_int<Q>::value) || float_simd_pair<Q, T>();
                if constexpr (!ok) {
                    return -3;
                } else {
                    auto run = [&](auto cmp) {
                        return iterate_typed<Q, T, decltype(cmp)>(
                            metric, static_cast<const T*>(data), n, d, graph, graph_stride, entry, n_entry,
                            static_cast<const Q*>(query), extra, batch_sizes, restart_before, n_batches, ids, dists, counts,
                            done_after
                        );
                    };
                    return metric == L2 ? run(std::less<>{}) : run(std::greater<>{});
                }
            });
        });
    } catch (...) { return -9; }
}

int svsref_has_avx512() {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512vnni");
}

float svsref_norm(const void* x, int dtype, size_t d) {
    return with_type(dtype, [&](auto* tag) {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(tag)>>;
        return query_norm(static_cast<const T*>(x), d);
    });
}

float svsref_distance(int metric, const void* a, int adt, const void* b, int bdt, size_t d, float a_norm) {
    return with_type(adt, [&](auto* ta) {
        using A = std::remove_cv_t<std::remove_pointer_t<decltype(ta)>>;
        return with_type(bdt, [&](auto* tb) {
            using B = std::remove_cv_t<std::remove_pointer_t<decltype(tb)>>;
            return kernel(metric, static_cast<const A*>(a), static_cast<const B*>(b), d, a_norm);
        });
    });
}

int svsref_distance_batch(int metric, const void* a, int adt, size_t na, const void* b, int bdt, size_t nb, size_t d, float* out) {
    size_t sa = adt == F32 ? 4 : adt == F16 ? 2 : 1, sb = bdt == F32 ? 4 : bdt == F16 ? 2 : 1;
    for (size_t i = 0; i < na; ++i) {
        const char* ai = static_cast<const char*>(a) + i * d * sa;
        float an = metric == COS ? svsref_norm(ai, adt, d) : 0.0f;
        for (size_t j = 0; j < nb; ++j) {
            out[i * nb + j] = svsref_distance(metric, ai, adt, static_cast<const char*>(b) + j * d * sb, bdt, d, an);
        }
    }
    return 0;
}

// FlatIndex::search (index/flat/flat.h:322-460) with the reference's own sorter_type = BulkInserter<Neighbor<size_t>, compare> (flat.h:173):
// data indices ascending (search_patch :437-457), one HeapInserter per query (std::pop_heap / push_heap, a candidate EQUAL to the heap's
// top still replaces it), scratch.cleanup() = std::sort_heap, results copied out in that order.  What ties do is therefore whatever
// libstdc++'s heap leaves -- this function is the pin for it.
extern "C++" {
template <typename Cmp>
int flat_typed(int metric, const void* data, int ddt, size_t n, const void* queries, int qdt, size_t nq, size_t d, size_t k, uint64_t* ids, float* dists) {
    using N = svs::Neighbor<size_t>;
    svs::index::flat::BulkInserter<N, Cmp> scratch{nq, k, Cmp{}};
    scratch.prepare();
    const size_t sq = qdt == F32 ? 4 : qdt == F16 ? 2 : 1, sd = ddt == F32 ? 4 : ddt == F16 ? 2 : 1;
    std::vector<float> norms(nq, 0.0f);
    for (size_t i = 0; i < nq; ++i)
        if (metric == COS) norms[i] = svsref_norm(static_cast<const char*>(queries) + i * d * sq, qdt, d);
    for (size_t j = 0; j < n; ++j) {
        const char* row = static_cast<const char*>(data) + j * d * sd;
        for (size_t i = 0; i < nq; ++i) {
            const float dist = svsref_distance(metric, static_cast<const char*>(queries) + i * d * sq, qdt, row, ddt, d, norms[i]);
            scratch.insert(i, N{j, dist});
        }
    }
    scratch.cleanup();
    for (size_t i = 0; i < nq; ++i) {
        const auto r = scratch.result(i);
        for (size_t j = 0; j < k; ++j) {
            ids[i * k + j] = r[j].id();
            dists[i * k + j] = r[j].distance();
        }
    }
    return 0;
}
} // extern "C++"

int svsref_flat_search(int metric, const void* data, int ddt, size_t n, const void* queries, int qdt, size_t nq, size_t d, size_t k, uint64_t* ids, float* dists) {
    if (metric == L2) return flat_typed<std::less<>>(metric, data, ddt, n, queries, qdt, nq, d, k, ids, dists);
    return flat_typed<std::greater<>>(metric, data, ddt, n, queries, qdt, nq, d, k, ids, dists);
}

// core/io/vecs.h: shape of a [fbi]vecs file (get_dims :97-124 with its element-size deduction) and its contents through VecsReader<T>
// (:140-190); returns 0, or -1 with nothing written when the reference throws.  VecsWriter<T> (:196-222) for the other direction.
int svsref_vecs_dims(const char* path, size_t elsize_hint, size_t* nvec, size_t* dims) {
    try {
        const auto nd = svs::io::vecs::get_dims(std::filesystem::path(path), elsize_hint);
        *nvec = nd.first;
        *dims = nd.second;
        return 0;
    } catch (const std::exception&) { return -1; }
}

extern "C++" {
template <typename T> int vecs_read_typed(const char* path, void* out, size_t capacity_elems) {
    try {
        svs::io::vecs::VecsReader<T> reader{std::string(path)};
        T* dst = static_cast<T*>(out);
        size_t at = 0;
        for (auto it = reader.begin(); it != reader.end(); ++it) {
            const auto v = *it;
            if (at + v.size() > capacity_elems) return -2;
            std::copy(v.begin(), v.end(), dst + at);
            at += v.size();
        }
        return 0;
    } catch (const std::exception&) { return -1; }
}
} // extern "C++"

int svsref_vecs_read(const char* path, int elsize, void* out, size_t capacity_elems) {
    if (elsize == 4) return vecs_read_typed<uint32_t>(path, out, capacity_elems); // fvecs / ivecs rows move as 32-bit words
    if (elsize == 1) return vecs_read_typed<uint8_t>(path, out, capacity_elems);
    return -3;
}

int svsref_vecs_write_u32(const char* path, const uint32_t* rows, size_t n, size_t d) {
    try {
        svs::io::vecs::VecsWriter<uint32_t> w{std::string(path), d};
        for (size_t i = 0; i < n; ++i) w.append(std::span<const uint32_t>(rows + i * d, d));
        w.flush();
        return 0;
    } catch (const std::exception&) { return -1; }
}

// Batch search through the reference's greedy_search/SearchBuffer.  stats (optional): 2 u64 per query
// {expansions, neighbor_slots}.
int svsref_search_batch(
    int metric, int data_dtype, const void* data, size_t n, size_t d, const uint32_t* graph, size_t graph_stride,
    const uint32_t* entry, size_t n_entry, const void* queries, int q_dtype, size_t nq, size_t k, size_t window,
    size_t capacity, int visited_set, int nthreads, uint32_t* ids, float* dists, uint64_t* stats
) {
    if (window > capacity) return -1;
    try {
        return with_type(q_dtype, [&](auto* tq) {
            using Q = std::remove_cv_t<std::remove_pointer_t<decltype(tq)>>;
            return with_type(data_dtype, [&](auto* tt) {
                using T = std::remove_cv_t<std::remove_pointer_t<decltype(tt)>>;
                constexpr bool ok = (std::is_same_v<Q, T> && is_small_int<Q>::value) || float_simd_pair<Q, T>();
                if constexpr (!ok) {
                    return -3;
                } else {
                    auto run = [&](auto cmp) {
                        return search_typed<Q, T, decltype(cmp)>(
                            metric, static_cast<const T*>(data), n, d, graph, graph_stride, entry, n_entry,
                            static_cast<const Q*>(queries), nq, k, window, capacity, visited_set, nthreads, ids,
                            dists, stats
                        );
                    };
                    return metric == L2 ? run(std::less<>{}) : run(std::greater<>{});
                }
            });
        });
    } catch (...) { return -9; }
}

// ---- SearchBuffer op-by-op access for trace tests --------------------------------------------------------
struct RefBuffer {
    bool greater;
    vamana::SearchBuffer<uint32_t, std::less<>> lt;
    vamana::SearchBuffer<uint32_t, std::greater<>> gt;
};

void* svsref_buffer_new(size_t window, size_t capacity, int greater, int visited_set) {
    try {
        auto* b = new RefBuffer{greater != 0, {}, {}};
        auto cfg = vamana::SearchBufferConfig(window, capacity);
        if (greater) b->gt = vamana::SearchBuffer<uint32_t, std::greater<>>(cfg, {}, visited_set != 0);
        else b->lt = vamana::SearchBuffer<uint32_t, std::less<>>(cfg, {}, visited_set != 0);
        return b;
    } catch (...) { return nullptr; }
}
void svsref_buffer_free(void* p) { delete static_cast<RefBuffer*>(p); }

#define REFBUF_DISPATCH(expr_lt, expr_gt) (b->greater ? (expr_gt) : (expr_lt))
void svsref_buffer_clear(void* p) {
    auto* b = static_cast<RefBuffer*>(p);
    if (b->greater) b->gt.clear(); else b->lt.clear();
}
void svsref_buffer_push_back(void* p, uint32_t id, float d) {
    auto* b = static_cast<RefBuffer*>(p);
    if (b->greater) b->gt.push_back({id, d}); else b->lt.push_back({id, d});
}
void svsref_buffer_sort(void* p) {
    auto* b = static_cast<RefBuffer*>(p);
    if (b->greater) b->gt.sort(); else b->lt.sort();
}
size_t svsref_buffer_insert(void* p, uint32_t id, float d) {
    auto* b = static_cast<RefBuffer*>(p);
    return REFBUF_DISPATCH(b->lt.insert({id, d}), b->gt.insert({id, d}));
}
int svsref_buffer_done(void* p) {
    auto* b = static_cast<RefBuffer*>(p);
    return REFBUF_DISPATCH(b->lt.done(), b->gt.done());
}
uint32_t svsref_buffer_next(void* p, float* dist) {
    auto* b = static_cast<RefBuffer*>(p);
    if (b->greater) {
        const auto& n = b->gt.next();
        *dist = n.distance();
        return n.id();
    }
    const auto& n = b->lt.next();
    *dist = n.distance();
    return n.id();
}
size_t svsref_buffer_size(void* p) {
    auto* b = static_cast<RefBuffer*>(p);
    return REFBUF_DISPATCH(b->lt.size(), b->gt.size());
}
size_t svsref_buffer_best_unvisited(void* p) {
    auto* b = static_cast<RefBuffer*>(p);
    return REFBUF_DISPATCH(b->lt.best_unvisited(), b->gt.best_unvisited());
}
void svsref_buffer_get(void* p, size_t i, uint32_t* id, float* dist, int* visited) {
    auto* b = static_cast<RefBuffer*>(p);
    if (b->greater) {
        *id = b->gt[i].id(); *dist = b->gt[i].distance(); *visited = b->gt[i].visited();
    } else {
        *id = b->lt[i].id(); *dist = b->lt[i].distance(); *visited = b->lt[i].visited();
    }
}
int svsref_buffer_emplace_visited(void* p, uint32_t id) {
    auto* b = static_cast<RefBuffer*>(p);
    return REFBUF_DISPATCH(b->lt.emplace_visited(id), b->gt.emplace_visited(id));
}

// ---- VisitedFilter<uint32_t,16> ---------------------------------------------------------------------------
void* svsref_filter_new() { return new vamana::VisitedFilter<uint32_t, 16>(); }
void svsref_filter_free(void* f) { delete static_cast<vamana::VisitedFilter<uint32_t, 16>*>(f); }
int svsref_filter_emplace(void* f, uint32_t k) { return static_cast<vamana::VisitedFilter<uint32_t, 16>*>(f)->emplace(k); }
int svsref_filter_contains(void* f, uint32_t k) { return static_cast<vamana::VisitedFilter<uint32_t, 16>*>(f)->contains(k); }
void svsref_filter_reset(void* f) { static_cast<vamana::VisitedFilter<uint32_t, 16>*>(f)->reset(); }

// lib/float16.h scalar conversions (for documentation of item 7 in SURVEY appendix B)
float svsref_half_to_float_hw(uint16_t h) {
    return _mm_cvtss_f32(_mm_cvtph_ps(_mm_cvtsi32_si128(h)));
}

} // extern "C"
