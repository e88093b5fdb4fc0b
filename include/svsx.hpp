// svsx.hpp -- header-only C++ facade over the C ABI (svsx.h), shaped like the reference operator for this path.
//
// Mirrors (paths relative to the reference root):
//   svs::index::vamana::VamanaIndex::search(QueryResultView<I>, queries, VamanaSearchParameters, cancel)
//                                                                          index/vamana/index.h:564-611
//   svs::index::vamana::VamanaSearchParameters / SearchBufferConfig        search_params.h:28-61, search_buffer.h:39-96
//   svs::QueryResult<size_t>                                               core/query_result.h:53-172
//   svs::distance::DistanceL2 / DistanceIP / DistanceCosineSimilarity      core/distance/*.h (compare, implicit_broadcast)
//   svs::ANNException                                                      lib/exception.h
//   svs::index::vamana::BatchIterator (one per query of a batch)           index/vamana/iterator.h:81-339
//   VamanaIndex::calibrate (search-buffer part), file readers, recall      svsx_io.hpp (index/vamana/calibrate.h, core/io/*.h, core/recall.h)
// so that a caller written against the reference reads the same:  index.search(result.view(), queries, params).
#pragma once

#include "svsx.h"
#include "svsx_io.hpp"

#include <array>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <functional>
#include <limits>
#include <span>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace svsx {

struct ANNException : std::runtime_error {
    int code;
    ANNException(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc != SVSX_SUCCESS) throw ANNException(rc, svsx_last_error());
}

struct Float16 { uint16_t bits; };
template <typename T> constexpr int dtype_of();
template <> constexpr int dtype_of<float>() { return SVSX_F32; }
template <> constexpr int dtype_of<Float16>() { return SVSX_F16; }
template <> constexpr int dtype_of<int8_t>() { return SVSX_I8; }
template <> constexpr int dtype_of<uint8_t>() { return SVSX_U8; }

// distance functors: same static surface as the reference's (compare, implicit_broadcast, name)
struct DistanceL2 { using compare = std::less<>; static constexpr bool implicit_broadcast = true; static constexpr int metric = SVSX_L2; static constexpr const char* name = "squared_l2"; };
struct DistanceIP { using compare = std::greater<>; static constexpr bool implicit_broadcast = true; static constexpr int metric = SVSX_IP; static constexpr const char* name = "inner_product"; };
struct DistanceCosineSimilarity { using compare = std::greater<>; static constexpr bool implicit_broadcast = false; static constexpr bool must_fix_argument = true; static constexpr int metric = SVSX_COSINE; static constexpr const char* name = "cosine_similarity"; };

// distance::compute for every pair (a_i, b_j); out is na x nb
template <typename Dist, typename Ea, typename Eb>
void compute(Dist, const Ea* a, size_t na, const Eb* b, size_t nb, size_t d, float* out, int device = 0) {
    check(svsx_distance_batch(Dist::metric, dtype_of<Ea>(), dtype_of<Eb>(), d, a, na, b, nb, out, device));
}

class SearchBufferConfig {
    size_t window_ = 0, capacity_ = 0;
  public:
    SearchBufferConfig() = default;
    SearchBufferConfig(size_t window) : window_(window), capacity_(window) {}
    SearchBufferConfig(size_t window, size_t capacity) : window_(window), capacity_(capacity) {
        if (window > capacity)
            throw ANNException(SVSX_INVALID_ARGUMENT, "Improper configuration for search buffer! Effective size (" + std::to_string(window) +
                                                          ") cannot be less than capacity (" + std::to_string(capacity) + ").");
    }
    size_t get_search_window_size() const { return window_; }
    size_t get_total_capacity() const { return capacity_; }
};

struct VamanaSearchParameters {
    SearchBufferConfig buffer_config_{};
    bool search_buffer_visited_set_ = false;
    size_t prefetch_lookahead_ = 4;
    size_t prefetch_step_ = 1;
    VamanaSearchParameters& buffer_config(SearchBufferConfig c) { buffer_config_ = c; return *this; }
    VamanaSearchParameters& search_buffer_visited_set(bool v) { search_buffer_visited_set_ = v; return *this; }
};

template <typename Idx = size_t> struct QueryResultView {
    Idx* indices; float* distances; size_t n_queries_, n_neighbors_;
    size_t n_queries() const { return n_queries_; }
    size_t n_neighbors() const { return n_neighbors_; }
};

template <typename Idx = size_t> class QueryResult {
    std::vector<Idx> ids_; std::vector<float> dists_; size_t nq_, k_;
  public:
    QueryResult(size_t nq, size_t k) : ids_(nq * k), dists_(nq * k), nq_(nq), k_(k) {}
    QueryResultView<Idx> view() { return {ids_.data(), dists_.data(), nq_, k_}; }
    Idx index(size_t i, size_t j) const { return ids_[i * k_ + j]; }
    float distance(size_t i, size_t j) const { return dists_[i * k_ + j]; }
    size_t n_queries() const { return nq_; }
    size_t n_neighbors() const { return k_; }
};

// Read-only Vamana index resident in one GPU's HBM.  Graph = SVS rows [deg, ids...] (core/graph/graph.h:32-48).
// svs::Flat (include/svs/orchestrators/exhaustive.h; index/flat/flat.h:322-460): exhaustive search over a data set resident in HBM.  Exact by
// proof (svsx.h): float16 MFMA shortlist, float32 re-rank in the reference's operation order, per-query certificate, exhaustive pass for the rest.
template <typename T, typename Dist> class FlatIndex {
    svsx_index* h_ = nullptr;
  public:
    FlatIndex(const T* data, size_t n, size_t d, Dist = {}, int device = 0) {
        svsx_index_desc desc{};
        desc.metric = Dist::metric; desc.data_dtype = dtype_of<T>(); desc.n = n; desc.d = d; desc.data = data; desc.device = device;
        check(svsx_index_create(&desc, &h_)); // graph == NULL: a flat index
    }
    FlatIndex(const FlatIndex&) = delete;
    FlatIndex& operator=(const FlatIndex&) = delete;
    FlatIndex(FlatIndex&& o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    ~FlatIndex() { svsx_index_destroy(h_); }
    template <typename Q> void search(QueryResultView<size_t> result, const Q* queries) const {
        static_assert(sizeof(size_t) == sizeof(uint64_t));
        check(svsx_flat_search(h_, queries, dtype_of<Q>(), result.n_queries(), result.n_neighbors(), reinterpret_cast<uint64_t*>(result.indices), result.distances));
    }
    template <typename Q> QueryResult<size_t> search(const Q* queries, size_t nq, size_t k) const {
        QueryResult<size_t> r(nq, k);
        search(r.view(), queries);
        return r;
    }
    const svsx_index* handle() const { return h_; }
};

template <typename T, typename Dist> class VamanaIndex {
    svsx_index* h_ = nullptr;
    VamanaSearchParameters defaults_{};
  public:
    VamanaIndex(const uint32_t* graph, size_t graph_row_stride, const T* data, size_t n, size_t d, std::span<const uint32_t> entry_points,
                Dist = {}, int device = 0, uint64_t id_offset = 0) {
        svsx_index_desc desc{};
        desc.metric = Dist::metric; desc.data_dtype = dtype_of<T>(); desc.n = n; desc.d = d; desc.data = data;
        desc.graph = graph; desc.graph_row_stride_u32 = graph_row_stride; desc.max_degree = static_cast<uint32_t>(graph_row_stride - 1);
        desc.entry_points = entry_points.data(); desc.n_entry = entry_points.size(); desc.device = device; desc.id_offset = id_offset;
        check(svsx_index_create(&desc, &h_));
    }
    VamanaIndex(const VamanaIndex&) = delete;
    VamanaIndex& operator=(const VamanaIndex&) = delete;
    VamanaIndex(VamanaIndex&& o) noexcept : h_(o.h_), defaults_(o.defaults_) { o.h_ = nullptr; }
    ~VamanaIndex() { svsx_index_destroy(h_); }

    VamanaSearchParameters get_search_parameters() const { return defaults_; }
    void set_search_parameters(const VamanaSearchParameters& p) { defaults_ = p; }

    // auto_assemble (index/vamana/index.h:1017-1046) for what VamanaIndex::save wrote (:825-840): svs_config.toml + the native v1 blobs of
    // graph and dataset.  Like the reference's loader it restores the saved DEFAULT SEARCH PARAMETERS (index.h:141-170,318) next to the entry
    // point; the saved build parameters are returned through `config` for callers that want them.
    static VamanaIndex assemble(const std::string& config_toml, const std::string& graph_blob, const std::string& data_blob, Dist dist = {}, int device = 0,
                                io::IndexConfig* config = nullptr) {
        const io::IndexConfig cfg = io::read_index_config(config_toml);
        const auto graph = io::read_svs_native<uint32_t>(graph_blob);
        const auto data = io::read_svs_native<T>(data_blob);
        if (graph.n != data.n) throw ANNException(SVSX_INVALID_ARGUMENT, "Wrong sizes!"); // index.h:420-422
        const uint32_t entry[1] = {cfg.entry_point};
        VamanaIndex index(graph.values.data(), graph.d, data.values.data(), data.n, data.d, std::span<const uint32_t>(entry, 1), dist, device);
        if (cfg.has_search) {
            index.defaults_.buffer_config_ = SearchBufferConfig(cfg.search_window_size, cfg.search_buffer_capacity);
            index.defaults_.search_buffer_visited_set_ = cfg.search_buffer_visited_set;
            index.defaults_.prefetch_lookahead_ = cfg.prefetch_lookahead;
            index.defaults_.prefetch_step_ = cfg.prefetch_step;
        }
        if (config) *config = cfg;
        return index;
    }

    // index/vamana/index.h:564-611.  Without a predicate the batch is simply enqueued.  With one, `cancel` (thread-safe by the
    // reference's contract, index.h:541-545) is polled from a helper thread while the batch runs; once it returns true the
    // library raises the device word the kernel polls at every hop (greedy_search.h:155) and the call returns early.
    template <typename Q>
    void search(QueryResultView<size_t> result, const Q* queries, const VamanaSearchParameters& sp) const {
        svsx_search_params p{sp.buffer_config_.get_search_window_size(), sp.buffer_config_.get_total_capacity(),
                             sp.search_buffer_visited_set_ ? 1 : 0, sp.prefetch_lookahead_, sp.prefetch_step_};
        static_assert(sizeof(size_t) == sizeof(uint64_t));
        check(svsx_search(h_, queries, dtype_of<Q>(), result.n_queries(), result.n_neighbors(), &p,
                          reinterpret_cast<uint64_t*>(result.indices), nullptr, result.distances, nullptr));
    }
    template <typename Q>
    void search(QueryResultView<size_t> result, const Q* queries, const VamanaSearchParameters& sp, const std::function<bool()>& cancel) const {
        if (cancel()) return; // extensions.h:575-578: nothing starts
        svsx_search_params p{sp.buffer_config_.get_search_window_size(), sp.buffer_config_.get_total_capacity(),
                             sp.search_buffer_visited_set_ ? 1 : 0, sp.prefetch_lookahead_, sp.prefetch_step_};
        // the word the library polls is a plain volatile int (what the C ABI declares); the helper thread is its only writer, through
        // an atomic_ref, so no object of another type is passed off as one
        alignas(std::atomic_ref<int>::required_alignment) volatile int flag = 0;
        std::atomic<int> finished{0};
        svsx_search_aux aux{};
        aux.struct_size = sizeof(aux);
        aux.cancel_flag = &flag;
        std::thread poller([&] {
            while (!finished.load(std::memory_order_acquire)) {
                if (cancel()) { std::atomic_ref<int>(const_cast<int&>(flag)).store(1, std::memory_order_release); return; }
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        });
        const int rc = svsx_search(h_, queries, dtype_of<Q>(), result.n_queries(), result.n_neighbors(), &p,
                                   reinterpret_cast<uint64_t*>(result.indices), nullptr, result.distances, &aux);
        finished.store(1, std::memory_order_release);
        poller.join();
        check(rc);
    }
    template <typename Q> QueryResult<size_t> search(const Q* queries, size_t nq, size_t k) const {
        QueryResult<size_t> r(nq, k);
        search(r.view(), queries, defaults_);
        return r;
    }
    // optional, result-neutral (svsx.h): 2 = graph-inline bound records sized from the data's spectrum (the default and the fast path of
    // bench.py), 32 / 64 / 128 = records of that size, 1 = one 128-byte code row per vector (float rows, L2, d = 96 / 128), 0 = off
    /// Returns false when the scheme asked for did not fit the free HBM and a smaller one was built instead (SVSX_FELL_BACK: not an error;
    /// `memory()[8]` / `[3]` say which scheme and record size are in effect, svsx_last_error() why).
    bool build_bound_codes(int mode = 2) {
        const int rc = svsx_index_build_bound_codes(h_, mode);
        if (rc == SVSX_FELL_BACK) return false;
        check(rc);
        return true;
    }
    std::array<uint64_t, 10> memory() const { std::array<uint64_t, 10> m{}; check(svsx_index_memory(h_, m.data())); return m; }
    // scalar-quantized rows under cosine similarity keep one norm per row (svsx.h): recompute them in the order of `hip_stream` after rewriting
    // rows of a device view, or release them (searches then take the norm per candidate, as the reference does; same results)
    void refresh_row_norms(void* hip_stream = nullptr) { check(svsx_index_refresh_row_norms(h_, hip_stream)); }
    void drop_row_norms() { check(svsx_index_drop_row_norms(h_)); }
    // runtime VamanaIndex::search(n, x, k, distances, labels, params, IDFilter* filter) (bindings/cpp/include/svs/runtime/vamana_index.h:75-83):
    // `is_member` = one bit per id ((size() + 63) / 64 words); unfilled result slots hold id all-ones / distance +inf; filter_stop and
    // filter_estimate_batch are SearchParams' fields of the same names (api_defs.h; the runtime defaults: 0 and true)
    template <typename Q>
    void search(QueryResultView<size_t> result, const Q* queries, const VamanaSearchParameters& sp, std::span<const uint64_t> is_member,
                float filter_stop = 0.0f, bool filter_estimate_batch = true) const {
        svsx_search_params p{sp.buffer_config_.get_search_window_size(), sp.buffer_config_.get_total_capacity(),
                             sp.search_buffer_visited_set_ ? 1 : 0, sp.prefetch_lookahead_, sp.prefetch_step_};
        if (is_member.size() * 64 < size()) throw ANNException(SVSX_INVALID_ARGUMENT, "is_member must hold one bit per vector");
        check(svsx_search_filtered_ex(h_, queries, dtype_of<Q>(), result.n_queries(), result.n_neighbors(), &p, is_member.data(), filter_stop,
                                      filter_estimate_batch ? 1 : 0, reinterpret_cast<uint64_t*>(result.indices), result.distances));
    }
    // runtime VamanaIndex::range_search(n, x, radius, ResultsAllocator, params, IDFilter* filter) (runtime/vamana_index.h:85-92): FAISS-style
    // result -- lims[q] .. lims[q + 1] index labels / distances of query q.  `is_member`: empty = no filter, else one bit per id.
    struct RangeResult {
        std::vector<size_t> lims;
        std::vector<uint64_t> labels;
        std::vector<float> distances;
    };
    template <typename Q>
    RangeResult range_search(const Q* queries, size_t nq, float radius, const VamanaSearchParameters& sp, std::span<const uint64_t> is_member = {}) const {
        svsx_search_params p{sp.buffer_config_.get_search_window_size(), sp.buffer_config_.get_total_capacity(),
                             sp.search_buffer_visited_set_ ? 1 : 0, sp.prefetch_lookahead_, sp.prefetch_step_};
        if (!is_member.empty() && is_member.size() * 64 < size()) throw ANNException(SVSX_INVALID_ARGUMENT, "is_member must hold one bit per vector");
        RangeResult r;
        auto allocate = [](void* user, const size_t* counts, size_t n, svsx_range_storage* storage) {
            auto* out = static_cast<RangeResult*>(user);
            out->lims.assign(n + 1, 0);
            for (size_t i = 0; i < n; ++i) out->lims[i + 1] = out->lims[i] + counts[i];
            out->labels.resize(out->lims[n]);
            out->distances.resize(out->lims[n]);
            *storage = svsx_range_storage{out->labels.data(), out->distances.data()};
        };
        check(svsx_range_search(h_, queries, dtype_of<Q>(), nq, radius, &p, is_member.empty() ? nullptr : is_member.data(), allocate, &r));
        if (r.lims.empty()) r.lims.assign(nq + 1, 0);
        return r;
    }
    // VamanaIndex::get_distance(id, query) (index/vamana/index.h:915-933; runtime/vamana_index.h:94-96)
    template <typename Q> double get_distance(size_t id, const Q* query) const {
        double d = 0.0;
        check(svsx_get_distance(h_, id, query, dtype_of<Q>(), &d));
        return d;
    }
    const svsx_index* handle() const { return h_; }
    size_t size() const { uint64_t info[6]; check(svsx_index_info(h_, info)); return info[0]; }

    // VamanaIndex::calibrate for the search-buffer parameters (index/vamana/calibrate.h:249-343,478-559): tunes this index's default search
    // parameters so that recall@k against `groundtruth` (row stride gt_stride) reaches target_recall, and returns them.  Each probe is one
    // batch search on the GPU.
    template <typename Q, typename G>
    VamanaSearchParameters calibrate(const Q* queries, size_t nq, const G* groundtruth, size_t gt_stride, size_t k, double target_recall,
                                     const CalibrationParameters& cp = {}) {
        VamanaSearchParameters start = cp.use_existing_parameter_values_ ? defaults_ : VamanaSearchParameters{};
        if (cp.search_buffer_optimization_ == CalibrationParameters::SearchBufferOptimization::Disable) return start;
        QueryResult<size_t> result(nq, k);
        auto compute_recall = [&](size_t window, size_t capacity) {
            VamanaSearchParameters p = start;
            p.buffer_config_ = SearchBufferConfig(window, capacity);
            search(result.view(), queries, p);
            return k_recall_at_n(groundtruth, gt_stride, result.view().indices, k, nq, k, k);
        };
        const BufferChoice best = optimize_search_buffer(cp, start.buffer_config_.get_search_window_size(), start.buffer_config_.get_total_capacity(),
                                                         k, target_recall, compute_recall);
        start.buffer_config_ = SearchBufferConfig(best.window, best.capacity);
        defaults_ = start;
        return start;
    }
};

// svs::Neighbor<size_t> as yielded by the iterator (core/query_result.h / lib/neighbor.h)
struct Neighbor {
    size_t id_;
    float distance_;
    size_t id() const { return id_; }
    float distance() const { return distance_; }
};

// index/vamana/iterator.h:81-339 for a BATCH of queries: query q behaves exactly like the reference's
// BatchIterator{index, queries[q], extra_search_buffer_capacity}; next() advances all of them with one kernel launch.
template <typename Q> class BatchIterator {
    svsx_batch_iterator* h_ = nullptr;
    size_t nq_ = 0, n_ = 0, stride_ = 0;
    std::vector<uint64_t> ids_;
    std::vector<float> dists_;
    std::vector<uint32_t> counts_;
    std::vector<Neighbor> results_;
    mutable std::vector<uint32_t> meta_;
    void refresh() const { meta_.resize(nq_ * 8); check(svsx_iterator_state(h_, meta_.data())); }
  public:
    static constexpr uint32_t skip = 0xFFFFFFFFu;
    template <typename T, typename Dist>
    BatchIterator(const VamanaIndex<T, Dist>& parent, const Q* queries, size_t nq,
                  size_t extra_search_buffer_capacity = std::numeric_limits<size_t>::max())
        : nq_(nq), n_(parent.size()), counts_(nq, 0) {
        check(svsx_iterator_create(parent.handle(), queries, dtype_of<Q>(), nq, extra_search_buffer_capacity, &h_));
    }
    BatchIterator(const BatchIterator&) = delete;
    BatchIterator& operator=(const BatchIterator&) = delete;
    ~BatchIterator() { svsx_iterator_destroy(h_); }

    // BatchIterator::next (:249-317); per_query (optional): batch size of each query, `skip` leaves a query untouched
    void next(size_t batch_size, const uint32_t* per_query = nullptr) {
        stride_ = batch_size;
        ids_.resize(nq_ * batch_size);
        dists_.resize(nq_ * batch_size);
        check(svsx_iterator_next(h_, batch_size, per_query, ids_.data(), dists_.data(), counts_.data()));
    }
    void update(const Q* queries) { check(svsx_iterator_update(h_, queries)); std::fill(counts_.begin(), counts_.end(), 0u); }
    void restart_next_search() { check(svsx_iterator_restart_next_search(h_)); }

    size_t size(size_t q) const { return counts_[q]; }                                       // :224
    std::span<const Neighbor> contents(size_t q) {                                           // :221
        results_.resize(counts_[q]);
        for (size_t j = 0; j < counts_[q]; ++j) results_[j] = Neighbor{static_cast<size_t>(ids_[q * stride_ + j]), dists_[q * stride_ + j]};
        return {results_.data(), results_.size()};
    }
    size_t batch_number(size_t q) const { refresh(); return meta_[q * 8 + 5]; }             // :227
    bool done(size_t q) const { refresh(); return meta_[q * 8 + 4] == n_ || (meta_[q * 8 + 6] & 2u); } // :230
    VamanaSearchParameters parameters_for_current_iteration(size_t q) const {               // :236-245
        refresh();
        VamanaSearchParameters p{};
        p.buffer_config_ = SearchBufferConfig(meta_[q * 8 + 2], meta_[q * 8 + 3]);
        return p;
    }
};

} // namespace svsx
