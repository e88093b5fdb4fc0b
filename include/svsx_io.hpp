// svsx_io.hpp -- header-only C++ host side of the rows SURVEY 8(f) ranks 2 and 3 for a C++ caller of libsvsx: the on-disk formats that
// feed the search path, recall, and search-parameter calibration.  Host logic only (no device code); the searches it drives go through
// svsx.hpp / the C ABI.  Mirrors (paths relative to the reference root):
//   [fbi]vecs                    core/io/vecs.h:37-273        each vector = int32 dimension + d elements (f32 / u8 / i32)
//   SVS native v1 (".svs")       core/io/native.h:307-344     1024-byte header {u64 magic, 16-byte uuid, u64 num_vectors, u64 dims}, dense payload;
//                                                             VamanaIndex::save writes the dataset and the graph (n x (R + 1) u32, slot 0 = degree,
//                                                             core/graph/graph.h:300-305) as two such blobs
//   svs_config.toml              index/vamana/index.h:88-97   the `entry_point` key of the saved index parameters
//   k_recall_at_n                core/recall.h:47-58
//   optimize_search_buffer       index/vamana/calibrate.h:249-343 (+ :150-169 for the window-below-k split); prefetch tuning (:345-470) has no
//                                GPU analogue and is omitted
// The Python host mirror has the same pieces (scalablevectorsearch_amd/io.py, calibrate.py); tests/cpp/io_smoke.cpp checks both against the
// reference's own data files (tests/golden/known_f32.{fvecs,svs}).
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace svsx::io {

template <typename T> struct Matrix { // dense row-major n x d
    size_t n = 0, d = 0;
    std::vector<T> values;
    const T* row(size_t i) const { return values.data() + i * d; }
    T* row(size_t i) { return values.data() + i * d; }
    const T* data() const { return values.data(); }
};

inline std::vector<unsigned char> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error(path + ": cannot open");
    const std::streamsize size = f.tellg();
    f.seekg(0);
    std::vector<unsigned char> raw(static_cast<size_t>(size));
    if (size > 0 && !f.read(reinterpret_cast<char*>(raw.data()), size)) throw std::runtime_error(path + ": short read");
    return raw;
}

// core/io/vecs.h:37-273: T = float (.fvecs), uint8_t (.bvecs), uint32_t / int32_t (.ivecs)
template <typename T> Matrix<T> read_vecs(const std::string& path) {
    const std::vector<unsigned char> raw = read_file(path);
    Matrix<T> m;
    if (raw.empty()) return m;
    int32_t d = 0;
    std::memcpy(&d, raw.data(), 4);
    const size_t row = 4 + static_cast<size_t>(d > 0 ? d : 0) * sizeof(T);
    if (d <= 0 || raw.size() % row != 0) throw std::runtime_error(path + ": not a well-formed vecs file");
    m.n = raw.size() / row;
    m.d = static_cast<size_t>(d);
    m.values.resize(m.n * m.d);
    for (size_t i = 0; i < m.n; ++i) {
        int32_t di = 0;
        std::memcpy(&di, raw.data() + i * row, 4);
        if (di != d) throw std::runtime_error(path + ": vectors of differing dimensionality");
        std::memcpy(m.row(i), raw.data() + i * row + 4, m.d * sizeof(T));
    }
    return m;
}

// QueryResult::save_vecs shape: one row of k ids per query
inline void write_ivecs(const std::string& path, const uint32_t* ids, size_t n, size_t k) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(path + ": cannot open for writing");
    const int32_t kk = static_cast<int32_t>(k);
    for (size_t i = 0; i < n; ++i) {
        f.write(reinterpret_cast<const char*>(&kk), 4);
        f.write(reinterpret_cast<const char*>(ids + i * k), static_cast<std::streamsize>(k * 4));
    }
}

constexpr uint64_t kSvsV1Magic = 0xCAD4A6B2579980FEull; // core/io/native.h:307-344
constexpr size_t kSvsV1Header = 1024;

// dense SVS v1 blob -> n x d matrix of T (f32 / f16-as-uint16_t / i8 / u8 data, u32 graph rows)
template <typename T> Matrix<T> read_svs_native(const std::string& path) {
    const std::vector<unsigned char> raw = read_file(path);
    if (raw.size() < kSvsV1Header) throw std::runtime_error(path + ": truncated header");
    uint64_t magic = 0, n = 0, d = 0;
    std::memcpy(&magic, raw.data(), 8);
    if (magic != kSvsV1Magic) throw std::runtime_error(path + ": bad magic");
    std::memcpy(&n, raw.data() + 24, 8);
    std::memcpy(&d, raw.data() + 32, 8);
    if (d != 0 && n > (raw.size() - kSvsV1Header) / (d * sizeof(T))) throw std::runtime_error(path + ": payload shorter than the header says");
    Matrix<T> m;
    m.n = static_cast<size_t>(n);
    m.d = static_cast<size_t>(d);
    m.values.resize(m.n * m.d);
    std::memcpy(m.values.data(), raw.data() + kSvsV1Header, m.values.size() * sizeof(T));
    return m;
}

template <typename T> void write_svs_native(const std::string& path, const T* values, size_t n, size_t d) {
    std::vector<unsigned char> head(kSvsV1Header, 0);
    const uint64_t nn = n, dd = d;
    std::memcpy(head.data(), &kSvsV1Magic, 8);
    std::memcpy(head.data() + 24, &nn, 8);
    std::memcpy(head.data() + 32, &dd, 8);
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(path + ": cannot open for writing");
    f.write(reinterpret_cast<const char*>(head.data()), static_cast<std::streamsize>(head.size()));
    f.write(reinterpret_cast<const char*>(values), static_cast<std::streamsize>(n * d * sizeof(T)));
}

// svs_config.toml as VamanaIndex::save writes it (index/vamana/index.h:88-97,141-170,825-840): the `vamana_index_parameters` object with its
// entry point, the build parameters (build_params.h:88-101) and the DEFAULT SEARCH PARAMETERS the index is reloaded with (search_params.h:79-126,
// index.h:318) -- plus the flat legacy layout of versions <= v0.0.2 (index.h:107-150: max_out_degree, construction_window_size, max_candidates,
// default_search_window_size, visited_set).  A scanner for the TOML subset the reference emits (tables, integer / float / boolean / quoted-string
// values, comments), not a general TOML parser.
struct IndexConfig {
    uint32_t entry_point = 0;
    bool has_build = false, has_search = false;
    // VamanaBuildParameters
    float alpha = 1.2f;
    size_t graph_max_degree = 0, window_size = 0, max_candidate_pool_size = 0, prune_to = 0;
    bool use_full_search_history = true;
    // VamanaSearchParameters
    size_t search_window_size = 0, search_buffer_capacity = 0, prefetch_lookahead = 4, prefetch_step = 1;
    bool search_buffer_visited_set = false;
};

inline IndexConfig read_index_config(const std::string& config_path) {
    std::ifstream f(config_path);
    if (!f) throw std::runtime_error(config_path + ": cannot open");
    auto trim = [](std::string v) {
        const size_t a = v.find_first_not_of(" \t\r"), b = v.find_last_not_of(" \t\r");
        return a == std::string::npos ? std::string() : v.substr(a, b - a + 1);
    };
    IndexConfig c;
    std::string line, table;
    bool have_entry = false, legacy_window = false;
    size_t legacy_w = 0;
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        line = trim(line);
        if (line.empty()) continue;
        if (line.front() == '[') { // [object.build_parameters]
            const size_t close = line.find(']');
            table = close == std::string::npos ? std::string() : trim(line.substr(1, close - 1));
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = trim(line.substr(0, eq));
        std::string val = trim(line.substr(eq + 1));
        if (!val.empty() && (val.front() == '\'' || val.front() == '"')) continue; // strings (names, schemas, versions) are not needed
        const bool truth = val == "true";
        auto as_size = [&] { return static_cast<size_t>(std::stoull(val)); };
        const bool in_build = table.size() >= 16 && table.compare(table.size() - 16, 16, "build_parameters") == 0;
        const bool in_search = table.size() >= 17 && table.compare(table.size() - 17, 17, "search_parameters") == 0;
        if (in_build) {
            c.has_build = true;
            if (key == "alpha") c.alpha = std::stof(val);
            else if (key == "graph_max_degree") c.graph_max_degree = as_size();
            else if (key == "window_size") c.window_size = as_size();
            else if (key == "max_candidate_pool_size") c.max_candidate_pool_size = as_size();
            else if (key == "prune_to") c.prune_to = as_size();
            else if (key == "use_full_search_history") c.use_full_search_history = truth;
        } else if (in_search) {
            c.has_search = true;
            if (key == "search_window_size") c.search_window_size = as_size();
            else if (key == "search_buffer_capacity") c.search_buffer_capacity = as_size();
            else if (key == "search_buffer_visited_set") c.search_buffer_visited_set = truth;
            else if (key == "prefetch_lookahead") c.prefetch_lookahead = as_size();
            else if (key == "prefetch_step") c.prefetch_step = as_size();
        } else { // the index object itself, and the flat legacy layout
            if (key == "entry_point") { c.entry_point = static_cast<uint32_t>(std::stoull(val)); have_entry = true; }
            else if (key == "alpha") { c.alpha = std::stof(val); c.has_build = true; }
            else if (key == "max_out_degree") { c.graph_max_degree = as_size(); c.has_build = true; if (c.prune_to == 0) c.prune_to = c.graph_max_degree; }
            else if (key == "construction_window_size") c.window_size = as_size();
            else if (key == "max_candidates") c.max_candidate_pool_size = as_size();
            else if (key == "prune_to") c.prune_to = as_size();
            else if (key == "use_full_search_history") c.use_full_search_history = truth;
            else if (key == "default_search_window_size") { legacy_w = as_size(); legacy_window = true; }
            else if (key == "visited_set") { c.search_buffer_visited_set = truth; c.has_search = true; }
        }
    }
    if (!have_entry) throw std::runtime_error(config_path + ": no entry_point key");
    if (legacy_window) { // load_legacy (index.h:107-150): SearchBufferConfig{default_search_window_size}, prefetch (4, 1)
        c.has_search = true;
        c.search_window_size = c.search_buffer_capacity = legacy_w;
    }
    if (c.search_window_size > c.search_buffer_capacity) throw std::runtime_error(config_path + ": search window larger than the buffer capacity (search_buffer.h:84-93)");
    return c;
}

// The one key the search_index CLI needs from svs_config.toml (index/vamana/index.h:88-97): `entry_point = <integer>`, wherever the saved object nests it.
// A line scanner, not a TOML parser: comments and other keys are skipped; a missing key throws.
inline uint32_t read_entry_point(const std::string& config_path) {
    std::ifstream f(config_path);
    if (!f) throw std::runtime_error(config_path + ": cannot open");
    std::string line;
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        const size_t key = line.find("entry_point");
        if (key == std::string::npos) continue;
        if (line.find_first_not_of(" \t") != key) continue; // a key at the start of its line, not part of another key or a value
        const size_t eq = line.find('=', key + 11);
        if (eq == std::string::npos || line.find_first_not_of(" \t", key + 11) != eq) continue;
        const size_t digit = line.find_first_of("0123456789", eq);
        if (digit == std::string::npos) continue;
        return static_cast<uint32_t>(std::stoull(line.substr(digit)));
    }
    throw std::runtime_error(config_path + ": no entry_point key");
}

} // namespace svsx::io

namespace svsx {

// core/recall.h:47-58: |first k of the groundtruth row  intersected with  first n of the result row| summed over the queries / (k * queries)
template <typename G, typename R>
double k_recall_at_n(const G* groundtruth, size_t gt_stride, const R* results, size_t res_stride, size_t n_queries, size_t k, size_t n) {
    size_t count = 0;
    for (size_t i = 0; i < n_queries; ++i)
        for (size_t a = 0; a < k; ++a)
            for (size_t b = 0; b < n; ++b)
                if (static_cast<uint64_t>(groundtruth[i * gt_stride + a]) == static_cast<uint64_t>(results[i * res_stride + b])) {
                    ++count;
                    break;
                }
    return static_cast<double>(count) / static_cast<double>(k * n_queries);
}

// index/vamana/calibrate.h:91-109 (the buffer-related fields)
struct CalibrationParameters {
    enum class SearchBufferOptimization { Disable, All, ROIOnly, ROITuneUp };
    size_t search_window_size_upper_ = 1000;
    size_t search_window_capacity_upper_ = 1000;
    SearchBufferOptimization search_buffer_optimization_ = SearchBufferOptimization::All;
    bool use_existing_parameter_values_ = true;
};

struct BufferChoice { // SearchBufferConfig as plain numbers (svsx.hpp turns it into VamanaSearchParameters)
    size_t window = 0, capacity = 0;
    bool converged = false;
};

namespace detail {
// std::lower_bound over the integer range [lo, hi): first v for which is_less(v) is false (hi if none)
template <typename F> size_t lower_bound_range(size_t lo, size_t hi, F&& is_less) {
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        if (is_less(mid)) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
} // namespace detail

// calibrate.h:249-343 for datasets without reranking: the smallest search window whose recall reaches the target (binary search; the recall is
// assumed monotone in the window, as there), then -- state All, window already equal to k -- a window below k with the capacity fixed to k
// (:150-169).  compute_recall(window, capacity) runs the search and scores it.
inline BufferChoice optimize_search_buffer(const CalibrationParameters& cp, size_t current_window, size_t current_capacity, size_t num_neighbors,
                                           double target_recall, const std::function<double(size_t, size_t)>& compute_recall) {
    using Opt = CalibrationParameters::SearchBufferOptimization;
    (void)current_window;
    const bool use_current_capacity = cp.search_buffer_optimization_ == Opt::ROITuneUp;
    double max_recall = -1.0;
    auto capacity_for = [&](size_t window) { return use_current_capacity ? std::max(window, current_capacity) : window; };
    const size_t lower = (use_current_capacity && current_capacity >= num_neighbors) ? 1 : num_neighbors;
    const size_t upper = cp.search_window_size_upper_;
    size_t window = detail::lower_bound_range(lower, upper, [&](size_t w) {
        const double r = compute_recall(w, capacity_for(w));
        max_recall = std::max(max_recall, r);
        return r < target_recall;
    });
    window = upper > lower ? std::min(window, upper - 1) : lower; // *lower_bound(...) of a UnitRange never dereferences end
    BufferChoice best{window, capacity_for(window), max_recall >= target_recall};
    if (cp.search_buffer_optimization_ != Opt::All || !best.converged || window != num_neighbors) return best;
    const size_t cap = best.capacity; // "maybe oversized": shrink the window below k, capacity stays k
    size_t w = detail::lower_bound_range(1, cap, [&](size_t ws) { return compute_recall(ws, cap) < target_recall; });
    w = cap > 1 ? std::min(w, cap - 1) : 1;
    best.window = w;
    return best;
}

} // namespace svsx
