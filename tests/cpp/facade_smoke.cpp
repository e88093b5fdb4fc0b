// Compiles the C++ facade (include/svsx.hpp) the way a reference user would call VamanaIndex::search and, when a GPU
// is present, runs a tiny search: ring graph of 8 points on a line; the nearest neighbours of q are known.
#include "svsx.hpp"

#include <algorithm>
#include <cstdio>
#include <utility>
#include <vector>

int main(int argc, char**) {
    const size_t n = 8, d = 4, R = 2;
    std::vector<float> data(n * d, 0.0f);
    for (size_t i = 0; i < n; ++i) data[i * d] = static_cast<float>(i);
    std::vector<uint32_t> graph(n * (R + 1));
    for (size_t i = 0; i < n; ++i) { graph[i * 3] = 2; graph[i * 3 + 1] = (i + 1) % n; graph[i * 3 + 2] = (i + n - 1) % n; }
    const uint32_t entry[1] = {0};
    std::vector<float> q = {5.2f, 0, 0, 0};
    try {
        svsx::SearchBufferConfig bad(10, 9);
        return 10;
    } catch (const svsx::ANNException&) {}
    if (argc > 1) return 0; // compile/argument check only
    try {
        svsx::VamanaIndex<float, svsx::DistanceL2> index(graph.data(), R + 1, data.data(), n, d, {entry, 1});
        svsx::QueryResult<size_t> result(1, 3);
        index.search(result.view(), q.data(), svsx::VamanaSearchParameters{}.buffer_config({8}));
        std::printf("%zu %zu %zu\n", result.index(0, 0), result.index(0, 1), result.index(0, 2));
        if (!(result.index(0, 0) == 5 && result.index(0, 1) == 6 && result.index(0, 2) == 4)) return 1;
        {   // svs::Flat-shaped facade over the same 8 points: exhaustive search, host pointers in and out
            svsx::FlatIndex<float, svsx::DistanceL2> flat(data.data(), n, d);
            const auto fr = flat.search(q.data(), 1, 3);
            if (!(fr.index(0, 0) == 5 && fr.index(0, 1) == 6 && fr.index(0, 2) == 4)) return 15;
        }
        // the cancel predicate of VamanaIndex::search (index.h:564-570): false -> same answer; true -> nothing is written
        svsx::QueryResult<size_t> r2(1, 3);
        index.search(r2.view(), q.data(), svsx::VamanaSearchParameters{}.buffer_config({8}), [] { return false; });
        if (!(r2.index(0, 0) == 5 && r2.index(0, 1) == 6 && r2.index(0, 2) == 4)) return 7;
        svsx::QueryResult<size_t> r3(1, 3);
        r3.view().indices[0] = 12345;
        index.search(r3.view(), q.data(), svsx::VamanaSearchParameters{}.buffer_config({8}), [] { return true; });
        if (r3.index(0, 0) != 12345) return 8;
        {   // graph-inline bound records (the default fast path; result-neutral) and the runtime's IDFilter argument on a 64-point ring in 32 dimensions
            const size_t n2 = 64, d2 = 32;
            std::vector<float> data2(n2 * d2, 0.0f);
            for (size_t i = 0; i < n2; ++i) { data2[i * d2] = static_cast<float>(i); data2[i * d2 + 1] = 0.25f * static_cast<float>(i % 5); }
            std::vector<uint32_t> graph2(n2 * (R + 1));
            for (size_t i = 0; i < n2; ++i) { graph2[i * 3] = 2; graph2[i * 3 + 1] = (i + 1) % n2; graph2[i * 3 + 2] = (i + n2 - 1) % n2; }
            std::vector<float> q2(d2, 0.0f);
            q2[0] = 40.2f;
            svsx::VamanaIndex<float, svsx::DistanceL2> index2(graph2.data(), R + 1, data2.data(), n2, d2, {entry, 1});
            const auto sp2 = svsx::VamanaSearchParameters{}.buffer_config({16}).search_buffer_visited_set(true);
            svsx::QueryResult<size_t> plain(1, 4), coded(1, 4), filtered(1, 4);
            index2.search(plain.view(), q2.data(), sp2);
            index2.build_bound_codes();   // mode 2
            index2.search(coded.view(), q2.data(), sp2);
            for (size_t j = 0; j < 4; ++j)
                if (plain.index(0, j) != coded.index(0, j) || plain.distance(0, j) != coded.distance(0, j)) return 11;
            if (plain.index(0, 0) != 40) return 12;
            std::vector<uint64_t> even(1, 0x5555555555555555ull); // ids 0, 2, 4, ... are members
            index2.search(filtered.view(), q2.data(), sp2, std::span<const uint64_t>(even.data(), even.size()));
            std::printf("filtered: %zu %zu %zu %zu\n", filtered.index(0, 0), filtered.index(0, 1), filtered.index(0, 2), filtered.index(0, 3));
            if (!(filtered.index(0, 0) == 40 && filtered.index(0, 1) == 42 && filtered.index(0, 2) == 38 && filtered.index(0, 3) == 44)) return 13;
            {   // runtime range_search / get_distance through the C ABI (svsx_range_search, svsx_get_distance): points within radius 2.5^2 + slack of x = 40.2
                const auto rr = index2.range_search(q2.data(), 1, 7.0f, sp2);
                std::printf("range: %zu results\n", rr.labels.size());
                if (rr.lims.size() != 2 || rr.lims[1] != rr.labels.size() || rr.labels.empty() || rr.labels[0] != 40) return 15;
                for (size_t j = 0; j < rr.labels.size(); ++j) {
                    if (!(rr.distances[j] < 7.0f)) return 16;
                    if (static_cast<float>(index2.get_distance(rr.labels[j], q2.data())) != rr.distances[j]) return 17;
                }
                const auto rf = index2.range_search(q2.data(), 1, 7.0f, sp2, std::span<const uint64_t>(even.data(), even.size()));
                for (const uint64_t id : rf.labels) if (id & 1u) return 18;
                if (rf.labels.empty() || rf.labels.size() >= rr.labels.size()) return 19;
            }
            // VamanaIndex::calibrate (search-buffer part, calibrate.h:249-343): exact top-4 by brute force as the groundtruth, target recall 1
            std::vector<std::pair<float, size_t>> all;
            for (size_t i = 0; i < n2; ++i) {
                float dd = 0.0f;
                for (size_t j = 0; j < d2; ++j) dd += (data2[i * d2 + j] - q2[j]) * (data2[i * d2 + j] - q2[j]);
                all.emplace_back(dd, i);
            }
            std::sort(all.begin(), all.end());
            const size_t gt[4] = {all[0].second, all[1].second, all[2].second, all[3].second};
            const auto tuned = index2.calibrate(q2.data(), 1, gt, 4, 4, 1.0);
            svsx::QueryResult<size_t> again(1, 4);
            index2.search(again.view(), q2.data(), index2.get_search_parameters());
            std::printf("calibrated: %zu %zu recall %.2f\n", tuned.buffer_config_.get_search_window_size(), tuned.buffer_config_.get_total_capacity(),
                        svsx::k_recall_at_n(gt, 4, again.view().indices, 4, 1, 4, 4));
            if (svsx::k_recall_at_n(gt, 4, again.view().indices, 4, 1, 4, 4) != 1.0) return 14;
        }
        // BatchIterator: 8 points on a ring, batches of 3 -> 5 6 4 | 7 3 2 | 1 0 | (done)
        svsx::BatchIterator<float> it(index, q.data(), 1);
        const size_t want[8] = {5, 6, 4, 7, 3, 2, 1, 0};
        size_t seen = 0;
        for (size_t batch = 1; !it.done(0); ++batch) {
            it.next(3);
            if (it.batch_number(0) != batch) return 3;
            for (const auto& nb : it.contents(0)) {
                std::printf("batch %zu: %zu %g\n", batch, nb.id(), nb.distance());
                if (seen >= 8 || nb.id() != want[seen++]) return 4;
            }
        }
        if (seen != 8) return 5;
        it.next(3);
        return it.size(0) == 0 ? 0 : 6;
    } catch (const svsx::ANNException& e) {
        std::printf("ANNException(%d): %s\n", e.code, e.what());
        return 2;
    }
}
