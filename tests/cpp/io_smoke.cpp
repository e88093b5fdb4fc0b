// Host-side checks of include/svsx_io.hpp (no GPU): the reference's own data files, the entry point key of a saved config, recall, and the
// calibration search over a synthetic recall model.  Prints one line per check; tests/test_abi.py compares them with the Python mirror.
#include "svsx_io.hpp"

#include <cmath>
#include <cstdio>

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const std::string fvecs = argv[1], svs = argv[2], tmp = argv[3];
    try {
        const auto a = svsx::io::read_vecs<float>(fvecs);
        const auto b = svsx::io::read_svs_native<float>(svs);
        double sa = 0.0, sb = 0.0;
        for (float v : a.values) sa += v;
        for (float v : b.values) sb += v;
        std::printf("fvecs %zu %zu %.9g\n", a.n, a.d, sa);
        std::printf("svs %zu %zu %.9g\n", b.n, b.d, sb);
        // round trips
        svsx::io::write_svs_native(tmp + "/rt.svs", a.data(), a.n, a.d);
        const auto c = svsx::io::read_svs_native<float>(tmp + "/rt.svs");
        std::printf("roundtrip %d\n", c.n == a.n && c.d == a.d && c.values == a.values ? 1 : 0);
        const uint32_t ids[6] = {1, 2, 3, 4, 5, 6};
        svsx::io::write_ivecs(tmp + "/ids.ivecs", ids, 2, 3);
        const auto d = svsx::io::read_vecs<uint32_t>(tmp + "/ids.ivecs");
        std::printf("ivecs %zu %zu %u\n", d.n, d.d, d.values[5]);
        // svs_config.toml as VamanaIndex::save nests it
        {
            std::ofstream f(tmp + "/svs_config.toml");
            f << "__version__ = 'v0.0.2'\n[object]\nalpha = 1.2   # entry_point = 9 in a comment does not count\nnot_the_entry_point = 3\n  entry_point = 4711\nname = 'vamana index parameters'\n";
        }
        std::printf("entry %u\n", svsx::io::read_entry_point(tmp + "/svs_config.toml"));
        // recall: 2 queries, k = n = 3: {1,2,3} vs {3,9,1} -> 2 hits; {4,5,6} vs {6,5,4} -> 3 hits
        const uint32_t gt[6] = {1, 2, 3, 4, 5, 6};
        const uint64_t res[6] = {3, 9, 1, 6, 5, 4};
        std::printf("recall %.6f\n", svsx::k_recall_at_n(gt, 3, res, 3, 2, 3, 3));
        // calibration over the recall model of tests/test_host_logic.py: recall = min(1, 0.5 + 0.02 w + 0.001 (c - w))
        int calls = 0;
        auto model = [&](size_t w, size_t c) { ++calls; return std::min(1.0, 0.5 + 0.02 * static_cast<double>(w) + 0.001 * static_cast<double>(c - w)); };
        svsx::CalibrationParameters cp;
        auto r1 = svsx::optimize_search_buffer(cp, 0, 0, 10, 0.9, model);
        std::printf("cal1 %zu %zu %d %d\n", r1.window, r1.capacity, r1.converged ? 1 : 0, calls <= 12 ? 1 : 0);
        svsx::CalibrationParameters cp30;
        cp30.search_window_size_upper_ = 30;
        auto r2 = svsx::optimize_search_buffer(cp30, 0, 0, 10, 1.5, model);
        std::printf("cal2 %zu %d\n", r2.window, r2.converged ? 1 : 0);
        auto r3 = svsx::optimize_search_buffer(cp, 0, 0, 10, 0.6, model);
        std::printf("cal3 %zu %zu %d\n", r3.window, r3.capacity, r3.converged ? 1 : 0);
        svsx::CalibrationParameters roi;
        roi.search_buffer_optimization_ = svsx::CalibrationParameters::SearchBufferOptimization::ROIOnly;
        auto r4 = svsx::optimize_search_buffer(roi, 0, 0, 10, 0.6, model);
        std::printf("cal4 %zu %zu\n", r4.window, r4.capacity);
        try {
            svsx::io::read_svs_native<float>(fvecs); // wrong magic
            return 3;
        } catch (const std::runtime_error&) {}
    } catch (const std::exception& e) {
        std::printf("error: %s\n", e.what());
        return 1;
    }
    return 0;
}
