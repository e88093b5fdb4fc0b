// search_index for the MI355X path in C++ -- the same eleven positional arguments as the reference's utils/search_index.cpp:89-178 -- over
// libsvsx's C ABI (include/svsx.h) and the header-only readers of include/svsx_io.hpp.  Reads what VamanaIndex::save wrote
// (index/vamana/index.h:825-840: svs_config.toml with the entry point, the native v1 blobs of dataset and graph; a directory is searched for
// its single *.svs blob), runs the batch search on the GPU and writes <prefix>_idx.ivecs like QueryResult::save_vecs.
//   g++ -std=c++20 -I include tools/search_index.cpp -L scalablevectorsearch_amd -lsvsx -Wl,-rpath,$PWD/scalablevectorsearch_amd -o search_index
// (scalablevectorsearch_amd/search_index.py is the same program in Python; tests/test_gpu_parity.py runs both against the oracle.)
#include "svsx.h"
#include "svsx_io.hpp"

#include <chrono>
#include <filesystem>
#include <iostream>
#include <map>

namespace {
const char* HELP = R"(
  search_index <query type> <data type> <queries.[fbi]vecs> <search window size> <neighbours> <threads (ignored)>
               <config dir | svs_config.toml> <graph dir | blob.svs> <data dir | blob.svs> <result prefix> <L2 | MIP | Cosine>

  query type   int8 | uint8 | float            element type the queries are converted to
  data type    int8 | uint8 | float | float16  element type of the saved dataset
  config       where VamanaIndex::save put svs_config.toml (the entry point is read from it)
  graph, data  SVS native v1 blobs (a directory is searched for its single *.svs file)
  result       <prefix>_idx.ivecs receives one row of neighbour ids per query
)";

std::string blob(const std::string& path) {
    namespace fs = std::filesystem;
    if (!fs::is_directory(path)) return path;
    std::vector<std::string> found;
    for (const auto& e : fs::directory_iterator(path))
        if (e.path().extension() == ".svs") found.push_back(e.path().string());
    if (found.size() != 1) throw std::runtime_error(path + ": expected exactly one .svs blob, found " + std::to_string(found.size()));
    return found[0];
}

bool ends_with(const std::string& s, const std::string& t) { return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }
} // namespace

int main(int argc, char** argv) {
    if (argc != 12) {
        std::cout << "Expected 11 arguments. Instead, got " << argc - 1 << ". The required positional arguments are given below.\n" << HELP << std::endl;
        return 1;
    }
    try {
        const std::string qt = argv[1], dt = argv[2], qfile = argv[3];
        const size_t window = std::stoull(argv[4]), k = std::stoull(argv[5]);
        const std::string config = argv[7], graph_path = argv[8], data_path = argv[9], prefix = argv[10], dist = argv[11];
        const std::map<std::string, int> dtypes{{"float", SVSX_F32}, {"float16", SVSX_F16}, {"int8", SVSX_I8}, {"uint8", SVSX_U8}};
        const std::map<std::string, int> metrics{{"L2", SVSX_L2}, {"MIP", SVSX_IP}, {"Cosine", SVSX_COSINE}};
        if (!metrics.count(dist)) throw std::runtime_error("Unsupported distance type. Valid values: L2/MIP/Cosine. Received: " + dist + "!");
        if (!dtypes.count(qt) || !dtypes.count(dt) || qt == "float16") throw std::runtime_error("Unsupported Query and Data type pair: (" + qt + ", " + dt + ")!");
        // dataset and graph: native v1 blobs; element sizes by type
        const auto graph = svsx::io::read_svs_native<uint32_t>(blob(graph_path));
        const size_t es = dt == "float" ? 4 : dt == "float16" ? 2 : 1;
        svsx::io::Matrix<unsigned char> data;
        {
            auto raw = svsx::io::read_file(blob(data_path));
            if (raw.size() < svsx::io::kSvsV1Header) throw std::runtime_error(data_path + ": truncated header");
            uint64_t magic = 0, n = 0, d = 0;
            std::memcpy(&magic, raw.data(), 8);
            std::memcpy(&n, raw.data() + 24, 8);
            std::memcpy(&d, raw.data() + 32, 8);
            if (magic != svsx::io::kSvsV1Magic || raw.size() - svsx::io::kSvsV1Header < n * d * es) throw std::runtime_error(data_path + ": not an SVS v1 blob of this element type");
            data.n = n;
            data.d = d;
            data.values.assign(raw.begin() + svsx::io::kSvsV1Header, raw.begin() + svsx::io::kSvsV1Header + n * d * es);
        }
        if (graph.n != data.n) throw std::runtime_error("graph and dataset differ in size");
        const std::string cfg = std::filesystem::is_directory(config) ? config + "/svs_config.toml" : config;
        const uint32_t entry = svsx::io::read_entry_point(cfg);
        svsx_index_desc desc{};
        desc.metric = metrics.at(dist);
        desc.data_dtype = dtypes.at(dt);
        desc.n = data.n;
        desc.d = data.d;
        desc.data = data.values.data();
        desc.graph = graph.data();
        desc.graph_row_stride_u32 = graph.d;
        desc.max_degree = static_cast<uint32_t>(graph.d - 1);
        desc.entry_points = &entry;
        desc.n_entry = 1;
        svsx_index* index = nullptr;
        if (svsx_index_create(&desc, &index) != SVSX_SUCCESS) throw std::runtime_error(svsx_last_error());
        // queries: [fbi]vecs, converted to the query element type when they differ
        std::vector<unsigned char> qbytes;
        size_t nq = 0, qd = 0;
        auto take = [&](const auto& m, auto convert) {
            nq = m.n;
            qd = m.d;
            using Out = decltype(convert(m.values[0]));
            qbytes.resize(nq * qd * sizeof(Out));
            Out* o = reinterpret_cast<Out*>(qbytes.data());
            for (size_t i = 0; i < nq * qd; ++i) o[i] = convert(m.values[i]);
        };
        auto load = [&](auto convert) {
            if (ends_with(qfile, ".fvecs")) take(svsx::io::read_vecs<float>(qfile), convert);
            else if (ends_with(qfile, ".bvecs")) take(svsx::io::read_vecs<uint8_t>(qfile), convert);
            else if (ends_with(qfile, ".ivecs")) take(svsx::io::read_vecs<uint32_t>(qfile), convert);
            else throw std::runtime_error("unsupported query file " + qfile);
        };
        if (qt == "float") load([](auto v) { return static_cast<float>(v); });
        else if (qt == "int8") load([](auto v) { return static_cast<int8_t>(v); });
        else load([](auto v) { return static_cast<uint8_t>(v); });
        if (qd != data.d) throw std::runtime_error("Query dimension " + std::to_string(qd) + " does not match index dimension " + std::to_string(data.d));
        svsx_search_params sp{};
        sp.search_window_size = window;
        sp.search_buffer_capacity = window;
        std::vector<uint32_t> ids(nq * k);
        std::vector<float> dists(nq * k);
        std::cout << "Running Search." << std::endl;
        const auto tic = std::chrono::steady_clock::now();
        if (svsx_search(index, qbytes.data(), dtypes.at(qt), nq, k, &sp, nullptr, ids.data(), dists.data(), nullptr) != SVSX_SUCCESS)
            throw std::runtime_error(svsx_last_error());
        std::cout << "Global search time: " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tic).count() << " seconds" << std::endl;
        svsx::io::write_ivecs(prefix + "_idx.ivecs", ids.data(), nq, k);
        svsx_index_destroy(index);
    } catch (const std::exception& e) {
        std::cerr << e.what() << std::endl;
        return 2;
    }
    return 0;
}
