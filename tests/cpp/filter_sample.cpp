// The draws of sample_filter_hits (bindings/cpp/src/svs_runtime_utils.h:445-471) by the standard library's own classes -- the checker of the
// Python restatement in scalablevectorsearch_amd/iterator.py (tests/test_host_logic.py); libsvsx uses these classes directly.
//   filter_sample <n> <count>   ->  <count> draws of std::uniform_int_distribution<size_t>(0, n - 1) over std::mt19937(42), one per line
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const size_t n = std::strtoull(argv[1], nullptr, 10), count = std::strtoull(argv[2], nullptr, 10);
    std::mt19937 rng(42);
    std::uniform_int_distribution<size_t> dist(0, n - 1);
    for (size_t i = 0; i < count; ++i) std::printf("%zu\n", dist(rng));
    return 0;
}
