// Resident single-wave workgroups per CU as a function of the kernel's VGPR / SGPR / LDS request (gfx950):
//   hipcc --offload-arch=gfx950 -O2 -o occupancy_probe occupancy_probe.hip && ./occupancy_probe
// Every wave stamps wall_clock64 (100 MHz) at start and end and idles ~150 us in between; the host counts the maximum overlap.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

template <int V, int S> __global__ void __launch_bounds__(64) probe(unsigned long long* t, int spin_ticks) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = wall_clock64();
    // touch the highest register of the request so that next_free_vgpr = V, next_free_sgpr >= S
    if constexpr (V == 32) asm volatile("v_mov_b32 v31, 0" ::: "v31");
    if constexpr (V == 56) asm volatile("v_mov_b32 v55, 0" ::: "v55");
    if constexpr (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if constexpr (V == 67) asm volatile("v_mov_b32 v66, 0" ::: "v66");
    if constexpr (V == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if constexpr (V == 73) asm volatile("v_mov_b32 v72, 0" ::: "v72");
    if constexpr (V == 80) asm volatile("v_mov_b32 v79, 0" ::: "v79");
    if constexpr (V == 81) asm volatile("v_mov_b32 v80, 0" ::: "v80");
    if constexpr (S == 64) asm volatile("s_mov_b32 s63, 0" ::: "s63");
    if constexpr (S == 80) asm volatile("s_mov_b32 s79, 0" ::: "s79");
    if constexpr (S == 96) asm volatile("s_mov_b32 s95, 0" ::: "s95");
    if constexpr (S == 100) asm volatile("s_mov_b32 s99, 0" ::: "s99");
    if (threadIdx.x == 0) smem[0] = 1;
    while (wall_clock64() - t0 < static_cast<unsigned long long>(spin_ticks)) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) {
        t[2 * blockIdx.x] = t0;
        t[2 * blockIdx.x + 1] = wall_clock64();
    }
}

template <int V, int S> static void run(size_t lds, unsigned long long* d_t, int nwg) {
    std::vector<unsigned long long> h(2 * nwg);
    auto k = probe<V, S>;
    int per_cu = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(k), 64, lds);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(64), lds, 0, d_t, 15000);
    if (hipDeviceSynchronize() != hipSuccess) { printf("V=%d S=%d lds=%zu: launch failed\n", V, S, lds); return; }
    (void)hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * 2 * nwg, hipMemcpyDeviceToHost);
    std::vector<std::pair<unsigned long long, int>> ev;
    for (int i = 0; i < nwg; ++i) { ev.push_back({h[2 * i], 1}); ev.push_back({h[2 * i + 1], -1}); }
    std::sort(ev.begin(), ev.end());
    int cur = 0, best = 0;
    for (auto& e : ev) { cur += e.second; best = std::max(best, cur); }
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k));
    printf("request V=%3d S=%3d lds=%6zu | compiled numRegs=%3d | runtime says %2d/CU | measured max resident %5d = %.2f/CU\n", V, S, lds, fa.numRegs, per_cu, best, best / 256.0);
}

int main() {
    const int nwg = 20000;
    unsigned long long* d_t;
    if (hipMalloc(&d_t, sizeof(unsigned long long) * 2 * nwg) != hipSuccess) return 1;
    run<32, 0>(64, d_t, nwg);
    run<56, 0>(64, d_t, nwg);
    run<64, 0>(64, d_t, nwg);
    run<67, 0>(64, d_t, nwg);
    run<72, 0>(64, d_t, nwg);
    run<73, 0>(64, d_t, nwg);
    run<80, 0>(64, d_t, nwg);
    run<81, 0>(64, d_t, nwg);
    run<32, 64>(64, d_t, nwg);
    run<32, 80>(64, d_t, nwg);
    run<32, 96>(64, d_t, nwg);
    run<32, 100>(64, d_t, nwg);
    run<67, 100>(64, d_t, nwg);
    run<67, 100>(5776, d_t, nwg);
    run<32, 0>(4096, d_t, nwg);
    run<32, 0>(5120, d_t, nwg);
    run<32, 0>(5776, d_t, nwg);
    run<32, 0>(6144, d_t, nwg);
    run<32, 0>(6400, d_t, nwg);
    run<32, 0>(6656, d_t, nwg);
    run<32, 0>(8192, d_t, nwg);
    return 0;
}
