// gather_bench.hip -- how fast can one MI355X gather random rows of 64 / 128 / 256 / 384 bytes from HBM / the Infinity Cache?
// Decides the bound-code row size: does a 64-B row cost half the fabric traffic of a 128-B row?
//   ./gather_bench [table_MiB=8192] [iters=3]
// Each "query" wave walks a pseudo-random chain: it draws 16 (or 64/G) row ids per step from a hash of (wave, step), G lanes
// load one row with 16-B loads, and the xor of what was read feeds the next step's seed (so steps are dependent, like hops).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// ROWB: bytes per row (multiple of 16), LPR = ROWB / 16 lanes per row (each lane one dwordx4) capped at 8 (then 2+ loads per lane)
template <int ROWB>
__global__ __launch_bounds__(64) void gather_kernel(const uint4* __restrict__ table, uint32_t n_rows, uint32_t steps, uint32_t* out) {
    constexpr int CH = ROWB / 16;                 // 16-B chunks per row
    constexpr int G = CH >= 8 ? 8 : (CH >= 4 ? 4 : CH);   // lanes per row
    constexpr int LPL = CH / G;                   // loads per lane
    constexpr int RPP = 64 / G;
    const uint32_t lane = threadIdx.x, sub = lane % G, grp = lane / G;
    uint32_t seed = blockIdx.x * 2654435761u + 12345u;
    uint32_t acc = 0;
    for (uint32_t s = 0; s < steps; ++s) {
        const uint32_t r = mix(seed + grp * 0x9e3779b9u + s * 0x85ebca6bu) % n_rows;
        const uint4* row = table + static_cast<size_t>(r) * CH + sub;
        uint4 v[LPL];
#pragma unroll
        for (int i = 0; i < LPL; ++i) v[i] = row[i * G];
        uint32_t x = 0;
#pragma unroll
        for (int i = 0; i < LPL; ++i) x ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        // wave-wide dependency: next step's seed depends on everything read in this one
        for (int off = 32; off >= 1; off >>= 1) x ^= __shfl_xor(x, off);
        seed = seed * 1664525u + 1013904223u + (x & 1u);
        acc ^= x;
    }
    if (lane == 0) out[blockIdx.x] = acc;
}

template <int ROWB> double run(const uint4* table, size_t table_bytes, uint32_t waves, uint32_t steps, int iters, uint32_t* out) {
    const uint32_t n_rows = static_cast<uint32_t>(table_bytes / ROWB);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(gather_kernel<ROWB>, dim3(waves), dim3(64), 0, 0, table, n_rows, steps, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(gather_kernel<ROWB>, dim3(waves), dim3(64), 0, 0, table, n_rows, steps, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    constexpr int CH = ROWB / 16; constexpr int G = CH >= 8 ? 8 : (CH >= 4 ? 4 : CH);
    const double rows = double(waves) * steps * (64 / G);
    printf("{\"row_bytes\": %d, \"table_MiB\": %zu, \"waves\": %u, \"steps\": %u, \"ms\": %.4f, \"Grows_per_s\": %.3f, \"useful_GBs\": %.1f}\n", ROWB, table_bytes >> 20, waves, steps,
           best, rows / best * 1e-6, rows * ROWB / best * 1e-6);
    fflush(stdout);
    return best;
}

int main(int argc, char** argv) {
    size_t mib = argc > 1 ? strtoull(argv[1], nullptr, 10) : 8192;
    int iters = argc > 2 ? atoi(argv[2]) : 3;
    size_t bytes = mib << 20;
    uint4* table; uint32_t* out;
    CK(hipMalloc(&table, bytes));
    CK(hipMemset(table, 1, bytes));
    CK(hipMalloc(&out, 1 << 22));
    for (uint32_t waves : {5120u, 8192u, 16384u}) {
        const uint32_t steps = 400;
        run<64>(table, bytes, waves, steps, iters, out);
        run<128>(table, bytes, waves, steps, iters, out);
        run<256>(table, bytes, waves, steps, iters, out);
        run<384>(table, bytes, waves, steps, iters, out);
    }
    return 0;
}
