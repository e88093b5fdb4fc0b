// How fast do v_mfma_f32_32x32x16_f16 chains run in the shapes the flat threshold kernel uses?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate tools/ubench/mfma_rate.hip && ./mfma_rate
// For each (threads per workgroup, workgroups, chains per wave, registers forced) prints TFLOP/s of dense f16 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16v_t __attribute__((ext_vector_type(16)));

template <int CHAINS, int PAD>
__global__ void mfma_loop(const _Float16* __restrict__ src, float* __restrict__ out, int iters) {
    half8_t a[CHAINS], b;
    for (int c = 0; c < CHAINS; ++c)
        for (int e = 0; e < 8; ++e) a[c][e] = src[(threadIdx.x * 8 + e + c) & 1023];
    for (int e = 0; e < 8; ++e) b[e] = src[(threadIdx.x + e) & 1023];
    float16v_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.0f;
    float pad[PAD > 0 ? PAD : 1]; // live registers that lower the occupancy like the real kernel's A fragments do
    for (int i = 0; i < PAD; ++i) pad[i] = static_cast<float>(src[(threadIdx.x + i) & 1023]);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[c], b, acc[c], 0, 0, 0);
        }
        if (PAD > 0) {
#pragma unroll
            for (int i = 0; i < PAD; ++i) asm volatile("" : "+v"(pad[i]));
        }
    }
    float s = 0.0f;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int i = 0; i < PAD; ++i) s += pad[i];
    if (s == 123.456f) out[0] = s;
}

template <int CHAINS, int PAD> static void run(const char* what, int threads, int blocks, const _Float16* src, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<CHAINS, PAD>), dim3(blocks), dim3(threads), 0, 0, src, out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<CHAINS, PAD>), dim3(blocks), dim3(threads), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double waves = static_cast<double>(blocks) * threads / 64;
    const double flop = waves * iters * 6.0 * CHAINS * 32768.0;
    hipFuncAttributes at;
    hipFuncGetAttributes(&at, reinterpret_cast<const void*>(mfma_loop<CHAINS, PAD>));
    printf("{\"what\": \"%s\", \"threads_per_wg\": %d, \"workgroups\": %d, \"chains_per_wave\": %d, \"vgprs\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", what, threads,
           blocks, CHAINS, at.numRegs, ms, flop / (ms * 1e-3) / 1e12);
}

int main() {
    _Float16* src;
    float* out;
    hipMalloc(&src, 1024 * 2);
    hipMalloc(&out, 64);
    std::vector<_Float16> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = static_cast<_Float16>(0.001f * (i % 17));
    hipMemcpy(src, h.data(), 2048, hipMemcpyHostToDevice);
    run<2, 0>("1-wave workgroups, 2 per SIMD if spread", 64, 2048, src, out);
    run<2, 0>("4-wave workgroups, same wave count", 256, 512, src, out);
    run<2, 0>("1-wave workgroups, 8 per SIMD", 64, 8192, src, out);
    run<4, 0>("1-wave workgroups, 4 chains", 64, 2048, src, out);
    run<2, 160>("1-wave workgroups, 2 chains, ~200 live registers", 64, 2048, src, out);
    run<2, 160>("4-wave workgroups, 2 chains, ~200 live registers", 256, 512, src, out);
    run<1, 0>("1-wave workgroups, 1 chain", 64, 2048, src, out);
    run<2, 0>("1-wave workgroups, 1 per SIMD if spread", 64, 1024, src, out);
    return 0;
}
