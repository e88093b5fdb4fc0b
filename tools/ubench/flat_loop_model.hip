// Step-by-step model of the flat threshold kernel's tile loop (64 queries x 32 rows x 96 dims per tile and wave): which ingredient costs what.
//   hipcc --offload-arch=gfx950 -O3 -o flat_loop_model tools/ubench/flat_loop_model.hip && ./flat_loop_model
// F bits: 1 = accumulators start from a per-tile constant and are read at the end of the tile (drain), 2 = B fragments re-read from LDS,
//         4 = tile staged into LDS from registers (ds_write), 8 = tile streamed from global memory (6 KB per tile), 16 = 32 compares + mask ORs,
//         32 = the row norm of every tile loaded from global memory (the accumulators' constant), 64 = the product's eight group masks with the
//         (never taken) append path behind them instead of one OR-ed mask.
// Occupancy is set like the product kernel's (2 waves per SIMD) through the dynamic LDS size.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16v_t __attribute__((ext_vector_type(16)));
constexpr int KS = 6;

template <int F>
__global__ __launch_bounds__(64) void tile_loop(const unsigned char* __restrict__ x, size_t x_bytes, const float* __restrict__ thr_in, float* __restrict__ out, int tiles, int share) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x, half = lane >> 5, col = lane & 31u;
    constexpr uint32_t ROWB = KS * 32u, PITCH = ROWB + 16u;
    unsigned char* btile = smem; // 2 x 32 x PITCH
    half8_t afr0[KS], afr1[KS];
    for (int s = 0; s < KS; ++s) {
        afr0[s] = *reinterpret_cast<const half8_t*>(x + (static_cast<size_t>(lane) * 192 + s * 32) % 4096);
        afr1[s] = *reinterpret_cast<const half8_t*>(x + (static_cast<size_t>(lane) * 192 + s * 32 + 16) % 4096);
    }
    float thr[2][16];
    for (int q = 0; q < 2; ++q)
        for (int r = 0; r < 16; ++r) thr[q][r] = thr_in[(q * 16 + r + lane) & 63];
    uint32_t lds_off[KS];
    for (int i = 0; i < KS; ++i) {
        const uint32_t g = static_cast<uint32_t>(i) * 1024u + lane * 16u;
        lds_off[i] = (g / ROWB) * PITCH + (g % ROWB);
    }
    // share = 0: every wave streams its own region (HBM-bound); share = S > 0: like the product, waves with the same blockIdx.x / (gridDim.x / S)
    // stream the same slice (157 query blocks per slice at S = 13)
    const size_t own = share > 0 ? (blockIdx.x / ((gridDim.x + share - 1) / share)) : blockIdx.x;
    const size_t wave_base = (own * 7919u * 6144u) % (x_bytes - static_cast<size_t>(tiles + 2) * 6144u);
    const unsigned char* base = x + (wave_base & ~size_t(15)) + lane * 16u;
    half8_t gnext[KS], bcur[KS];
    for (int i = 0; i < KS; ++i) gnext[i] = *reinterpret_cast<const half8_t*>(base + i * 1024);
    for (int i = 0; i < KS; ++i) *reinterpret_cast<half8_t*>(&btile[lds_off[i]]) = gnext[i];
    for (int i = 0; i < KS; ++i) bcur[i] = gnext[i];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    float16v_t acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.0f; acc1[i] = 0.0f; }
    unsigned long long hits = 0;
    uint32_t buf = 0;
    float xn = thr_in[lane];
    for (int t = 0; t < tiles; ++t, buf ^= 1u) {
        float xn_next = xn;
        if (F & 8) {
            const unsigned char* p = base + static_cast<size_t>(t + 1) * 6144u;
#pragma unroll
            for (int i = 0; i < KS; ++i) gnext[i] = *reinterpret_cast<const half8_t*>(p + i * 1024);
        }
        if (F & 32) xn_next = -0.5f * reinterpret_cast<const float*>(x)[((wave_base >> 2) + static_cast<size_t>(t + 1) * 32u + col) & ((x_bytes >> 2) - 1)];
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (F & 2) {
#pragma unroll
            for (int s = 0; s < KS; ++s) bcur[s] = *reinterpret_cast<const half8_t*>(&btile[buf * 32 * PITCH + col * PITCH + 32u * s + 16u * half]);
        }
        if (F & 1) {
            float16v_t cin;
#pragma unroll
            for (int i = 0; i < 16; ++i) cin[i] = xn;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[0], bcur[0], cin, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[0], bcur[0], cin, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[0], bcur[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[0], bcur[0], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[s], bcur[s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[s], bcur[s], acc1, 0, 0, 0);
        }
        if (F & 64) {
            unsigned long long gm[2][4], anym = 0;
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float16v_t& ac = q == 0 ? acc0 : acc1;
                    gm[q][g4] = __ballot(ac[4 * g4] >= thr[q][4 * g4]) | __ballot(ac[4 * g4 + 1] >= thr[q][4 * g4 + 1]) | __ballot(ac[4 * g4 + 2] >= thr[q][4 * g4 + 2]) |
                                __ballot(ac[4 * g4 + 3] >= thr[q][4 * g4 + 3]);
                    anym |= gm[q][g4];
                }
            if (anym != 0ull) {
                uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem + 2 * 32 * PITCH);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        if (gm[q][g4] == 0ull) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float v = q == 0 ? acc0[4 * g4 + j] : acc1[4 * g4 + j];
                            if (v >= thr[q][4 * g4 + j]) {
                                const uint32_t pos = atomicAdd(&lcnt[q * 32 + 8 * g4 + j + 4 * half], 1u);
                                if (pos < 512u) out[64 + ((blockIdx.x * 64u + q * 32 + 8 * g4 + j + 4 * half) * 512u + pos) % 4096u] = v;
                            }
                        }
                    }
                hits += 1;
            }
        } else if (F & 16) {
            unsigned long long m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) m |= __ballot(acc0[r] >= thr[0][r]) | __ballot(acc1[r] >= thr[1][r]);
            hits += m != 0ull ? 1u : 0u;
        } else if (F & 1) {
            if (acc0[0] + acc1[15] == 123.456f) hits += 1;
        }
        if (F & 4) {
#pragma unroll
            for (int i = 0; i < KS; ++i) *reinterpret_cast<half8_t*>(&btile[(buf ^ 1u) * 32 * PITCH + lds_off[i]]) = gnext[i];
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        if (F & 32) xn = xn_next;
        else if (F & 1) xn = xn * 1.0001f;
    }
    float s = static_cast<float>(hits);
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < KS; ++i) s += static_cast<float>(gnext[i][0]);
    if (s == 123.456f) out[0] = s;
}

template <int F> static void run(const unsigned char* x, size_t x_bytes, const float* thr, float* out, int share = 0) {
    const int tiles = 2400, blocks = 2048;
    const size_t lds = 20 * 1024; // 8 workgroups per CU = 2 waves per SIMD
    auto kern = tile_loop<F>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, x, x_bytes, thr, out, 100, share);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, x, x_bytes, thr, out, tiles, share);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kern));
    const double flop = static_cast<double>(blocks) * tiles * 12.0 * 32768.0;
    printf("{\"share\": %d, \"features\": %d, \"norm_load\": %d, \"group_masks\": %d, \"drain\": %d, \"lds_read\": %d, \"lds_write\": %d, \"global\": %d, \"compares\": %d, \"vgprs\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"ns_per_tile_per_wave\": %.1f}\n", share, F, (F >> 5) & 1, (F >> 6) & 1,
           F & 1, (F >> 1) & 1, (F >> 2) & 1, (F >> 3) & 1, (F >> 4) & 1, at.numRegs, ms, flop / (ms * 1e-3) / 1e12, ms * 1e6 / tiles);
}

// software-pipelined form: the MFMAs of tile t + 1 are issued before the compares of tile t (two accumulator sets, two tiles per trip)
template <int F>
__global__ __launch_bounds__(64) void tile_loop_pipelined(const unsigned char* __restrict__ x, size_t x_bytes, const float* __restrict__ thr_in, float* __restrict__ out, int tiles, int share) {
    extern __shared__ __align__(16) unsigned char smem[];
    const uint32_t lane = threadIdx.x, half = lane >> 5, col = lane & 31u;
    constexpr uint32_t ROWB = KS * 32u, PITCH = ROWB + 16u;
    unsigned char* btile = smem;
    half8_t afr0[KS], afr1[KS];
    for (int s = 0; s < KS; ++s) {
        afr0[s] = *reinterpret_cast<const half8_t*>(x + (static_cast<size_t>(lane) * 192 + s * 32) % 4096);
        afr1[s] = *reinterpret_cast<const half8_t*>(x + (static_cast<size_t>(lane) * 192 + s * 32 + 16) % 4096);
    }
    uint32_t lds_off[KS];
    for (int i = 0; i < KS; ++i) {
        const uint32_t g = static_cast<uint32_t>(i) * 1024u + lane * 16u;
        lds_off[i] = (g / ROWB) * PITCH + (g % ROWB);
    }
    float* ltau = reinterpret_cast<float*>(smem + 2 * 32 * PITCH);
    ltau[lane] = thr_in[lane];
    const size_t own = share > 0 ? (blockIdx.x / ((gridDim.x + share - 1) / share)) : blockIdx.x;
    const size_t wave_base = (own * 7919u * 6144u) % (x_bytes - static_cast<size_t>(tiles + 4) * 6144u);
    const unsigned char* base = x + (wave_base & ~size_t(15)) + lane * 16u;
    half8_t gnext[KS];
    for (int i = 0; i < KS; ++i) gnext[i] = *reinterpret_cast<const half8_t*>(base + i * 1024);
    for (int i = 0; i < KS; ++i) *reinterpret_cast<half8_t*>(&btile[lds_off[i]]) = gnext[i];
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    unsigned long long hits = 0;
    float xn = thr_in[lane];
    auto mfma_tile = [&](uint32_t buf, float16v_t& acc0, float16v_t& acc1) {
        float16v_t cin;
#pragma unroll
        for (int i = 0; i < 16; ++i) cin[i] = xn;
        half8_t b = *reinterpret_cast<const half8_t*>(&btile[buf * 32 * PITCH + col * PITCH + 16u * half]);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[0], b, cin, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[0], b, cin, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            b = *reinterpret_cast<const half8_t*>(&btile[buf * 32 * PITCH + col * PITCH + 32u * s + 16u * half]);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[s], b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[s], b, acc1, 0, 0, 0);
        }
    };
    auto compares = [&](const float16v_t& acc0, const float16v_t& acc1) {
        unsigned long long m = 0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const float4 t0 = *reinterpret_cast<const float4*>(&ltau[8 * g4 + 4 * static_cast<int>(half)]);
            const float4 t1 = *reinterpret_cast<const float4*>(&ltau[32 + 8 * g4 + 4 * static_cast<int>(half)]);
            m |= __ballot(acc0[4 * g4] >= t0.x) | __ballot(acc0[4 * g4 + 1] >= t0.y) | __ballot(acc0[4 * g4 + 2] >= t0.z) | __ballot(acc0[4 * g4 + 3] >= t0.w);
            m |= __ballot(acc1[4 * g4] >= t1.x) | __ballot(acc1[4 * g4 + 1] >= t1.y) | __ballot(acc1[4 * g4 + 2] >= t1.z) | __ballot(acc1[4 * g4 + 3] >= t1.w);
        }
        hits += m != 0ull ? 1u : 0u;
    };
    auto advance = [&](int t, uint32_t buf) { // stage the tile that is in registers, request the one after
#pragma unroll
        for (int i = 0; i < KS; ++i) *reinterpret_cast<half8_t*>(&btile[buf * 32 * PITCH + lds_off[i]]) = gnext[i];
        const unsigned char* p = base + static_cast<size_t>(t) * 6144u;
#pragma unroll
        for (int i = 0; i < KS; ++i) gnext[i] = *reinterpret_cast<const half8_t*>(p + i * 1024);
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
    };
    float16v_t a0, a1, b0, b1;
    {
        const unsigned char* p = base + 6144u;
        for (int i = 0; i < KS; ++i) gnext[i] = *reinterpret_cast<const half8_t*>(p + i * 1024);
    }
    mfma_tile(0u, a0, a1);
    for (int t = 0; t + 1 < tiles; t += 2) {
        advance(t + 2, 1u);
        mfma_tile(1u, b0, b1);
        compares(a0, a1);
        xn = xn * 1.0001f;
        advance(t + 3, 0u);
        mfma_tile(0u, a0, a1);
        compares(b0, b1);
    }
    float s = static_cast<float>(hits);
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + b0[i] + b1[i];
    for (int i = 0; i < KS; ++i) s += static_cast<float>(gnext[i][0]);
    if (s == 123.456f) out[0] = s;
}

template <int F> static void run_pipelined(const unsigned char* x, size_t x_bytes, const float* thr, float* out, int share) {
    const int tiles = 2400, blocks = 2048;
    const size_t lds = 20 * 1024;
    auto kern = tile_loop_pipelined<F>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, x, x_bytes, thr, out, 100, share);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), lds, 0, x, x_bytes, thr, out, tiles, share);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    hipFuncAttributes at;
    (void)hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kern));
    const double flop = static_cast<double>(blocks) * tiles * 12.0 * 32768.0;
    printf("{\"share\": %d, \"form\": \"pipelined: MFMAs of tile t+1 before the compares of tile t, thresholds from LDS\", \"vgprs\": %d, \"ms\": %.3f, \"tflops\": %.1f}\n", share, at.numRegs, ms,
           flop / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t x_bytes = size_t(192) << 20; // the float16 shadow of 1 M x 96
    unsigned char* x;
    float *thr, *out;
    (void)hipMalloc(&x, x_bytes);
    (void)hipMalloc(&thr, 256);
    (void)hipMalloc(&out, 64 * 4 + 4096 * 4);
    (void)hipMemset(x, 0x11, x_bytes);
    std::vector<float> h(64, 1.0e30f);
    (void)hipMemcpy(thr, h.data(), 256, hipMemcpyHostToDevice);
    run<0>(x, x_bytes, thr, out);
    run<1>(x, x_bytes, thr, out);
    run<1 | 2>(x, x_bytes, thr, out);
    run<1 | 2 | 4>(x, x_bytes, thr, out);
    run<1 | 2 | 4 | 8>(x, x_bytes, thr, out);
    run<1 | 2 | 4 | 8 | 16>(x, x_bytes, thr, out);
    run<1 | 16>(x, x_bytes, thr, out);
    run<1 | 8>(x, x_bytes, thr, out);
    run<8>(x, x_bytes, thr, out);
    run<1 | 2 | 4 | 8>(x, x_bytes, thr, out, 13);
    run<1 | 2 | 4 | 8 | 16>(x, x_bytes, thr, out, 13);
    run<1 | 2 | 4 | 8>(x, x_bytes, thr, out, 1);
    run<1 | 2 | 4 | 8>(x, x_bytes, thr, out, 64);
    run<1 | 2 | 4 | 8>(x, x_bytes, thr, out, 256);
    run<1 | 2 | 4 | 8 | 16 | 32>(x, x_bytes, thr, out, 13);
    run<1 | 2 | 4 | 8 | 64>(x, x_bytes, thr, out, 13);
    run<1 | 2 | 4 | 8 | 32 | 64>(x, x_bytes, thr, out, 13);
    run_pipelined<0>(x, x_bytes, thr, out, 13);
    run_pipelined<0>(x, x_bytes, thr, out, 0);
    return 0;
}
