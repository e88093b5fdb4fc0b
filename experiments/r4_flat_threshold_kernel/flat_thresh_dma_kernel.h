// Round-4 experiment, not in libsvsx.so: the threshold kernel with B tiles DMA-ed straight into LDS (global_load_lds_dwordx4, source-side
// swizzle), three waves per SIMD.  Bit-exact (48 flat GPU tests green), slower: 6.5 ms (2 waves) / 7.3 ms (3 waves) per 10 000 x 1 M x 96 search
// against 5.1-5.2 ms for the register-staged kernel -- the kernel is bound by bytes through the vector-memory path per MFMA, not by latency
// (profiles/r04_notes.md section 7).  Drop-in next to flat_thresh_kernel in svsx_flat.hip.
template <int METRIC, int KS>
__global__ __launch_bounds__(64, 3) void flat_thresh_dma_kernel(const FlatThreshArgs a) {
    __shared__ uint32_t lcnt[64];
    __shared__ __align__(16) float ltau[64];
    // B tiles go global -> LDS directly (global_load_lds_dwordx4: KS instructions of 1 KB per tile, no staging registers and no ds_write
    // pass), which leaves room for three waves per SIMD.  The LDS image of such a load is lane-linear, so the conflict-free layout comes
    // from the SOURCE side: 16-byte piece p of row r (P = 2 KS pieces per row) is stored at slot r * P + (p + r) mod P, i.e. the lane that
    // writes slot s fetches piece (s mod P - s / P) mod P of row s / P; the fragment reads apply the same rotation.
    __shared__ __align__(16) unsigned char btile[2][32 * KS * 32];
    __shared__ float bnorm[2][64];
    const uint32_t lane = threadIdx.x;
    const uint32_t q0 = blockIdx.x * 64u;
    const uint32_t slice = blockIdx.y;
    const uint32_t dpad = a.dpad;
    const uint32_t rows_per_slice = ((a.n + a.n_slices - 1) / a.n_slices + 31u) & ~31u;
    const uint32_t x_lo = slice * rows_per_slice;
    const uint32_t x_hi = (x_lo + rows_per_slice) < a.n ? (x_lo + rows_per_slice) : a.n;
    const uint32_t xs = 32u * (a.tile_step ? a.tile_step : 1u);
    const uint32_t x_first = ((x_lo + xs - 1u) / xs) * xs;
    lcnt[lane] = 0u;
    lds_order();
    const uint32_t half = lane >> 5, col = lane & 31u;
    const _Float16* xh = static_cast<const _Float16*>(a.xh);
    const _Float16* qh = static_cast<const _Float16*>(a.qh);

    if (x_lo < x_hi && x_first < x_hi) {
        half8_t afr0[KS], afr1[KS];
        {
            const uint32_t r0 = q0 + col, r1 = q0 + 32u + col;
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) {
                half8_t z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = static_cast<_Float16>(0.0f);
                afr0[s2] = r0 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r0) * dpad + 16 * s2 + 8u * half) : z;
                afr1[s2] = r1 < a.nq ? *reinterpret_cast<const half8_t*>(qh + static_cast<size_t>(r1) * dpad + 16 * s2 + 8u * half) : z;
            }
        }
        // Thresholds in the accumulator's domain, so that the accept test is ONE compare per cell: the accumulators start from the row's
        // constant (L2: -|x|^2 / 2, through the MFMA's C operand) and a cell is accepted iff acc >= thr,
        //   L2: key = |x|^2 - 2 q.x = -2 acc <= tau  <=>  acc >= -tau / 2;   IP: key = -acc <= tau  <=>  acc >= -tau;
        //   cosine: key = -(acc * rinv) <= tau  <=>  acc * rinv >= -tau.
        // The scalings are exact, so the accepted set is exactly {key <= tau} for the key that is stored and selected from.
        // -inf for padding queries' tau (thr = +inf: never accepted).  A lane's 32 cells read their thresholds as 8 float4 from LDS (registers are kept for a third wave per SIMD).
        {
            const float t = q0 + lane < a.nq ? a.tau[static_cast<size_t>(q0 + lane) * a.tau_stride] : -__builtin_inff();
            ltau[lane] = METRIC == M_L2 ? -0.5f * t : -t;
#ifdef SVSX_FLAT_DIAG_NO_HITS   // timing builds only: nothing is ever accepted at the full-set level
            if (a.tile_step == 1) ltau[lane] = __builtin_inff();
#endif
        }
        lds_order();
        constexpr uint32_t ROWB = KS * 32u;       // bytes of one f16 row (dpad * 2)
        constexpr uint32_t P = 2u * KS;           // 16-byte pieces per row
        uint32_t src_off[KS];                      // byte offset inside the tile of the piece this lane fetches with instruction i
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const uint32_t slot = static_cast<uint32_t>(i) * 64u + lane;
            const uint32_t r = slot / P, qpos = slot % P;
            src_off[i] = r * ROWB + 16u * ((qpos + P - r % P) % P);
        }
        uint32_t frag_off[KS];                     // where this lane's fragment of k-step s2 lives in a tile image
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) frag_off[s2] = 16u * (col * P + (2u * static_cast<uint32_t>(s2) + half + col % P) % P);
        const unsigned char* xbytes = reinterpret_cast<const unsigned char*>(xh);
        typedef __attribute__((address_space(3))) void* lds_ptr_t;
        typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
        // a tile is 32 whole rows = KS contiguous KB (the shadow is padded by one tile: the last, partial tile reads initialised bytes; its
        // surplus rows are masked when a cell is appended)
        auto request_tile = [&](uint32_t xt, uint32_t buf) {
            const unsigned char* base = xbytes + static_cast<size_t>(xt) * ROWB;
#pragma unroll
            for (int i = 0; i < KS; ++i)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + src_off[i]), (lds_ptr_t)(&btile[buf][static_cast<uint32_t>(i) * 1024u]), 16, 0, 0);
            if constexpr (METRIC != M_IP) {
                const uint32_t r = xt + col;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(a.xnorm2 + (r < a.n ? r : a.n - 1u)), (lds_ptr_t)(&bnorm[buf][0]), 4, 0, 0);
            }
        };
        auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        auto fragments = [&](uint32_t buf, half8_t (&dst)[KS], float& xnv) {
#pragma unroll
            for (int s2 = 0; s2 < KS; ++s2) dst[s2] = *reinterpret_cast<const half8_t*>(&btile[buf][frag_off[s2]]);
            if constexpr (METRIC != M_IP) xnv = bnorm[buf][col];
        };
        const size_t stride_q = static_cast<size_t>(a.n_slices) * a.cap;                       // list stride between queries
        const size_t lane_base = (static_cast<size_t>(q0 + 4u * half) * a.n_slices + slice) * a.cap; // this lane's query group
        auto consume = [&](uint32_t x0, const half8_t (&b)[KS], float xn) {
            float16v_t acc0, acc1;
            {
                const float c0 = METRIC == M_L2 ? -0.5f * xn : 0.0f;
                float16v_t cin;
#pragma unroll
                for (int i = 0; i < 16; ++i) cin[i] = c0;
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[0], b[0], cin, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[0], b[0], cin, 0, 0, 0);
            }
#pragma unroll
            for (int s2 = 1; s2 < KS; ++s2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr0[s2], b[s2], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(afr1[s2], b[s2], acc1, 0, 0, 0);
            }
            float rinv = 1.0f;
            if constexpr (METRIC == M_COS) rinv = 1.0f / sqrtf(xn > 0.0f ? xn : 1.0f);
            auto score = [&](float v) -> float { // what is compared with thr
                if constexpr (METRIC == M_COS) return v * rinv;
                else return v;
            };
            auto keyof = [&](float v) -> float { // the key that is stored (smaller = better), an exact scaling of the score
                if constexpr (METRIC == M_L2) return -2.0f * v;
                else return -score(v);
            };
            // accept test: one compare per cell, folded into 8 wave-level masks (4 cells each); the append path only
            // revisits the groups in which some lane hit
            unsigned long long gm[2][4];
            unsigned long long anym = 0ull;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    // four independent compares OR-ed as masks (no short circuit: `||` compiles to exec-mask branches here)
                    const float4 t4 = *reinterpret_cast<const float4*>(&ltau[qt * 32 + 8 * g4 + 4 * static_cast<int>(half)]);
                    const bool h0 = score(qt == 0 ? acc0[4 * g4 + 0] : acc1[4 * g4 + 0]) >= t4.x;
                    const bool h1 = score(qt == 0 ? acc0[4 * g4 + 1] : acc1[4 * g4 + 1]) >= t4.y;
                    const bool h2 = score(qt == 0 ? acc0[4 * g4 + 2] : acc1[4 * g4 + 2]) >= t4.z;
                    const bool h3 = score(qt == 0 ? acc0[4 * g4 + 3] : acc1[4 * g4 + 3]) >= t4.w;
                    gm[qt][g4] = __ballot(h0) | __ballot(h1) | __ballot(h2) | __ballot(h3);
                    anym |= gm[qt][g4];
                }
            }
            if (anym == 0ull) return;
            const uint32_t row = x0 + col;
            const bool valid = row < x_hi; // lanes of a partial last tile hold rows of the next slice / the padding
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    if (gm[qt][g4] == 0ull) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = 4 * g4 + j;
                        const float v = qt == 0 ? acc0[r] : acc1[r];
                        const uint32_t cell = static_cast<uint32_t>(qt) * 32u + static_cast<uint32_t>(j + 8 * g4);
                        const bool hit = valid && score(v) >= ltau[cell + 4u * half];
                        if (hit) {
                            const uint32_t pos = atomicAdd(&lcnt[cell + 4u * half], 1u);
                            if (pos < a.cap) {
                                const size_t o = lane_base + static_cast<size_t>(cell) * stride_q + pos; // cell * stride_q is wave-uniform
                                a.cand[o] = row;
                                if (a.cand_keys != nullptr) a.cand_keys[o] = keyof(v);
                            }
                        }
                    }
                }
            }
        };
        half8_t bcur[KS];
        float xn_cur = 0.0f;
        request_tile(x_first, 0u);
        uint32_t buf = 0;
        for (uint32_t x0 = x_first; x0 < x_hi; x0 += xs, buf ^= 1u) {
            landed();                                                            // this tile (and the appends of the last one)
            fragments(buf, bcur, xn_cur);
            lds_order();
            // next tile in flight during this tile's MFMAs and epilogue (past the end: a harmless re-read of the first); the buffer it
            // lands in was read two tiles ago
            request_tile(x0 + xs < x_hi ? x0 + xs : x_first, buf ^ 1u);
            consume(x0, bcur, xn_cur);
        }
        landed();
    }
    lds_order();
    if (q0 + lane < a.nq) a.cnt[static_cast<size_t>(q0 + lane) * a.n_slices + slice] = lcnt[lane];
}

