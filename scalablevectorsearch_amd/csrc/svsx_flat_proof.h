// Flat search, stages 4 and 5 (svsx_flat.hip): proof that the re-ranked float16 shortlist is the exhaustive answer, and the exhaustive
// float32 pass for the queries it cannot be proven for.  Host-side declarations only (svsx_api.hip <-> svsx_flat.hip).
#pragma once
#include "svsx_kernels.h"

namespace svsx {

// exhaustive pass: only the queries with redo[q] != 0 are recomputed, against all n rows (r.cand_* unused)
struct FlatExactArgs {
    RerankArgs r;
    const uint32_t* redo;
    uint32_t n;
};
hipError_t launch_flat_exact(int data_dtype, int metric, int qmode, const FlatExactArgs& a, hipStream_t stream);
// Proof of exactness of the float16 shortlist, one query at a time (svsx_flat.hip stage 4): redo[q] = 1 unless every row that was NOT
// re-ranked provably has a reference distance strictly worse than the k-th re-ranked one.
struct FlatProofArgs {
    int metric;
    uint32_t nq, k, kk, d, dpad;
    const void* queries;
    size_t q_row_bytes;
    int q_dtype;
    int sq_l2;                   // the reference compresses the query too (EuclideanCompressed, scalar.h:88-94): data_dtype = code type
    int data_dtype;
    float sq_scale, sq_bias;
    const uint64_t* out_ids;     // [nq][k] re-ranked answer
    const float* out_dists;
    const float* short_keys;     // [nq][kk] merged shortlist keys (ascending) and ids (~0 = empty slot)
    const uint64_t* short_ids;
    const float* tau;            // threshold of the last threshold pass per query, or NULL (shortlist pipeline)
    const uint32_t* cnt;         // [nq][n_lists] list lengths of that pass (overflow => the query used the shortlist)
    uint32_t cap, n_lists;
    float xmax2, xmin2;          // max / min squared norm of the float16 shadow rows
    uint32_t* redo;              // [nq] out
    uint32_t* redo_count;        // optional: += number of queries flagged
};
hipError_t launch_flat_certify(const FlatProofArgs& a, hipStream_t stream);
// the overflow gate with the re-rank capacity of a kk-entry shortlist (max(2048, 32 kk) candidates per query)
hipError_t launch_flat_gate_kk(const uint32_t* cnt, uint32_t nq, uint32_t n_slices, uint32_t cap, uint32_t kk, uint32_t* block_gate, hipStream_t stream);
// which padded dimensions the threshold pipeline takes (d <= 128, or a multiple of 128 up to 768) and at how many waves per SIMD
bool flat_thresh_supports(uint32_t dpad);
uint32_t flat_thresh_waves_per_simd(uint32_t dpad);
uint32_t flat_thresh_slices(int metric, uint32_t dpad, size_t nq);
uint32_t flat_thresh_unit_rows(uint32_t dpad);        // rows of one sampled unit (a 32-row tile, or a 64- / 128-row step)
uint32_t flat_thresh_lists_per_slice(uint32_t dpad);  // candidate lists per (query, slice)
constexpr size_t kShadowPadRows = 160; // rows of zeros behind the float16 shadow: a partial last 128-row step (+ a 32-row tile) stays inside the allocation
hipError_t launch_norm_range(const float* norm2, uint32_t n, uint32_t* d_max_min_bits, hipStream_t stream); // [0] = max bits (init 0), [1] = min bits (init ~0)

} // namespace svsx
