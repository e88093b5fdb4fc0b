// svsx_host.h -- what the host-side translation units of libsvsx.so share (svsx_api.hip: index life cycle and the search entry points;
// svsx_records.hip: bound records / code rows; svsx_flat_host.hip: the flat-search pipeline): the index object, the per-call staging
// contexts, error and allocation helpers.  Internal: nothing outside csrc/ includes it.
#pragma once

#include "../../include/svsx.h"
#include "svsx_device.h"
#include "svsx_kernels.h"
#include "svsx_flat_proof.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <random>
#include <string>
#include <vector>

using namespace svsx;

struct svsx_index {
    int metric = 0;
    int data_dtype = 0;
    size_t n = 0, d = 0;
    void* d_data = nullptr;
    size_t row_bytes = 0;
    uint32_t* d_graph = nullptr;
    size_t gstride = 0;
    uint32_t* d_entry = nullptr;
    size_t n_entry = 0;
    int sq_enabled = 0;
    float sq_scale = 1.0f, sq_bias = 0.0f;
    float* d_sq_row_norm = nullptr; // scalar-quantized rows under cosine: sqrtf(||decompressed row||^2), reference order (launch_sq_row_norms)
    int device = 0;
    uint64_t id_offset = 0;
    bool owns = true;
    // lazily built float16 shadow for the MFMA flat-search filter (svsx_flat.hip)
    mutable std::mutex shadow_mutex;
    mutable void* d_half = nullptr;
    mutable float* d_norm2 = nullptr;
    mutable float shadow_max2 = 0.0f, shadow_min2 = 0.0f; // range of d_norm2 (flat search: exactness proof)
    // bound codes (svsx_index_build_bound_codes): 128 B per vector
    void* d_codes = nullptr;
    float code_scale = 1.0f, code_bias = 0.0f, code_eq_abs = 0.0f;
    // graph-inline records (svsx_index_build_bound_codes mode 2; svsx_codes.hip)
    void* d_fat = nullptr;
    size_t fat_stride = 0;
    uint32_t fat_rb = 32;
    uint32_t fat_degshift = 0; // 26 when the blocks' id words carry the neighbours' degree classes (n <= 2^26), else 0
    float* d_pca = nullptr;      // [32][dpad] leading directions (float32, row-major, zero padded) | [dpad] mean | [32] bias
    float pca_step = 1.0f, pca_tail_unit = 1.0f, pca_delta = 0.0f, pca_eq_abs = 0.0f, pca_ref_extra = 0.0f, pca_mean_norm = 0.0f;
    uint32_t pca_mean_dir = 0;   // similarity metrics: coordinate 0 of the basis is the unit mean direction; the kernel keeps the query's coordinate 0 in float32
    float calib_ms_records = 0.0f, calib_ms_plain = 0.0f; // mode 3's timing of records against the plain kernel (svsx_records.hip)
    // host-pointer calls (svsx_search): persistent staging contexts, one per concurrent caller (HostCallContext below)
    mutable std::mutex ctx_mutex;
    mutable std::vector<struct HostCallContext*> ctx_free;
};

namespace svsx { // defined in svsx_api.hip
int set_error(int code, const char* fmt, ...);
void index_facts(const svsx_index* ix, int* metric, int* device, size_t* d);
} // namespace svsx

constexpr uint32_t kDefaultFilterEntries = 2048;
constexpr uint32_t kPcaRows = 128;  // rows of the float32 device copy of the directions (rows >= dc are zero)

// records the message for svsx_last_error() (thread-local) and returns `code`
template <typename... A> inline int fail(int code, const char* fmt, A... a) { return svsx::set_error(code, fmt, a...); }


#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : dt == DT_F16 ? 2 : (dt == DT_I8 || dt == DT_U8) ? 1 : 0; }

// Which kernel family serves (query dtype, data dtype) -- the reference's AVX-512 specialisation table
// (euclidean.h:318-358 and the VNNI pairs :291-313); everything else has only a scalar loop there (generic_l2 and its
// siblings: utils/search_index.cpp:147-162 offers six such pairs) and takes the scalar-loop kernels here (QM_SCALAR, round 5).
inline int select_qmode(int metric, int q_dtype, int data_dtype, int sq_enabled) {
    if (sq_enabled) {
        if (data_dtype != DT_I8 && data_dtype != DT_U8) return -1;
        if (q_dtype != DT_F32 && q_dtype != DT_F16) return -1;
        return metric == M_L2 ? QM_SQ_L2 : metric == M_IP ? QM_SQ_IP : QM_SQ_COS;
    }
    if ((q_dtype == DT_I8 && data_dtype == DT_I8) || (q_dtype == DT_U8 && data_dtype == DT_U8)) return QM_INT;
    if (q_dtype == DT_F32) return QM_FLOAT;
    if (q_dtype == DT_F16 && (data_dtype == DT_F32 || data_dtype == DT_F16)) return QM_FLOAT;
    return QM_SCALAR;
}

inline size_t padded_row_bytes(size_t d, int dt) { return ((d + 15) & ~size_t(15)) * dtype_size(dt); }

struct DeviceBuffer {
    void* p = nullptr;
    ~DeviceBuffer() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

// Temporary buffer allocated and released in stream order (no device-wide synchronisation).
struct StreamBuffer {
    void* p = nullptr;
    hipStream_t s = nullptr;
    ~StreamBuffer() {
        if (p) (void)hipFreeAsync(p, s);
    }
    hipError_t alloc(size_t bytes, hipStream_t stream) {
        s = stream;
        return hipMallocAsync(&p, bytes ? bytes : 1, stream);
    }
};

// A private stream per host call: concurrent callers (index.h:541-545 contract) never serialise on the default stream.
struct CallStream {
    hipStream_t s = nullptr;
    ~CallStream() {
        if (s) (void)hipStreamDestroy(s);
    }
    hipError_t create() { return hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
};


// Everything a host-pointer svsx_search needs between calls: streams and device staging, grown on demand and kept with the index -- the
// per-call hipStreamCreate / hipMallocAsync of round 2 cost 0.65 ms per 10 000-query batch.
// The batch is cut into chunks, each on its own stream: H2D of chunk i+1 and D2H of chunk i-1 overlap the kernel of chunk i (queries are
// independent, index/vamana/index.h:571-574, so the results are those of one launch).
struct HostCallContext {
    static constexpr int kMaxChunks = 8;
    hipStream_t streams[kMaxChunks] = {};
    void* d_in = nullptr;   size_t d_in_cap = 0;    // queries
    void* d_out = nullptr;  size_t d_out_cap = 0;   // ids64 | ids32 | dists | stats | history ids | history dists | history len
    bool dirty = false;                             // an error return left work in flight on the streams: drained before the context is reused
    static constexpr size_t kKeepBytes = size_t(64) << 20; // staging beyond this is given back when a call that needed it has returned
    ~HostCallContext() {
        for (auto st : streams) if (st) (void)hipStreamDestroy(st);
        release();
    }
    void release() {
        if (d_in) (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        d_in = d_out = nullptr;
        d_in_cap = d_out_cap = 0;
    }
    void drain() { // every stream idle: nothing reads or writes the staging any more
        for (auto st : streams) if (st) (void)hipStreamSynchronize(st);
        dirty = false;
    }
    void trim() {  // one huge batch must not keep gigabytes of staging for the life of the index
        if (d_in_cap + d_out_cap > 2 * kKeepBytes) { drain(); release(); }
    }
    hipError_t reserve(size_t in_bytes, size_t out_bytes, int n_streams) {
        for (int i = 0; i < n_streams; ++i)
            if (!streams[i]) { hipError_t e = hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking); if (e != hipSuccess) return e; }
        auto grow = [](void*& p, size_t& cap, size_t want) -> hipError_t {
            if (want <= cap) return hipSuccess;
            if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
            const size_t bytes = want + want / 4 + 4096;
            hipError_t e = hipMalloc(&p, bytes);
            if (e == hipSuccess) cap = bytes;
            return e;
        };
        hipError_t e = grow(d_in, d_in_cap, in_bytes);
        if (e == hipSuccess) e = grow(d_out, d_out_cap, out_bytes);
        return e;
    }
    hipError_t stream0() { return streams[0] ? hipSuccess : hipStreamCreateWithFlags(&streams[0], hipStreamNonBlocking); }
};


// borrowed for the duration of one call; concurrent callers get one each (index/vamana/index.h:541-545)
struct ContextLease {
    const svsx_index* ix;
    HostCallContext* c = nullptr;
    explicit ContextLease(const svsx_index* index) : ix(index) {
        {
            std::lock_guard<std::mutex> lock(ix->ctx_mutex);
            if (!ix->ctx_free.empty()) { c = ix->ctx_free.back(); ix->ctx_free.pop_back(); }
        }
        if (!c) c = new (std::nothrow) HostCallContext();
        if (c && c->dirty) c->drain();
    }
    ~ContextLease() {
        if (!c) return;
        if (c->dirty) c->drain(); // an early return with copies / kernels still in flight: the next caller must not find an active DMA on the staging
        c->trim();
        std::lock_guard<std::mutex> lock(ix->ctx_mutex);
        ix->ctx_free.push_back(c);
    }
};

inline int validate_common(int metric, int dtype, size_t n, size_t d) {
    if (metric < 0 || metric > 2) return fail(SVSX_INVALID_ARGUMENT, "unknown metric %d", metric);
    if (dtype_size(dtype) == 0) return fail(SVSX_INVALID_ARGUMENT, "unknown data dtype %d", dtype);
    if (n == 0 || d == 0) return fail(SVSX_INVALID_ARGUMENT, "empty dataset (n=%zu, d=%zu)", n, d);
    if (n >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "n=%zu exceeds 2^31-1 ids per index (shard it)", n);
    return SVSX_SUCCESS;
}



// argument checks shared by every search entry point
inline int check_search_args(const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, bool need_graph = true) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (need_graph && !ix->d_graph) return fail(SVSX_INVALID_ARGUMENT, "this index was created without a graph (flat search only)");
    if (k == 0) return fail(SVSX_INVALID_ARGUMENT, "k must be greater than 0");
    if (dtype_size(q_dtype) == 0) return fail(SVSX_INVALID_ARGUMENT, "unknown query dtype %d", q_dtype);
    if (nq > 0 && !queries) return fail(SVSX_INVALID_ARGUMENT, "null queries");
    if (nq >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "too many queries in one batch");
    return SVSX_SUCCESS;
}

