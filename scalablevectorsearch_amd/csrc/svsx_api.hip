// svsx_api.hip -- host side of the C ABI declared in include/svsx.h (libsvsx.so).
//
// Owns the HBM-resident index (re-laid-out copies of the SVS data/graph arrays), validates arguments the way the
// reference's runtime layer does (bindings/cpp/src/vamana_index.cpp:56-72, vamana_index_impl.h:109-138) and
// launches the gfx950 kernels.  There is deliberately NO CPU fallback: without a HIP device every compute entry
// point fails with SVSX_RUNTIME_ERROR.
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
    float pca_step = 1.0f, pca_tail_unit = 1.0f, pca_delta = 0.0f, pca_eq_abs = 0.0f, pca_ref_extra = 0.0f;
    // host-pointer calls (svsx_search): persistent staging contexts, one per concurrent caller (HostCallContext below)
    mutable std::mutex ctx_mutex;
    mutable std::vector<struct HostCallContext*> ctx_free;
};

namespace {

thread_local std::string g_error;
constexpr uint32_t kDefaultFilterEntries = 2048;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}

} // namespace

namespace svsx { // for the other translation units of the library (svsx_sharded.hip)
int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
void index_facts(const svsx_index* ix, int* metric, int* device, size_t* d) {
    *metric = ix->metric;
    *device = ix->device;
    *d = ix->d;
}
} // namespace svsx

namespace {

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : dt == DT_F16 ? 2 : (dt == DT_I8 || dt == DT_U8) ? 1 : 0; }

// Which kernel family serves (query dtype, data dtype) -- the reference's AVX-512 specialisation table
// (euclidean.h:318-358 and the VNNI pairs :291-313); everything else has only a scalar loop there (generic_l2 and its
// siblings: utils/search_index.cpp:147-162 offers six such pairs) and takes the scalar-loop kernels here (QM_SCALAR, round 5).
int select_qmode(int metric, int q_dtype, int data_dtype, int sq_enabled) {
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

size_t padded_row_bytes(size_t d, int dt) { return ((d + 15) & ~size_t(15)) * dtype_size(dt); }

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

} // namespace

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

namespace {

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

int validate_common(int metric, int dtype, size_t n, size_t d) {
    if (metric < 0 || metric > 2) return fail(SVSX_INVALID_ARGUMENT, "unknown metric %d", metric);
    if (dtype_size(dtype) == 0) return fail(SVSX_INVALID_ARGUMENT, "unknown data dtype %d", dtype);
    if (n == 0 || d == 0) return fail(SVSX_INVALID_ARGUMENT, "empty dataset (n=%zu, d=%zu)", n, d);
    if (n >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "n=%zu exceeds 2^31-1 ids per index (shard it)", n);
    return SVSX_SUCCESS;
}

} // namespace

extern "C" {

const char* svsx_last_error(void) { return g_error.c_str(); }
const char* svsx_version(void) { return "svsx 0.5.0 (gfx950)"; }
int svsx_abi_version(void) { return SVSX_ABI_VERSION; }
const char* svsx_last_kernel_info(void) { return last_search_instantiation(); }

// CosineSimilarityCompressed divides by the norm of the decompressed row (scalar.h:175-184, cosine.h:334-335): a property of the row alone,
// computed once by the search kernel's own accumulation routine (same bits) instead of once per candidate and query.  Stream-ordered: the
// kernel runs on `stream`; `wait` makes the call return only when the norms are there (index creation: the NULL stream, which joins every
// blocking stream -- rows written on a NON-blocking stream must be complete before the view is created, or be followed by
// svsx_index_refresh_row_norms on that stream).  No device-wide synchronisation.
static int build_sq_row_norms(svsx_index* ix, hipStream_t stream, bool wait) {
    if (!ix->sq_enabled || ix->metric != M_COS || ix->n == 0) return SVSX_SUCCESS;
    if (!ix->d_sq_row_norm) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ix->d_sq_row_norm), ix->n * sizeof(float)));
    hipError_t e = launch_sq_row_norms(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), static_cast<uint32_t>(ix->d), ix->sq_scale, ix->sq_bias,
                                       ix->d_sq_row_norm, stream);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "row-norm kernel failed: %s", hipGetErrorString(e));
    if (wait) HIP_TRY(hipStreamSynchronize(stream));
    return SVSX_SUCCESS;
}

int svsx_index_refresh_row_norms(svsx_index* ix, void* hip_stream) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    HIP_TRY(hipSetDevice(ix->device));
    return build_sq_row_norms(ix, static_cast<hipStream_t>(hip_stream), false);
}

int svsx_index_drop_row_norms(svsx_index* ix) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    HIP_TRY(hipSetDevice(ix->device));
    if (ix->d_sq_row_norm) {
        HIP_TRY(hipDeviceSynchronize()); // searches in flight may still read them
        (void)hipFree(ix->d_sq_row_norm);
        ix->d_sq_row_norm = nullptr;
    }
    return SVSX_SUCCESS;
}

int svsx_index_create(const svsx_index_desc* desc, svsx_index** out) {
    if (!desc || !out) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (int rc = validate_common(desc->metric, desc->data_dtype, desc->n, desc->d)) return rc;
    const bool flat_only = desc->graph == nullptr; // FlatIndex (index/flat/flat.h): data only
    if (!desc->data) return fail(SVSX_INVALID_ARGUMENT, "data is required");
    if (!flat_only && (!desc->entry_points || desc->n_entry == 0))
        return fail(SVSX_INVALID_ARGUMENT, "data, graph and at least one entry point are required");
    if (!flat_only && desc->graph_row_stride_u32 < 2) return fail(SVSX_INVALID_ARGUMENT, "graph row stride must be >= 2 (degree slot + ids)");
    if (desc->sq_enabled && desc->data_dtype != SVSX_I8 && desc->data_dtype != SVSX_U8)
        return fail(SVSX_INVALID_ARGUMENT, "scalar-quantized data must be int8 or uint8");
    for (size_t i = 0; !flat_only && i < desc->n_entry; ++i)
        if (desc->entry_points[i] >= desc->n) return fail(SVSX_INVALID_ARGUMENT, "entry point %u out of range", desc->entry_points[i]);
    HIP_TRY(hipSetDevice(desc->device));

    svsx_index* ix = new (std::nothrow) svsx_index();
    if (!ix) return fail(SVSX_RUNTIME_ERROR, "out of host memory");
    ix->metric = desc->metric;
    ix->data_dtype = desc->data_dtype;
    ix->n = desc->n;
    ix->d = desc->d;
    ix->n_entry = flat_only ? 0 : desc->n_entry;
    ix->sq_enabled = desc->sq_enabled;
    ix->sq_scale = desc->sq_scale;
    ix->sq_bias = desc->sq_bias;
    ix->device = desc->device;
    ix->id_offset = desc->id_offset;
    ix->owns = true;

    auto cleanup = [&](int rc) {
        svsx_index_destroy(ix);
        return rc;
    };
    // data: rows zero padded to a multiple of 16 elements (16-B aligned chunks for every lane of a row group)
    const size_t es = dtype_size(desc->data_dtype);
    const size_t src_stride = desc->data_row_stride_bytes ? desc->data_row_stride_bytes : desc->d * es;
    if (src_stride < desc->d * es) return cleanup(fail(SVSX_INVALID_ARGUMENT, "data row stride smaller than a row"));
    ix->row_bytes = padded_row_bytes(desc->d, desc->data_dtype);
    const size_t data_bytes = ix->row_bytes * desc->n + 64; // slack
    if (hipMalloc(&ix->d_data, data_bytes) != hipSuccess) return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for data failed", data_bytes));
    if (ix->row_bytes != desc->d * es) {
        if (hipMemset(ix->d_data, 0, data_bytes) != hipSuccess) return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMemset failed"));
    }
    if (hipMemcpy2D(ix->d_data, ix->row_bytes, desc->data, src_stride, desc->d * es, desc->n, hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(SVSX_RUNTIME_ERROR, "copying data to the device failed"));

    if (int rc = build_sq_row_norms(ix, nullptr, true)) return cleanup(rc);
    if (flat_only) {
        *out = ix;
        return SVSX_SUCCESS;
    }
    // graph: SVS rows -> packed rows of `gstride` ids (multiple of 16 u32 = 64 B) padded with 0xFFFFFFFF
    size_t maxdeg = desc->max_degree ? desc->max_degree : desc->graph_row_stride_u32 - 1;
    if (maxdeg > desc->graph_row_stride_u32 - 1) maxdeg = desc->graph_row_stride_u32 - 1;
    ix->gstride = (maxdeg + 15) & ~size_t(15);
    {
        DeviceBuffer tmp;
        const size_t src_bytes = desc->n * desc->graph_row_stride_u32 * sizeof(uint32_t);
        if (tmp.alloc(src_bytes) != hipSuccess) return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for graph staging failed", src_bytes));
        if (hipMemcpy(tmp.p, desc->graph, src_bytes, hipMemcpyHostToDevice) != hipSuccess)
            return cleanup(fail(SVSX_RUNTIME_ERROR, "copying graph to the device failed"));
        if (hipMalloc(reinterpret_cast<void**>(&ix->d_graph), desc->n * ix->gstride * sizeof(uint32_t)) != hipSuccess)
            return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMalloc for graph failed"));
        DeviceBuffer bad;
        if (bad.alloc(8) != hipSuccess || hipMemset(bad.p, 0, 8) != hipSuccess) return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMalloc failed"));
        if (launch_pack_graph(static_cast<const uint32_t*>(tmp.p), static_cast<uint32_t>(desc->graph_row_stride_u32), static_cast<uint32_t>(desc->n),
                              ix->d_graph, static_cast<uint32_t>(ix->gstride), static_cast<uint32_t*>(bad.p), nullptr) != hipSuccess)
            return cleanup(fail(SVSX_RUNTIME_ERROR, "graph re-layout kernel failed: %s", hipGetErrorString(hipGetLastError())));
        uint32_t hbad[2] = {0, 0};
        if (hipMemcpy(hbad, bad.p, 8, hipMemcpyDeviceToHost) != hipSuccess) return cleanup(fail(SVSX_RUNTIME_ERROR, "graph re-layout kernel failed"));
        if (hbad[0] || hbad[1])
            return cleanup(fail(SVSX_INVALID_ARGUMENT, "malformed graph: %u rows with a degree larger than the row, %u neighbour ids >= n", hbad[0], hbad[1]));
    }
    if (hipMalloc(reinterpret_cast<void**>(&ix->d_entry), desc->n_entry * sizeof(uint32_t)) != hipSuccess)
        return cleanup(fail(SVSX_RUNTIME_ERROR, "hipMalloc for entry points failed"));
    if (hipMemcpy(ix->d_entry, desc->entry_points, desc->n_entry * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(SVSX_RUNTIME_ERROR, "copying entry points failed"));
    *out = ix;
    return SVSX_SUCCESS;
}

int svsx_index_create_device_view(const svsx_device_index_desc* desc, svsx_index** out) {
    if (!desc || !out) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (int rc = validate_common(desc->metric, desc->data_dtype, desc->n, desc->d)) return rc;
    if (!desc->d_data || !desc->d_graph || !desc->d_entry_points || desc->n_entry == 0)
        return fail(SVSX_INVALID_ARGUMENT, "device data, graph and entry points are required");
    if (desc->data_row_stride_bytes < padded_row_bytes(desc->d, desc->data_dtype) || (desc->data_row_stride_bytes & 15))
        return fail(SVSX_INVALID_ARGUMENT, "device rows must be padded to 16 elements and 16-byte aligned");
    if (desc->graph_row_stride_u32 == 0 || (desc->graph_row_stride_u32 & 15))
        return fail(SVSX_INVALID_ARGUMENT, "device graph rows must be a multiple of 16 ids");
    svsx_index* ix = new (std::nothrow) svsx_index();
    if (!ix) return fail(SVSX_RUNTIME_ERROR, "out of host memory");
    ix->metric = desc->metric;
    ix->data_dtype = desc->data_dtype;
    ix->n = desc->n;
    ix->d = desc->d;
    ix->d_data = const_cast<void*>(desc->d_data);
    ix->row_bytes = desc->data_row_stride_bytes;
    ix->d_graph = const_cast<uint32_t*>(desc->d_graph);
    ix->gstride = desc->graph_row_stride_u32;
    ix->d_entry = const_cast<uint32_t*>(desc->d_entry_points);
    ix->n_entry = desc->n_entry;
    ix->sq_enabled = desc->sq_enabled;
    ix->sq_scale = desc->sq_scale;
    ix->sq_bias = desc->sq_bias;
    ix->device = desc->device;
    ix->id_offset = desc->id_offset;
    ix->owns = false;
    if (hipSetDevice(ix->device) != hipSuccess) { delete ix; return fail(SVSX_RUNTIME_ERROR, "hipSetDevice(%d) failed", desc->device); }
    if (int rc = build_sq_row_norms(ix, nullptr, true)) { svsx_index_destroy(ix); return rc; }
    *out = ix;
    return SVSX_SUCCESS;
}

void svsx_index_destroy(svsx_index* ix) {
    if (!ix) return;
    if (ix->owns) {
        (void)hipSetDevice(ix->device);
        if (ix->d_data) (void)hipFree(ix->d_data);
        if (ix->d_graph) (void)hipFree(ix->d_graph);
        if (ix->d_entry) (void)hipFree(ix->d_entry);
    }
    if (ix->d_sq_row_norm) (void)hipFree(ix->d_sq_row_norm);
    if (ix->d_half) (void)hipFree(ix->d_half);
    if (ix->d_norm2) (void)hipFree(ix->d_norm2);
    if (ix->d_codes) (void)hipFree(ix->d_codes);
    if (ix->d_fat) (void)hipFree(ix->d_fat);
    if (ix->d_pca) (void)hipFree(ix->d_pca);
    for (HostCallContext* c : ix->ctx_free) delete c;
    delete ix;
}

int svsx_index_info(const svsx_index* ix, uint64_t info[6]) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    info[0] = ix->n;
    info[1] = ix->d;
    info[2] = ix->row_bytes * ix->n;
    info[3] = ix->gstride * ix->n * sizeof(uint32_t);
    info[4] = ix->gstride;
    info[5] = ix->row_bytes;
    return SVSX_SUCCESS;
}

} // extern "C"

namespace {

// cyclic Jacobi eigen-decomposition of a symmetric d x d matrix (row-major, f64): A -> diagonal, V columns = eigenvectors
void jacobi_eigh(std::vector<double>& A, int d, std::vector<double>& V) {
    V.assign(static_cast<size_t>(d) * d, 0.0);
    for (int i = 0; i < d; ++i) V[static_cast<size_t>(i) * d + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < d; ++i)
            for (int j = 0; j < d; ++j) (i == j ? diag : off) += A[static_cast<size_t>(i) * d + j] * A[static_cast<size_t>(i) * d + j];
        if (off <= 1e-30 * diag || off == 0.0) break;
        for (int p = 0; p < d - 1; ++p) {
            for (int q = p + 1; q < d; ++q) {
                const double apq = A[static_cast<size_t>(p) * d + q];
                if (apq == 0.0) continue;
                const double theta = (A[static_cast<size_t>(q) * d + q] - A[static_cast<size_t>(p) * d + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < d; ++k) { // columns p, q
                    const double akp = A[static_cast<size_t>(k) * d + p], akq = A[static_cast<size_t>(k) * d + q];
                    A[static_cast<size_t>(k) * d + p] = c * akp - sn * akq;
                    A[static_cast<size_t>(k) * d + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < d; ++k) { // rows p, q
                    const double apk = A[static_cast<size_t>(p) * d + k], aqk = A[static_cast<size_t>(q) * d + k];
                    A[static_cast<size_t>(p) * d + k] = c * apk - sn * aqk;
                    A[static_cast<size_t>(q) * d + k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < d; ++k) {
                    const double vkp = V[static_cast<size_t>(k) * d + p], vkq = V[static_cast<size_t>(k) * d + q];
                    V[static_cast<size_t>(k) * d + p] = c * vkp - sn * vkq;
                    V[static_cast<size_t>(k) * d + q] = sn * vkp + c * vkq;
                }
            }
        }
    }
}

float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (h & 0x8000u) << 16, ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t u;
    if (ex == 0) {
        if (man == 0) u = sign;
        else {
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            u = sign | ((127 - 15 - e) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (ex == 31) u = sign | 0x7F800000u | (man << 13);
    else u = sign | ((ex + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

constexpr uint32_t kPcaRows = 128;  // rows of the float32 device copy of the directions (rows >= dc are zero)

// modified Gram-Schmidt on the m columns of V (d x m, column j at V[i * m + j]); twice for orthogonality to rounding
void orthonormalize_columns(std::vector<double>& V, uint32_t d, uint32_t m) {
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t j = 0; j < m; ++j) {
            for (uint32_t k = 0; k < j; ++k) {
                double dot = 0.0;
                for (uint32_t i = 0; i < d; ++i) dot += V[static_cast<size_t>(i) * m + j] * V[static_cast<size_t>(i) * m + k];
                for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] -= dot * V[static_cast<size_t>(i) * m + k];
            }
            double nrm = 0.0;
            for (uint32_t i = 0; i < d; ++i) nrm += V[static_cast<size_t>(i) * m + j] * V[static_cast<size_t>(i) * m + j];
            nrm = std::sqrt(nrm);
            if (!(nrm > 1e-300)) { // degenerate direction: replace by a unit vector and let the next repetition clean it up
                for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] = (i == j % d) ? 1.0 : 0.0;
                nrm = 1.0;
            }
            for (uint32_t i = 0; i < d; ++i) V[static_cast<size_t>(i) * m + j] /= nrm;
        }
}

// graph-inline records: sample -> leading principal directions (host, f64) -> per-vector records (device) -> node rows (device)
int build_fat_rows(svsx_index* ix, uint32_t forced_rb) {
    const uint32_t d = static_cast<uint32_t>(ix->d), n = static_cast<uint32_t>(ix->n);
    const size_t es = dtype_size(ix->data_dtype);
    const bool l2 = ix->metric == M_L2, cosine = ix->metric == M_COS;
    const uint32_t meta = l2 ? 4u : 6u; // bytes of float16 fields at the end of a record
    // 1. sample (evenly strided) -> host, the vector the metric sees (decompressed / normalised), f64
    const size_t want = d <= 128 ? 65536 : (d <= 256 ? 32768 : 8192);
    const size_t ns = n < want ? n : want;
    const size_t step = n / ns;
    std::vector<unsigned char> raw(ns * d * es);
    HIP_TRY(hipMemcpy2D(raw.data(), d * es, ix->d_data, step * ix->row_bytes, d * es, ns, hipMemcpyDeviceToHost));
    std::vector<double> xs(ns * d);
    for (size_t i = 0; i < ns * d; ++i) {
        float f;
        switch (ix->data_dtype) {
            case DT_F32: f = reinterpret_cast<const float*>(raw.data())[i]; break;
            case DT_F16: f = half_bits_to_float(reinterpret_cast<const uint16_t*>(raw.data())[i]); break;
            case DT_I8: f = static_cast<float>(reinterpret_cast<const int8_t*>(raw.data())[i]); break;
            default: f = static_cast<float>(raw[i]); break;
        }
        if (ix->sq_enabled) {
            const float t = ix->sq_scale * f;
            f = t + ix->sq_bias;
        }
        xs[i] = f;
        if (!(xs[i] - xs[i] == 0.0)) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    }
    if (cosine)
        for (size_t r = 0; r < ns; ++r) {
            double nrm = 0.0;
            for (uint32_t j = 0; j < d; ++j) nrm += xs[r * d + j] * xs[r * d + j];
            nrm = nrm > 0.0 ? 1.0 / std::sqrt(nrm) : 0.0;
            for (uint32_t j = 0; j < d; ++j) xs[r * d + j] *= nrm;
        }
    const uint32_t pd = (d + 15u) & ~15u;
    std::vector<float> mean_f(pd, 0.0f);
    if (l2) {
        std::vector<double> mean(d, 0.0);
        for (size_t r = 0; r < ns; ++r)
            for (uint32_t j = 0; j < d; ++j) mean[j] += xs[r * d + j];
        for (uint32_t j = 0; j < d; ++j) mean_f[j] = static_cast<float>(mean[j] / static_cast<double>(ns));
        for (size_t r = 0; r < ns; ++r)
            for (uint32_t j = 0; j < d; ++j) xs[r * d + j] -= static_cast<double>(mean_f[j]);
    }
    // 2. second-moment matrix (upper triangle, then mirrored)
    std::vector<double> cov(static_cast<size_t>(d) * d, 0.0);
    for (size_t r = 0; r < ns; ++r) {
        const double* row = &xs[r * d];
        for (uint32_t i = 0; i < d; ++i) {
            const double ri = row[i];
            double* c = &cov[static_cast<size_t>(i) * d];
            for (uint32_t j = i; j < d; ++j) c[j] += ri * row[j];
        }
    }
    for (uint32_t i = 0; i < d; ++i)
        for (uint32_t j = i; j < d; ++j) {
            const double v = cov[static_cast<size_t>(i) * d + j] / static_cast<double>(ns);
            cov[static_cast<size_t>(i) * d + j] = v;
            cov[static_cast<size_t>(j) * d + i] = v;
        }
    // 3. leading directions by subspace iteration + Rayleigh-Ritz (ANY orthonormal set is valid for the bound; a good one
    //    makes it tight).  m >= dc columns for convergence margin.
    // candidate record sizes: a record larger than a third of the row it protects cannot pay for itself
    uint32_t rb_max = 32;
    if (ix->row_bytes >= 3 * 64) rb_max = 64;
    if (ix->row_bytes >= 3 * 128) rb_max = 128;
    const uint32_t m = d < rb_max - meta + 12u ? d : rb_max - meta + 12u;
    std::vector<double> V(static_cast<size_t>(d) * m), W(static_cast<size_t>(d) * m);
    uint64_t lcg = 0x9E3779B97F4A7C15ull;
    for (auto& v : V) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        v = static_cast<double>(static_cast<int64_t>(lcg >> 11) % 2000001 - 1000000) * 1e-6;
    }
    orthonormalize_columns(V, d, m);
    for (int it = 0; it < 48; ++it) {
        std::fill(W.begin(), W.end(), 0.0);
        for (uint32_t i = 0; i < d; ++i) {
            double* w = &W[static_cast<size_t>(i) * m];
            const double* c = &cov[static_cast<size_t>(i) * d];
            for (uint32_t k = 0; k < d; ++k) {
                const double cik = c[k];
                const double* v = &V[static_cast<size_t>(k) * m];
                for (uint32_t j = 0; j < m; ++j) w[j] += cik * v[j];
            }
        }
        V.swap(W);
        orthonormalize_columns(V, d, m);
    }
    // Rayleigh-Ritz: T = V^T C V (m x m), eigenvectors Z, V <- V Z ordered by decreasing eigenvalue
    std::fill(W.begin(), W.end(), 0.0);
    for (uint32_t i = 0; i < d; ++i)
        for (uint32_t k = 0; k < d; ++k) {
            const double cik = cov[static_cast<size_t>(i) * d + k];
            for (uint32_t j = 0; j < m; ++j) W[static_cast<size_t>(i) * m + j] += cik * V[static_cast<size_t>(k) * m + j];
        }
    std::vector<double> T(static_cast<size_t>(m) * m, 0.0), Z;
    for (uint32_t a = 0; a < m; ++a)
        for (uint32_t b = 0; b < m; ++b) {
            double t = 0.0;
            for (uint32_t i = 0; i < d; ++i) t += V[static_cast<size_t>(i) * m + a] * W[static_cast<size_t>(i) * m + b];
            T[static_cast<size_t>(a) * m + b] = t;
        }
    for (uint32_t a = 0; a < m; ++a)
        for (uint32_t b = a + 1; b < m; ++b) {
            const double t = 0.5 * (T[static_cast<size_t>(a) * m + b] + T[static_cast<size_t>(b) * m + a]);
            T[static_cast<size_t>(a) * m + b] = T[static_cast<size_t>(b) * m + a] = t;
        }
    jacobi_eigh(T, static_cast<int>(m), Z);
    std::vector<uint32_t> order(m);
    for (uint32_t i = 0; i < m; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return T[static_cast<size_t>(x) * m + x] > T[static_cast<size_t>(y) * m + y]; });
    // Record size by expected lines per hop: block lines (4 + RB bytes per slot) + the rows of the candidates the bound cannot
    // exclude.  How many it excludes follows the share f of the sample's second moment the coded directions carry -- the remainder
    // only enters the bound through a product of norms, which is loose: measured (profiles/r02_notes.md) the bound is useless
    // below f ~ 0.6 and rejects ~88 % of the candidates at f ~ 0.9 -- modelled as rej = 0.88 clamp((f - 0.6) / 0.3, 0, 1) -- and a little more
    // towards f = 1 (+ 0.05 clamp((f - 0.9) / 0.1, 0, 1): 0.86 -> 0.926 measured between 64- and 128-byte records on config 4's rows,
    // profiles/r05_notes.md).  A row of scalar-quantized codes costs far more than its lines: every element is decompressed and multiplied
    // (2.5 vector instructions, scalar.h:155-195) on a kernel whose few resident waves are bound by their instruction count, not by lines: a
    // 768-byte row is ~34 wave instructions, doubling the records of a 128-slot row ~50 -- weighted 8x (round 5: 128-byte records beat 64-byte
    // ones by 18 % on 768-byte rows although they double the block).  A larger record must win by 10 % (the 32-byte form has
    // static-dimension kernels and the fewest registers).
    double trace = 0.0;
    for (uint32_t i = 0; i < d; ++i) trace += cov[static_cast<size_t>(i) * d + i];
    const double row_lines = static_cast<double>((ix->row_bytes + 127) / 128) * (ix->sq_enabled ? 8.0 : 1.0);
    const double cand = 0.9 * static_cast<double>(ix->gstride);
    uint32_t rb = 32;
    double best_cost = 0.0;
    for (uint32_t r = 32; r <= rb_max; r *= 2) {
        double got = 0.0;
        for (uint32_t c = 0; c < r - meta && c < m; ++c) got += std::max(T[static_cast<size_t>(order[c]) * m + order[c]], 0.0);
        const double f = trace > 0.0 ? got / trace : 1.0;
        const double rej = 0.88 * std::min(std::max((f - 0.6) / 0.3, 0.0), 1.0) + 0.05 * std::min(std::max((f - 0.9) / 0.1, 0.0), 1.0);
        const double cost = static_cast<double>(ix->gstride) * (4.0 + r) / 128.0 + cand * (1.0 - rej) * row_lines;
        if (r == 32 || cost < 0.9 * best_cost) {
            best_cost = cost;
            rb = r;
        }
    }
    if (forced_rb != 0) { // the caller asked for this record size (svsx_index_build_bound_codes mode 32 / 64 / 128)
        if (forced_rb > rb_max) return fail(SVSX_INVALID_ARGUMENT, "records of %u bytes exceed a third of the %zu-byte rows they protect", forced_rb, ix->row_bytes);
        rb = forced_rb;
    }
    const uint32_t dc = rb - meta;
    const uint32_t dcu = dc < m ? dc : m; // directions really used (d < dc cannot happen: d >= 32 and m >= dc unless d is smaller)
    std::vector<double> Q(static_cast<size_t>(d) * dcu), sigma(dcu);
    for (uint32_t c = 0; c < dcu; ++c) {
        for (uint32_t i = 0; i < d; ++i) {
            double v = 0.0;
            for (uint32_t j = 0; j < m; ++j) v += V[static_cast<size_t>(i) * m + j] * Z[static_cast<size_t>(j) * m + order[c]];
            Q[static_cast<size_t>(i) * dcu + c] = v;
        }
        sigma[c] = std::sqrt(std::max(T[static_cast<size_t>(order[c]) * m + order[c]], 0.0));
    }
    orthonormalize_columns(Q, d, dcu);
    // device copies: float32 rows for the query side (+ delta = how far they are from the f64 directions), f64 rows for the data side
    std::vector<float> host(static_cast<size_t>(kPcaRows) * pd + pd + 128, 0.0f); // P | mean | bias
    std::vector<double> q64(static_cast<size_t>(dcu) * d);
    double delta = 0.0;
    for (uint32_t c = 0; c < dcu; ++c) {
        double e2 = 0.0;
        for (uint32_t j = 0; j < d; ++j) {
            const double q = Q[static_cast<size_t>(j) * dcu + c];
            const float f = static_cast<float>(q);
            host[static_cast<size_t>(c) * pd + j] = f;
            q64[static_cast<size_t>(c) * d + j] = q;
            e2 += (static_cast<double>(f) - q) * (static_cast<double>(f) - q);
        }
        delta = std::max(delta, std::sqrt(e2));
    }
    for (uint32_t j = 0; j < d; ++j) host[static_cast<size_t>(kPcaRows) * pd + j] = mean_f[j];
    if (ix->d_pca) { (void)hipFree(ix->d_pca); ix->d_pca = nullptr; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ix->d_pca), host.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(ix->d_pca, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    DeviceBuffer dq;
    HIP_TRY(dq.alloc(q64.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(dq.p, q64.data(), q64.size() * sizeof(double), hipMemcpyHostToDevice));
    // 4. per-coordinate range of y over ALL vectors (device), clipped to +-8 sigma; step, biases, tail unit
    DeviceBuffer st;
    HIP_TRY(st.alloc((2 * 128 + 2) * 4));
    std::vector<uint32_t> init(2 * 128 + 2, 0u);
    for (uint32_t i = 0; i < 128; ++i) init[i] = 0xFFFFFFFFu;
    HIP_TRY(hipMemcpy(st.p, init.data(), init.size() * 4, hipMemcpyHostToDevice));
    PcaArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.data = ix->d_data; pa.row_bytes = ix->row_bytes; pa.dtype = ix->data_dtype; pa.n = n; pa.d = d; pa.dc = dcu; pa.rb = rb;
    pa.decomp = ix->sq_enabled; pa.sq_scale = ix->sq_scale; pa.sq_bias = ix->sq_bias; pa.normalize = cosine ? 1 : 0; pa.signed_codes = l2 ? 0 : 1;
    pa.q64 = static_cast<const double*>(dq.p);
    pa.mean = ix->d_pca + static_cast<size_t>(kPcaRows) * pd; pa.bias = pa.mean + pd;
    pa.stat_min = static_cast<uint32_t*>(st.p); pa.stat_max = pa.stat_min + 128; pa.stat_tail = pa.stat_max + 128; pa.stat_bad = pa.stat_tail + 1;
    HIP_TRY(launch_pca_stats(pa, nullptr));
    std::vector<uint32_t> stats(2 * 128 + 2);
    HIP_TRY(hipMemcpy(stats.data(), st.p, stats.size() * 4, hipMemcpyDeviceToHost));
    if (stats[2 * 128 + 1] != 0) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    double range = 0.0, bmax = 0.0;
    std::vector<float> bias(128, 0.0f);
    for (uint32_t i = 0; i < dcu; ++i) {
        double lo = pca_key_to_float(stats[i]), hi = pca_key_to_float(stats[128 + i]);
        if (sigma[i] > 0.0) { // outliers beyond 8 sigma clamp (their coding error is in their own ex)
            lo = std::max(lo, -8.0 * sigma[i]);
            hi = std::min(hi, 8.0 * sigma[i]);
        }
        if (!(hi > lo)) hi = lo;
        if (l2) {
            bias[i] = static_cast<float>(lo);
            range = std::max(range, hi - lo);
            bmax = std::max(bmax, std::fabs(static_cast<double>(bias[i])));
        } else {
            range = std::max(range, std::max(std::fabs(lo), std::fabs(hi)));
        }
    }
    float stepf = static_cast<float>(range / (l2 ? 255.0 : 127.0));
    if (!(stepf > 0.0f) || !(stepf < 3.0e38f)) stepf = 1.0f;
    float tail_max = pca_key_to_float(stats[2 * 128]);
    if (!(tail_max > 0.0f) || !(tail_max < 3.0e38f)) tail_max = 1.0f;
    const float tail_unit = tail_max / 60000.0f;
    HIP_TRY(hipMemcpy(ix->d_pca + static_cast<size_t>(kPcaRows) * pd + pd, bias.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    // 5. records, then node rows
    DeviceBuffer rec;
    HIP_TRY(rec.alloc(static_cast<size_t>(n) * rb + 128));
    pa.step = stepf; pa.tail_unit = tail_unit; pa.records = static_cast<uint32_t*>(rec.p);
    HIP_TRY(launch_pca_encode(pa, nullptr));
    const size_t fat_stride = (ix->gstride * (4 + rb) + 127) & ~size_t(127);
    void* fat = nullptr;
    if (hipMalloc(&fat, static_cast<size_t>(n) * fat_stride + 128) != hipSuccess)
        return fail(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for the graph-inline records failed", static_cast<size_t>(n) * fat_stride);
    // degree-aware block loads (svsx_search_kernel.h): ids below 2^26 leave bits [26, 31) of a buffer word for ceil(degree / 8)
    constexpr uint32_t kDegShift = 26;
    const bool deg_aware = n <= (1u << kDegShift) && ix->gstride <= 248; // class = ceil(degree / 8) <= 31: five bits
    DeviceBuffer cls;
    if (deg_aware && cls.alloc(n) != hipSuccess) {
        (void)hipFree(fat);
        return fail(SVSX_RUNTIME_ERROR, "hipMalloc(%u) for the degree classes failed", n);
    }
    hipError_t e = launch_fat_build(ix->d_graph, static_cast<uint32_t>(ix->gstride), n, rec.p, rb, fat, fat_stride, deg_aware ? static_cast<uint8_t*>(cls.p) : nullptr,
                                    kDegShift, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(fat);
        return fail(SVSX_RUNTIME_ERROR, "graph-inline record kernels failed: %s", hipGetErrorString(e));
    }
    ix->d_fat = fat;
    ix->fat_stride = fat_stride;
    ix->fat_rb = rb;
    ix->fat_degshift = deg_aware ? kDegShift : 0u;
    ix->pca_step = stepf;
    ix->pca_tail_unit = tail_unit;
    ix->pca_delta = static_cast<float>(delta * 1.01 + 1e-9);
    // rounding of fl(b + s c) (L2) / fl(s c) (signed codes) in the query's residual: <= 2^-23 (|b| + 255 s) per coordinate
    ix->pca_eq_abs = static_cast<float>(std::sqrt(static_cast<double>(dcu)) * 1.2e-7 * (bmax + 255.0 * stepf) * 1.01);
    ix->pca_ref_extra = ix->sq_enabled ? static_cast<float>(2.0 * std::fabs(static_cast<double>(ix->sq_bias)) * std::sqrt(static_cast<double>(d)) * 1.01) : 0.0f;
    return SVSX_SUCCESS;
}

} // namespace

extern "C" int svsx_index_build_bound_codes(svsx_index* ix, int mode) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    uint32_t forced_rb = 0;
    if (mode == 32 || mode == 64 || mode == 128) { // graph-inline records of exactly this size
        forced_rb = static_cast<uint32_t>(mode);
        mode = 2;
    }
    if (mode < 0 || mode > 2) return fail(SVSX_INVALID_ARGUMENT, "bound-code mode must be 0 (off), 1 (128-B rows), 2 (graph-inline records) or 32 / 64 / 128 (inline records of that size)");
    HIP_TRY(hipSetDevice(ix->device));
    std::lock_guard<std::mutex> lock(ix->shadow_mutex);
    HIP_TRY(hipDeviceSynchronize());
    // one scheme at a time: drop what the new mode does not use
    if (mode != 1 && ix->d_codes) { (void)hipFree(ix->d_codes); ix->d_codes = nullptr; }
    if (mode != 2 && ix->d_fat) { (void)hipFree(ix->d_fat); ix->d_fat = nullptr; }
    if (mode == 0) return SVSX_SUCCESS;
    if (mode == 2) {
        // float32 / float16 rows (any metric) and scalar-quantized rows under inner product / cosine (the compressed L2 distance
        // is taken between CODED query and data, scalar.h:58-105, and the raw int8 x int8 kernels are exact integer paths)
        if (ix->d_fat && (forced_rb == 0 || forced_rb == ix->fat_rb)) return SVSX_SUCCESS;
        if (ix->d_fat) { (void)hipFree(ix->d_fat); ix->d_fat = nullptr; }
        const bool float_rows = (ix->data_dtype == DT_F32 || ix->data_dtype == DT_F16) && !ix->sq_enabled;
        const bool sq_rows = ix->sq_enabled && ix->metric != M_L2;
        if (!float_rows && !sq_rows) return fail(SVSX_NOT_IMPLEMENTED, "graph-inline records cover float32 / float16 rows and scalar-quantized rows under inner product or cosine");
        if (ix->d < 32 || ix->d > 4096 || !ix->d_graph) return fail(SVSX_NOT_IMPLEMENTED, "graph-inline records cover graph indexes of dimension 32..4096");
        return build_fat_rows(ix, forced_rb);
    }
    if (ix->metric != M_L2 || (ix->data_dtype != DT_F32 && ix->data_dtype != DT_F16) || ix->sq_enabled)
        return fail(SVSX_NOT_IMPLEMENTED, "128-byte bound-code rows cover float32 / float16 L2 indexes");
    if (ix->d_codes) return SVSX_SUCCESS;
    if (ix->d != 96 && ix->d != 128) return fail(SVSX_NOT_IMPLEMENTED, "128-byte bound-code rows cover dimension 96 or 128");
    const uint32_t dc = static_cast<uint32_t>(ix->d < 120 ? ix->d : 120);
    DeviceBuffer mm;
    HIP_TRY(mm.alloc(12));
    HIP_TRY(launch_codes_minmax(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), dc, static_cast<uint32_t*>(mm.p), nullptr));
    uint32_t h[3];
    HIP_TRY(hipMemcpy(h, mm.p, sizeof(h), hipMemcpyDeviceToHost));
    if (h[2] != 0) return fail(SVSX_INVALID_ARGUMENT, "bound codes need finite data");
    const float lo = codes_key_to_float(h[0]), hi = codes_key_to_float(h[1]);
    float scale = (hi - lo) / 255.0f;
    if (!(scale > 0.0f) || !(scale < 3.0e38f)) scale = 1.0f;
    void* codes = nullptr;
    HIP_TRY(hipMalloc(&codes, ix->n * 128 + 128));
    hipError_t e = launch_codes_build(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), dc, scale, lo, codes, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(codes);
        return fail(SVSX_RUNTIME_ERROR, "bound code kernel failed: %s", hipGetErrorString(e));
    }
    ix->code_scale = scale;
    ix->code_bias = lo;
    ix->code_eq_abs = static_cast<float>(std::sqrt(static_cast<double>(dc)) * 2.4e-7 * (std::fabs(static_cast<double>(lo)) + 255.0 * scale) * 1.01);
    ix->d_codes = codes;
    return SVSX_SUCCESS;
}

extern "C" {

static SearchArgs base_search_args(const svsx_index* ix, const void* d_queries, int q_dtype, size_t q_row_bytes, size_t nq) {
    SearchArgs a;
    memset(&a, 0, sizeof(a));
    a.data = ix->d_data;
    a.row_bytes = ix->row_bytes;
    a.graph = ix->d_graph;
    a.gstride = static_cast<uint32_t>(ix->gstride);
    a.n = static_cast<uint32_t>(ix->n);
    a.d = static_cast<uint32_t>(ix->d);
    a.entry = ix->d_entry;
    a.n_entry = static_cast<uint32_t>(ix->n_entry);
    a.sq_scale = ix->sq_scale;
    a.sq_bias = ix->sq_bias;
    a.sq_row_norm = ix->d_sq_row_norm;
    a.id_offset = ix->id_offset;
    a.queries = d_queries;
    a.q_dtype = q_dtype;
    a.q_row_bytes = q_row_bytes;
    a.nq = static_cast<uint32_t>(nq);
    a.general = ix->n_entry > 16 ? 1u : 0u; // std::sort of > 16 entry points is not a stable insertion sort
    // scalar-quantized rows under cosine without their precomputed norms (svsx_index_drop_row_norms): the runtime-d kernel variant computes the
    // norm per candidate, the reference's own operation order (same bits, ~14 % slower: profiles/r04_notes.md section 3)
    if (ix->sq_enabled && ix->metric == M_COS && ix->d_sq_row_norm == nullptr) a.general = 1u;
    // the scalar-loop type pairs exist as the runtime-d variant only; no bound records either (their slack terms are derived for the SIMD tree)
    if (select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled) == QM_SCALAR) a.general = 1u;
    if (ix->d_codes && !a.general) { // result-neutral lower-bound rejection (float queries on float32 L2 rows)
        a.codes = static_cast<const unsigned char*>(ix->d_codes);
        a.code_scale = ix->code_scale;
        a.code_bias = ix->code_bias;
        a.code_eq_abs = ix->code_eq_abs;
    }
    if (ix->d_fat && !a.general) { // graph-inline records: same guarantee, one contiguous block per expansion
        a.fat = static_cast<const unsigned char*>(ix->d_fat);
        a.fat_stride = ix->fat_stride;
        a.fat_rb = ix->fat_rb;
        a.fat_degshift = ix->fat_degshift;
        a.pca_pd = static_cast<uint32_t>((ix->d + 15) & ~size_t(15));
        a.pca_t = ix->d_pca;
        a.pca_mean = ix->d_pca + static_cast<size_t>(kPcaRows) * a.pca_pd;
        a.pca_bias = a.pca_mean + a.pca_pd;
        a.pca_step = ix->pca_step;
        a.pca_tail_unit = ix->pca_tail_unit;
        a.pca_delta = ix->pca_delta;
        a.pca_ref_extra = ix->pca_ref_extra;
        a.pca_eq_abs = ix->pca_eq_abs;
    }
    return a;
}

// default LDS visited filter: 4 KB per wave (the measured optimum): 4096 entries with 8-bit tags when they identify
// the id exactly ((n >> 12) < 255), else 2048 entries with 16-bit tags; grown until tags are exact, else off
static void default_filter(const svsx_index* ix, SearchArgs& a, uint32_t fe) {
    while (fe <= 16384 && (ix->n >> __builtin_ctz(fe)) >= 0xFFFFu) fe <<= 1;
    a.filter_entries = fe <= 16384 ? fe : 0;
    a.filter_tag8 = (a.filter_entries && (ix->n >> __builtin_ctz(a.filter_entries)) < 0xFFu) ? 1u : 0u;
    if (a.fat != nullptr) a.filter_tag8 = 0; // the inline-record kernel keeps only fetched rows in the filter: 16-bit tags, no variant
}

// Search buffers beyond what a wave's LDS holds (16 000 entries).  The reference only asks for window <= capacity (search_buffer.h:84-93), so
// nothing is refused: the buffers live in HBM, one row of a state table per query, and the GENERAL kernel variant works on them in place
// exactly as it does for a BatchIterator whose buffer has outgrown LDS (svsx_iterator_next_device) -- a plain search is one iterator step
// that starts from the entry points with (window - k, capacity - k) + a batch of k.  Queries run in groups that keep the table under 2 GB.
__global__ void fill_spill_meta_kernel(uint32_t* meta, uint32_t nq, uint32_t window_minus_k, uint32_t capacity_minus_k) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    uint32_t* m = meta + static_cast<size_t>(q) * 8;
    m[0] = 0u; m[1] = 0u; m[2] = window_minus_k; m[3] = capacity_minus_k; m[4] = 0u; m[5] = 0u; m[6] = 1u /* start from the entry points */; m[7] = 0u;
}

static int search_spilled(const svsx_index* ix, int qmode, const SearchArgs& plain, hipStream_t stream) {
    const size_t nq = plain.nq, k = plain.k, estride = (static_cast<size_t>(plain.capacity) + 1 + 1) & ~size_t(1);
    size_t group = (size_t(2) << 30) / (estride * sizeof(uint2));
    if (group == 0) group = 1;
    if (group > nq) group = nq;
    StreamBuffer entries, meta, yielded, out_i, out_d, counts;
    HIP_TRY(entries.alloc(group * estride * sizeof(uint2), stream));
    HIP_TRY(meta.alloc(group * 8 * sizeof(uint32_t), stream));
    HIP_TRY(yielded.alloc(group * k * sizeof(uint32_t), stream));
    HIP_TRY(out_i.alloc(group * k * sizeof(uint64_t), stream));
    HIP_TRY(out_d.alloc(group * k * sizeof(float), stream));
    HIP_TRY(counts.alloc(group * sizeof(uint32_t), stream));
    for (size_t lo = 0; lo < nq; lo += group) {
        const size_t cnt = nq - lo < group ? nq - lo : group;
        SearchArgs a = plain;
        a.queries = static_cast<const unsigned char*>(plain.queries) + lo * plain.q_row_bytes;
        a.nq = static_cast<uint32_t>(cnt);
        if (plain.ids64) a.ids64 = plain.ids64 + lo * k;
        if (plain.ids32) a.ids32 = plain.ids32 + lo * k;
        if (plain.dists) a.dists = plain.dists + lo * k;
        if (plain.stats) a.stats = plain.stats + lo * 6;
        if (plain.history_ids) a.history_ids = plain.history_ids + lo * plain.history_cap;
        if (plain.history_dists) a.history_dists = plain.history_dists + lo * plain.history_cap;
        if (plain.history_len) a.history_len = plain.history_len + lo;
        a.general = 1;
        a.codes = nullptr;
        a.fat = nullptr;
        a.filter_tag8 = (a.filter_entries && (ix->n >> __builtin_ctz(a.filter_entries)) < 0xFFu) ? 1u : 0u;
        hipLaunchKernelGGL(fill_spill_meta_kernel, dim3(static_cast<uint32_t>((cnt + 255) / 256)), dim3(256), 0, stream, static_cast<uint32_t*>(meta.p), static_cast<uint32_t>(cnt),
                           plain.window - static_cast<uint32_t>(k), plain.capacity - static_cast<uint32_t>(k));
        a.window = 1;   // per-query values come from the state table; these only size the (unused) LDS buffer
        a.capacity = 1;
        a.it_spill = 1u;
        a.it_entries = static_cast<uint2*>(entries.p);
        a.it_estride = static_cast<uint32_t>(estride);
        a.it_meta = static_cast<uint32_t*>(meta.p);
        a.it_yielded = static_cast<uint32_t*>(yielded.p);
        a.it_ystride = static_cast<uint32_t>(k);
        a.it_batch_sizes = nullptr;
        a.it_batch = static_cast<uint32_t>(k);
        a.it_out_ids = static_cast<uint64_t*>(out_i.p);
        a.it_out_dists = static_cast<float*>(out_d.p);
        a.it_counts = static_cast<uint32_t*>(counts.p);
        hipError_t e = launch_search(ix->data_dtype, ix->metric, qmode, a, stream);
        if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "search kernel launch failed: %s", hipGetErrorString(e));
    }
    return SVSX_SUCCESS;
}

static int search_common(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* d_ids64, uint32_t* d_ids32, float* d_dists, const svsx_search_aux* aux, hipStream_t stream
) {
    size_t window = params ? params->search_window_size : 0;
    size_t capacity = params ? params->search_buffer_capacity : 0;
    if (capacity == 0) capacity = window;
    if (window > capacity)
        return fail(SVSX_INVALID_ARGUMENT, "Improper configuration for search buffer! Effective size (%zu) cannot be less than capacity (%zu).", window, capacity);
    if (capacity < k) { // index/vamana/index.h:590-592
        window = k;
        capacity = k;
    }
    if (capacity == 0) return fail(SVSX_INVALID_ARGUMENT, "search window size must be positive");
    if (d_ids32 != nullptr && ix->id_offset + ix->n > (uint64_t(1) << 32))
        return fail(SVSX_INVALID_ARGUMENT, "32-bit ids cannot hold id_offset + n = %llu: ask for 64-bit ids", static_cast<unsigned long long>(ix->id_offset + ix->n));
    if (capacity > (size_t(1) << 27)) return fail(SVSX_INVALID_ARGUMENT, "search buffer capacity %zu is out of range", capacity);
    const int qmode = select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled);
    if (qmode < 0)
        return fail(SVSX_INVALID_ARGUMENT, "unsupported (query dtype %d, data dtype %d%s) pair", q_dtype, ix->data_dtype, ix->sq_enabled ? ", SQ" : "");
    SearchArgs a = base_search_args(ix, d_queries, q_dtype, (aux && aux->query_row_stride_bytes) ? aux->query_row_stride_bytes : ix->d * dtype_size(q_dtype), nq);
    a.window = static_cast<uint32_t>(window);
    a.capacity = static_cast<uint32_t>(capacity);
    a.k = static_cast<uint32_t>(k);
    // visited_set: 0 = off (reference default, search_params.h:42); 1 = on with the default LDS filter size;
    // >= 64 = explicit number of filter entries (rounded down to a power of two, at most 16384).
    if (params && params->visited_set > 0) {
        uint32_t fe;
        if (params->visited_set == 1) {
            // default: 4 KB of LDS per wave (the measured optimum): 4096 entries with 8-bit tags when they identify the
            // id exactly ((n >> 12) < 255), else 2048 entries with 16-bit tags
            fe = ((ix->n >> 12) < 0xFFu && !ix->d_fat) ? 4096u : kDefaultFilterEntries;
        } else {
            fe = static_cast<uint32_t>(params->visited_set);
            if (fe < 64) fe = 64;
            if (fe > 16384) fe = 16384;
            while (fe & (fe - 1)) fe &= fe - 1;
        }
        // tags must identify the id exactly: grow the table until (n >> log2 entries) < 0xFFFF, else stay off
        default_filter(ix, a, fe);
    }
    a.ids64 = d_ids64;
    a.ids32 = d_ids32;
    a.dists = d_dists;
    if (aux) {
        a.stats = aux->stats;
        a.history_ids = aux->history_ids;
        a.history_dists = aux->history_dists;
        a.history_len = aux->history_len;
        a.history_cap = static_cast<uint32_t>(aux->history_cap);
        a.cancel = const_cast<const int*>(aux->cancel_flag); // device word here (svsx_search mirrors the caller's host word into one)
    }
    if (capacity > 16000) return search_spilled(ix, qmode, a, stream);
    hipError_t e = launch_search(ix->data_dtype, ix->metric, qmode, a, stream);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "search kernel launch failed: %s", hipGetErrorString(e));
    return SVSX_SUCCESS;
}

static int check_search_args(const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, bool need_graph = true) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (need_graph && !ix->d_graph) return fail(SVSX_INVALID_ARGUMENT, "this index was created without a graph (flat search only)");
    if (k == 0) return fail(SVSX_INVALID_ARGUMENT, "k must be greater than 0");
    if (dtype_size(q_dtype) == 0) return fail(SVSX_INVALID_ARGUMENT, "unknown query dtype %d", q_dtype);
    if (nq > 0 && !queries) return fail(SVSX_INVALID_ARGUMENT, "null queries");
    if (nq >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "too many queries in one batch");
    return SVSX_SUCCESS;
}

// the caller's aux through the size it was compiled with (svsx.h: struct_size)
constexpr size_t kAuxSizeAbi3 = 64; // struct_size, four pointers, two sizes, the cancel word: the struct as ABI 3 introduced it
static_assert(sizeof(svsx_search_aux) >= kAuxSizeAbi3, "svsx_search_aux only grows at its end");
static const svsx_search_aux* normalise_aux(const svsx_search_aux* in, svsx_search_aux& store) {
    if (!in) return nullptr;
    const size_t have = in->struct_size == 0 ? kAuxSizeAbi3 : in->struct_size; // 0 = the ABI-3 size, a constant (not sizeof of this build)
    if (have >= sizeof(svsx_search_aux)) return in;
    memset(&store, 0, sizeof(store));
    memcpy(&store, in, have < sizeof(size_t) ? sizeof(size_t) : have);
    return &store;
}

int svsx_search_device(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* d_ids64, uint32_t* d_ids32, float* d_dists, const svsx_search_aux* d_aux_in, void* hip_stream
) {
    svsx_search_aux aux_store;
    const svsx_search_aux* d_aux = normalise_aux(d_aux_in, aux_store);
    if (int rc = check_search_args(ix, d_queries, q_dtype, nq, k)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    HIP_TRY(hipSetDevice(ix->device));
    return search_common(ix, d_queries, q_dtype, nq, k, params, d_ids64, d_ids32, d_dists, d_aux, static_cast<hipStream_t>(hip_stream));
}


// Host pointers in, host results out (SURVEY 8d's QPS definition; benchmark/include/svs-benchmark/search.h:209-219 times exactly this call).
// The form that also serves the side outputs (statistics, search history): H2D -> kernel -> D2H with persistent device staging, in two
// chunks on two streams (below).
static int search_host_pipelined(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* ids64, uint32_t* ids32, float* dists, const svsx_search_aux* aux, size_t q_stride
) {
    ContextLease lease(ix);
    HostCallContext* c = lease.c;
    if (!c) return fail(SVSX_RUNTIME_ERROR, "out of host memory");
    // chunk boundaries: small first / last chunks for batches worth pipelining, one chunk otherwise
    size_t bounds[HostCallContext::kMaxChunks + 1];
    int nc = 1;
    bounds[0] = 0;
    // Two launches on two streams: the first 4 096 queries start as soon as their 1.5 MB have crossed PCIe, the rest is copied while they
    // run.  More chunks lose: a later launch only gets compute units when an earlier one has drained, so every extra launch adds a drain
    // (five chunks: 1.43-1.50 ms against 1.15 ms for the kernel alone; profiles/r03_notes.md).  The copies go straight between the caller's
    // (pageable) arrays and the device staging: the runtime's own pipelined bounce (1.5 MB in 41 us, 3.8 MB in 92 us on the bench host)
    // beats memcpy into a page-locked mirror followed by the DMA (33 + 37 us, 79 + 76 us; profiles/r04_notes.md section 6).
    // (three chunks 2 048 + 4 096 + rest with direct copies: 1.30-1.31 ms against 1.25-1.27, profiles/r04_notes.md section 6)
    size_t first = nq > 4096 ? 4096 : nq;
    if (first < nq) {
        nc = 2;
        bounds[1] = first;
        bounds[2] = nq;
    } else {
        bounds[1] = nq;
    }
    // layout of the output staging: one region per output array
    const size_t hcap = aux ? aux->history_cap : 0;
    struct Region { void* host; size_t row_bytes; size_t offset; };
    Region regions[7];
    int nr = 0;
    size_t out_bytes = 0;
    auto add_region = [&](void* host, size_t row_bytes) -> int {
        if (!host || row_bytes == 0) return -1;
        regions[nr] = Region{host, row_bytes, out_bytes};
        out_bytes += (row_bytes * nq + 255) & ~size_t(255);
        return nr++;
    };
    const int r64 = add_region(ids64, k * 8), r32 = add_region(ids32, k * 4), rdd = add_region(dists, k * 4);
    const int rst = add_region(aux ? aux->stats : nullptr, 24);
    const int rhi = add_region(aux ? aux->history_ids : nullptr, hcap * 4), rhd = add_region(aux ? aux->history_dists : nullptr, hcap * 4);
    const int rhl = add_region(aux ? aux->history_len : nullptr, 4);
    HIP_TRY(c->reserve(q_stride * nq, out_bytes, nc));
    const size_t q_row = ix->d * dtype_size(q_dtype);
    c->dirty = true; // from here on an early return leaves copies / kernels in flight: the lease drains the streams before the context is reused
    auto dptr = [&](int r, size_t row) -> void* { return r < 0 ? nullptr : static_cast<unsigned char*>(c->d_out) + regions[r].offset + regions[r].row_bytes * row; };
    for (int i = 0; i < nc; ++i) {
        const size_t lo = bounds[i], cnt = bounds[i + 1] - lo;
        if (cnt == 0) continue;
        hipStream_t st = c->streams[i];
        void* dq = static_cast<unsigned char*>(c->d_in) + q_stride * lo;
        // never past the caller's last row
        HIP_TRY(hipMemcpyAsync(dq, static_cast<const unsigned char*>(queries) + q_stride * lo, q_stride * (cnt - 1) + q_row, hipMemcpyHostToDevice, st));
        svsx_search_aux daux;
        memset(&daux, 0, sizeof(daux));
        daux.query_row_stride_bytes = q_stride;
        daux.history_cap = hcap;
        daux.stats = static_cast<uint32_t*>(dptr(rst, lo));
        daux.history_ids = static_cast<uint32_t*>(dptr(rhi, lo));
        daux.history_dists = static_cast<float*>(dptr(rhd, lo));
        daux.history_len = static_cast<uint32_t*>(dptr(rhl, lo));
        if (int rc = search_common(ix, dq, q_dtype, cnt, k, params, static_cast<uint64_t*>(dptr(r64, lo)), static_cast<uint32_t*>(dptr(r32, lo)),
                                   static_cast<float*>(dptr(rdd, lo)), &daux, st)) {
            for (int j = 0; j <= i; ++j) (void)hipStreamSynchronize(c->streams[j]);
            return rc;
        }
    }
    // results: a copy into pageable memory returns when it has happened, so they are requested only after every launch is enqueued
    for (int i = 0; i < nc; ++i) {
        const size_t lo = bounds[i], cnt = bounds[i + 1] - lo;
        if (cnt == 0) continue;
        for (int r = 0; r < nr; ++r) {
            const size_t off = regions[r].offset + regions[r].row_bytes * lo;
            HIP_TRY(hipMemcpyAsync(static_cast<unsigned char*>(regions[r].host) + regions[r].row_bytes * lo, static_cast<unsigned char*>(c->d_out) + off,
                                   regions[r].row_bytes * cnt, hipMemcpyDeviceToHost, c->streams[i]));
        }
        HIP_TRY(hipStreamSynchronize(c->streams[i]));
    }
    c->dirty = false; // every stream has been synchronised
    return SVSX_SUCCESS;
}

int svsx_search(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    uint64_t* ids64, uint32_t* ids32, float* dists, const svsx_search_aux* aux_in
) {
    svsx_search_aux aux_store;
    const svsx_search_aux* aux = normalise_aux(aux_in, aux_store);
    if (!ix) {
        // vamana_index_impl.h:115-123: outputs filled with the "unspecified" sentinels
        if (dists) for (size_t i = 0; i < nq * k; ++i) dists[i] = __builtin_inff();
        if (ids64) for (size_t i = 0; i < nq * k; ++i) ids64[i] = ~0ull;
        if (ids32) for (size_t i = 0; i < nq * k; ++i) ids32[i] = 0xFFFFFFFFu;
        return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    }
    if (int rc = check_search_args(ix, queries, q_dtype, nq, k)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    HIP_TRY(hipSetDevice(ix->device));
    const size_t q_stride = (aux && aux->query_row_stride_bytes) ? aux->query_row_stride_bytes : ix->d * dtype_size(q_dtype);
    if (!(aux && aux->cancel_flag)) return search_host_pipelined(ix, queries, q_dtype, nq, k, params, ids64, ids32, dists, aux, q_stride);
    // ---- with a cancel word: one chunk on one stream, the host thread polls the caller's word while the stream drains -------------------
    CallStream cs;
    HIP_TRY(cs.create());
    StreamBuffer dq, d64, d32, dd, dst, dhi, dhd, dhl;
    HIP_TRY(dq.alloc(q_stride * nq, cs.s));
    HIP_TRY(hipMemcpyAsync(dq.p, queries, q_stride * nq, hipMemcpyHostToDevice, cs.s));
    // rows of queries the kernel never starts (or abandons) must not carry stale pool contents: "unspecified" sentinels, like
    // vamana_index_impl.h:115-123 fills them
    if (ids64) { HIP_TRY(d64.alloc(nq * k * 8, cs.s)); HIP_TRY(hipMemsetAsync(d64.p, 0xFF, nq * k * 8, cs.s)); }
    if (ids32) { HIP_TRY(d32.alloc(nq * k * 4, cs.s)); HIP_TRY(hipMemsetAsync(d32.p, 0xFF, nq * k * 4, cs.s)); }
    if (dists) { HIP_TRY(dd.alloc(nq * k * 4, cs.s)); HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dd.p), 0x7F800000, nq * k, cs.s)); }
    svsx_search_aux daux;
    memset(&daux, 0, sizeof(daux));
    daux.query_row_stride_bytes = q_stride;
    if (aux) {
        daux.history_cap = aux->history_cap;
        if (aux->stats) { HIP_TRY(dst.alloc(nq * 24, cs.s)); daux.stats = static_cast<uint32_t*>(dst.p); }
        if (aux->history_ids && aux->history_cap) { HIP_TRY(dhi.alloc(nq * aux->history_cap * 4, cs.s)); daux.history_ids = static_cast<uint32_t*>(dhi.p); }
        if (aux->history_dists && aux->history_cap) { HIP_TRY(dhd.alloc(nq * aux->history_cap * 4, cs.s)); daux.history_dists = static_cast<float*>(dhd.p); }
        if (aux->history_len) { HIP_TRY(dhl.alloc(nq * 4, cs.s)); daux.history_len = static_cast<uint32_t*>(dhl.p); }
    }
    StreamBuffer dcancel;
    const volatile int* host_cancel = aux ? aux->cancel_flag : nullptr;
    if (host_cancel) {
        HIP_TRY(dcancel.alloc(4, cs.s));
        HIP_TRY(hipMemsetAsync(dcancel.p, 0, 4, cs.s));
        if (*host_cancel) HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(dcancel.p), 1, 1, cs.s)); // already requested: no query starts
        daux.cancel_flag = static_cast<const volatile int*>(dcancel.p);
    }
    if (int rc = search_common(ix, dq.p, q_dtype, nq, k, params, static_cast<uint64_t*>(d64.p), static_cast<uint32_t*>(d32.p), static_cast<float*>(dd.p), &daux, cs.s))
        return rc;
    if (ids64) HIP_TRY(hipMemcpyAsync(ids64, d64.p, nq * k * 8, hipMemcpyDeviceToHost, cs.s));
    if (ids32) HIP_TRY(hipMemcpyAsync(ids32, d32.p, nq * k * 4, hipMemcpyDeviceToHost, cs.s));
    if (dists) HIP_TRY(hipMemcpyAsync(dists, dd.p, nq * k * 4, hipMemcpyDeviceToHost, cs.s));
    if (aux) {
        if (aux->stats) HIP_TRY(hipMemcpyAsync(aux->stats, dst.p, nq * 24, hipMemcpyDeviceToHost, cs.s));
        if (daux.history_ids) HIP_TRY(hipMemcpyAsync(aux->history_ids, dhi.p, nq * aux->history_cap * 4, hipMemcpyDeviceToHost, cs.s));
        if (daux.history_dists) HIP_TRY(hipMemcpyAsync(aux->history_dists, dhd.p, nq * aux->history_cap * 4, hipMemcpyDeviceToHost, cs.s));
        if (aux->history_len) HIP_TRY(hipMemcpyAsync(aux->history_len, dhl.p, nq * 4, hipMemcpyDeviceToHost, cs.s));
    }
    if (host_cancel) {
        // `cancel` of VamanaIndex::search (index.h:541-545): the caller may raise its word from any thread while the batch runs.
        // Poll it while the stream drains and forward it ONCE through a second stream (a one-thread kernel doing an agent-scope
        // store); the search kernel polls the device word once per hop with an agent-scope load.
        CallStream side;
        bool raised = *host_cancel != 0;
        while (hipStreamQuery(cs.s) == hipErrorNotReady) {
            if (!raised && *host_cancel != 0) {
                raised = true;
                if (!side.s) HIP_TRY(side.create());
                HIP_TRY(launch_set_word(static_cast<int*>(dcancel.p), 1, side.s));
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        if (side.s) HIP_TRY(hipStreamSynchronize(side.s));
    }
    HIP_TRY(hipStreamSynchronize(cs.s));
    return SVSX_SUCCESS;
}

int svsx_distance_batch(
    int metric, int a_dtype, int b_dtype, size_t d, const void* a, size_t na, const void* b, size_t nb, float* out, int device
) {
    if (metric < 0 || metric > 2) return fail(SVSX_INVALID_ARGUMENT, "unknown metric %d", metric);
    if (!dtype_size(a_dtype) || !dtype_size(b_dtype)) return fail(SVSX_INVALID_ARGUMENT, "unknown dtype");
    if (d == 0) return fail(SVSX_INVALID_ARGUMENT, "d must be positive");
    if (na == 0 || nb == 0) return SVSX_SUCCESS;
    if (!a || !b || !out) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    const int qmode = select_qmode(metric, a_dtype, b_dtype, 0);
    if (qmode < 0) return fail(SVSX_INVALID_ARGUMENT, "unsupported (%d, %d) type pair", a_dtype, b_dtype);
    HIP_TRY(hipSetDevice(device));
    DeviceBuffer da, db, dout;
    const size_t a_bytes = na * d * dtype_size(a_dtype);
    const size_t b_row = padded_row_bytes(d, b_dtype);
    HIP_TRY(da.alloc(a_bytes));
    HIP_TRY(db.alloc(b_row * nb + 64));
    HIP_TRY(dout.alloc(na * nb * 4));
    HIP_TRY(hipMemcpy(da.p, a, a_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(db.p, 0, b_row * nb + 64));
    HIP_TRY(hipMemcpy2D(db.p, b_row, b, d * dtype_size(b_dtype), d * dtype_size(b_dtype), nb, hipMemcpyHostToDevice));
    DistanceArgs args;
    args.a = da.p;
    args.b = db.p;
    args.b_row_bytes = b_row;
    args.a_dtype = a_dtype;
    args.na = static_cast<uint32_t>(na);
    args.nb = static_cast<uint32_t>(nb);
    args.d = static_cast<uint32_t>(d);
    args.out = static_cast<float*>(dout.p);
    hipError_t e = launch_distance(b_dtype, metric, qmode, args, nullptr);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "distance kernel launch failed: %s", hipGetErrorString(e));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, dout.p, na * nb * 4, hipMemcpyDeviceToHost));
    return SVSX_SUCCESS;
}

// rank at which the threshold levels take their next threshold: k + 4, at least 12, at most the shortlist length (round 4: k + 8, at least 16;
// with the cheaper appends of round 5 the ~16 kt appended rows per query weigh more: k + 4 is worth 4-8 % of a search, k + 2 one more per cent,
// and no query of the bench sets needs the exhaustive pass at either; profiles/r05_notes.md section 5)
#ifndef SVSX_FLAT_KT
#define SVSX_FLAT_KT(k, kk) ((k) + 4 > 12 ? ((k) + 4 < (kk) ? (k) + 4 : (kk)) : (12 < (kk) ? size_t(12) : (kk)))
#endif
int svsx_flat_search_device(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, void* hip_stream
) {
    return svsx_flat_search_device_counted(ix, d_queries, q_dtype, nq, k, d_ids64, d_dists, nullptr, hip_stream);
}

// host pointers in, host results out (svs::Flat::search through the C ABI): staging in stream order, one synchronisation at the end
int svsx_flat_search(const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, uint64_t* ids64, float* dists) {
    if (int rc = check_search_args(ix, queries, q_dtype, nq, k, false)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    if (!ids64) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    CallStream cs;
    HIP_TRY(cs.create());
    StreamBuffer dq, di, dd;
    const size_t q_bytes = nq * ix->d * dtype_size(q_dtype);
    HIP_TRY(dq.alloc(q_bytes, cs.s));
    HIP_TRY(di.alloc(nq * k * 8, cs.s));
    HIP_TRY(dd.alloc(nq * k * 4, cs.s));
    HIP_TRY(hipMemcpyAsync(dq.p, queries, q_bytes, hipMemcpyHostToDevice, cs.s));
    if (int rc = svsx_flat_search_device_counted(ix, dq.p, q_dtype, nq, k, static_cast<uint64_t*>(di.p), static_cast<float*>(dd.p), nullptr, cs.s)) {
        (void)hipStreamSynchronize(cs.s);
        return rc;
    }
    HIP_TRY(hipMemcpyAsync(ids64, di.p, nq * k * 8, hipMemcpyDeviceToHost, cs.s));
    if (dists) HIP_TRY(hipMemcpyAsync(dists, dd.p, nq * k * 4, hipMemcpyDeviceToHost, cs.s));
    HIP_TRY(hipStreamSynchronize(cs.s));
    return SVSX_SUCCESS;
}

int svsx_flat_search_device_counted(
    const svsx_index* ix, const void* d_queries, int q_dtype, size_t nq, size_t k, uint64_t* d_ids64, float* d_dists, uint32_t* d_recomputed,
    void* hip_stream
) {
    if (int rc = check_search_args(ix, d_queries, q_dtype, nq, k, false)) return rc;
    if (nq == 0) return SVSX_SUCCESS;
    if (!d_ids64) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    const int qmode = select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled);
    if (qmode < 0) return fail(SVSX_INVALID_ARGUMENT, "unsupported (query dtype %d, data dtype %d) pair", q_dtype, ix->data_dtype);
    if (qmode == QM_SCALAR)
        return fail(SVSX_NOT_IMPLEMENTED, "flat search serves the (query, data) type pairs with a SIMD kernel in the reference; (%d, %d) has the scalar loop only: graph search and svsx_distance_batch cover it", q_dtype, ix->data_dtype);
    HIP_TRY(hipSetDevice(ix->device));
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    // FlatIndex::search (index/flat/flat.h:322-460) bounds neither k nor the dimension.  The float16 pipeline below keeps k + 32 <= 256 shortlist
    // entries per query (and, beyond d = 128, the 64 queries of a wave) in LDS; what does not fit it -- k > 224, or a dimension whose 64
    // float16 queries plus k + 8 entries exceed 160 KB (d >= ~1 200) -- is answered by the exhaustive float32 pass alone: every query
    // against every row with the reference-order routines and a k-entry list (stage 5 of the pipeline, exact by construction; one wave per
    // query, so it is the slow path: 10 000 x 1 M x 96 takes ~1 s instead of 4 ms).
    auto exhaustive_only = [&]() -> int {
        const size_t lds = ((ix->d + 15) & ~size_t(15)) * 4 + k * 8;
        if (lds > 160 * 1024) return fail(SVSX_INVALID_ARGUMENT, "flat search: k = %zu at d = %zu needs %zu bytes of LDS for one query's list (160 KB available)", k, ix->d, lds);
        StreamBuffer redo;
        HIP_TRY(redo.alloc(nq * 4, stream));
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(redo.p), 1, nq, stream));
        FlatExactArgs xa;
        memset(&xa, 0, sizeof(xa));
        xa.r.data = ix->d_data; xa.r.row_bytes = ix->row_bytes; xa.r.d = static_cast<uint32_t>(ix->d); xa.r.sq_scale = ix->sq_scale; xa.r.sq_bias = ix->sq_bias;
        xa.r.id_offset = ix->id_offset; xa.r.queries = d_queries; xa.r.q_row_bytes = ix->d * dtype_size(q_dtype); xa.r.q_dtype = q_dtype;
        xa.r.nq = static_cast<uint32_t>(nq); xa.r.kk = static_cast<uint32_t>(k); xa.r.k = static_cast<uint32_t>(k);
        xa.r.ids64 = d_ids64; xa.r.dists = d_dists;
        xa.redo = static_cast<const uint32_t*>(redo.p); xa.n = static_cast<uint32_t>(ix->n);
        hipError_t e = launch_flat_exact(ix->data_dtype, ix->metric, qmode, xa, stream);
        if (e == hipSuccess && d_recomputed != nullptr) e = launch_add_word(d_recomputed, static_cast<uint32_t>(nq), stream);
        if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "flat search launch failed: %s", hipGetErrorString(e));
        return SVSX_SUCCESS;
    };
    if (k > 224) return exhaustive_only();
    // rows of the float16 shadow: padded to 16 elements, between 129 and 768 dimensions to 128 (the threshold kernel for those takes the
    // dimension in chunks of 128)
    const uint32_t dpad = static_cast<uint32_t>(ix->d > 128 && ix->d <= 768 ? (ix->d + 127) & ~size_t(127) : (ix->d + 15) & ~size_t(15));
    // shortlist length: max(4k, k + 32) up to 128 entries; beyond k = 96 (the reference's own ground-truth files hold 100 neighbours per
    // query, data/test_dataset/groundtruth_*.ivecs) k + 32, the general insert path of the shortlist kernel, up to 256 entries
    size_t kk = 4 * k > k + 32 ? 4 * k : k + 32;
    if (kk > 128) kk = k + 32 > 128 ? k + 32 : 128;
    if (kk > ix->n) kk = ix->n < k ? k : ix->n;
    {   // the shortlist kernel keeps 64 shortlists (and, beyond d = 128, the 64 queries) of a wave in LDS; where that does not fit the
        // shortlist shrinks towards k + 8 -- the answer stays exact, a shorter shortlist only makes the proof of stage 4 fail more often
        const size_t fixed = (dpad > 128 ? 64 * (static_cast<size_t>(dpad) * 2 + 16) : 0) + 256;
        while (fixed + 64 * kk * 8 > 160 * 1024 && kk > k + 8) kk -= 4;
        if (kk < k) kk = k;
        const size_t need = fixed + 64 * kk * 8;
        // (decided BEFORE the float16 shadow is built: the exhaustive pass never reads it, and at 1 M x 1 536 it would be 3 GB of HBM kept for the index's life -- ADVICE r5)
        if (need > 160 * 1024) return exhaustive_only(); // the 64 float16 queries of a wave + its shortlists do not fit (d >= ~1 200, or a long list at a large d)
    }
    {   // float16 shadow of the database + squared norms, built once
        std::lock_guard<std::mutex> lock(ix->shadow_mutex);
        if (!ix->d_half) {
            void* h = nullptr;
            float* nrm = nullptr;
            HIP_TRY(hipMalloc(&h, (ix->n + kShadowPadRows) * dpad * 2 + 64)); // padding: the threshold kernels load whole 32-row tiles / 128-row steps
            if (hipMalloc(reinterpret_cast<void**>(&nrm), ix->n * 4) != hipSuccess) { (void)hipFree(h); return fail(SVSX_RUNTIME_ERROR, "hipMalloc failed"); }
            // the padding tile is read (masked) by the last tile of the threshold kernel: zeros, not whatever the allocation held
            hipError_t e = hipMemsetAsync(static_cast<unsigned char*>(h) + ix->n * static_cast<size_t>(dpad) * 2, 0, kShadowPadRows * static_cast<size_t>(dpad) * 2 + 64, stream);
            if (e == hipSuccess)
                e = launch_to_half(ix->data_dtype, ix->d_data, ix->row_bytes, static_cast<uint32_t>(ix->n), static_cast<uint32_t>(ix->d), dpad,
                                   ix->sq_enabled, ix->sq_scale, ix->sq_bias, h, nrm, stream);
            uint32_t* mm = nullptr; // {max, min} bits of the squared norms
            if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&mm), 8);
            uint32_t range[2] = {0u, 0xFFFFFFFFu};
            if (e == hipSuccess) e = hipMemcpyAsync(mm, range, 8, hipMemcpyHostToDevice, stream);
            if (e == hipSuccess) e = launch_norm_range(nrm, static_cast<uint32_t>(ix->n), mm, stream);
            if (e == hipSuccess) e = hipMemcpyAsync(range, mm, 8, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (mm) (void)hipFree(mm);
            if (e != hipSuccess) { (void)hipFree(h); (void)hipFree(nrm); return fail(SVSX_RUNTIME_ERROR, "float16 shadow conversion failed: %s", hipGetErrorString(e)); }
            memcpy(&ix->shadow_max2, &range[0], 4);
            memcpy(&ix->shadow_min2, &range[1], 4);
            ix->d_half = h;
            ix->d_norm2 = nrm;
        }
    }
    const size_t nqb = (nq + 63) / 64;
    // one round of resident waves: LDS (64 shortlists of kk entries) bounds the waves per CU; a second round would
    // double the time, more slices than needed only add shortlist inserts (measured: 14.1 ms at 80 % of the slots)
    const size_t flat_lds = 64 * kk * 8 + 256 + (dpad > 128 ? 64 * (static_cast<size_t>(dpad) * 2 + 16) : 0); // shortlists + (beyond d = 128) the wave's 64 queries
    size_t waves_per_cu = (160 * 1024) / flat_lds;
    if (waves_per_cu > 8) waves_per_cu = 8;
    if (waves_per_cu < 1) waves_per_cu = 1;
    size_t target_waves = 256 * waves_per_cu * 4 / 5;
    size_t slices = (target_waves + nqb - 1) / nqb; // thresholds are shared across the slices of a query block
    if (slices > 64) slices = 64;
    const size_t max_slices = (ix->n + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const size_t q_stride = ix->d * dtype_size(q_dtype);
    StreamBuffer bqh, bsk, bsi, bmk, bmi, bredo, bdist;
    HIP_TRY(bredo.alloc(nq * 4, stream));
    if (!d_dists) { // the proof needs the k-th exact distance
        HIP_TRY(bdist.alloc(nq * k * 4, stream));
        d_dists = static_cast<float*>(bdist.p);
    }
    HIP_TRY(bqh.alloc(nq * dpad * 2 + 64, stream));
    HIP_TRY(bsk.alloc(slices * nq * kk * 4, stream));
    HIP_TRY(bsi.alloc(slices * nq * kk * 8, stream));
    HIP_TRY(bmk.alloc(nq * kk * 4, stream));
    HIP_TRY(bmi.alloc(nq * kk * 8, stream));
    void *qh = bqh.p, *sk = bsk.p, *si = bsi.p, *mk = bmk.p, *mi = bmi.p;
    hipError_t e = launch_to_half(q_dtype, d_queries, q_stride, static_cast<uint32_t>(nq), static_cast<uint32_t>(ix->d), dpad, 0, 1.0f, 0.0f, qh, nullptr, stream);
    FlatArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.xh = ix->d_half; fa.xnorm2 = ix->d_norm2; fa.n = static_cast<uint32_t>(ix->n); fa.dpad = dpad; fa.qh = qh;
    fa.nq = static_cast<uint32_t>(nq); fa.kk = static_cast<uint32_t>(kk); fa.n_slices = static_cast<uint32_t>(slices);
    fa.out_keys = static_cast<float*>(sk); fa.out_ids = static_cast<uint64_t*>(si);
    fa.tile_step = 1;
    StreamBuffer bgt;
    HIP_TRY(bgt.alloc(nq * 4, stream));
    fa.gtau = static_cast<uint32_t*>(bgt.p);
    MergeArgs ma;
    memset(&ma, 0, sizeof(ma));
    ma.dists = static_cast<const float*>(sk); ma.ids = static_cast<const uint64_t*>(si);
    ma.out_dists = static_cast<float*>(mk); ma.out_ids = static_cast<uint64_t*>(mi);
    ma.n_shards = static_cast<uint32_t>(slices); ma.nq = static_cast<uint32_t>(nq); ma.k = static_cast<uint32_t>(kk);
    RerankArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.data = ix->d_data; ra.row_bytes = ix->row_bytes; ra.d = static_cast<uint32_t>(ix->d); ra.sq_scale = ix->sq_scale; ra.sq_bias = ix->sq_bias;
    ra.id_offset = ix->id_offset; ra.queries = d_queries; ra.q_row_bytes = q_stride; ra.q_dtype = q_dtype;
    ra.nq = static_cast<uint32_t>(nq); ra.kk = static_cast<uint32_t>(kk); ra.k = static_cast<uint32_t>(k);
    ra.cand_ids = static_cast<const uint64_t*>(mi); ra.ids64 = d_ids64; ra.dists = d_dists;
    // Large sets with d <= 128 take the threshold pipeline (svsx_flat.hip stage 1b): nested strided samples of the 32-row
    // tiles, every 16^j-th tile for j = L .. 1, each ~16x larger than the one before.  The smallest (>= 256 rows) is
    // taken whole (threshold +inf); at every level the threshold kernel appends the rows at or below the current
    // threshold (about 16 kk per query) and the select kernel takes the kk-th best key of that list as the next, tighter
    // threshold -- which at least kk rows of the next level satisfy, because the samples are nested.  The last level is
    // the full set; its lists (>= kk rows, ~16 kk expected) are re-ranked exactly.  Queries whose lists overflow
    // (unrepresentative samples) are recomputed by the full shortlist kernel, gated per 64-query block.
    FlatProofArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.metric = ix->metric; pa.nq = ra.nq; pa.k = ra.k; pa.kk = ra.kk; pa.d = ra.d; pa.dpad = dpad;
    pa.queries = d_queries; pa.q_row_bytes = q_stride; pa.q_dtype = q_dtype;
    pa.sq_l2 = qmode == QM_SQ_L2 ? 1 : 0; pa.data_dtype = ix->data_dtype; pa.sq_scale = ix->sq_scale; pa.sq_bias = ix->sq_bias;
    pa.out_ids = d_ids64; pa.out_dists = d_dists;
    pa.short_keys = static_cast<const float*>(mk); pa.short_ids = static_cast<const uint64_t*>(mi);
    pa.xmax2 = ix->shadow_max2; pa.xmin2 = ix->shadow_min2;
    pa.redo = static_cast<uint32_t*>(bredo.p); pa.redo_count = d_recomputed;
    bool thresh_path = flat_thresh_supports(dpad) && ix->n >= 65536 && nq <= 65536;
#ifdef SVSX_DIAG
    if (getenv("SVSX_FLAT_SHORTLIST_ONLY")) thresh_path = false; // A/B builds only
#endif
    StreamBuffer bcand, bkeys, bcnt, bgate, btau_a, btau_b;
    if (thresh_path) {
        // sampling unit: a 32-row tile (d > 128) or a 64- / 128-row step of the four-wave kernel (d <= 128)
        const size_t unit = flat_thresh_unit_rows(dpad);
        const size_t units = (ix->n + unit - 1) / unit;
        uint32_t step = 1;
        while ((units / (static_cast<size_t>(step) * 16)) * unit >= 128) step *= 16; // coarsest level: 128 ... 2 047 rows
        size_t s2 = flat_thresh_slices(ix->metric, dpad, nq); // one generation of resident workgroups
        if (s2 > ix->n / 4096) s2 = ix->n / 4096;
        if (s2 < 1) s2 = 1;
        const size_t lps = flat_thresh_lists_per_slice(dpad); // candidate lists per (query, slice)
        if (s2 * lps > 1024) s2 = 1024 / lps;
        const size_t lists = s2 * lps;
        // rows per list: ~16 kk / lists expected at every level, at most the coarsest level's rows; 4x head room
        size_t cap = 512;
        // The thresholds are taken at rank kt < kk: with the proof of stage 4 behind it, the threshold pipeline needs no float16 safety margin
        // in its candidate lists -- ~16 kt rows per query (224 for k = 10) are appended and re-ranked instead of ~16 kk (640), and whether
        // that was enough is decided per query by the proof, not by the margin.  (The shortlist kernel keeps kk: its proof needs the gap
        // between the k-th and the kk-th key.)
        const size_t kt = SVSX_FLAT_KT(k, kk);
        while (cap < 4 * (16 * kt / lists + 1)) cap *= 2;
        const size_t coarse_rows = (units / step + 1) * unit; // all of them are appended at the first level
        while (cap < 2 * coarse_rows / s2 && cap < 8192) cap *= 2;
        HIP_TRY(bcand.alloc(nq * lists * cap * 4, stream));
        HIP_TRY(bkeys.alloc(nq * lists * cap * 4, stream));
        HIP_TRY(bcnt.alloc(nq * lists * 4, stream));
        HIP_TRY(bgate.alloc(nqb * 4, stream));
        HIP_TRY(btau_a.alloc(nq * 4, stream));
        HIP_TRY(btau_b.alloc(nq * 4, stream));
        float* tau_cur = static_cast<float*>(btau_a.p);
        float* tau_nxt = static_cast<float*>(btau_b.p);
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(tau_cur), 0x7F800000, nq, stream); // +inf
        FlatThreshArgs ta;
        memset(&ta, 0, sizeof(ta));
        ta.xh = ix->d_half; ta.xnorm2 = ix->d_norm2; ta.n = static_cast<uint32_t>(ix->n); ta.dpad = dpad; ta.qh = qh;
        ta.nq = static_cast<uint32_t>(nq);
        ta.n_slices = static_cast<uint32_t>(s2);
        ta.tau_stride = 1;
        ta.cand = static_cast<uint32_t*>(bcand.p);
        ta.cnt = static_cast<uint32_t*>(bcnt.p); ta.cap = static_cast<uint32_t>(cap);
        for (; step > 1; step /= 16) {
            ta.tile_step = step;
            ta.tau = tau_cur;
            ta.cand_keys = static_cast<float*>(bkeys.p);
            if (e == hipSuccess) e = launch_flat_thresh(ix->metric, ta, stream);
            if (e == hipSuccess) e = launch_flat_select(ta.cand_keys, ta.cnt, ta.nq, static_cast<uint32_t>(lists), ta.cap, static_cast<uint32_t>(kt), tau_cur, 1, tau_nxt, stream);
            float* t = tau_cur; tau_cur = tau_nxt; tau_nxt = t;
        }
        ta.tile_step = 1;
        ta.tau = tau_cur;
        ta.cand_keys = nullptr;
        if (e == hipSuccess) e = launch_flat_thresh(ix->metric, ta, stream);
        if (e == hipSuccess) e = launch_flat_gate_kk(ta.cnt, ta.nq, static_cast<uint32_t>(lists), ta.cap, static_cast<uint32_t>(kk), static_cast<uint32_t*>(bgate.p), stream);
        // gated fallback: full shortlist pass and merge only for the query blocks that overflowed
        fa.tile_step = 1;
        fa.block_gate = static_cast<const uint32_t*>(bgate.p);
        ma.gate64 = fa.block_gate;
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bgt.p), 0xFF800000, nq, stream);
        if (e == hipSuccess) e = launch_flat_mfma(ix->metric, fa, stream);
        if (e == hipSuccess) e = launch_merge(M_L2, ma, stream);
        ra.cand32 = ta.cand; ra.cnt = ta.cnt; ra.cap = ta.cap; ra.n_lists = static_cast<uint32_t>(lists);
        if (e == hipSuccess) e = launch_rerank2(ix->data_dtype, ix->metric, qmode, ra, stream);
        pa.tau = tau_cur; pa.cnt = ta.cnt; pa.cap = ta.cap; pa.n_lists = static_cast<uint32_t>(lists);
        if (e == hipSuccess) e = launch_flat_certify(pa, stream);
    } else {
        if (e == hipSuccess) e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(bgt.p), 0xFF800000, nq, stream); // +inf, order-encoded
        if (e == hipSuccess) e = launch_flat_mfma(ix->metric, fa, stream);
        if (e == hipSuccess) e = launch_merge(M_L2, ma, stream); // keys: smaller is better for every metric
        if (e == hipSuccess) e = launch_rerank(ix->data_dtype, ix->metric, qmode, ra, stream);
        if (e == hipSuccess) e = launch_flat_certify(pa, stream);
    }
    // the queries whose shortlist is not PROVEN to hold the exact answer are recomputed exhaustively in float32 (svsx_flat.hip stages 4, 5)
    FlatExactArgs xa;
    xa.r = ra; xa.redo = pa.redo; xa.n = static_cast<uint32_t>(ix->n);
    if (e == hipSuccess) e = launch_flat_exact(ix->data_dtype, ix->metric, qmode, xa, stream);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "flat search launch failed: %s", hipGetErrorString(e));
    return SVSX_SUCCESS;
}

int svsx_get_distance(const svsx_index* ix, uint64_t id, const void* query, int q_dtype, double* distance) {
    if (int rc = check_search_args(ix, query, q_dtype, 1, 1, false)) return rc;
    if (!distance) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    const uint64_t local = id - ix->id_offset;
    if (id < ix->id_offset || local >= ix->n) // index/vamana/index.h:916-919
        return fail(SVSX_INVALID_ARGUMENT, "ID %llu is out of bounds for index of size %zu!", static_cast<unsigned long long>(id), static_cast<size_t>(ix->n));
    const int qmode = select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled);
    if (qmode < 0) return fail(SVSX_INVALID_ARGUMENT, "unsupported (query dtype %d, data dtype %d%s) pair", q_dtype, ix->data_dtype, ix->sq_enabled ? ", SQ" : "");
    HIP_TRY(hipSetDevice(ix->device));
    CallStream cs;
    HIP_TRY(cs.create());
    const size_t q_row = ix->d * dtype_size(q_dtype);
    if (qmode == QM_SCALAR) { // the scalar-loop pairs: the distance plugin kernel on this one row
        StreamBuffer dq1, dd1;
        HIP_TRY(dq1.alloc(q_row, cs.s));
        HIP_TRY(dd1.alloc(4, cs.s));
        HIP_TRY(hipMemcpyAsync(dq1.p, query, q_row, hipMemcpyHostToDevice, cs.s));
        DistanceArgs da;
        da.a = dq1.p; da.b = static_cast<const unsigned char*>(ix->d_data) + local * ix->row_bytes; da.b_row_bytes = ix->row_bytes; da.a_dtype = q_dtype;
        da.na = 1; da.nb = 1; da.d = static_cast<uint32_t>(ix->d); da.out = static_cast<float*>(dd1.p);
        const hipError_t e1 = launch_distance(ix->data_dtype, ix->metric, qmode, da, cs.s);
        if (e1 != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "distance kernel launch failed: %s", hipGetErrorString(e1));
        float f1 = 0.0f;
        HIP_TRY(hipMemcpyAsync(&f1, dd1.p, 4, hipMemcpyDeviceToHost, cs.s));
        HIP_TRY(hipStreamSynchronize(cs.s));
        *distance = static_cast<double>(f1);
        return SVSX_SUCCESS;
    }
    StreamBuffer dq, dc, di, dd;
    HIP_TRY(dq.alloc(q_row, cs.s));
    HIP_TRY(dc.alloc(8, cs.s));
    HIP_TRY(di.alloc(8, cs.s));
    HIP_TRY(dd.alloc(4, cs.s));
    HIP_TRY(hipMemcpyAsync(dq.p, query, q_row, hipMemcpyHostToDevice, cs.s));
    HIP_TRY(hipMemcpyAsync(dc.p, &local, 8, hipMemcpyHostToDevice, cs.s));
    // one query, one candidate: the exact-order re-rank kernel of the flat search is the index's distance kernel applied to chosen rows
    RerankArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.data = ix->d_data; ra.row_bytes = ix->row_bytes; ra.d = static_cast<uint32_t>(ix->d); ra.sq_scale = ix->sq_scale; ra.sq_bias = ix->sq_bias;
    ra.id_offset = ix->id_offset; ra.queries = dq.p; ra.q_row_bytes = q_row; ra.q_dtype = q_dtype; ra.nq = 1; ra.kk = 1; ra.k = 1;
    ra.cand_ids = static_cast<const uint64_t*>(dc.p); ra.ids64 = static_cast<uint64_t*>(di.p); ra.dists = static_cast<float*>(dd.p);
    const hipError_t e = launch_rerank(ix->data_dtype, ix->metric, qmode, ra, cs.s);
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "distance kernel launch failed: %s", hipGetErrorString(e));
    float f = 0.0f;
    HIP_TRY(hipMemcpyAsync(&f, dd.p, 4, hipMemcpyDeviceToHost, cs.s));
    HIP_TRY(hipStreamSynchronize(cs.s));
    *distance = static_cast<double>(f);
    return SVSX_SUCCESS;
}

int svsx_prune_device(
    const svsx_index* ix, const int32_t* d_node_ids, const int32_t* d_pool_ids, const float* d_pool_dists, size_t n_nodes,
    size_t pool_size, size_t max_out, float alpha, int32_t* d_out, size_t out_stride, void* hip_stream
) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (n_nodes == 0) return SVSX_SUCCESS;
    if (!d_node_ids || !d_pool_ids || !d_pool_dists || !d_out) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    if (pool_size == 0 || max_out == 0 || out_stride < max_out) return fail(SVSX_INVALID_ARGUMENT, "pool_size, max_out must be positive and out_stride >= max_out");
    HIP_TRY(hipSetDevice(ix->device));
    PruneArgs a;
    a.data = ix->d_data;
    a.row_bytes = ix->row_bytes;
    a.d = static_cast<uint32_t>(ix->d);
    a.sq_scale = ix->sq_scale;
    a.sq_bias = ix->sq_bias;
    a.node_ids = d_node_ids;
    a.pool_ids = d_pool_ids;
    a.pool_dists = d_pool_dists;
    a.n_nodes = static_cast<uint32_t>(n_nodes);
    a.pool_size = static_cast<uint32_t>(pool_size);
    a.max_out = static_cast<uint32_t>(max_out);
    a.alpha = alpha;
    a.out = d_out;
    a.out_stride = static_cast<uint32_t>(out_stride);
    hipError_t e = launch_prune(ix->data_dtype, ix->metric, ix->sq_enabled != 0, a, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "prune kernel launch failed: %s", hipGetErrorString(e));
    return SVSX_SUCCESS;
}

} // extern "C"

namespace svsx { // svsx_builder.hip
struct BuildTarget {
    const svsx_index* index;
    const void* d_data;
    size_t row_bytes;
    int data_dtype, metric;
    uint32_t n, d;
    int sq_enabled;
    float sq_scale, sq_bias;
    uint32_t* d_graph;
    uint32_t gstride;
};
struct BuildParams {
    float alpha;
    uint32_t graph_max_degree, window_size, max_candidate_pool_size, prune_to;
    int use_full_search_history;
};
int build_vamana(const BuildTarget& t, const BuildParams& p, hipStream_t stream, std::string& err);
int find_medoid_device(const BuildTarget& t, uint32_t* medoid, hipStream_t stream, std::string& err);
} // namespace svsx

static svsx::BuildTarget build_target(const svsx_index* ix) {
    svsx::BuildTarget t;
    t.index = ix;
    t.d_data = ix->d_data;
    t.row_bytes = ix->row_bytes;
    t.data_dtype = ix->data_dtype;
    t.metric = ix->metric;
    t.n = static_cast<uint32_t>(ix->n);
    t.d = static_cast<uint32_t>(ix->d);
    t.sq_enabled = ix->sq_enabled;
    t.sq_scale = ix->sq_scale;
    t.sq_bias = ix->sq_bias;
    t.d_graph = ix->d_graph;
    t.gstride = static_cast<uint32_t>(ix->gstride);
    return t;
}

extern "C" {

int svsx_vamana_build_device(svsx_index* ix, const svsx_build_params* params, void* hip_stream) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (!params) return fail(SVSX_INVALID_ARGUMENT, "null build parameters");
    if (!ix->d_graph || ix->n_entry != 1) return fail(SVSX_INVALID_ARGUMENT, "the build needs a graph to fill and exactly one entry point");
    if (ix->n >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "too many vectors for one graph");
    if (params->graph_max_degree == 0 || params->graph_max_degree > ix->gstride)
        return fail(SVSX_INVALID_ARGUMENT, "graph_max_degree %zu does not fit the graph rows of %zu ids", params->graph_max_degree, ix->gstride);
    if (params->window_size == 0 || params->window_size > 8000) return fail(SVSX_INVALID_ARGUMENT, "build window_size out of range");
    if (!(params->alpha > 0.0f)) return fail(SVSX_INVALID_ARGUMENT, "alpha must be positive");
    if (ix->d_fat || ix->d_codes) return fail(SVSX_INVALID_ARGUMENT, "bound codes describe a finished graph: build first");
    HIP_TRY(hipSetDevice(ix->device));
    svsx::BuildParams p;
    p.alpha = params->alpha;
    p.graph_max_degree = static_cast<uint32_t>(params->graph_max_degree);
    p.window_size = static_cast<uint32_t>(params->window_size);
    p.use_full_search_history = params->use_full_search_history ? 1 : 0;
    p.max_candidate_pool_size = static_cast<uint32_t>(params->max_candidate_pool_size ? params->max_candidate_pool_size : 750);
    p.prune_to = static_cast<uint32_t>(params->prune_to);
    std::string err;
    const int rc = svsx::build_vamana(build_target(ix), p, static_cast<hipStream_t>(hip_stream), err);
    if (rc != SVSX_SUCCESS) return fail(rc, "%s", err.c_str());
    return SVSX_SUCCESS;
}

int svsx_find_medoid_device(const svsx_index* ix, uint32_t* medoid) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (!medoid) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(ix->device));
    std::string err;
    const int rc = svsx::find_medoid_device(build_target(ix), medoid, nullptr, err);
    if (rc != SVSX_SUCCESS) return fail(rc, "%s", err.c_str());
    return SVSX_SUCCESS;
}

int svsx_merge_topk_strided_device(
    int metric, size_t n_shards, size_t nq, size_t k, const float* d_dists, size_t dist_shard_stride, const uint64_t* d_ids,
    size_t id_shard_stride, float* d_out_dists, uint64_t* d_out_ids, void* hip_stream
) {
    if (metric < 0 || metric > 2) return fail(SVSX_INVALID_ARGUMENT, "unknown metric %d", metric);
    if (n_shards == 0 || n_shards > 64) return fail(SVSX_INVALID_ARGUMENT, "n_shards must be in [1, 64]");
    if (k == 0 || k > 65535) return fail(SVSX_INVALID_ARGUMENT, "k out of range");
    if (nq == 0) return SVSX_SUCCESS;
    if (!d_dists || !d_ids || !d_out_dists || !d_out_ids) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    MergeArgs a;
    memset(&a, 0, sizeof(a));
    a.dists = d_dists;
    a.ids = d_ids;
    a.out_dists = d_out_dists;
    a.out_ids = d_out_ids;
    a.n_shards = static_cast<uint32_t>(n_shards);
    a.nq = static_cast<uint32_t>(nq);
    a.k = static_cast<uint32_t>(k);
    a.dist_shard_stride = dist_shard_stride;
    a.id_shard_stride = id_shard_stride;
    hipError_t e = launch_merge(metric, a, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "merge kernel launch failed: %s", hipGetErrorString(e));
    return SVSX_SUCCESS;
}

int svsx_merge_topk_device(
    int metric, size_t n_shards, size_t nq, size_t k, const float* d_dists, const uint64_t* d_ids, float* d_out_dists,
    uint64_t* d_out_ids, void* hip_stream
) {
    return svsx_merge_topk_strided_device(metric, n_shards, nq, k, d_dists, 0, d_ids, 0, d_out_dists, d_out_ids, hip_stream);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// Batched BatchIterator (index/vamana/iterator.h:81-339): see include/svsx.h
struct svsx_batch_iterator {
    const svsx_index* ix = nullptr;
    int q_dtype = 0;
    int qmode = 0;
    size_t nq = 0;
    size_t extra = 0;
    size_t q_stride = 0;
    void* d_queries = nullptr;
    uint2* d_entries = nullptr;
    size_t estride = 0;       // entries per query
    uint32_t* d_meta = nullptr;
    uint32_t* d_yielded = nullptr;
    size_t ystride = 0;
    size_t max_capacity = 0;  // upper bound of any query's capacity
    size_t yield_bound = 0;   // upper bound of any query's yielded count
};

namespace {

int iterator_reset_meta(svsx_batch_iterator* it, hipStream_t stream) {
    std::vector<uint32_t> meta(it->nq * 8, 0u);
    for (size_t i = 0; i < it->nq; ++i) {
        meta[i * 8 + 3] = static_cast<uint32_t>(it->extra); // initialize_buffer: SearchBufferConfig{0, extra} (iterator.h:127-131)
        meta[i * 8 + 6] = 1u;                               // restart_search_ = true
    }
    HIP_TRY(hipMemcpyAsync(it->d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    it->max_capacity = it->extra;
    it->yield_bound = 0;
    return SVSX_SUCCESS;
}

// grow a [nq][stride] device table, keeping the first `keep` elements of every row
template <typename T> int grow_rows(T*& p, size_t& stride, size_t need, size_t nq, size_t keep) {
    if (need <= stride) return SVSX_SUCCESS;
    size_t ns = stride * 2 > need ? stride * 2 : need;
    ns = (ns + 63) & ~size_t(63);
    T* np = nullptr;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&np), ns * nq * sizeof(T)));
    if (p && keep) {
        hipError_t e = hipMemcpy2D(np, ns * sizeof(T), p, stride * sizeof(T), keep * sizeof(T), nq, hipMemcpyDeviceToDevice);
        if (e != hipSuccess) {
            (void)hipFree(np);
            return fail(SVSX_RUNTIME_ERROR, "iterator state copy failed: %s", hipGetErrorString(e));
        }
    }
    if (p) (void)hipFree(p);
    p = np;
    stride = ns;
    return SVSX_SUCCESS;
}

} // namespace

int svsx_iterator_create(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t extra, svsx_batch_iterator** out
) {
    if (!out) return fail(SVSX_INVALID_ARGUMENT, "null output handle");
    *out = nullptr;
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (!ix->d_graph) return fail(SVSX_INVALID_ARGUMENT, "this index was created without a graph (flat search only)");
    if (dtype_size(q_dtype) == 0) return fail(SVSX_INVALID_ARGUMENT, "unknown query dtype %d", q_dtype);
    if (nq == 0 || !queries) return fail(SVSX_INVALID_ARGUMENT, "null queries");
    if (nq >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "too many queries in one batch");
    const int qmode = select_qmode(ix->metric, q_dtype, ix->data_dtype, ix->sq_enabled);
    if (qmode < 0)
        return fail(SVSX_INVALID_ARGUMENT, "unsupported (query dtype %d, data dtype %d%s) pair", q_dtype, ix->data_dtype, ix->sq_enabled ? ", SQ" : "");
    if (extra == ~size_t(0)) extra = 100; // lib/preprocessor.h:177,187
    if (extra >= (size_t(1) << 31)) return fail(SVSX_INVALID_ARGUMENT, "extra search buffer capacity %zu out of range", extra); // beyond one wave's LDS the buffers live in HBM
    HIP_TRY(hipSetDevice(ix->device));
    svsx_batch_iterator* it = new (std::nothrow) svsx_batch_iterator();
    if (!it) return fail(SVSX_RUNTIME_ERROR, "out of memory");
    it->ix = ix;
    it->q_dtype = q_dtype;
    it->qmode = qmode;
    it->nq = nq;
    it->extra = extra;
    it->q_stride = ix->d * dtype_size(q_dtype);
    int rc = SVSX_SUCCESS;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&it->d_queries, it->q_stride * nq));
        HIP_TRY(hipMemcpy(it->d_queries, queries, it->q_stride * nq, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&it->d_meta), nq * 32));
        return iterator_reset_meta(it, nullptr);
    };
    rc = body();
    if (rc != SVSX_SUCCESS) {
        svsx_iterator_destroy(it);
        return rc;
    }
    *out = it;
    return SVSX_SUCCESS;
}

void svsx_iterator_destroy(svsx_batch_iterator* it) {
    if (!it) return;
    (void)hipSetDevice(it->ix->device);
    if (it->d_queries) (void)hipFree(it->d_queries);
    if (it->d_entries) (void)hipFree(it->d_entries);
    if (it->d_meta) (void)hipFree(it->d_meta);
    if (it->d_yielded) (void)hipFree(it->d_yielded);
    delete it;
}

int svsx_iterator_update(svsx_batch_iterator* it, const void* queries) {
    if (!it) return fail(SVSX_NOT_INITIALIZED, "Iterator not initialized");
    if (!queries) return fail(SVSX_INVALID_ARGUMENT, "null queries");
    HIP_TRY(hipSetDevice(it->ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(it->d_queries, queries, it->q_stride * it->nq, hipMemcpyHostToDevice));
    return iterator_reset_meta(it, nullptr);
}

int svsx_iterator_restart_next_search(svsx_batch_iterator* it) {
    if (!it) return fail(SVSX_NOT_INITIALIZED, "Iterator not initialized");
    HIP_TRY(hipSetDevice(it->ix->device));
    HIP_TRY(hipDeviceSynchronize());
    std::vector<uint32_t> meta(it->nq * 8);
    HIP_TRY(hipMemcpy(meta.data(), it->d_meta, meta.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < it->nq; ++i) meta[i * 8 + 6] |= 1u;
    HIP_TRY(hipMemcpy(it->d_meta, meta.data(), meta.size() * 4, hipMemcpyHostToDevice));
    return SVSX_SUCCESS;
}

int svsx_iterator_state(const svsx_batch_iterator* it, uint32_t* meta) {
    if (!it) return fail(SVSX_NOT_INITIALIZED, "Iterator not initialized");
    if (!meta) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    HIP_TRY(hipSetDevice(it->ix->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(meta, it->d_meta, it->nq * 32, hipMemcpyDeviceToHost));
    return SVSX_SUCCESS;
}

int svsx_iterator_next_device(
    svsx_batch_iterator* it, size_t batch_size, const uint32_t* d_batch_sizes, uint64_t* d_ids64, float* d_dists,
    uint32_t* d_counts, void* hip_stream
) {
    if (!it) return fail(SVSX_NOT_INITIALIZED, "Iterator not initialized");
    if ((batch_size && (!d_ids64 || !d_dists)) || !d_counts) return fail(SVSX_INVALID_ARGUMENT, "null output");
    if (batch_size >= 0xFFFFFFFFull) return fail(SVSX_INVALID_ARGUMENT, "batch size too large");
    const svsx_index* ix = it->ix;
    const size_t cap = it->max_capacity + batch_size;
    // up to 16 000 entries the search buffer of a query lives in its wave's LDS; beyond that (low-hit-rate filters ask for batches
    // up to n, vamana_index_impl.h:176-213) the kernel works in place on the query's row of the HBM state table
    const bool spill = cap > 16000;
    if (cap > (size_t(1) << 27)) return fail(SVSX_INVALID_ARGUMENT, "search buffer capacity %zu is out of range", cap);
    HIP_TRY(hipSetDevice(ix->device));
    size_t ybound = it->yield_bound + batch_size;
    if (ybound > ix->n) ybound = ix->n;
    if (int rc = grow_rows(it->d_entries, it->estride, cap + 1, it->nq, it->estride)) return rc;
    if (int rc = grow_rows(it->d_yielded, it->ystride, ybound ? ybound : 1, it->nq, it->ystride)) return rc;
    SearchArgs a = base_search_args(ix, it->d_queries, it->q_dtype, it->q_stride, it->nq);
    a.general = 1;
    a.codes = nullptr;
    a.fat = nullptr; // the iterator kernel reads neither code rows nor inline records: the default filter below then keeps its 8-bit tags
    a.window = static_cast<uint32_t>(spill ? 1 : cap); // per-query values come from the state table; this sizes the LDS buffer
    a.capacity = static_cast<uint32_t>(spill ? 1 : cap);
    a.it_spill = spill ? 1u : 0u;
    default_filter(ix, a, (ix->n >> 12) < 0xFFu ? 4096u : kDefaultFilterEntries); // result-neutral within one call
    a.it_entries = it->d_entries;
    a.it_estride = static_cast<uint32_t>(it->estride);
    a.it_meta = it->d_meta;
    a.it_yielded = it->d_yielded;
    a.it_ystride = static_cast<uint32_t>(it->ystride);
    a.it_batch_sizes = d_batch_sizes;
    a.it_batch = static_cast<uint32_t>(batch_size);
    a.it_out_ids = d_ids64;
    a.it_out_dists = d_dists;
    a.it_counts = d_counts;
    hipError_t e = launch_search(ix->data_dtype, ix->metric, it->qmode, a, static_cast<hipStream_t>(hip_stream));
    if (e != hipSuccess) return fail(SVSX_RUNTIME_ERROR, "iterator kernel launch failed: %s", hipGetErrorString(e));
    it->max_capacity = cap;
    it->yield_bound = ybound;
    return SVSX_SUCCESS;
}

int svsx_iterator_next(
    svsx_batch_iterator* it, size_t batch_size, const uint32_t* batch_sizes, uint64_t* ids64, float* dists, uint32_t* counts
) {
    if (!it) return fail(SVSX_NOT_INITIALIZED, "Iterator not initialized");
    if ((batch_size && (!ids64 || !dists)) || !counts) return fail(SVSX_INVALID_ARGUMENT, "null output");
    HIP_TRY(hipSetDevice(it->ix->device));
    CallStream cs;
    HIP_TRY(cs.create());
    StreamBuffer dbs, d64, dd, dc;
    const size_t nq = it->nq;
    if (batch_sizes) {
        HIP_TRY(dbs.alloc(nq * 4, cs.s));
        HIP_TRY(hipMemcpyAsync(dbs.p, batch_sizes, nq * 4, hipMemcpyHostToDevice, cs.s));
    }
    HIP_TRY(d64.alloc(nq * batch_size * 8, cs.s));
    HIP_TRY(dd.alloc(nq * batch_size * 4, cs.s));
    HIP_TRY(dc.alloc(nq * 4, cs.s));
    if (int rc = svsx_iterator_next_device(it, batch_size, static_cast<const uint32_t*>(dbs.p), static_cast<uint64_t*>(d64.p),
                                           static_cast<float*>(dd.p), static_cast<uint32_t*>(dc.p), cs.s))
        return rc;
    if (batch_size) {
        HIP_TRY(hipMemcpyAsync(ids64, d64.p, nq * batch_size * 8, hipMemcpyDeviceToHost, cs.s));
        HIP_TRY(hipMemcpyAsync(dists, dd.p, nq * batch_size * 4, hipMemcpyDeviceToHost, cs.s));
    }
    HIP_TRY(hipMemcpyAsync(counts, dc.p, nq * 4, hipMemcpyDeviceToHost, cs.s));
    HIP_TRY(hipStreamSynchronize(cs.s));
    return SVSX_SUCCESS;
}


// ---------------------------------------------------------------------------------------------------------------
// Batch search with an id filter: the runtime's VamanaIndex::search(n, x, k, distances, labels, params, IDFilter*)
// (bindings/cpp/include/svs/runtime/vamana_index.h:75-83; algorithm bindings/cpp/src/vamana_index_impl.h:140-218 with
// filter_estimate_batch = false, helpers svs_runtime_utils.h:411-434).  Per query: pull adaptive batches from its
// BatchIterator until k members of the filter are found, the iterator is done, or the observed hit rate falls below
// filter_stop.  All queries advance together, one kernel launch per round (per-query batch sizes, finished queries skipped).
extern "C" int svsx_search_filtered_ex(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    const uint64_t* id_bitmap, float filter_stop, int filter_estimate_batch, uint64_t* ids64, float* dists
);

extern "C" int svsx_search_filtered(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    const uint64_t* id_bitmap, float filter_stop, uint64_t* ids64, float* dists
) {
    return svsx_search_filtered_ex(ix, queries, q_dtype, nq, k, params, id_bitmap, filter_stop, 0, ids64, dists);
}

extern "C" int svsx_search_filtered_ex(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params,
    const uint64_t* id_bitmap, float filter_stop, int filter_estimate_batch, uint64_t* ids64, float* dists
) {
    if (int rc = check_search_args(ix, queries, q_dtype, nq, k)) return rc;
    if (!id_bitmap || !ids64 || !dists) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    if (nq == 0) return SVSX_SUCCESS;
    for (size_t i = 0; i < nq * k; ++i) { // Unspecify<size_t>() / Unspecify<float>() padding (vamana_index_impl.h:205-209)
        ids64[i] = ~0ull;
        dists[i] = __builtin_inff();
    }
    const size_t sws = params ? params->search_window_size : 0;
    const size_t max_batch = ix->n;
    const size_t q_row = ix->d * dtype_size(q_dtype);
    size_t initial_batch = std::max(k, sws);
    if (filter_estimate_batch) {
        // Pre-search filter sampling (vamana_index_impl.h:153-175; sample_filter_hits, svs_runtime_utils.h:445-471): the hit rate of up to
        // max(200, 1 / filter_stop) ids drawn by std::mt19937(42) through std::uniform_int_distribution<size_t>(0, n - 1) -- the standard
        // library's own classes, so the draws are the reference's whenever both are built against the same libstdc++ -- decides whether to
        // search at all and the first batch size of every query.
        const size_t sample_size = filter_stop > 0.0f ? std::max<size_t>(200, static_cast<size_t>(1.0f / filter_stop)) : 200;
        const size_t target = std::min(sample_size, static_cast<size_t>(ix->n));
        std::mt19937 rng(42);
        std::uniform_int_distribution<size_t> draw(0, ix->n - 1);
        size_t hits = 0, checked = 0;
        for (size_t tries = 0; checked < target && tries < target * 4; ++tries) {
            const size_t id = draw(rng);
            hits += (id_bitmap[id >> 6] >> (id & 63)) & 1ull;
            ++checked;
        }
        if (filter_stop > 0.0f && checked != 0 && (hits == 0 || static_cast<float>(hits) / checked < filter_stop)) return SVSX_SUCCESS; // all padding
        if (hits != 0 && hits < k) {
            const float bs = static_cast<float>(k - hits) * checked / hits;
            initial_batch = std::min(std::max(static_cast<size_t>(bs), size_t(1)), max_batch);
        } else {
            initial_batch = std::min(initial_batch, max_batch);
        }
    }
    auto predict = [](size_t processed, size_t hits, size_t goal, size_t hint, size_t max_value) -> size_t { // svs_runtime_utils.h:411-419
        if (hits == 0 || hits >= goal) return std::min(hint, max_value);
        const float bs = static_cast<float>(goal - hits) * processed / hits;
        return std::min(std::max(static_cast<size_t>(bs), size_t(1)), max_value);
    };
    // One group of queries advancing together, one kernel launch per round.  Every buffer of a group is sized by the largest batch /
    // capacity ANY of its queries asks for (iterator state table: m * (capacity + 1) * 8 bytes), and one low-hit-rate query can ask for
    // batches of up to n: a query whose search buffer would outgrow `cap_limit` entries is therefore taken out of the group and handed
    // back in `postponed`; the caller runs those in small groups of their own, from the start -- a query's batch schedule depends on
    // nothing but its own hits, so the replay gives the answer the reference's per-query loop (vamana_index_impl.h:176-213) gives.
    auto run_group = [&](const std::vector<size_t>& qidx, size_t cap_limit, std::vector<size_t>* postponed) -> int {
        const size_t m = qidx.size();
        std::vector<unsigned char> qbuf(m * q_row);
        for (size_t i = 0; i < m; ++i) memcpy(&qbuf[i * q_row], static_cast<const unsigned char*>(queries) + qidx[i] * q_row, q_row);
        svsx_batch_iterator* it = nullptr;
        if (int rc = svsx_iterator_create(ix, qbuf.data(), q_dtype, m, ~size_t(0), &it)) return rc;
        std::vector<size_t> found(m, 0), checked(m, 0), batch(m, initial_batch), grown(m, 0);
        std::vector<char> active(m, 1);
        std::vector<uint32_t> per_query(m), counts(m), meta(m * 8);
        std::vector<uint64_t> b_ids;
        std::vector<float> b_dists;
        int rc = SVSX_SUCCESS;
        size_t n_active = m;
        while (n_active > 0 && rc == SVSX_SUCCESS) {
            size_t round_max = 0;
            for (size_t i = 0; i < m; ++i) {
                if (!active[i]) { per_query[i] = 0xFFFFFFFFu; continue; }
                batch[i] = predict(checked[i], found[i], k, batch[i], max_batch);
                if (postponed != nullptr && grown[i] + batch[i] > cap_limit) { // would outgrow this group's memory: replayed on its own later
                    for (size_t j = 0; j < k; ++j) {
                        ids64[qidx[i] * k + j] = ~0ull;
                        dists[qidx[i] * k + j] = __builtin_inff();
                    }
                    postponed->push_back(qidx[i]);
                    active[i] = 0;
                    --n_active;
                    per_query[i] = 0xFFFFFFFFu;
                    continue;
                }
                grown[i] += batch[i];
                per_query[i] = static_cast<uint32_t>(std::min<size_t>(batch[i], 0xFFFFFFFEu));
                round_max = std::max(round_max, batch[i]);
            }
            if (n_active == 0) break;
            b_ids.resize(m * round_max);
            b_dists.resize(m * round_max);
            rc = svsx_iterator_next(it, round_max, per_query.data(), b_ids.data(), b_dists.data(), counts.data());
            if (rc != SVSX_SUCCESS) break;
            rc = svsx_iterator_state(it, meta.data());
            if (rc != SVSX_SUCCESS) break;
            for (size_t i = 0; i < m; ++i) {
                if (!active[i]) continue;
                const size_t o = qidx[i] * k;
                const size_t c = counts[i];
                checked[i] += c;
                for (size_t j = 0; j < c && found[i] < k; ++j) {
                    const uint64_t id = b_ids[i * round_max + j];
                    const uint64_t local = id - ix->id_offset;
                    if (local < ix->n && ((id_bitmap[local >> 6] >> (local & 63)) & 1ull)) {
                        ids64[o + found[i]] = id;
                        dists[o + found[i]] = b_dists[i * round_max + j];
                        ++found[i];
                    }
                }
                bool stop = false; // should_stop_filtered_search (svs_runtime_utils.h:424-434)
                if (filter_stop > 0.0f && checked[i] != 0)
                    stop = found[i] == 0 || static_cast<float>(found[i]) / checked[i] < filter_stop;
                const bool done = meta[i * 8 + 4] == ix->n || (meta[i * 8 + 6] & 2u) != 0u;
                if (stop) {
                    for (size_t j = 0; j < k; ++j) {
                        ids64[o + j] = ~0ull;
                        dists[o + j] = __builtin_inff();
                    }
                    found[i] = 0;
                }
                if (stop || found[i] >= k || done) {
                    active[i] = 0;
                    --n_active;
                }
            }
        }
        svsx_iterator_destroy(it);
        return rc;
    };
    // memory budget: ~128 M buffer entries per group (state table 8 B + yielded list 4 B + batch outputs 12 B each, host and device)
    constexpr size_t kBudgetEntries = size_t(128) << 20;
    std::vector<size_t> all(nq), heavy;
    for (size_t i = 0; i < nq; ++i) all[i] = i;
    const size_t cap_limit = std::min<size_t>(ix->n, std::max<size_t>(kBudgetEntries / nq, 1024));
    if (int rc = run_group(all, cap_limit, &heavy)) return rc;
    const size_t per_group = std::max<size_t>(1, kBudgetEntries / std::max<size_t>(ix->n, 1));
    for (size_t lo = 0; lo < heavy.size(); lo += per_group) {
        std::vector<size_t> grp(heavy.begin() + lo, heavy.begin() + std::min(heavy.size(), lo + per_group));
        if (int rc = run_group(grp, 0, nullptr)) return rc;
    }
    return SVSX_SUCCESS;
}

// Runtime VamanaIndex::range_search (bindings/cpp/src/vamana_index_impl.h:220-326) inside the library: every query owns a BatchIterator that is
// advanced in batches of max(window, 10); a query ends at its first out-of-range neighbour or when its iterator is done.  All queries of a group
// advance together (one launch per round); a query whose buffer would outgrow the group's budget (a radius that takes in a large part of the
// index) is replayed in a small group of its own -- its rounds depend on nothing but its own results.
extern "C" int svsx_range_search(
    const svsx_index* ix, const void* queries, int q_dtype, size_t nq, float radius, const svsx_search_params* params,
    const uint64_t* id_bitmap, svsx_results_allocator allocate, void* user
) {
    if (int rc = check_search_args(ix, queries, q_dtype, nq, 1)) return rc;
    if (!(radius > 0.0f)) return fail(SVSX_INVALID_ARGUMENT, "radius must be greater than 0"); // vamana_index_impl.h:227-230
    if (!allocate) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    if (nq == 0) return SVSX_SUCCESS;
    const size_t sws = params ? params->search_window_size : 0;
    const size_t batch = std::max<size_t>(sws, 10); // :271-274
    const size_t q_row = ix->d * dtype_size(q_dtype);
    const bool greater = ix->metric != M_L2;         // svs::distance::comparator: std::less for L2, std::greater for inner product / cosine
    std::vector<std::vector<std::pair<uint64_t, float>>> found(nq);
    auto run_group = [&](const std::vector<size_t>& qidx, size_t cap_limit, std::vector<size_t>* postponed) -> int {
        const size_t m = qidx.size();
        std::vector<unsigned char> qbuf(m * q_row);
        for (size_t i = 0; i < m; ++i) memcpy(&qbuf[i * q_row], static_cast<const unsigned char*>(queries) + qidx[i] * q_row, q_row);
        svsx_batch_iterator* it = nullptr;
        if (int rc = svsx_iterator_create(ix, qbuf.data(), q_dtype, m, ~size_t(0), &it)) return rc;
        std::vector<char> active(m, 1);
        std::vector<size_t> grown(m, 0);
        std::vector<uint32_t> per_query(m), counts(m), meta(m * 8);
        std::vector<uint64_t> b_ids(m * batch);
        std::vector<float> b_dists(m * batch);
        int rc = SVSX_SUCCESS;
        size_t n_active = m;
        while (n_active > 0 && rc == SVSX_SUCCESS) {
            for (size_t i = 0; i < m; ++i) {
                if (active[i] && postponed != nullptr && grown[i] + batch > cap_limit) {
                    found[qidx[i]].clear();
                    postponed->push_back(qidx[i]);
                    active[i] = 0;
                    --n_active;
                }
                if (active[i]) grown[i] += batch;
                per_query[i] = active[i] ? static_cast<uint32_t>(batch) : 0xFFFFFFFFu;
            }
            if (n_active == 0) break;
            rc = svsx_iterator_next(it, batch, per_query.data(), b_ids.data(), b_dists.data(), counts.data());
            if (rc != SVSX_SUCCESS) break;
            rc = svsx_iterator_state(it, meta.data());
            if (rc != SVSX_SUCCESS) break;
            for (size_t i = 0; i < m; ++i) {
                if (!active[i]) continue;
                bool in_range = true;
                for (size_t j = 0; j < counts[i]; ++j) {
                    const float dd = b_dists[i * batch + j];
                    in_range = greater ? dd > radius : dd < radius;
                    if (!in_range) break; // the iterator's results are ordered by distance (:291-293)
                    const uint64_t id = b_ids[i * batch + j];
                    const uint64_t local = id - ix->id_offset;
                    if (id_bitmap == nullptr || (local < ix->n && ((id_bitmap[local >> 6] >> (local & 63)) & 1ull))) found[qidx[i]].emplace_back(id, dd);
                }
                const bool done = meta[i * 8 + 4] == ix->n || (meta[i * 8 + 6] & 2u) != 0u;
                if (!in_range || done) {
                    active[i] = 0;
                    --n_active;
                }
            }
        }
        svsx_iterator_destroy(it);
        return rc;
    };
    constexpr size_t kBudgetEntries = size_t(128) << 20;
    std::vector<size_t> all(nq), heavy;
    for (size_t i = 0; i < nq; ++i) all[i] = i;
    const size_t cap_limit = std::min<size_t>(ix->n + batch, std::max<size_t>(kBudgetEntries / nq, 1024));
    if (int rc = run_group(all, cap_limit, &heavy)) return rc;
    const size_t per_group = std::max<size_t>(1, kBudgetEntries / std::max<size_t>(ix->n, 1));
    for (size_t lo = 0; lo < heavy.size(); lo += per_group) {
        std::vector<size_t> grp(heavy.begin() + lo, heavy.begin() + std::min(heavy.size(), lo + per_group));
        if (int rc = run_group(grp, 0, nullptr)) return rc;
    }
    std::vector<size_t> result_counts(nq);
    size_t total = 0;
    for (size_t i = 0; i < nq; ++i) total += (result_counts[i] = found[i].size());
    svsx_range_storage st{nullptr, nullptr};
    allocate(user, result_counts.data(), nq, &st);
    if (total != 0 && (!st.labels || !st.distances)) return fail(SVSX_INVALID_ARGUMENT, "the results allocator returned no storage for %zu results", total);
    for (size_t q = 0, ofs = 0; q < nq; ++q) {
        for (const auto& [id, dd] : found[q]) {
            st.labels[ofs] = id;
            st.distances[ofs] = dd;
            ++ofs;
        }
    }
    return SVSX_SUCCESS;
}
