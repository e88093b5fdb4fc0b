// svsx_api.hip -- host side of the C ABI declared in include/svsx.h (libsvsx.so): index life cycle, the search entry points (device- and
// host-pointer), distance plugin, iterator, filtered and range search.  Bound records: svsx_records.hip; flat search: svsx_flat_host.hip;
// shared host-side types: svsx_host.h.
//
//
// Owns the HBM-resident index (re-laid-out copies of the SVS data/graph arrays), validates arguments the way the
// reference's runtime layer does (bindings/cpp/src/vamana_index.cpp:56-72, vamana_index_impl.h:109-138) and
// launches the gfx950 kernels.  There is deliberately NO CPU fallback: without a HIP device every compute entry
// point fails with SVSX_RUNTIME_ERROR.
#include "svsx_host.h"

namespace {
thread_local std::string g_error;
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

int svsx_index_memory(const svsx_index* ix, uint64_t mem[10]) {
    if (!ix) return fail(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (!mem) return fail(SVSX_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(ix->shadow_mutex);
    const size_t dpad_h = ix->d > 128 && ix->d <= 768 ? (ix->d + 127) & ~size_t(127) : (ix->d + 15) & ~size_t(15);
    mem[0] = ix->row_bytes * ix->n;
    mem[1] = ix->d_graph ? ix->gstride * ix->n * sizeof(uint32_t) : 0;
    mem[2] = ix->d_fat ? ix->n * ix->fat_stride + 128 : 0;
    mem[3] = ix->d_fat ? ix->fat_rb : 0;
    mem[4] = ix->d_codes ? ix->n * 128 + 128 : 0;
    mem[5] = ix->d_half ? (ix->n + kShadowPadRows) * dpad_h * 2 + 64 + ix->n * 4 : 0;
    mem[6] = ix->d_sq_row_norm ? ix->n * 4 : 0;
    mem[7] = (ix->owns ? mem[0] + mem[1] : 0) + mem[2] + mem[4] + mem[5] + mem[6]; // what the library itself allocated (a device view borrows rows and graph)
    mem[8] = ix->d_fat ? 2 : ix->d_codes ? 1 : 0;
    mem[9] = ix->owns ? 1 : 0;
    return SVSX_SUCCESS;
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
        a.pca_mean_norm = ix->pca_mean_norm;
        a.pca_mean_dir = ix->pca_mean_dir;
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
#ifndef SVSX_HOST_FIRST
#define SVSX_HOST_FIRST 4096
#endif
    size_t first = nq > SVSX_HOST_FIRST ? SVSX_HOST_FIRST : nq;
#ifdef SVSX_HOST_THIRD  // diagnostic builds: a small third chunk, so that the last result copy (the exposed one) is short
    if (first + SVSX_HOST_THIRD + 1024 < nq) {
        nc = 3;
        bounds[1] = first;
        bounds[2] = nq - SVSX_HOST_THIRD;
        bounds[3] = nq;
    } else
#endif
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
#ifdef SVSX_DIAG_HOST
    const auto t_begin = std::chrono::steady_clock::now();
    double stamp_us[16];
    int n_stamp = 0;
#define SVSX_HOST_STAMP() stamp_us[n_stamp++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count()
#else
#define SVSX_HOST_STAMP()
#endif
    for (int i = 0; i < nc; ++i) {
        const size_t lo = bounds[i], cnt = bounds[i + 1] - lo;
        if (cnt == 0) continue;
        hipStream_t st = c->streams[i];
        void* dq = static_cast<unsigned char*>(c->d_in) + q_stride * lo;
        // never past the caller's last row
        HIP_TRY(hipMemcpyAsync(dq, static_cast<const unsigned char*>(queries) + q_stride * lo, q_stride * (cnt - 1) + q_row, hipMemcpyHostToDevice, st));
        SVSX_HOST_STAMP();
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
        SVSX_HOST_STAMP();
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
        SVSX_HOST_STAMP();
        HIP_TRY(hipStreamSynchronize(c->streams[i]));
        SVSX_HOST_STAMP();
    }
#ifdef SVSX_DIAG_HOST
    {   // diagnostic builds only: host-side time line of the call, us since its start: per chunk [copy in, launch], then per chunk [results requested, stream idle]
        static std::atomic<int> calls{0};
        if (calls.fetch_add(1) % 4 == 3) {
            fprintf(stderr, "[svsx host] nc=%d", nc);
            for (int j = 0; j < n_stamp; ++j) fprintf(stderr, " %.0f", stamp_us[j]);
            fprintf(stderr, "\n");
        }
    }
#endif
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
