// svsx_sharded.hip -- the sharded search of SURVEY 8e behind the C ABI: read-only database shards on the GPUs of one node, per-shard top-k,
// ONE all-gather over RCCL / xGMI, S-way merge.  New design (the reference has no sharded mode; the closest anchor is its partition of
// the QUERIES over threads, index/vamana/index.h:571-574); a C++ host needs nothing but libsvsx.so and an RCCL communicator.
//
// One object per process / GPU.  This rank's shards are global shards rank * n_local + s (contiguous id ranges, svsx_index_desc.id_offset).
// Per chunk of the query batch (one chunk, or two for batches of >= 8 192 queries when an exchange follows):
//   1. svsx_search_device on every local shard, writing (64-bit global ids | distances) straight into its slot of the packed buffer;
//      several local shards alternate between two internal streams, so the drain of one shard's launch overlaps the start of the next
//   2. n_local > 1: local merge (ties -> lower shard, then position) into the rank's send slot   svsx::merge_kernel
//   3. world > 1: ONE ncclAllGather of the chunk's slot per rank (1.2 MB at 10 000 x 10: latency-bound, no ring tuning)
//   4. world-way merge with the same tie rule = one global merge over all shards in shard order  svsx::merge_kernel
// Steps 3-4 of a chunk run on an internal stream while the caller's stream searches the next chunk; the caller's stream joins at the end,
// so the call is still "enqueued on hip_stream".  Slot layout [m u64 ids | m f32 distances], padded to 16 bytes: any nq * k.
// The staging buffers belong to the object: calls are serialised by a mutex and chained through an event, so calls on different
// streams (or a reallocation) never touch staging an earlier call still uses.
// RCCL is resolved at run time (dlopen "librccl.so.1": inside a PyTorch process that is the copy torch
// already loaded), so libsvsx.so itself has no link-time dependency on it and single-GPU users never load it.
#include "../../include/svsx.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace svsx {
int set_error(int code, const char* fmt, ...);                      // svsx_api.hip: thread-local text behind svsx_last_error
void index_facts(const svsx_index* ix, int* metric, int* device, size_t* d); // svsx_api.hip
} // namespace svsx

namespace {

// the five RCCL entry points used, by their NCCL names (rccl/rccl.h); ncclComm_t is an opaque pointer, ncclUniqueId 128 bytes
struct Id128 { char b[128]; };
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, Id128 /* ncclUniqueId, by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* why = "";
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.why = "librccl.so.1 not found"; return; }
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.handle, "ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.handle, "ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.handle, "ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.handle, "ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) {
            r.why = "librccl.so.1 lacks an expected symbol";
            r.handle = nullptr;
        }
    });
    return r;
}

constexpr int kNcclChar = 0; // ncclInt8 / ncclChar (rccl.h ncclDataType_t)

} // namespace

constexpr int kMaxChunks = 2;

struct svsx_sharded {
    std::vector<const svsx_index*> shards;
    int metric = 0, device = 0, rank = 0, world = 1;
    size_t d = 0;
    void* comm = nullptr;   // ncclComm_t
    bool owns_comm = false;
    // packed staging, grown on demand: per slot [m u64 | m f32] (16-byte padded), one region per chunk
    void* d_local = nullptr;  size_t local_cap = 0;   // n_local slots per chunk (only when n_local > 1)
    void* d_send = nullptr;   size_t send_cap = 0;    // 1 slot per chunk
    void* d_recv = nullptr;   size_t recv_cap = 0;    // world slots per chunk
    // serialisation of calls and the internal streams / events of the pipeline
    std::mutex mu;
    hipStream_t side[2] = {nullptr, nullptr};  // local shard searches alternate between them; side[0] also carries a chunk's exchange
    hipEvent_t ev_start = nullptr, ev_join[2] = {nullptr, nullptr}, ev_ready[kMaxChunks] = {nullptr, nullptr}, ev_done[kMaxChunks] = {nullptr, nullptr};
    hipEvent_t ev_last = nullptr;              // end of the previous call (staging reuse across streams)
    bool have_last = false;
    // optional phase timing of the last call (svsx_sharded_set_timing / _phase_times): per chunk, events at search start, searches done,
    // local merge done, all-gather done, world merge done -- recorded only while enabled
    bool timing = false, timed_exchange = false;
    int timed_chunks = 0;
    hipEvent_t tev[kMaxChunks][5] = {};
};

namespace {

int grow(svsx_sharded* s, void*& p, size_t& cap, size_t want) {
    if (want <= cap) return SVSX_SUCCESS;
    if (p) {
        if (s->have_last) (void)hipEventSynchronize(s->ev_last); // an earlier call may still be reading the old buffer
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    if (hipMalloc(&p, want) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for the shard lists failed", want);
    cap = want;
    return SVSX_SUCCESS;
}

int make_pipeline(svsx_sharded* s) {
    if (s->ev_last) return SVSX_SUCCESS;
    bool ok = true;
    for (hipStream_t& st : s->side) ok = ok && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    auto ev = [&](hipEvent_t& e) { ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess; };
    ev(s->ev_start);
    for (hipEvent_t& e : s->ev_join) ev(e);
    for (hipEvent_t& e : s->ev_ready) ev(e);
    for (hipEvent_t& e : s->ev_done) ev(e);
    ev(s->ev_last);
    if (!ok) return svsx::set_error(SVSX_RUNTIME_ERROR, "could not create the streams / events of the sharded search");
    return SVSX_SUCCESS;
}

int create_common(const svsx_index* const* local_shards, size_t n_local, int rank, int world, svsx_sharded** out) {
    if (!out) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!local_shards || n_local == 0 || n_local > 64) return svsx::set_error(SVSX_INVALID_ARGUMENT, "a rank owns 1..64 shards");
    if (world < 1 || rank < 0 || rank >= world) return svsx::set_error(SVSX_INVALID_ARGUMENT, "rank %d outside a world of %d", rank, world);
    if (static_cast<size_t>(world) > 64) return svsx::set_error(SVSX_INVALID_ARGUMENT, "at most 64 ranks");
    svsx_sharded* s = new (std::nothrow) svsx_sharded();
    if (!s) return svsx::set_error(SVSX_RUNTIME_ERROR, "out of host memory");
    for (size_t i = 0; i < n_local; ++i) {
        if (!local_shards[i]) { delete s; return svsx::set_error(SVSX_NOT_INITIALIZED, "shard %zu: Index not initialized", i); }
        int metric = 0, device = 0;
        size_t d = 0;
        svsx::index_facts(local_shards[i], &metric, &device, &d);
        if (i == 0) { s->metric = metric; s->device = device; s->d = d; }
        else if (metric != s->metric || device != s->device || d != s->d) { delete s; return svsx::set_error(SVSX_INVALID_ARGUMENT, "the shards of a rank share one metric, one dimension and one device"); }
        s->shards.push_back(local_shards[i]);
    }
    s->rank = rank;
    s->world = world;
    *out = s;
    return SVSX_SUCCESS;
}

} // namespace

extern "C" {

int svsx_sharded_unique_id(void* id128) {
    if (!id128) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    Rccl& r = rccl();
    if (!r.handle) return svsx::set_error(SVSX_RUNTIME_ERROR, "RCCL is not available: %s", r.why);
    const int rc = r.GetUniqueId(id128);
    if (rc != 0) return svsx::set_error(SVSX_RUNTIME_ERROR, "ncclGetUniqueId failed: %s", r.GetErrorString(rc));
    return SVSX_SUCCESS;
}

int svsx_sharded_create(const svsx_index* const* local_shards, size_t n_local, const void* id128, int rank, int world, svsx_sharded** out) {
    if (int rc = create_common(local_shards, n_local, rank, world, out)) return rc;
    svsx_sharded* s = *out;
    if (id128 == nullptr) {
        if (world != 1) { svsx_sharded_destroy(s); *out = nullptr; return svsx::set_error(SVSX_INVALID_ARGUMENT, "a world of %d ranks needs the unique id of its communicator", world); }
        return SVSX_SUCCESS; // one process, no exchange step
    }
    Rccl& r = rccl();
    if (!r.handle) { svsx_sharded_destroy(s); *out = nullptr; return svsx::set_error(SVSX_RUNTIME_ERROR, "RCCL is not available: %s", r.why); }
    if (hipSetDevice(s->device) != hipSuccess) { svsx_sharded_destroy(s); *out = nullptr; return svsx::set_error(SVSX_RUNTIME_ERROR, "hipSetDevice(%d) failed", s->device); }
    Id128 id;
    memcpy(id.b, id128, 128);
    const int rc = r.CommInitRank(&s->comm, world, id, rank); // collective: returns once every rank has joined
    if (rc != 0) { svsx_sharded_destroy(s); *out = nullptr; return svsx::set_error(SVSX_RUNTIME_ERROR, "ncclCommInitRank failed: %s", r.GetErrorString(rc)); }
    s->owns_comm = true;
    return SVSX_SUCCESS;
}

int svsx_sharded_create_with_comm(const svsx_index* const* local_shards, size_t n_local, void* nccl_comm, int rank, int world, svsx_sharded** out) {
    if (int rc = create_common(local_shards, n_local, rank, world, out)) return rc;
    if (world > 1) {
        if (!nccl_comm) { svsx_sharded_destroy(*out); *out = nullptr; return svsx::set_error(SVSX_INVALID_ARGUMENT, "null communicator"); }
        Rccl& r = rccl();
        if (!r.handle) { svsx_sharded_destroy(*out); *out = nullptr; return svsx::set_error(SVSX_RUNTIME_ERROR, "RCCL is not available: %s", r.why); }
    }
    (*out)->comm = nccl_comm;
    return SVSX_SUCCESS;
}

void svsx_sharded_destroy(svsx_sharded* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->comm && s->owns_comm) (void)rccl().CommDestroy(s->comm);
    if (s->d_local) (void)hipFree(s->d_local);
    if (s->d_send) (void)hipFree(s->d_send);
    if (s->d_recv) (void)hipFree(s->d_recv);
    for (hipStream_t st : s->side) if (st) (void)hipStreamDestroy(st);
    for (hipEvent_t e : {s->ev_start, s->ev_join[0], s->ev_join[1], s->ev_ready[0], s->ev_ready[1], s->ev_done[0], s->ev_done[1], s->ev_last}) if (e) (void)hipEventDestroy(e);
    for (auto& row : s->tev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    delete s;
}

int svsx_sharded_info(const svsx_sharded* s, uint64_t info[4]) {
    if (!s || !info) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    info[0] = s->shards.size();
    info[1] = static_cast<uint64_t>(s->world);
    info[2] = static_cast<uint64_t>(s->rank);
    info[3] = s->comm ? 1u : 0u;
    return SVSX_SUCCESS;
}

// ---- the pipeline as DATA: svsx_sharded_plan lists, in enqueue order, every operation a call puts on its three streams (0 = the caller's,
// 1 / 2 = the internal ones) -- searches, merges, the all-gather, event records and waits -- and svsx_sharded_search_device executes exactly that
// list.  A pure host function: tests/test_sharded_plan.py replays the list through a happens-before model (every read after its write, the
// second chunk's searches NOT ordered behind the first chunk's exchange, the caller's stream joined at the end, calls chained through
// ev_last) without a GPU -- the part of the N > 1 path that a one-GPU box cannot time is at least checked for order.
int svsx_sharded_plan(size_t n_local, int exchange, size_t nq, int have_last, svsx_sharded_op* ops, size_t cap, size_t* n_ops) {
    if (!n_ops || n_local == 0 || nq == 0) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null / empty argument");
    size_t n = 0;
    auto put = [&](int kind, int stream, int chunk, int arg) {
        if (ops && n < cap) ops[n] = svsx_sharded_op{kind, stream, chunk, arg};
        ++n;
    };
    const bool fan_out = n_local > 1;
    const int nchunk = (exchange && nq >= 8192) ? 2 : 1; // the exchange + world merge of chunk c runs under the searches of chunk c + 1
    // the staging may still be in use by the previous call (made on another stream): chain behind it
    if (have_last) put(SVSX_OP_WAIT, 0, -1, SVSX_EV_LAST);
    if (fan_out || nchunk > 1) {
        put(SVSX_OP_RECORD, 0, -1, SVSX_EV_START);
        put(SVSX_OP_WAIT, 1, -1, SVSX_EV_START);
        if (fan_out) put(SVSX_OP_WAIT, 2, -1, SVSX_EV_START); // (one local shard: the second internal stream carries nothing)
    }
    for (int c = 0; c < nchunk; ++c) {
        if (!fan_out) {
            put(SVSX_OP_STAMP, 0, c, 0);
            put(SVSX_OP_SEARCH, 0, c, 0);
            put(SVSX_OP_STAMP, 0, c, 1);
            put(SVSX_OP_STAMP, 0, c, 2); // no local merge: an empty phase
        } else {
            put(SVSX_OP_STAMP, 1, c, 0);
            for (size_t i = 0; i < n_local; ++i) put(SVSX_OP_SEARCH, 1 + static_cast<int>(i & 1), c, static_cast<int>(i));
            for (int j = 0; j < 2; ++j) {
                put(SVSX_OP_RECORD, 1 + j, c, SVSX_EV_JOIN0 + j);
                put(SVSX_OP_WAIT, 0, c, SVSX_EV_JOIN0 + j);
            }
            put(SVSX_OP_STAMP, 0, c, 1);
            put(SVSX_OP_LOCAL_MERGE, 0, c, 0);
            put(SVSX_OP_STAMP, 0, c, 2);
        }
        if (!exchange) continue;
        // exchange + world-way merge.  One local shard, two chunks: on an internal stream, so that the caller's stream goes straight on to the
        // search of the next chunk.  With local fan-out the searches already run on the internal streams (the next chunk's are not held up
        // by anything queued here), so the exchange stays on the caller's stream.
        int xs = 0;
        if (nchunk > 1 && !fan_out) {
            xs = 1;
            put(SVSX_OP_RECORD, 0, c, SVSX_EV_READY0 + c);
            put(SVSX_OP_WAIT, xs, c, SVSX_EV_READY0 + c);
        }
        put(SVSX_OP_ALLGATHER, xs, c, 0);
        put(SVSX_OP_STAMP, xs, c, 3);
        put(SVSX_OP_WORLD_MERGE, xs, c, 0);
        put(SVSX_OP_STAMP, xs, c, 4);
        if (xs != 0) put(SVSX_OP_RECORD, xs, c, SVSX_EV_DONE0 + c);
    }
    if (exchange && nchunk > 1 && !fan_out) // the caller's stream joins the exchanges only now: every search is already queued in front
        for (int c = 0; c < nchunk; ++c) put(SVSX_OP_WAIT, 0, c, SVSX_EV_DONE0 + c);
    put(SVSX_OP_RECORD, 0, -1, SVSX_EV_LAST);
    *n_ops = n;
    if (ops && n > cap) return svsx::set_error(SVSX_INVALID_ARGUMENT, "the plan has %zu operations, room for %zu", n, cap);
    return SVSX_SUCCESS;
}

int svsx_sharded_set_timing(svsx_sharded* s, int enabled) {
    if (!s) return svsx::set_error(SVSX_NOT_INITIALIZED, "Index not initialized");
    std::lock_guard<std::mutex> lock(s->mu);
    s->timing = enabled != 0;
    s->timed_chunks = 0;
    return SVSX_SUCCESS;
}

int svsx_sharded_phase_times(svsx_sharded* s, float ms[5]) {
    if (!s || !ms) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->timing || s->timed_chunks == 0 || !s->have_last) return svsx::set_error(SVSX_INVALID_ARGUMENT, "no timed call to report: svsx_sharded_set_timing(s, 1), then search");
    if (hipSetDevice(s->device) != hipSuccess || hipEventSynchronize(s->ev_last) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "the timed call did not complete");
    for (int i = 0; i < 5; ++i) ms[i] = 0.0f;
    for (int c = 0; c < s->timed_chunks; ++c) {
        for (int ph = 0; ph < 4; ++ph) {
            if (ph >= 2 && !s->timed_exchange) break;
            float t = 0.0f;
            if (hipEventElapsedTime(&t, s->tev[c][ph], s->tev[c][ph + 1]) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipEventElapsedTime failed");
            ms[ph] += t;
        }
    }
    float span = 0.0f;
    if (hipEventElapsedTime(&span, s->tev[0][0], s->tev[s->timed_chunks - 1][s->timed_exchange ? 4 : 2]) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipEventElapsedTime failed");
    ms[4] = span;
    return SVSX_SUCCESS;
}

int svsx_sharded_search_device(
    svsx_sharded* s, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params, uint64_t* d_ids64,
    float* d_dists, void* hip_stream
) {
    if (!s) return svsx::set_error(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (k == 0) return svsx::set_error(SVSX_INVALID_ARGUMENT, "k must be greater than 0");
    if (nq == 0) return SVSX_SUCCESS;
    if (!d_queries || !d_ids64 || !d_dists) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    static const size_t esize[4] = {4, 2, 1, 1};
    if (q_dtype < 0 || q_dtype > 3) return svsx::set_error(SVSX_INVALID_ARGUMENT, "unknown query dtype %d", q_dtype);
    if (hipSetDevice(s->device) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipSetDevice(%d) failed", s->device);
    std::lock_guard<std::mutex> lock(s->mu);
    if (int rc = make_pipeline(s)) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const size_t n_local = s->shards.size();
    const bool exchange = s->world > 1 || s->comm != nullptr;
    const size_t q_row = s->d * esize[q_dtype];
    const int nchunk = (exchange && nq >= 8192) ? 2 : 1; // (the same rule as svsx_sharded_plan)
    size_t lo[kMaxChunks + 1], slot[kMaxChunks], off[kMaxChunks + 1];
    off[0] = 0;
    for (int c = 0; c < nchunk; ++c) {
        lo[c] = nq * static_cast<size_t>(c) / nchunk;
        lo[c + 1] = nq * static_cast<size_t>(c + 1) / nchunk;
        slot[c] = ((lo[c + 1] - lo[c]) * k * 12 + 15) & ~size_t(15); // [m u64 | m f32]
        off[c + 1] = off[c] + slot[c];
    }
    const size_t all = off[nchunk];
    if (exchange) {
        if (int rc = grow(s, s->d_send, s->send_cap, all)) return rc;
        if (int rc = grow(s, s->d_recv, s->recv_cap, all * static_cast<size_t>(s->world))) return rc;
    }
    if (n_local > 1)
        if (int rc = grow(s, s->d_local, s->local_cap, all * n_local)) return rc;
    std::vector<svsx_sharded_op> plan(64 + 4 * n_local);
    size_t n_ops = 0;
    if (int rc = svsx_sharded_plan(n_local, exchange ? 1 : 0, nq, s->have_last ? 1 : 0, plan.data(), plan.size(), &n_ops)) return rc;
#define SH_TRY(expr)                                                                                                         \
    do {                                                                                                                     \
        hipError_t e_ = (expr);                                                                                              \
        if (e_ != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
    // An error return below leaves work queued on the caller's stream and on the internal ones without ev_last recorded behind it: the next
    // call would neither chain behind that work nor know that the staging is still being written.  Every exit that is not the
    // successful one therefore drains the three streams before the staging is handed on.
    struct DrainOnError {
        svsx_sharded* s;
        hipStream_t stream;
        bool ok = false;
        ~DrainOnError() {
            if (ok) return;
            for (hipStream_t st : s->side) (void)hipStreamSynchronize(st);
            (void)hipStreamSynchronize(stream);
            s->have_last = false;
        }
    } drain{s, stream};
    if (s->timing) {
        for (int c = 0; c < nchunk; ++c)
            for (hipEvent_t& e : s->tev[c])
                if (!e) SH_TRY(hipEventCreate(&e));
    }
    hipStream_t streams[3] = {stream, s->side[0], s->side[1]};
    hipEvent_t events[8] = {s->ev_start, s->ev_join[0], s->ev_join[1], s->ev_ready[0], s->ev_ready[1], s->ev_done[0], s->ev_done[1], s->ev_last};
    Rccl& r = rccl();
    for (size_t oi = 0; oi < n_ops; ++oi) {
        const svsx_sharded_op& op = plan[oi];
        hipStream_t st = streams[op.stream];
        const int c = op.chunk < 0 ? 0 : op.chunk;
        const size_t rows = lo[c + 1] - lo[c], m = rows * k;
        const unsigned char* q = static_cast<const unsigned char*>(d_queries) + lo[c] * q_row;
        auto ids_of = [&](void* base, size_t region_slots, size_t i) {  // slot i of chunk c in a staging buffer holding `region_slots` slots per chunk
            return reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(base) + off[c] * region_slots + slot[c] * i);
        };
        auto dist_of = [&](void* base, size_t region_slots, size_t i) { return reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(ids_of(base, region_slots, i)) + m * 8); };
        // where this rank's merged list of the chunk goes: the send slot when an exchange follows, else straight to the caller's rows
        uint64_t* own_i = exchange ? ids_of(s->d_send, 1, 0) : d_ids64 + lo[c] * k;
        float* own_d = exchange ? dist_of(s->d_send, 1, 0) : d_dists + lo[c] * k;
        switch (op.kind) {
            case SVSX_OP_WAIT: SH_TRY(hipStreamWaitEvent(st, events[op.arg], 0)); break;
            case SVSX_OP_RECORD: SH_TRY(hipEventRecord(events[op.arg], st)); break;
            case SVSX_OP_STAMP:
                if (s->timing) SH_TRY(hipEventRecord(s->tev[c][op.arg], st));
                break;
            case SVSX_OP_SEARCH: {
                const size_t i = static_cast<size_t>(op.arg);
                uint64_t* oi_ = n_local > 1 ? ids_of(s->d_local, n_local, i) : own_i;
                float* od_ = n_local > 1 ? dist_of(s->d_local, n_local, i) : own_d;
                if (int rc = svsx_search_device(s->shards[i], q, q_dtype, rows, k, params, oi_, nullptr, od_, nullptr, st)) return rc;
                break;
            }
            case SVSX_OP_LOCAL_MERGE:
                if (int rc = svsx_merge_topk_strided_device(s->metric, n_local, rows, k, dist_of(s->d_local, n_local, 0), slot[c] / 4, ids_of(s->d_local, n_local, 0), slot[c] / 8, own_d, own_i, st)) return rc;
                break;
            case SVSX_OP_ALLGATHER: {
                const int rc = r.AllGather(ids_of(s->d_send, 1, 0), ids_of(s->d_recv, static_cast<size_t>(s->world), 0), slot[c], kNcclChar, s->comm, st);
                if (rc != 0) return svsx::set_error(SVSX_RUNTIME_ERROR, "ncclAllGather failed: %s", r.GetErrorString(rc));
                break;
            }
            case SVSX_OP_WORLD_MERGE:
                if (int rc2 = svsx_merge_topk_strided_device(s->metric, static_cast<size_t>(s->world), rows, k, dist_of(s->d_recv, static_cast<size_t>(s->world), 0), slot[c] / 4,
                                                             ids_of(s->d_recv, static_cast<size_t>(s->world), 0), slot[c] / 8, d_dists + lo[c] * k, d_ids64 + lo[c] * k, st))
                    return rc2;
                break;
            default: return svsx::set_error(SVSX_RUNTIME_ERROR, "unknown operation %d in the sharded plan", op.kind);
        }
    }
    s->have_last = true;
    s->timed_chunks = s->timing ? nchunk : 0;
    s->timed_exchange = exchange;
    drain.ok = true;
#undef SH_TRY
    return SVSX_SUCCESS;
}

} // extern "C"
