// svsx_sharded.hip -- the sharded search of SURVEY 8e behind the C ABI: read-only database shards on the GPUs of one node, per-shard top-k,
// ONE all-gather over RCCL / xGMI, S-way merge.  New design (the reference has no sharded mode; the closest anchor is its partition of
// the QUERIES over threads, index/vamana/index.h:571-574); a C++ host needs nothing but libsvsx.so and an RCCL communicator.
//
// One object per process / GPU.  This rank's shards are global shards rank * n_local + s (contiguous id ranges, svsx_index_desc.id_offset).
//   1. svsx_search_device on every local shard, writing (distances | 64-bit global ids) straight into its slot of the packed send buffer
//   2. n_local > 1: local merge (ties -> lower shard, then position) into the rank's slot        svsx::merge_kernel
//   3. world > 1: ONE ncclAllGather of nq * k * 12 bytes per rank (1.2 MB at 10 000 x 10: latency-bound, no ring tuning)
//   4. world-way merge with the same tie rule = one global merge over all shards in shard order  svsx::merge_kernel
// all on the caller's stream.  RCCL is resolved at run time (dlopen "librccl.so.1": inside a PyTorch process that is the copy torch
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
void index_facts(const svsx_index* ix, int* metric, int* device);   // svsx_api.hip
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

struct svsx_sharded {
    std::vector<const svsx_index*> shards;
    int metric = 0, device = 0, rank = 0, world = 1;
    void* comm = nullptr;   // ncclComm_t
    bool owns_comm = false;
    // packed staging, grown on demand: per slot [nq * k f32 | nq * k u64]
    void* d_local = nullptr;  size_t local_cap = 0;   // n_local slots (only when n_local > 1)
    void* d_send = nullptr;   size_t send_cap = 0;    // 1 slot
    void* d_recv = nullptr;   size_t recv_cap = 0;    // world slots
};

namespace {

int grow(void*& p, size_t& cap, size_t want) {
    if (want <= cap) return SVSX_SUCCESS;
    if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
    if (hipMalloc(&p, want) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipMalloc(%zu) for the shard lists failed", want);
    cap = want;
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
        svsx::index_facts(local_shards[i], &metric, &device);
        if (i == 0) { s->metric = metric; s->device = device; }
        else if (metric != s->metric || device != s->device) { delete s; return svsx::set_error(SVSX_INVALID_ARGUMENT, "the shards of a rank share one metric and one device"); }
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

int svsx_sharded_search_device(
    svsx_sharded* s, const void* d_queries, int q_dtype, size_t nq, size_t k, const svsx_search_params* params, uint64_t* d_ids64,
    float* d_dists, void* hip_stream
) {
    if (!s) return svsx::set_error(SVSX_NOT_INITIALIZED, "Index not initialized");
    if (k == 0) return svsx::set_error(SVSX_INVALID_ARGUMENT, "k must be greater than 0");
    if (nq == 0) return SVSX_SUCCESS;
    if (!d_queries || !d_ids64 || !d_dists) return svsx::set_error(SVSX_INVALID_ARGUMENT, "null argument");
    if ((nq * k) % 2 != 0) return svsx::set_error(SVSX_INVALID_ARGUMENT, "nq * k must be even (8-byte alignment of the packed ids)");
    if (hipSetDevice(s->device) != hipSuccess) return svsx::set_error(SVSX_RUNTIME_ERROR, "hipSetDevice(%d) failed", s->device);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    const size_t m = nq * k, slot = m * 12; // [m f32 | m u64]
    const size_t n_local = s->shards.size();
    const bool exchange = s->world > 1 || s->comm != nullptr;
    auto dist_of = [&](void* base, size_t i) { return reinterpret_cast<float*>(static_cast<unsigned char*>(base) + slot * i); };
    auto ids_of = [&](void* base, size_t i) { return reinterpret_cast<uint64_t*>(static_cast<unsigned char*>(base) + slot * i + m * 4); };
    // where this rank's merged list goes: the send slot when an exchange follows, else straight to the caller
    if (exchange) {
        if (int rc = grow(s->d_send, s->send_cap, slot)) return rc;
        if (int rc = grow(s->d_recv, s->recv_cap, slot * static_cast<size_t>(s->world))) return rc;
    }
    float* own_d = exchange ? dist_of(s->d_send, 0) : d_dists;
    uint64_t* own_i = exchange ? ids_of(s->d_send, 0) : d_ids64;
    if (n_local == 1) {
        if (int rc = svsx_search_device(s->shards[0], d_queries, q_dtype, nq, k, params, own_i, nullptr, own_d, nullptr, stream)) return rc;
    } else {
        if (int rc = grow(s->d_local, s->local_cap, slot * n_local)) return rc;
        for (size_t i = 0; i < n_local; ++i)
            if (int rc = svsx_search_device(s->shards[i], d_queries, q_dtype, nq, k, params, ids_of(s->d_local, i), nullptr, dist_of(s->d_local, i), nullptr, stream)) return rc;
        if (int rc = svsx_merge_topk_strided_device(s->metric, n_local, nq, k, dist_of(s->d_local, 0), slot / 4, ids_of(s->d_local, 0), slot / 8, own_d, own_i, stream)) return rc;
    }
    if (!exchange) return SVSX_SUCCESS;
    Rccl& r = rccl();
    const int rc = r.AllGather(s->d_send, s->d_recv, slot, kNcclChar, s->comm, stream);
    if (rc != 0) return svsx::set_error(SVSX_RUNTIME_ERROR, "ncclAllGather failed: %s", r.GetErrorString(rc));
    return svsx_merge_topk_strided_device(s->metric, static_cast<size_t>(s->world), nq, k, dist_of(s->d_recv, 0), slot / 4, ids_of(s->d_recv, 0), slot / 8, d_dists, d_ids64, stream);
}

} // extern "C"
