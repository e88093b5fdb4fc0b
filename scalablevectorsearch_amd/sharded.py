"""Read-only database sharding across the GPUs of one node (new design, the reference has no such mode).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank owns an independent Vamana
sub-index over a contiguous id range (own graph, own entry point); a query batch is searched on every shard and the
per-shard top-k -- `nq*k*(4+8)` bytes per shard, 1.2 MB at nq=10^4, k=10, distances and ids packed into one buffer --
are exchanged with ONE all-gather and merged per query with the metric's comparator (ties: lower shard, then position), so every rank ends with the same
global top-k.  No other data-path collective exists: graph and vectors never move.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from .distance import as_functor


def shard_ranges(n_total: int, n_shards: int):
    """Contiguous, balanced id ranges: shard s owns [lo, hi)."""
    base, extra = divmod(n_total, n_shards)
    out, lo = [], 0
    for s in range(n_shards):
        hi = lo + base + (1 if s < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def hip_merge(metric: int, gathered_d, gathered_i, k: int):
    """S-way merge on the GPU (svsx_merge_topk_device).  Tensors must live in HBM -- there is no CPU merge here."""
    import torch

    if not gathered_d.is_cuda:
        raise _lib.SvsxError(_lib.RUNTIME_ERROR, "shard merge runs on the GPU only (no CPU fallback)")
    S, nq, kk = gathered_d.shape
    for t in (gathered_d, gathered_i):  # rows of a shard dense, shards at any stride (views into a packed gather buffer)
        if t.stride(2) != 1 or t.stride(1) != kk:
            raise _lib.SvsxError(_lib.INVALID_ARGUMENT, "shard lists must be dense [nq, k] blocks")
    out_d = torch.empty((nq, k), dtype=torch.float32, device=gathered_d.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=gathered_d.device)
    _lib.check(_lib.lib().svsx_merge_topk_strided_device(
        int(metric), S, nq, k, gathered_d.data_ptr(), gathered_d.stride(0) if S > 1 else 0, gathered_i.data_ptr(),
        gathered_i.stride(0) if S > 1 else 0, out_d.data_ptr(), out_i.data_ptr(),
        C.c_void_p(torch.cuda.current_stream(gathered_d.device).cuda_stream)))
    return out_d, out_i


def slot_words(nq: int, k: int) -> int:
    """int32 words of one packed shard list [nq*k i64 ids | nq*k f32 distances], padded to 8 bytes (any nq * k; the layout svsx_sharded.hip uses)."""
    return (3 * nq * k + 1) & ~1


def packed_views(buf, world: int, nq: int, k: int):
    """Views into a packed (ids | distances) buffer of `world` shards: int32 [world * slot_words] -> (f32 [world,nq,k], i64 [world,nq,k])."""
    import torch

    m = nq * k
    b = buf.view(world, slot_words(nq, k))
    i = b[:, : 2 * m].view(torch.int64).view(world, nq, k)
    d = b[:, 2 * m : 3 * m].view(torch.float32).view(world, nq, k)
    return d, i


class ShardedSearcher:
    """Orchestrates search over `world` shards.

    local_search(queries, k) -> (dists [nq,k] f32, ids [nq,k] int64 GLOBAL ids, best-first, padded with -1/+inf)
    merge(metric, gathered_d [S,nq,k], gathered_i [S,nq,k], k) -> (dists, ids)           default: the HIP kernel
    """

    def __init__(self, local_search, distance, group=None, merge=hip_merge):
        import torch.distributed as dist

        self.local_search = local_search
        self.metric = int(as_functor(distance).metric)
        self.group = group
        self.merge = merge
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def search(self, queries, k: int):
        import torch
        import torch.distributed as dist

        d_local, i_local = self.local_search(queries, k)
        if self.world == 1:
            return self.merge(self.metric, d_local[None], i_local[None], k)
        nq = d_local.shape[0]
        # one packed buffer per shard -> ONE collective: [nq*k i64 ids | nq*k f32 distances] (+ padding to 8 bytes) as int32 words
        w = slot_words(nq, k)
        loc = torch.zeros(w, dtype=torch.int32, device=d_local.device)
        loc_d, loc_i = packed_views(loc, 1, nq, k)
        loc_d[0].copy_(d_local)
        loc_i[0].copy_(i_local)
        allb = torch.empty(self.world * w, dtype=torch.int32, device=d_local.device)
        dist.all_gather_into_tensor(allb, loc, group=self.group)
        all_d, all_i = packed_views(allb, self.world, nq, k)
        return self.merge(self.metric, all_d, all_i, k)


class MultiShardSearcher(ShardedSearcher):
    """Fixed-shard-count form ("each GPU holds 8/G shards"): this rank owns SEVERAL shards -- contiguous shard
    ids, shard `rank * len(local_searches) + s` -- searches them one after the other, merges their lists locally (ties ->
    lower shard, then position: the same rule, so merging local shards first and ranks second equals one global merge over
    all shards in shard order), and ships ONE packed list through the same single all-gather as ShardedSearcher.

    local_searches: one callable per local shard, each `(queries, k) -> (dists [nq,k], GLOBAL ids [nq,k])`."""

    def __init__(self, local_searches, distance, group=None, merge=hip_merge):
        import torch

        self.local_searches = list(local_searches)

        def local(queries, k):
            parts = [f(queries, k) for f in self.local_searches]
            if len(parts) == 1:
                return parts[0]
            return self.merge(self.metric, torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), k)

        super().__init__(local, distance, group=group, merge=merge)

    @property
    def shards_total(self) -> int:
        return self.world * len(self.local_searches)


class PipelinedShardSearch:
    """Throughput form of the sharded search for a stream of equally shaped query batches: the gather + merge leg of batch i
    (latency-bound: one small collective, one small kernel) runs on a second stream while the search kernel of batch i+1
    already occupies the GPU.  `depth` packed buffers rotate; `submit()` enqueues one batch and returns the index of the
    buffer set its merged result will land in, `drain()` waits for everything.  Same collective sequence on every rank.

    index: VamanaIndex (this rank's shard, ids offset to global); results: `out_d[slot]`, `out_i[slot]` ([nq, k] each)."""

    def __init__(self, index, metric: int, nq: int, k: int, group=None, depth: int = 2, device=None):
        import torch
        import torch.distributed as dist

        self.index, self.metric, self.nq, self.k, self.group = index, int(metric), nq, k, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = dist.is_initialized()
        w = slot_words(nq, k)
        self.depth = depth
        self.loc = [torch.zeros(w, dtype=torch.int32, device=device) for _ in range(depth)]
        self.allb = [torch.empty(self.world * w, dtype=torch.int32, device=device) for _ in range(depth)]
        self.out_d = [torch.empty((nq, k), dtype=torch.float32, device=device) for _ in range(depth)]
        self.out_i = [torch.empty((nq, k), dtype=torch.int64, device=device) for _ in range(depth)]
        self.s_merge = torch.cuda.Stream(device=device)
        self.ev_search = [torch.cuda.Event() for _ in range(depth)]
        self.ev_done = [torch.cuda.Event() for _ in range(depth)]
        self.used = [False] * depth
        self.n = 0

    def submit(self, d_queries, search_parameters, events=None) -> int:
        import torch

        slot = self.n % self.depth
        self.n += 1
        cur = torch.cuda.current_stream(d_queries.device)
        if self.used[slot]:
            cur.wait_event(self.ev_done[slot])  # the merge that last read this buffer set has finished
        loc = self.loc[slot]
        if events is not None:
            events[0].record(cur)
        loc_d, loc_i = packed_views(loc, 1, self.nq, self.k)
        self.index.search_device(d_queries, self.k, search_parameters, d_ids64=loc_i[0], d_dists=loc_d[0])
        if events is not None:
            events[1].record(cur)
        self.ev_search[slot].record(cur)
        with torch.cuda.stream(self.s_merge):
            self.s_merge.wait_event(self.ev_search[slot])
            if self.collective:
                import torch.distributed as dist

                dist.all_gather_into_tensor(self.allb[slot], loc, group=self.group)
                src = self.allb[slot]
            else:
                src = loc
            all_d, all_i = packed_views(src, self.world, self.nq, self.k)
            _lib.check(_lib.lib().svsx_merge_topk_strided_device(
                self.metric, self.world, self.nq, self.k, all_d.data_ptr(), all_d.stride(0) if self.world > 1 else 0, all_i.data_ptr(),
                all_i.stride(0) if self.world > 1 else 0, self.out_d[slot].data_ptr(), self.out_i[slot].data_ptr(),
                C.c_void_p(self.s_merge.cuda_stream)))
            self.ev_done[slot].record(self.s_merge)
        self.used[slot] = True
        return slot

    def drain(self) -> None:
        import torch

        torch.cuda.current_stream().synchronize()
        self.s_merge.synchronize()


class NativeShardedSearcher:
    """The same search inside libsvsx (svsx_sharded_*: per-shard searches, local merge, ONE ncclAllGather on an RCCL communicator the
    library joins itself, world-way merge -- all on one stream, no torch collective on the data path).  What a C++ host of the
    reference would call; `torch.distributed` is only used here to hand rank 0's 128-byte unique id to the other ranks.

    indexes: this rank's VamanaIndex objects, global shards rank * len(indexes) + s (their id_offset makes the ids global).
    single_rank_comm: join a real one-rank RCCL communicator also when the world has one rank (exercises ncclAllGather on one GPU)"""

    def __init__(self, indexes, group=None, single_rank_comm: bool = False):
        import torch
        import torch.distributed as dist

        self.indexes = list(indexes)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        L = _lib.lib()
        id_ptr = None
        self._id = None
        if self.world > 1 or single_rank_comm:
            buf = (C.c_char * 128)()
            if self.rank == 0:
                _lib.check(L.svsx_sharded_unique_id(buf))
            if self.world > 1:  # hand the id to every rank through the process group that exists anyway (works on nccl and gloo groups)
                dev = torch.device("cuda", self.indexes[0].device) if dist.get_backend(group) == "nccl" else torch.device("cpu")
                t = torch.frombuffer(bytearray(bytes(buf)), dtype=torch.uint8).clone().to(dev)
                dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
                raw = bytes(t.cpu().numpy().tobytes())
                buf = (C.c_char * 128).from_buffer_copy(raw)
            self._id = buf
            id_ptr = C.cast(buf, C.c_void_p)
        arr = (C.c_void_p * len(self.indexes))(*[ix._h for ix in self.indexes])
        self._h = C.c_void_p()
        _lib.check(L.svsx_sharded_create(arr, len(self.indexes), id_ptr, self.rank, self.world, C.byref(self._h)))

    def info(self) -> dict:
        a = (C.c_uint64 * 4)()
        _lib.check(_lib.lib().svsx_sharded_info(self._h, a))
        return dict(local_shards=int(a[0]), world=int(a[1]), rank=int(a[2]), communicator=bool(a[3]))

    @property
    def shards_total(self) -> int:
        return self.world * len(self.indexes)

    def search(self, d_queries, k: int, search_parameters=None, d_ids64=None, d_dists=None, stream=None):
        """queries: torch CUDA tensor [nq, d]; returns (dists f32 [nq,k], ids int64 [nq,k]) -- same order as ShardedSearcher.search"""
        import torch

        nq = d_queries.shape[0]
        code = {torch.float32: 0, torch.float16: 1, torch.int8: 2, torch.uint8: 3}[d_queries.dtype]
        if d_ids64 is None:
            d_ids64 = torch.empty((nq, k), dtype=torch.int64, device=d_queries.device)
        if d_dists is None:
            d_dists = torch.empty((nq, k), dtype=torch.float32, device=d_queries.device)
        sp = (search_parameters or self.indexes[0]._params)._c()
        s = stream if stream is not None else torch.cuda.current_stream(d_queries.device).cuda_stream
        if d_queries.stride(0) != d_queries.shape[1] or d_queries.stride(1) != 1:
            d_queries = d_queries.contiguous()
        _lib.check(_lib.lib().svsx_sharded_search_device(self._h, d_queries.data_ptr(), code, nq, k, C.byref(sp), d_ids64.data_ptr(), d_dists.data_ptr(), C.c_void_p(s)))
        return d_dists, d_ids64

    def set_timing(self, enabled: bool) -> None:
        """svsx_sharded_set_timing: record phase-boundary events in the calls that follow (off by default)"""
        _lib.check(_lib.lib().svsx_sharded_set_timing(self._h, int(bool(enabled))))

    def phase_times(self) -> dict:
        """svsx_sharded_phase_times of the last (timed) call, ms: shard searches, local merge, all-gather (incl. waiting for the slowest rank),
        world merge -- summed over the call's chunks -- and the span from the first search to the last merge"""
        ms = (C.c_float * 5)()
        _lib.check(_lib.lib().svsx_sharded_phase_times(self._h, ms))
        return dict(search_ms=float(ms[0]), local_merge_ms=float(ms[1]), all_gather_ms=float(ms[2]), world_merge_ms=float(ms[3]), span_ms=float(ms[4]))

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            _lib.lib().svsx_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
