"""Read-only database sharding across the GPUs of one node (new design, the reference has no such mode).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Every rank owns an independent Vamana
sub-index over a contiguous id range (own graph, own entry point); a query batch is searched on every shard and the
per-shard top-k -- `nq*k*(4+8)` bytes per shard, 1.2 MB at nq=10^4, k=10 -- are exchanged with ONE all-gather and
merged per query with the metric's comparator (ties: lower shard, then position), so every rank ends with the same
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
    out_d = torch.empty((nq, k), dtype=torch.float32, device=gathered_d.device)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=gathered_d.device)
    _lib.check(_lib.lib().svsx_merge_topk_device(int(metric), S, nq, k, gathered_d.data_ptr(), gathered_i.data_ptr(), out_d.data_ptr(), out_i.data_ptr(),
                                                 C.c_void_p(torch.cuda.current_stream(gathered_d.device).cuda_stream)))
    return out_d, out_i


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
        all_d = torch.empty((self.world, nq, k), dtype=d_local.dtype, device=d_local.device)
        all_i = torch.empty((self.world, nq, k), dtype=i_local.dtype, device=i_local.device)
        dist.all_gather_into_tensor(all_d.view(-1), d_local.contiguous().view(-1), group=self.group)
        dist.all_gather_into_tensor(all_i.view(-1), i_local.contiguous().view(-1), group=self.group)
        return self.merge(self.metric, all_d, all_i, k)
