"""CPU check of the ALGORITHMS of staging/r3/builder/svsx_builder.hip (restated in numpy -- not the HIP code itself) against the round-2
torch-glue builder (scalablevectorsearch_amd/build.py): reverse-edge bookkeeping (edge keys -> sort -> group-leader scatter -> overflow
lists) and pool assembly (dedupe by id keeping the first occurrence, order by (distance, id), truncate).
    python staging/r3/builder/check_algorithms.py"""
import os
import struct
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from scalablevectorsearch_amd.build import _make_pool  # noqa: E402

INV = -1
NO = (1 << 64) - 1


def torch_reverse(graph, new_rows, lo, R, over_cap):
    """build.py's add_reverse_edges bookkeeping (up to the overflow candidate table), on CPU tensors"""
    graph = graph.clone()
    B, gs = new_rows.shape
    graph[lo:lo + B] = new_rows
    node_ids = torch.arange(lo, lo + B)
    src = node_ids[:, None].expand(B, gs).reshape(-1)
    dst = new_rows.reshape(-1).to(torch.int64)
    keep = dst >= 0
    src, dst = src[keep], dst[keep]
    present = (graph[dst, :R].to(torch.int64) == src[:, None]).any(1)
    src, dst = src[~present], dst[~present]
    over = {}
    if dst.numel() > 0:
        order = torch.argsort(dst, stable=True)
        src, dst = src[order], dst[order]
        uniq, inv, cnt = torch.unique_consecutive(dst, return_inverse=True, return_counts=True)
        start = torch.cumsum(cnt, 0) - cnt
        rank = torch.arange(dst.numel()) - start[inv]
        deg = (graph[uniq, :R] >= 0).sum(1)
        slot = deg[inv] + rank
        fits = slot < R
        graph[dst[fits], slot[fits]] = src[fits].to(torch.int32)
        ov = ~fits
        o_dst, o_src, o_rank = dst[ov], src[ov], slot[ov] - R
        ok = o_rank < over_cap
        for d_, s_ in zip(o_dst[ok].tolist(), o_src[ok].tolist()):
            over.setdefault(d_, []).append(s_)
    return graph, over


def kernel_reverse(graph, new_rows, lo, R, over_cap):
    """edge_keys_kernel + radix sort + reverse_scatter_kernel"""
    g = graph.clone().numpy()
    B, _ = new_rows.shape
    g[lo:lo + B] = new_rows.numpy()
    keys = []
    for v in range(B):
        for j in range(R):
            src = lo + v
            dst = int(g[src, j])
            key = NO
            if dst != INV and not (g[dst, :((R + 3) // 4) * 4] == src).any():
                key = (dst << 32) | src
            keys.append(key)
    keys = sorted(keys)
    m = len(keys)
    over = {}
    for i, key in enumerate(keys):
        if key == NO:
            continue
        dst = key >> 32
        if i > 0 and (keys[i - 1] >> 32) == dst:
            continue
        cnt = 1
        while i + cnt < m and (keys[i + cnt] >> 32) == dst and keys[i + cnt] != NO:
            cnt += 1
        row = g[dst]
        lo_, hi_ = 0, R
        while lo_ < hi_:
            mid = (lo_ + hi_) >> 1
            if row[mid] != INV:
                lo_ = mid + 1
            else:
                hi_ = mid
        deg = lo_
        fit = min(cnt, R - deg)
        for r in range(fit):
            row[deg + r] = keys[i + r] & 0xFFFFFFFF
        if cnt > fit:
            over[dst] = [keys[i + fit + t] & 0xFFFFFFFF for t in range(min(cnt - fit, over_cap))]
    return torch.from_numpy(g), over


def order_bits(x):
    x = np.float32(x)
    if x == 0:
        x = np.float32(0.0)
    u = struct.unpack("<I", struct.pack("<f", x))[0]
    return (~u) & 0xFFFFFFFF if u & 0x80000000 else u | 0x80000000


def order_bits_inv(e):
    u = (e & 0x7FFFFFFF) if e & 0x80000000 else (~e) & 0xFFFFFFFF
    return struct.unpack("<f", struct.pack("<I", u))[0]


def kernel_pool(ids, dists, greater, pool_size, N):
    """pool_kernel: sort (id, position), drop later duplicates, sort (order bits of the distance, id), truncate"""
    total = len(ids)
    keys = sorted([(int(ids[i]) << 32) | i if i < total else NO for i in range(N)])
    out = []
    for i, k in enumerate(keys):
        if k == NO:
            out.append(NO)
            continue
        idd = k >> 32
        drop = i > 0 and keys[i - 1] != NO and (keys[i - 1] >> 32) == idd
        e = order_bits(dists[k & 0xFFFFFFFF])
        e = (~e) & 0xFFFFFFFF if greater else e
        out.append(NO if drop else (e << 32) | idd)
    out = sorted(out)[:pool_size]
    worst = -np.inf if greater else np.inf
    pi = [(-1 if k == NO else k & 0xFFFFFFFF) for k in out]
    pd = [(worst if k == NO else order_bits_inv(((~(k >> 32)) & 0xFFFFFFFF) if greater else (k >> 32))) for k in out]
    return np.array(pi, np.int64), np.array(pd, np.float32)


def main():
    rng = np.random.default_rng(0)
    for trial in range(30):
        n = int(rng.integers(60, 300))
        R = int(rng.choice([4, 8, 12]))
        gs = (R + 15) // 16 * 16
        over_cap = int(rng.choice([2, 64]))
        graph = torch.full((n, gs), INV, dtype=torch.int32)
        for v in range(n):  # random packed rows without self loops / duplicates
            deg = int(rng.integers(0, R + 1))
            graph[v, :deg] = torch.from_numpy(rng.choice(np.delete(np.arange(n), v), deg, replace=False).astype(np.int32))
        lo = int(rng.integers(0, n - 20))
        B = int(rng.integers(5, min(60, n - lo)))
        new_rows = torch.full((B, gs), INV, dtype=torch.int32)
        for v in range(B):
            deg = int(rng.integers(0, R + 1))
            new_rows[v, :deg] = torch.from_numpy(rng.choice(np.delete(np.arange(n), lo + v), deg, replace=False).astype(np.int32))
        ga, oa = torch_reverse(graph, new_rows, lo, R, over_cap)
        gb, ob = kernel_reverse(graph, new_rows, lo, R, over_cap)
        assert torch.equal(ga, gb) and oa == ob, trial
    print("reverse edges: restated kernels == torch builder on 30 random cases (graph bytes and overflow lists)")
    for trial in range(200):
        greater = bool(trial & 1)
        tot, pool = int(rng.integers(1, 120)), int(rng.integers(4, 90))
        ids = rng.integers(0, 50, tot).astype(np.int64)
        dists = np.round(rng.normal(size=50).astype(np.float32), 1)[ids]  # duplicates of an id carry the same distance; ties between ids exist
        pad = int(rng.integers(0, 10))
        ri, rd = _make_pool(torch.from_numpy(np.concatenate([ids, -np.ones(pad, np.int64)]))[None],
                            torch.from_numpy(np.concatenate([dists, np.zeros(pad, np.float32)]))[None], greater, pool)
        ki, kd = kernel_pool(ids, dists, greater, pool, 128)
        L = min(ri.shape[1], pool)
        assert (ri[0].numpy()[:L] == ki[:L]).all() and (rd[0].numpy()[:L] == kd[:L]).all(), trial
    print("pool assembly: restated kernel == torch _make_pool on 200 random cases")


if __name__ == "__main__":
    main()
