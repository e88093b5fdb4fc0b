#!/usr/bin/env python3
"""bench.py -- QPS of the batched Vamana greedy search on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

One "step" = one batch search (10 000 queries, k=10, search_window=128) over a deep-1M-like index
(1 000 000 x 96 float32, L2, Vamana graph R=64) resident in HBM; queries/results are device tensors.
N > 1 (launched by torch.distributed.run, one rank per GPU): every rank owns one read-only shard of the same
size (weak scaling), the query batch is the same on all ranks, per-shard top-k are exchanged with one RCCL
all-gather and merged on the GPU.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def deep_like_device(n, d, latent, seed, device, sigma=0.1):
    """SURVEY 8d generator G(latent, sigma): x = normalize(z A + sigma eps), A seeded by 7."""
    import torch

    ga = torch.Generator(device=device).manual_seed(7)
    a = torch.randn((latent, d), generator=ga, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, d), device=device)
    for lo in range(0, n, 1 << 20):
        m = min(1 << 20, n - lo)
        z = torch.randn((m, latent), generator=g, device=device)
        e = torch.randn((m, d), generator=g, device=device)
        x = z @ a + sigma * e
        out[lo : lo + m] = x / x.norm(dim=1, keepdim=True)
    return out


def brute_force_topk(x, q, k, chunk=1 << 18):
    """Exact L2 ground truth (plumbing, torch GEMM): ids [nq,k]."""
    import torch

    qn = (q * q).sum(1, keepdim=True)
    best_d = torch.full((q.shape[0], k), float("inf"), device=q.device)
    best_i = torch.zeros((q.shape[0], k), dtype=torch.int64, device=q.device)
    for lo in range(0, x.shape[0], chunk):
        xs = x[lo : lo + chunk]
        dist = qn + (xs * xs).sum(1)[None, :] - 2.0 * (q @ xs.T)
        dd, ii = torch.topk(dist, k, dim=1, largest=False)
        cat_d = torch.cat([best_d, dd], 1)
        cat_i = torch.cat([best_i, ii + lo], 1)
        sel = torch.topk(cat_d, k, dim=1, largest=False)
        best_d = sel.values
        best_i = torch.gather(cat_i, 1, sel.indices)
    return best_i


def recall_at(gt, ids, k):
    import torch

    hit = (ids[:, :k, None].to(torch.int64) == gt[:, None, :k]).any(2).sum().item()
    return hit / (gt.shape[0] * k)


_REAL_STDOUT = None


def _claim_stdout():
    """Keep fd 1 for the ONE JSON line: everything else that writes to stdout (RCCL's version banner, C libraries)
    is redirected to stderr for the lifetime of the process."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def _emit(line: str):
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (line + "\n").encode())


def main():
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1_000_000, help="vectors per GPU shard (default: the BASELINE config)")
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--window", type=int, default=128)
    ap.add_argument("--degree", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="exercise the multi-GPU code path (RCCL all-gather + merge) even at world size 1")
    ap.add_argument("--visited", type=int, default=1, help="visited filter: 0 off, 1 default size, >=64 explicit LDS entries (result-neutral)")
    ap.add_argument("--bound-codes", type=int, default=1, help="1: build the 128-B bound-code rows (result-neutral lower-bound rejection); 0: off")
    ap.add_argument("--pipeline", action="store_true", help="multi-GPU: overlap the gather + merge of a batch with the next search (PipelinedShardSearch; measured neutral on one GPU)")
    ap.add_argument("--graph-cache", default="", help="load/save the built graph here (lets a profiler run skip the build)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import scalablevectorsearch_amd as svs
    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    n, d, nq, k, R = args.n, 96, args.nq, 10, args.degree
    t0 = time.time()
    # shard `rank` of a (world x n) database: independent seeds per shard, global ids = local + rank*n
    x = deep_like_device(n, d, 24, seed=1 + 1000 * rank, device=device)
    q = deep_like_device(nq, d, 24, seed=2, device=device)  # identical on every rank
    torch.cuda.synchronize()
    t_gen = time.time() - t0

    t0 = time.time()
    bp = VamanaBuildParameters(alpha=1.2, graph_max_degree=R, window_size=2 * R, max_candidate_pool_size=750, prune_to=R - 4)
    cache = f"{args.graph_cache}.r{rank}" if args.graph_cache else ""
    if cache and os.path.exists(cache):
        blob = torch.load(cache, map_location=device)
        graph, entry = blob["graph"], blob["entry"]
    else:
        graph, entry = build_vamana_device(x, n, d, svs.DistanceL2(), bp, device=local_rank)
        if cache:
            torch.save({"graph": graph, "entry": entry}, cache)
    t_build = time.time() - t0
    deg_mean = float((graph[:, :R] >= 0).sum(1).float().mean())
    if rank == 0:
        log(f"[bench] data {t_gen:.1f}s, graph build {t_build:.1f}s (R={R}, mean degree {deg_mean:.1f}), entry {entry}")

    d_entry = torch.tensor([entry], dtype=torch.int32, device=device)
    index = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), d_entry, n, d, np.float32, device=local_rank, id_offset=rank * n)
    if args.bound_codes:
        index.build_bound_codes(args.bound_codes)
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(args.window), search_buffer_visited_set=args.visited)

    ids = torch.empty((nq, k), dtype=torch.int64, device=device)  # written as uint64
    dists = torch.empty((nq, k), dtype=torch.float32, device=device)
    stats = torch.zeros((nq, 6), dtype=torch.int32, device=device)
    if multi:
        # every rank ships ONE packed buffer [nq*k f32 distances | nq*k i64 ids]; the search writes straight into it
        from scalablevectorsearch_amd.sharded import packed_views

        loc_buf = torch.empty(3 * nq * k, dtype=torch.int32, device=device)
        dists = loc_buf[:nq * k].view(torch.float32).view(nq, k)
        ids = loc_buf[nq * k:].view(torch.int64).view(nq, k)
        all_buf = torch.empty(world * 3 * nq * k, dtype=torch.int32, device=device)
        all_d, all_i = packed_views(all_buf, world, nq, k)
        shard_stride_d, shard_stride_i = (all_d.stride(0), all_i.stride(0)) if world > 1 else (0, 0)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=device)

    import ctypes as C

    from scalablevectorsearch_amd import _lib

    def step(with_stats=False):
        index.search_device(q, k, sp, d_ids64=ids, d_dists=dists, d_stats=stats if with_stats else None)
        if multi:
            gather_and_merge()

    def gather_and_merge():
        dist.all_gather_into_tensor(all_buf, loc_buf)
        _lib.check(_lib.lib().svsx_merge_topk_strided_device(0, world, nq, k, all_d.data_ptr(), shard_stride_d, all_i.data_ptr(), shard_stride_i,
                                                             out_d.data_ptr(), out_i.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    # ---- correctness of what is being timed: recall against exact ground truth ---------------------------------
    step(with_stats=True)
    torch.cuda.synchronize()
    # exact ground truth by this repo's exhaustive search (MFMA f16 shortlist + exact re-rank, svsx_flat_search_device);
    # ids are local to the shard here (a fresh view without id_offset)
    flat = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), d_entry, n, d, np.float32, device=local_rank)
    gt_local, _ = flat.flat_search_device(q, k)
    torch.cuda.synchronize()
    del flat
    gt_local = gt_local + rank * n
    if multi:
        # global ground truth = merge of per-shard exact top-k
        gd = ((q[:, None, :] - x[(gt_local - rank * n)]) ** 2).sum(-1)
        g_all_d = torch.empty((world, nq, k), device=device)
        g_all_i = torch.empty((world, nq, k), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(g_all_d.view(-1), gd.contiguous().view(-1))
        dist.all_gather_into_tensor(g_all_i.view(-1), gt_local.contiguous().view(-1))
        cd = g_all_d.permute(1, 0, 2).reshape(nq, world * k)
        ci = g_all_i.permute(1, 0, 2).reshape(nq, world * k)
        sel = torch.topk(cd, k, dim=1, largest=False).indices
        gt = torch.gather(ci, 1, sel)
        recall = recall_at(gt, out_i, k)
    else:
        gt = gt_local
        recall = recall_at(gt, ids, k)
    st = stats.to(torch.int64)
    exp_total = int(st[:, 0].sum())
    slots_total = int(st[:, 1].sum())
    # SURVEY 8(d) algorithmic bytes: sum_v [4(1+deg v) + deg v * d * 4] + n_entry*d*4 + d*4 + k*(8+4) per query
    alg_bytes = 4 * (exp_total + slots_total) + slots_total * d * 4 + nq * (d * 4 + d * 4 + k * 12)

    # ---- timed region ---------------------------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    if multi and args.pipeline:
        # throughput form: batch i's gather + merge (second stream) overlaps batch i+1's search kernel; all K batches
        # are complete before the closing synchronize + barrier
        from scalablevectorsearch_amd.sharded import PipelinedShardSearch

        pipe = PipelinedShardSearch(index, 0, nq, k, depth=2, device=device)
        for _ in range(2):
            pipe.submit(q, sp)
        pipe.drain()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            slot = pipe.submit(q, sp, events=ev[i])
        pipe.drain()
        assert torch.equal(pipe.out_i[slot], out_i) and torch.equal(pipe.out_d[slot], out_d), "pipelined result differs from the step-by-step one"
    else:
        for i in range(args.steps):
            ev[i][0].record()
            index.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
            ev[i][1].record()
            if multi:
                gather_and_merge()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    if rank != 0:
        if multi:
            dist.destroy_process_group()
        return

    # practical roof: device-to-device copy bandwidth on this box (read + write bytes / time)
    cp_src = torch.empty(1 << 29, dtype=torch.float32, device=device)  # 2 GiB
    cp_dst = torch.empty_like(cp_src)
    cp_dst.copy_(cp_src)
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(5):
        cp_dst.copy_(cp_src)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 5 * 2 * cp_src.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del cp_src, cp_dst

    ms_per_step = 1e3 * elapsed / args.steps
    shard_queries_per_s = world * nq * args.steps / elapsed
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic(n, nq, args.window, R, args.visited, args.bound_codes) if not multi else (None, None)
    result = {
        "metric": "QPS at recall@10>=0.9 (batch=10k); achieved HBM GB/s vs roofline",
        "value": shard_queries_per_s,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"deep-1M-like d=96 float32 L2, batch={nq}, search_window={args.window}, k=10, 1xMI355X per shard",
            "vectors_per_gpu": n,
            "graph": f"Vamana R={R} L={2*R} alpha=1.2 (GPU-built, mean degree {deg_mean:.1f})",
            "recall_at_10": recall,
            "sharded_index_qps": nq * args.steps / elapsed,
            "shards": world,
            "expansions_per_query": exp_total / nq,
            "neighbor_slots_per_query": slots_total / nq,
            "build_seconds": t_build,
            "visited_filter": args.visited,
            "distance_evals_per_query": int(st[:, 2].sum()) / nq,
            "partial_distance_rejects_per_query": int(st[:, 4].sum()) / nq,
            "accepted_inserts_per_query": int(st[:, 3].sum()) / nq,
            "bound_codes": args.bound_codes,
            "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_rate": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
            "frac_traffic": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "measured_copy_GBs": copy_gbs,
            "note": "achieved = ALGORITHMIC bytes (reference traversal: visited set off, whole rows; SURVEY 8d) / kernel time; "
                    "the kernel skips result-neutral work (visited filter, bound-code / partial-distance rejection), so measured traffic "
                    "is below the algorithmic bytes and frac can exceed 1; traffic_rate = measured fabric bytes / kernel time",
            "kernel": "svsx::search_kernel<f32,L2,float,96%s>" % (",bound-codes" if args.bound_codes else ""),
            "kernel_ms": kern_ms,
            "algorithmic_bytes_per_launch": alg_bytes,
        },
    }

    if not multi:
        # smallest window with recall@10 >= 0.9 (binary search of index/vamana/calibrate.h:286-300), timed the same way
        lo_w, hi_w = k, args.window
        while lo_w < hi_w:
            mid = (lo_w + hi_w) // 2
            index.search_device(q, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(mid), args.visited), d_ids64=ids, d_dists=dists)
            torch.cuda.synchronize()
            if recall_at(gt, ids, k) >= 0.9:
                hi_w = mid
            else:
                lo_w = mid + 1
        sp_m = svs.VamanaSearchParameters(svs.SearchBufferConfig(lo_w), args.visited)
        for _ in range(args.warmup):
            index.search_device(q, k, sp_m, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            index.search_device(q, k, sp_m, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        matched = {"window": lo_w, "recall_at_10": recall_at(gt, ids, k), "gpu_qps": nq * args.steps / dt}
        ids_matched = ids.clone()
        result["config"]["matched_recall_point"] = matched
        index.search_device(q, k, sp, d_ids64=ids, d_dists=dists)  # restore window-128 results for the parity sample
        torch.cuda.synchronize()
    if not multi and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(x, q, graph, entry, R, k, args.window, ids)
        cm = cpu_baseline(x, q, graph, entry, R, k, matched["window"], ids_matched)
        matched["cpu_qps"] = cm["value"]
        matched["cpu_cores"] = cm["cores"]
        matched["gpu_over_cpu"] = matched["gpu_qps"] / cm["value"]
        matched["ids_identical_on_sample"] = "identical on sample: True" in cm["sample"]
    _emit(json.dumps(result))
    if multi:
        dist.destroy_process_group()


def measured_traffic(n, nq, window, degree, visited, bound_codes=0):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/rNN_pmc.json, produced by tools/collect_pmc.sh: separate --pmc passes, gfx950 FETCH_SIZE x2 correction
    calibrated on a known stream).  bench.py cannot host the profiler itself; returns (None, None) if no pass matches."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            j = json.load(open(f))
            want = {"n": n, "d": 96, "nq": nq, "window": window, "degree": degree, "visited": visited}
            if bound_codes:
                want["bound_codes"] = bound_codes
            if j["workload"] == want:
                return float(j["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT)
        except Exception:
            continue
    return None, None


def host_cores():
    """Threads the host leg may really use: min(affinity mask, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(x, q, graph, entry, R, k, window, gpu_ids):
    """Time the CPU checker on this box's host cores over a bounded sample of the SAME workload and check that the
    GPU ids are identical on that sample.  kind="reference": oracle/_ref (the reference's own greedy_search /
    SearchBuffer / generic_simd_op, AVX-512, static partition over threads); kind="port": the scalar C oracle."""
    from scalablevectorsearch_amd.build import device_graph_to_svs

    from oracle.oracle import L2, Oracle, Ref

    xh = x.cpu().numpy()
    qh = q.cpu().numpy()
    gh = device_graph_to_svs(graph, R)
    eh = np.array([entry], np.uint32)
    ncores = host_cores()
    if Ref.available():
        r = Ref()
        kind, threads = "reference", ncores
        run = lambda qq: r.search(L2, xh, gh, eh, qq, k, window, nthreads=threads)
    else:
        o = Oracle()
        kind, threads = "port", 1
        run = lambda qq: o.search(L2, xh, gh, eh, qq, k, window)
    # size the sample for ~10-20 s of CPU work
    probe = qh[: min(len(qh), 64 * threads)]
    t0 = time.perf_counter()
    run(probe)
    rate = len(probe) / (time.perf_counter() - t0)
    ns = int(min(len(qh), max(len(probe), rate * 5.0)))
    best = None
    ids_cpu = None
    t_all = time.perf_counter()
    for _ in range(3):
        t0 = time.perf_counter()
        ids_cpu, _ = run(qh[:ns])
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
        if time.perf_counter() - t_all > 20:
            break
    same = bool((gpu_ids[:ns].cpu().numpy().astype(np.uint64) == ids_cpu.astype(np.uint64)).all())
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {
        "value": ns / best,
        "unit": "queries/s",
        "cores": threads,
        "host": {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": ncores},
        "kind": kind,
        "sample": f"first {ns} of the {len(qh)} queries, same graph/data/window, best of <=3 runs; GPU ids identical on sample: {same}",
    }


if __name__ == "__main__":
    main()
