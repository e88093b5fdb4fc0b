#!/usr/bin/env python3
"""bench.py -- QPS of the batched Vamana greedy search on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (driver contract; prints ONE JSON line on rank 0)

Default workload = BASELINE config 2: one "step" = one batch search (10 000 queries, k = 10, search_window = 128) over a
deep-1M-like index (1 000 000 x 96 float32, L2, Vamana R = 64) resident in HBM, queries / results device resident.
  N > 1 (torch.distributed.run, one rank per GPU): REPLICAS -- every rank holds the same 1M-vector index and answers its own
        10 000-query batch; no data-path collective; `value` = answers per second of the whole job (N = 1 equals the single-GPU line).
  Every line also carries `config.sharded_c5`: the sharded search of SURVEY 8(e) -- a FIXED set of 8 shards, 8/N per GPU, per-shard
        top-k -> local merge -> ONE RCCL all-gather -> global merge, all inside libsvsx (svsx_sharded_*), merged answers per second.
  --config c5: the same sharded search as the headline, at --n vectors per shard (default 12 500 000: BASELINE config 5).
"""
from __future__ import annotations

import argparse
import glob
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured by a float4 copy)
PREWARM_MIN, PREWARM_MAX = 24, 64  # untimed steps in front of --warmup (clock ramp after idle; see the timed region in main)
LINE = 128             # bytes per fabric request: a 64-byte gather costs the same as a 128-byte one (tools/ubench/gather_bench)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


# ---- own counter-based generator (no library normal_distribution) ------------------------------------------------------
def _mix64(x):
    """splitmix64 finalizer on int64 tensors (wrapping arithmetic, logical shifts)."""

    def lsr(v, s):
        return (v >> s) & ((1 << (64 - s)) - 1)

    x = (x ^ lsr(x, 30)) * -4658895280553007687   # 0xBF58476D1CE4E5B9
    x = (x ^ lsr(x, 27)) * -7723592293110705685   # 0x94D049BB133111EB
    return x ^ lsr(x, 31)


def counter_normal(n_rows, n_cols, seed, stream, device, row0=0):
    """N(0,1) samples addressed by (seed, stream, row, col): value = Box-Muller of two 24-bit uniforms hashed from the
    counter.  Pure integer hashing + float64 log / cos: the same numbers on every torch version and every device."""
    import torch

    r = torch.arange(row0, row0 + n_rows, dtype=torch.int64, device=device)[:, None]
    c = torch.arange(n_cols, dtype=torch.int64, device=device)[None, :]
    ctr = (r * 1_000_003 + c) * 2 + (seed * 0x9E3779B97F4A7C15 + stream * 0xD1B54A32D192ED03) % (1 << 62)
    h1, h2 = _mix64(ctr), _mix64(ctr + 1)
    u1 = (((h1 >> 40) & 0xFFFFFF).to(torch.float64) + 0.5) / 16777216.0
    u2 = (((h2 >> 40) & 0xFFFFFF).to(torch.float64) + 0.5) / 16777216.0
    return (torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(6.283185307179586 * u2)).to(torch.float32)


def deep_like_device(n, d, latent, seed, device, sigma=0.1, normalize=True, row0=0):
    """SURVEY 8d generator G(latent, sigma): x = normalize(z A + sigma eps), A from seed 7, own counter-based RNG.  Rows row0 .. row0 + n of
    the stream (a pure function of the row index: chunked generation equals one-shot generation)."""
    import torch

    a = counter_normal(max(latent, 1), d, 7, 0, device)
    out = torch.empty((n, d), device=device)
    for lo in range(0, n, 1 << 18):
        m = min(1 << 18, n - lo)
        z = counter_normal(m, max(latent, 1), seed, 1, device, row0=row0 + lo)
        e = counter_normal(m, d, seed, 2, device, row0=row0 + lo)
        x = (z @ a + sigma * e) if latent > 0 else e   # latent = 0: no low-rank part at all -- isotropic rows (a flat spectrum)
        out[lo : lo + m] = x / x.norm(dim=1, keepdim=True) if normalize else x
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


def kernel_src_sha() -> str:
    """Hash of the sources the search kernel and its inline records are built from: a PMC summary is only reused for the kernel it was measured on
    (flat search, builder, sharded host logic and the C ABI file do not enter: they cannot change the search kernel's traffic)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "scalablevectorsearch_amd", "csrc")
    names = ["svsx_device.h", "svsx_kernels.h", "svsx_search_kernel.h", "svsx_search.hip", "svsx_search_f32.hip", "svsx_search_f16.hip", "svsx_search_i8.hip",
             "svsx_search_u8.hip", "svsx_codes.hip", "Makefile"]
    for f in names:
        h.update(open(os.path.join(csrc, f), "rb").read())
    return h.hexdigest()[:16]


def stat3(ms_list, nq):
    """benchmark/include/svs-benchmark/search.h:209-219: report min / median / max over the timed runs."""
    a = np.sort(np.asarray(ms_list, np.float64))
    return {"runs": len(a), "min_ms": float(a[0]), "median_ms": float(np.median(a)), "max_ms": float(a[-1]),
            "qps_at_median": nq / (float(np.median(a)) * 1e-3), "qps_best": nq / (float(a[0]) * 1e-3)}


def designed_lines(stats_i64, gstride, row_bytes, bound_codes, d, k, record_bytes=32, degree_aware=True):
    """128-byte lines one launch is DESIGNED to request (what `roofline.achieved` prices): per expansion the adjacency block
    (ids, + inline records in mode 2), one code line per evaluated candidate in mode 1, the full row of every candidate
    that reaches the exact path (entry point, survivors of the bound / all evaluated candidates without a bound), the
    query and the result rows.  Counted from the kernel's own per-query statistics.
    degree_aware (round 5, inline records of indexes up to 2^26 nodes): an expansion requests only the 8-slot groups of its block that hold
    neighbours -- one line per group and record plane, plus the id lines in front of them; the number of groups is taken from the mean
    number of neighbour slots per expansion (ceil(degree / 8) averages (degree + 3.5) / 8 over a spread of degrees)."""
    exp = int(stats_i64[:, 0].sum())
    evals = int(stats_i64[:, 2].sum())
    early = int(stats_i64[:, 4].sum())
    rej = int(stats_i64[:, 5].sum())
    nq = stats_i64.shape[0]
    row_lines = -(-row_bytes // LINE)
    if bound_codes == 2:
        if degree_aware and exp > 0:
            groups = min((int(stats_i64[:, 1].sum()) / exp + 3.5) / 8.0, gstride / 8.0)
            full_chunks = int(groups // 8)
            rem = groups - 8 * full_chunks
            id_lines = 2 * full_chunks + (-(-int(round(rem * 32)) // LINE) if rem > 0 else 0)
            adj = int(round(exp * (id_lines + (record_bytes / 16.0) * groups)))  # (28-byte records: a 16- and a 12-byte plane = 1.75 lines per group of 8 slots)
        else:
            adj = exp * (-(-(gstride * (4 + record_bytes)) // LINE))
        code = 0
    else:
        adj = exp * (-(-(gstride * 4) // LINE))
        code = (evals - nq) if bound_codes == 1 else 0  # every non-entry candidate reads its code line once the buffer is full (upper bound)
    rows_full = evals - rej - early
    rows = rows_full * row_lines + early * min(2, row_lines)  # partial-distance rejects stop after two lines
    fixed = nq * (-(-(d * 4) // LINE) + -(-(k * 12) // LINE))
    return {"adjacency": adj, "codes": code, "rows": rows, "query_result": fixed, "total": adj + code + rows + fixed}


def measured_traffic(workload_key):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r*_pmc.json,
    produced by tools/collect_pmc.sh: separate --pmc passes, gfx950 FETCH_SIZE x2 correction calibrated on a known stream).
    Only a summary measured on THESE kernel sources (kernel_src_sha) for THIS workload is used; otherwise (None, None, None)."""
    sha = kernel_src_sha()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            j = json.load(open(f))
            if j.get("workload_key") == workload_key and j.get("kernel_src_sha") == sha:
                return float(j["traffic_bytes_per_launch"]), os.path.relpath(f, ROOT), j
        except Exception:
            continue
    return None, None, None


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
    ns = int(min(len(qh), max(len(probe), rate * 3.0)))
    times = []
    ids_cpu = None
    t_all = time.perf_counter()
    for _ in range(6):  # 1 warm-up + 5 timed (search.h:209-219), bounded to ~20 s
        t0 = time.perf_counter()
        ids_cpu, _ = run(qh[:ns])
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 20 and len(times) >= 3:
            break
    timed = times[1:] if len(times) > 1 else times
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
        "value": ns / min(timed),
        "unit": "queries/s",
        "cores": threads,
        "cores_present": os.cpu_count(),
        "host": {"cpu_model": model, "logical_cpus": os.cpu_count(), "usable_cpus": ncores},
        "kind": kind,
        "qps_min_median_max": [ns / max(timed), ns / float(np.median(timed)), ns / min(timed)],
        "sample": f"first {ns} of the {len(qh)} queries, same graph/data/window, 1 warm-up + {len(timed)} timed runs (best reported); "
                  f"{threads} threads = every CPU the container's cgroup grants of {os.cpu_count()} logical CPUs present; GPU ids identical on sample: {same}",
    }


def cpu_scaling(x, q, graph, entry, R, k, windows):
    """QPS of the CPU leg at 1 / 2 / 4 / ... threads (the reference's own code, oracle/_ref; static partition of the queries over threads like
    index/vamana/index.h:571-574), 1 warm-up + 3 timed runs each (best reported; benchmark/include/svs-benchmark/search.h:209-219), samples sized
    for ~1.5 s per run: how far from linear the host leg scales, next to the speed-up that assumes it."""
    from scalablevectorsearch_amd.build import device_graph_to_svs

    from oracle.oracle import L2, Ref

    if not Ref.available():
        return None
    r = Ref()
    xh, qh, gh, eh = x.cpu().numpy(), q.cpu().numpy(), device_graph_to_svs(graph, R), np.array([entry], np.uint32)
    ncores = host_cores()
    counts = [t for t in (1, 2, 4, 8, 16, 32, 64, 128, 256) if t <= ncores]
    if counts[-1] != ncores:
        counts.append(ncores)
    out = {}
    for w in windows:
        t0 = time.perf_counter()
        r.search(L2, xh, gh, eh, qh[:32], k, w, nthreads=1)
        per_thread_rate = 32 / (time.perf_counter() - t0)
        row = {}
        for t in counts:
            ns = int(min(len(qh), max(32 * t, per_thread_rate * t * 1.5)))
            best = 0.0
            for rep in range(4):
                t0 = time.perf_counter()
                r.search(L2, xh, gh, eh, qh[:ns], k, w, nthreads=t)
                dt = time.perf_counter() - t0
                if rep:
                    best = max(best, ns / dt)
            row[str(t)] = best
        out[str(w)] = {"qps_by_threads": row, "efficiency_at_max_threads": row[str(counts[-1])] / (row["1"] * counts[-1])}
    return out


def build_or_load_graph(svs, x, n, d, R, local_rank, cache, distance=None, alpha=1.2, data_dtype=np.float32, sq=None, build_window=0):
    import torch

    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device

    t0 = time.time()
    if cache and os.path.exists(cache):
        blob = torch.load(cache, map_location=x.device)
        graph, entry = blob["graph"], blob["entry"]
    else:
        bp = VamanaBuildParameters(alpha=alpha, graph_max_degree=R, window_size=build_window or 2 * R, max_candidate_pool_size=750, prune_to=R - 4)
        graph, entry = build_vamana_device(x, n, d, distance if distance is not None else svs.DistanceL2(), bp, data_dtype=data_dtype, device=local_rank, sq=sq)
        if cache:
            torch.save({"graph": graph, "entry": entry}, cache)
    return graph, entry, time.time() - t0


def sharded_shard_size(args, world, total_hbm):
    """Vectors per shard of the `sharded_search` point: --c5-n if given, else BASELINE config 5's 12 500 000 at EVERY N (the
    1 -> N curve must be one workload) -- 8 / N shards per GPU, with their inline records where those fit (N >= 2: 37 GB per shard), with
    128-byte code rows where they do not (N = 1: all eight shards on one GPU are 77 GB + 12.8 GB).  1 000 000 only if even that does not fit."""
    if args.c5_n:
        return args.c5_n
    per = max(1, 8 // world)
    floor_bytes = per * 12_500_000 * (96 * 4 + 64 * 4 + 128 + 64) * 1.15  # vectors, graph, code rows, slack
    return 12_500_000 if floor_bytes < 0.8 * total_hbm else 1_000_000


def sharded_c5_point(args, svs, q, k, R, d, rank, world, local_rank, device, reuse_shard0, steps, warmup):
    """SURVEY 8(e) as a measurement inside the default line: a FIXED set of 8 shards of --c5-n vectors (contiguous id ranges, global
    ids), 8 / N per GPU, every rank passes the same query batch to svsx_sharded_search_device: per-shard searches, local merge, ONE
    ncclAllGather on a communicator the library joined itself, world-way merge.  Collective: every rank calls this.  Returns a dict
    (merged answers per second over max-over-ranks time, recall against the global exact top-k)."""
    import torch
    import torch.distributed as dist

    from scalablevectorsearch_amd.sharded import NativeShardedSearcher

    total = 8
    if total % world:
        return {"skipped": f"8 shards do not divide over {world} GPUs"}
    per = total // world
    nq = q.shape[0]
    # --c5-n 0 (default): BASELINE config 5's own shard, 12 500 000 vectors, whenever the 8 / N shards of a GPU fit its HBM with their inline
    # records (N >= 2: 37 GB per shard), else -- one GPU holding all eight -- 1 000 000 vectors per shard so that the default N = 1 line stays
    # within minutes (the 100 M index on one GPU is `--config c5`: profiles/r03_bench_c5_100M_8shards_1gpu.json)
    total_hbm = torch.cuda.get_device_properties(device).total_memory
    n5 = sharded_shard_size(args, world, total_hbm)
    gs5 = (R + 15) // 16 * 16
    mode5 = args.bound_codes
    if mode5 >= 2 and per * n5 * (d * 4 + gs5 * 4 + gs5 * 36 + 64) > 0.75 * total_hbm - torch.cuda.memory_allocated(device):
        mode5 = 1  # the inline records of this GPU's shards do not fit next to their vectors: 128-byte code rows (the same choice `--config c5` makes)
    t0 = time.time()
    keep, indexes = [], []
    for s_ in range(per):
        sid = rank * per + s_
        if sid == 0 and reuse_shard0 is not None and n5 == reuse_shard0["n"]:
            sh = reuse_shard0  # shard 0 IS the replica's database (seed 1, id offset 0)
        else:
            x = deep_like_device(n5, d, 24, seed=1 + 1000 * sid, device=device)
            graph, entry, _ = build_or_load_graph(svs, x, n5, d, R, local_rank, "")
            d_entry = torch.tensor([entry], dtype=torch.int32, device=device)
            index = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), d_entry, n5, d, np.float32, device=local_rank, id_offset=sid * n5)
            if mode5:
                index.build_bound_codes(mode5)
            sh = dict(x=x, graph=graph, entry=entry, d_entry=d_entry, index=index, sid=sid, n=n5)
        keep.append(sh)
        indexes.append(sh["index"])
    torch.cuda.synchronize()
    t_build = time.time() - t0
    ns = NativeShardedSearcher(indexes, single_rank_comm=True)  # a real RCCL communicator at every N, also N = 1
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(args.window), search_buffer_visited_set=args.visited)
    out_i = torch.empty((nq, k), dtype=torch.int64, device=device)
    out_d = torch.empty((nq, k), dtype=torch.float32, device=device)
    for _ in range(max(5, warmup)):
        ns.search(q, k, sp, d_ids64=out_i, d_dists=out_d)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ns.search(q, k, sp, d_ids64=out_i, d_dists=out_d)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # where a batch's time goes, from the library's own phase events (a few extra, untimed calls): what explains the scaling curve at N > 1
    phases = None
    try:
        ns.set_timing(True)
        acc = []
        for _ in range(3):
            ns.search(q, k, sp, d_ids64=out_i, d_dists=out_d)
            acc.append(ns.phase_times())
        ns.set_timing(False)
        phases = {kk_: float(np.median([a_[kk_] for a_ in acc])) for kk_ in acc[0]}
        if world > 1:  # every rank's view: the all-gather phase of a fast rank contains its wait for the slowest one
            t = torch.tensor([phases[kk_] for kk_ in sorted(phases)], device=device, dtype=torch.float64)
            allt = torch.empty((world, t.numel()), device=device, dtype=torch.float64)
            dist.all_gather_into_tensor(allt.view(-1), t)
            phases = {"rank0": phases, "max_over_ranks": {kk_: float(allt[:, i].max()) for i, kk_ in enumerate(sorted(phases))},
                      "min_over_ranks": {kk_: float(allt[:, i].min()) for i, kk_ in enumerate(sorted(phases))}}
    except Exception as e:  # noqa: BLE001 (informational)
        phases = {"error": f"{type(e).__name__}: {e}"}
    # recall against the exact global top-k: exact per-shard lists (MFMA flat search + exact re-rank), gathered, best k overall
    gd, gi = [], []
    for sh in keep:
        flat = svs.VamanaIndex.from_device(sh["graph"], sh["x"], svs.DistanceL2(), sh["d_entry"], n5, d, np.float32, device=local_rank)
        ids_, _ = flat.flat_search_device(q, k)
        torch.cuda.synchronize()
        del flat
        gd.append(((q[:, None, :] - sh["x"][ids_]) ** 2).sum(-1))
        gi.append(ids_ + sh["sid"] * n5)
    cd, ci = torch.cat(gd, 1), torch.cat(gi, 1)
    if world > 1:
        a_d = torch.empty((world,) + tuple(cd.shape), device=device)
        a_i = torch.empty((world,) + tuple(ci.shape), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(a_d.view(-1), cd.contiguous().view(-1))
        dist.all_gather_into_tensor(a_i.view(-1), ci.contiguous().view(-1))
        cd, ci = a_d.permute(1, 0, 2).reshape(nq, -1), a_i.permute(1, 0, 2).reshape(nq, -1)
    gt = torch.gather(ci, 1, torch.topk(cd, k, dim=1, largest=False).indices)
    info = ns.info()
    res = {"shards_total": total, "shards_per_gpu": per, "vectors_per_shard": n5, "n_gpus": world, "batch": nq, "window_per_shard": args.window,
           "merged_answers_per_s": nq * steps / elapsed, "ms_per_batch": 1e3 * elapsed / steps, "steps": steps, "recall_at_10": recall_at(gt, out_i, k),
           "rccl_communicator": info["communicator"], "rccl_ranks": info["world"], "build_seconds_this_rank": t_build, "bound_codes": mode5,
           "phases_ms": phases,
           "phases_note": "svsx_sharded_phase_times (events inside the library, median of 3 extra calls): search = the per-shard searches of this GPU, all_gather "
                          "includes waiting for the slowest rank; span < sum of phases when the second chunk's searches overlap the first chunk's exchange",
           "expected_at_8_gpus": "DESIGN.md section 6: one 12.5 M-vector shard per GPU ~1.17 ms per 10 000 queries + an exposed half-batch exchange and two merges "
                                 "~0.25 ms => ~7 M merged answers/s; falsified or confirmed by the first 8-GPU run of this line",
           "path": "svsx_sharded_search_device (libsvsx: per-shard search -> local merge -> ONE ncclAllGather -> world-way merge, one stream)",
           "scaling": "strong (the set of 8 shards is fixed as N grows)",
           "shard_size_note": ("BASELINE config 5's shard size at every N (inline records where the GPU's shards fit with them, 128-byte code rows where not: bound_codes)"
                               if n5 == 12_500_000 else f"{n5} vectors per shard (--c5-n, or 12.5 M-vector shards do not fit this GPU)")}
    ns.close()
    return res


def spectrum_points(args, svs, device, local_rank, q_seed, k, R, headline):
    """how the headline moves with the data's spectrum.  The inline records bound a distance from the coordinates along the
    leading principal directions; G(24, 0.1) puts nearly all variance into 24 of them.  Two more sets of the same shape (1 M x 96 float32 unit
    rows, L2, batch, window, graph parameters of the headline): G(48, 0.2) and isotropic rows (no low-rank part: the flattest spectrum there
    is, where a 32-byte record says almost nothing and the size rule has to pay for 128-byte ones).  Each: QPS, recall, record size chosen,
    rejects and lines per query, the plain kernel on the same graph, ids against the reference on a query sample."""
    import torch

    from scalablevectorsearch_amd.build import device_graph_to_svs

    n, d, nq = 1_000_000, 96, args.nq
    out = [headline]
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(args.window), search_buffer_visited_set=args.visited)
    for name, latent, sigma in (("G(48, 0.2)", 48, 0.2), ("isotropic", 0, 0.0)):
        t0 = time.time()
        x = deep_like_device(n, d, latent, seed=1, device=device, sigma=sigma)
        q = deep_like_device(nq, d, latent, seed=q_seed, device=device, sigma=sigma)
        graph, entry, tb = build_or_load_graph(svs, x, n, d, R, local_rank, "")
        d_entry = torch.tensor([entry], dtype=torch.int32, device=device)
        index = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), d_entry, n, d, np.float32, device=local_rank)
        gt, _ = index.flat_search_device(q, k)
        torch.cuda.synchronize()
        ids = torch.empty((nq, k), dtype=torch.int64, device=device)
        dd = torch.empty((nq, k), dtype=torch.float32, device=device)
        st = torch.zeros((nq, 6), dtype=torch.int32, device=device)

        def timed(with_stats=False):
            index.search_device(q, k, sp, d_ids64=ids, d_dists=dd, d_stats=st if with_stats else None)
            for _ in range(max(args.warmup, 12)):
                index.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
            torch.cuda.synchronize()
            ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(5, args.steps // 2))]
            for a_, b_ in ev_:
                a_.record()
                index.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
                b_.record()
            torch.cuda.synchronize()
            return float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev_]))

        ms_plain = timed()
        ids_plain = ids.clone()
        index.build_bound_codes(2)
        ms = timed(with_stats=True)
        kinfo = index.last_kernel_info()
        rb = int(index.memory()["record_size"])
        lines = designed_lines(st.to(torch.int64), (R + 15) // 16 * 16, d * 4, 2, d, k, rb)
        point = {"generator": f"{name}, unit norm, base seed 1, query seed {q_seed}", "qps": nq / (ms * 1e-3), "kernel_ms": ms, "recall_at_10": recall_at(gt, ids, k),
                 "record_bytes": rb, "kernel": "svsx::" + kinfo, "expansions_per_query": int(st[:, 0].sum()) / nq,
                 "distance_evals_per_query": int(st[:, 2].sum()) / nq, "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq,
                 "reject_share": float(st[:, 5].sum()) / max(float(st[:, 2].sum()), 1.0), "designed_lines_per_query": lines["total"] / nq,
                 "frac_designed": lines["total"] * LINE / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_traffic": None,
                 "plain_kernel_ms": ms_plain, "speedup_over_plain_kernel": ms_plain / ms, "ids_equal_plain_kernel": bool((ids == ids_plain).all()),
                 "build_seconds": tb, "seconds": None}
        try:  # ids against the reference's own code on a query sample (oracle/_ref; the C oracle where it is not available)
            from oracle.oracle import L2, Oracle, Ref

            ns = 256
            xh, gh = x.cpu().numpy(), device_graph_to_svs(graph, R)
            if Ref.available():
                ri, _ = Ref().search(L2, xh, gh, np.array([entry], np.uint32), q[:ns].cpu().numpy(), k, args.window, nthreads=host_cores())
                who = "oracle/_ref"
            else:
                ri, _ = Oracle().search(L2, xh, gh, np.array([entry], np.uint32), q[:ns].cpu().numpy(), k, args.window)
                who = "C oracle"
            point["reference_sample"] = {"queries": ns, "checker": who, "ids_identical": bool((ri.astype(np.int64) == ids[:ns].cpu().numpy()).all())}
            del xh, gh
        except Exception as e:  # noqa: BLE001
            point["reference_sample"] = {"error": f"{type(e).__name__}: {e}"}
        point["seconds"] = time.time() - t0
        out.append(point)
        del index, x, q, graph, gt
        torch.cuda.empty_cache()
    return out


def regression_uint8_point(args, svs, device, local_rank, k):
    """The data TYPE of the reference's own regression inputs (tools/benchmark_inputs/vamana/static-regression.toml: bigann uint8, L2, graph R = 128,
    L = 200, prune_to 124): 1 M x 128 sift-like non-negative uint8 rows, uint8 queries -- the VNNI pair L2VNNIOp<int16_t,32> (euclidean.h:265-313), exact
    int32 sums -- behind inline records over float(element) (round 6).  A row is ONE 128-byte line here, so a record saves one line per rejected
    neighbour, not three."""
    import torch

    from scalablevectorsearch_amd.build import device_graph_to_svs

    n, d, nq, R = 1_000_000, 128, args.nq, 128
    t0 = time.time()
    xf = deep_like_device(n, d, 32, seed=41, device=device, normalize=False).abs()
    qf = deep_like_device(nq, d, 32, seed=42, device=device, normalize=False).abs()
    gain = 218.0 / float(xf.max())  # sift's values reach 218
    x = torch.clamp(torch.round(xf * gain), 0, 255).to(torch.uint8)
    q = torch.clamp(torch.round(qf * gain), 0, 255).to(torch.uint8)
    del xf, qf
    graph, entry, tb = build_or_load_graph(svs, x, n, d, R, local_rank, "", data_dtype=np.uint8, build_window=200)
    index = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), torch.tensor([entry], dtype=torch.int32, device=device), n, d, np.uint8, device=local_rank)
    gt, _ = index.flat_search_device(q, k)
    torch.cuda.synchronize()
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(args.window), search_buffer_visited_set=args.visited)
    ids = torch.empty((nq, k), dtype=torch.int64, device=device)
    dd = torch.empty((nq, k), dtype=torch.float32, device=device)
    st = torch.zeros((nq, 6), dtype=torch.int32, device=device)

    def timed(with_stats=False):
        index.search_device(q, k, sp, d_ids64=ids, d_dists=dd, d_stats=st if with_stats else None)
        for _ in range(max(args.warmup, 12)):
            index.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
        torch.cuda.synchronize()
        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(5, args.steps // 2))]
        for a_, b_ in ev_:
            a_.record()
            index.search_device(q, k, sp, d_ids64=ids, d_dists=dd)
            b_.record()
        torch.cuda.synchronize()
        return float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev_]))

    ms_plain = timed(with_stats=True)
    ids_plain = ids.clone()
    # mode 3 (round 6): inline records only where the size rule expects them to pay.  This synthetic set is |G(32, 0.1)| -- taking absolute values
    # makes it full-rank -- and a 32-byte record (all a 128-byte row allows) rejects 0.07 % of its neighbours: forced records run at 0.80 x the plain
    # kernel's rate (profiles/r06_notes.md), so the rule declines them and the point reports the plain kernel, which is then what a host gets.
    chosen = index.build_bound_codes(3)
    ms = timed(with_stats=True) if chosen else ms_plain
    rb = int(index.memory()["record_size"])
    lines = designed_lines(st.to(torch.int64), 128, d, 2 if chosen else 0, d, k, rb or 32)
    point = {"bound_scheme_chosen_by_mode_3": "inline records" if chosen else "none (the size rule expects no saving on this spectrum)",
             "workload": f"sift-like 1 M x 128 uint8 rows, uint8 queries, L2, batch={nq}, search_window={args.window}, k=10, graph R=128 L=200 prune_to 124 alpha 1.2 (static-regression.toml's data type), 1xMI355X",
             "qps": nq / (ms * 1e-3), "kernel_ms": ms, "recall_at_10": recall_at(gt, ids, k), "kernel": "svsx::" + index.last_kernel_info(), "record_bytes": rb,
             "expansions_per_query": int(st[:, 0].sum()) / nq, "distance_evals_per_query": int(st[:, 2].sum()) / nq, "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq,
             "reject_share": float(st[:, 5].sum()) / max(float(st[:, 2].sum()), 1.0), "designed_lines_per_query": lines["total"] / nq,
             "frac_designed": lines["total"] * LINE / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "plain_kernel_ms": ms_plain, "speedup_over_plain_kernel": ms_plain / ms,
             "ids_equal_plain_kernel": bool((ids == ids_plain).all()), "build_seconds": tb}
    try:
        from oracle.oracle import L2, Oracle, Ref

        ns = 256
        xh, gh = x.cpu().numpy(), device_graph_to_svs(graph, R)
        chk = Ref() if Ref.available() else Oracle()
        kw = {"nthreads": host_cores()} if Ref.available() else {}
        ri, rd = chk.search(L2, xh, gh, np.array([entry], np.uint32), q[:ns].cpu().numpy(), k, args.window, **kw)
        point["reference_sample"] = {"queries": ns, "checker": "oracle/_ref" if Ref.available() else "C oracle",
                                     "ids_identical": bool((ri.astype(np.int64) == ids[:ns].cpu().numpy()).all()),
                                     "distance_bits_identical": bool((rd.view(np.uint32) == dd[:ns].cpu().numpy().view(np.uint32)).all())}
    except Exception as e:  # noqa: BLE001
        point["reference_sample"] = {"error": f"{type(e).__name__}: {e}"}
    point["seconds"] = time.time() - t0
    return point


def config4_point(args, svs, device, local_rank, k, t_run0):
    """BASELINE config 4 inside the default line: synthetic N x 768 scalar-quantized int8 rows (SQDataset-style global scale / bias, scalar.h:454-462),
    float32 queries, cosine, batch 4 096, the reference's regression graph parameters (R = 128, L = 200, prune_to 124, alpha 0.95).  N = --c4-n
    (10 M) when the GPU has the memory and the run the time (build ~340 s), else 2 M -- the size is in the point.  Smallest window with
    recall@10 >= 0.9 by binary search, timed like the headline; ids of a query sample checked against the CPU oracle."""
    import torch

    from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device, device_graph_to_svs

    d4, nq4, R4 = 768, 4096, 128
    total_hbm = torch.cuda.get_device_properties(device).total_memory
    free_hbm = torch.cuda.mem_get_info(device)[0]
    n4 = args.c4_n
    left = args.budget_s - (time.time() - t_run0)
    need = lambda n_: n_ * (d4 + 129 * 4 + 128 * 132 + 2 * d4 + 8) * 1.08  # codes, graph, 128-byte inline records, float16 shadow, row norms
    why = None
    if n4 > 2_000_000 and (need(n4) > free_hbm or left < 60 + 40e-6 * n4 * 1.3):
        why = f"{n4} vectors need {need(n4) / 1e9:.0f} GB and ~{40e-6 * n4:.0f} s; {free_hbm / 1e9:.0f} GB free and {left:.0f} s left"
        n4 = 2_000_000
    t0 = time.time()
    fn = svs.DistanceCosineSimilarity()
    mn, mx = None, None
    x4 = torch.empty((n4, d4), dtype=torch.int8, device=device)
    # two passes over the generator keep the float32 rows out of memory: global min / max first (scalar.h:454-462), then the codes
    for lo in range(0, n4, 1 << 20):
        xf = deep_like_device(min(1 << 20, n4 - lo), d4, 64, seed=1, device=device, normalize=False, row0=lo)
        mn = float(xf.min()) if mn is None else min(mn, float(xf.min()))
        mx = float(xf.max()) if mx is None else max(mx, float(xf.max()))
    scale = (mx - mn) / 255.0
    bias = mn + 128.0 * scale
    for lo in range(0, n4, 1 << 20):
        xf = deep_like_device(min(1 << 20, n4 - lo), d4, 64, seed=1, device=device, normalize=False, row0=lo)
        x4[lo : lo + xf.shape[0]] = torch.clamp(torch.round((xf - bias) / scale), -128, 127).to(torch.int8)
    del xf
    q4 = deep_like_device(nq4, d4, 64, seed=2, device=device, normalize=False)
    sq = (scale, bias)
    t_gen = time.time() - t0
    t0 = time.time()
    g4, e4 = build_vamana_device(x4, n4, d4, fn, VamanaBuildParameters(alpha=0.95, graph_max_degree=R4, window_size=200, prune_to=R4 - 4), data_dtype=np.int8, device=local_rank, sq=sq)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    i4 = svs.VamanaIndex.from_device(g4, x4, fn, torch.tensor([e4], dtype=torch.int32, device=device), n4, d4, np.int8, device=local_rank, sq=sq)
    gt4, _ = i4.flat_search_device(q4, k)  # exhaustive ground truth (MFMA shortlist + exact re-rank) before the inline records take their memory
    torch.cuda.synchronize()
    i4.build_bound_codes(2)
    ids4 = torch.empty((nq4, k), dtype=torch.int64, device=device)
    dd4 = torch.empty((nq4, k), dtype=torch.float32, device=device)
    st4 = torch.zeros((nq4, 6), dtype=torch.int32, device=device)
    lo_w, hi_w = 64, 1024
    while lo_w < hi_w:  # calibrate.h:286-300
        mid = (lo_w + hi_w) // 2
        i4.search_device(q4, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(mid), 1), d_ids64=ids4, d_dists=dd4)
        torch.cuda.synchronize()
        if recall_at(gt4, ids4, k) >= 0.9:
            hi_w = mid
        else:
            lo_w = mid + 1
    sp4 = svs.VamanaSearchParameters(svs.SearchBufferConfig(lo_w), 1)
    i4.search_device(q4, k, sp4, d_ids64=ids4, d_dists=dd4, d_stats=st4)
    for _ in range(max(args.warmup, 12)):  # informational points warm up on their own: at least 12 launches whatever --warmup says
        i4.search_device(q4, k, sp4, d_ids64=ids4, d_dists=dd4)
    torch.cuda.synchronize()
    ev4 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, args.steps // 2))]
    for a_, b_ in ev4:
        a_.record()
        i4.search_device(q4, k, sp4, d_ids64=ids4, d_dists=dd4)
        b_.record()
    torch.cuda.synchronize()
    ms4 = float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev4]))
    k4 = i4.last_kernel_info()
    rb4 = {"codes=2>": 32, "codes=3>": 64, "codes=4>": 128, "codes=5>": 28}.get(k4[-8:], 128)
    l4 = designed_lines(st4.to(torch.int64), 128, d4, 2, d4, k, rb4)
    point = {"workload": f"synthetic {n4} x 768 scalar-quantized int8 (global scale / bias), float32 queries, cosine, batch={nq4}, k=10, graph R=128 L=200 prune_to 124 alpha 0.95, 1xMI355X",
             "vectors": n4, "size_note": why or "BASELINE config 4's size", "window_for_recall_0.9": lo_w, "recall_at_10": recall_at(gt4, ids4, k), "qps": nq4 / (ms4 * 1e-3), "kernel_ms": ms4,
             "kernel": "svsx::" + k4, "expansions_per_query": int(st4[:, 0].sum()) / nq4, "distance_evals_per_query": int(st4[:, 2].sum()) / nq4,
             "bound_code_rejects_per_query": int(st4[:, 5].sum()) / nq4, "designed_lines_per_query": l4["total"] / nq4,
             "frac_designed": l4["total"] * LINE / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, "generate_seconds": t_gen, "build_seconds": t_build,
             "bound": "vector ALU (decompress + dot product per element, scalar.h:155-195); row norms precomputed at index creation since round 4"}
    try:  # the same searches on the host by the CPU oracle for a query sample: ids identical?
        from oracle.oracle import COS, Oracle

        ns = 256  # (round 5: 8 queries, ids only)
        host_need = 4 * (n4 * d4 + n4 * (R4 + 1) * 4)  # host copies of codes and graph, plus the temporaries of the layout conversion
        try:
            lim = open("/sys/fs/cgroup/memory.max").read().strip()
            host_have = int(lim) if lim != "max" else 1 << 62
        except OSError:
            host_have = 1 << 62
        import psutil

        host_have = min(host_have, psutil.virtual_memory().available)
        if host_need > 0.5 * host_have:
            raise MemoryError(f"host copies need {host_need / 1e9:.0f} GB, {host_have / 1e9:.0f} GB available: sample check skipped")
        o = Oracle()
        xh = x4.cpu().numpy()
        gh = device_graph_to_svs(g4, R4)
        oi, od = o.search(COS, xh, gh, np.array([e4], np.uint32), q4[:ns].cpu().numpy(), k, lo_w, sq=sq)
        point["oracle_sample"] = {"queries": ns, "window": lo_w, "ids_identical": bool((oi.astype(np.int64) == ids4[:ns].cpu().numpy()).all()),
                                  "distance_bits_identical": bool((od.view(np.uint32) == dd4[:ns].cpu().numpy().view(np.uint32)).all())}
        del xh, gh
    except Exception as e:  # noqa: BLE001
        point["oracle_sample"] = {"error": f"{type(e).__name__}: {e}"}
    del i4, gt4
    torch.cuda.empty_cache()
    try:  # the SAME rows taken as raw int8 vectors, int8 queries, cosine (K6: cosine.h:261-326) -- the exact integer path
        left = args.budget_s - (time.time() - t_run0)
        if left < 150:
            raise TimeoutError(f"{left:.0f} s of the run's budget left")
        point["raw_int8"] = config4b_point(args, svs, device, local_rank, k, x4, g4, e4, q4, sq, n4)
    except Exception as e:  # noqa: BLE001  (informational: never fail the line over it)
        point["raw_int8"] = {"error": f"{type(e).__name__}: {e}"}
    del x4, g4, ids4, dd4, st4
    return point


def config4b_point(args, svs, device, local_rank, k, x4, g4, e4, q4f, sq, n4):
    """raw int8 rows x int8 queries, cosine, batch 4 096 -- `CosineSimilarity` over the VNNI loops (cosine.h:261-326):
    float(sum ab) / (||a|| sqrt(float(sum bb))), exact int32 sums.  Rows = config 4's codes read as plain int8 vectors, queries = the float
    queries quantized with the same scale / bias; the graph is config 4's (built under the scalar-quantized cosine of the same codes: the two
    metrics differ by the bias term only, |bias| / scale is reported).  Inline records over float(row element), round 6."""
    import torch

    d4, nq4, R4 = 768, q4f.shape[0], 128
    scale, bias = sq
    fn = svs.DistanceCosineSimilarity()
    q8 = torch.clamp(torch.round((q4f - bias) / scale), -128, 127).to(torch.int8)
    ib = svs.VamanaIndex.from_device(g4, x4, fn, torch.tensor([e4], dtype=torch.int32, device=device), n4, d4, np.int8, device=local_rank)
    gtb, _ = ib.flat_search_device(q8, k)
    torch.cuda.synchronize()
    t0 = time.time()
    ib.build_bound_codes(2)
    torch.cuda.synchronize()
    t_rec = time.time() - t0
    mem = ib.memory()
    ids = torch.empty((nq4, k), dtype=torch.int64, device=device)
    dd = torch.empty((nq4, k), dtype=torch.float32, device=device)
    st = torch.zeros((nq4, 6), dtype=torch.int32, device=device)
    lo_w, hi_w = 32, 1024
    while lo_w < hi_w:  # calibrate.h:286-300
        mid = (lo_w + hi_w) // 2
        ib.search_device(q8, k, svs.VamanaSearchParameters(svs.SearchBufferConfig(mid), 1), d_ids64=ids, d_dists=dd)
        torch.cuda.synchronize()
        if recall_at(gtb, ids, k) >= 0.9:
            hi_w = mid
        else:
            lo_w = mid + 1
    spb = svs.VamanaSearchParameters(svs.SearchBufferConfig(lo_w), 1)
    ib.search_device(q8, k, spb, d_ids64=ids, d_dists=dd, d_stats=st)
    for _ in range(max(args.warmup, 12)):
        ib.search_device(q8, k, spb, d_ids64=ids, d_dists=dd)
    torch.cuda.synchronize()
    evb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, args.steps // 2))]
    for a_, b_ in evb:
        a_.record()
        ib.search_device(q8, k, spb, d_ids64=ids, d_dists=dd)
        b_.record()
    torch.cuda.synchronize()
    msb = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evb]))
    kb = ib.last_kernel_info()
    rbb = int(mem["record_size"]) or 32
    lb = designed_lines(st.to(torch.int64), 128, d4, 2, d4, k, rbb)
    wkey = f"c4b:n={n4}:nq={nq4}:window={lo_w}:R=128:visited=1:codes=2"
    tb, tb_src, _ = measured_traffic(wkey)
    # the same launch on the plain kernel (no records: every neighbour's 768-byte row is read, the reference's own traffic)
    ib.build_bound_codes(0)
    for _ in range(3):
        ib.search_device(q8, k, spb, d_ids64=ids, d_dists=dd)
    torch.cuda.synchronize()
    evp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    ids_p = torch.empty_like(ids)
    for a_, b_ in evp:
        a_.record()
        ib.search_device(q8, k, spb, d_ids64=ids_p, d_dists=dd)
        b_.record()
    torch.cuda.synchronize()
    msp = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evp]))
    out = {"workload": f"synthetic {n4} x 768 raw int8 rows, int8 queries, cosine (VNNI pair, cosine.h:261-326), batch={nq4}, k=10, graph: config 4's (R=128), 1xMI355X",
           "vectors": n4, "window_for_recall_0.9": lo_w, "recall_at_10": recall_at(gtb, ids, k), "qps": nq4 / (msb * 1e-3), "kernel_ms": msb, "kernel": "svsx::" + kb,
           "record_bytes": rbb, "inline_record_GB": mem["inline_record_bytes"] / 1e9, "record_build_seconds": t_rec,
           "expansions_per_query": int(st[:, 0].sum()) / nq4, "distance_evals_per_query": int(st[:, 2].sum()) / nq4,
           "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq4, "reject_share": float(st[:, 5].sum()) / max(float(st[:, 2].sum()), 1.0),
           "designed_lines_per_query": lb["total"] / nq4, "frac_designed": lb["total"] * LINE / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "traffic": tb, "traffic_source": tb_src, "frac_traffic": (tb / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS) if tb else None,
           "plain_kernel_ms": msp, "speedup_over_plain_kernel": msp / msb, "ids_equal_plain_kernel": bool((ids_p == ids).all()),
           "bias_over_scale": abs(bias) / scale}
    del ib, gtb
    return out


def main():
    t_run0 = time.time()
    _claim_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps in front of the timed region (default 20: the first ~10 launches after an idle period run 10-15 %% slower -- clock ramp, profiles/r05_notes.md section 6)")
    ap.add_argument("--config", default="c2", choices=["c2", "c5"], help="c2: BASELINE config 2 (default); c5: fixed-shard-count sharded index (SURVEY 8e)")
    ap.add_argument("--n", type=int, default=0, help="vectors per shard (default: 1 000 000 for c2, 12 500 000 for c5)")
    ap.add_argument("--shards", type=int, default=8, help="c5: total number of shards (fixed as the number of GPUs varies)")
    ap.add_argument("--nq", type=int, default=10_000)
    ap.add_argument("--window", type=int, default=128)
    ap.add_argument("--degree", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dist", action="store_true", help="exercise the multi-GPU code path (RCCL all-gather + merge) even at world size 1")
    ap.add_argument("--visited", type=int, default=1, help="visited filter: 0 off, 1 default size, >=64 explicit LDS entries (result-neutral)")
    ap.add_argument("--bound-codes", type=int, default=32, help="result-neutral lower-bound rejection: 32 graph-inline records of 32 bytes (default: the headline of rounds 2-6), "
                    "2 inline records of the size the library's rule picks (28 bytes on this set: other_configs.c2_records_rule), 28 / 64 / 128 that size, 1 128-B code rows, 0 off")
    ap.add_argument("--graph-cache", default="", help="load/save the built graph(s) here (lets a profiler run skip the build)")
    ap.add_argument("--c5-n", type=int, default=0, help="vectors per shard of the sharded_search point (0 = BASELINE config 5's 12 500 000 at every N; -1 = skip the point)")
    ap.add_argument("--sharded-timeout", type=float, default=0.0, help="watchdog of the sharded_search point in seconds (0 = scaled with shards per GPU x shard size); when it fires the line is printed with sharded_search.failed and the run exits 3")
    ap.add_argument("--c4-n", type=int, default=10_000_000, help="vectors of the config-4 point (10 M = BASELINE config 4; falls back to 2 M when time or memory is short; 0 = skip)")
    ap.add_argument("--budget-s", type=float, default=1500.0, help="wall-clock budget of the whole run: the long extras (config 4 at 10 M) are sized against what is left")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational points (sharded_c5, other configs, 4x batch)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import scalablevectorsearch_amd as svs

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE")
    if os.environ.get("SVSX_BENCH_SINGLE_DEVICE"):  # rehearsal of the N > 1 control flow on a one-GPU box (with SVSX_BENCH_BACKEND=gloo)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    c5 = args.config == "c5"
    multi = args.force_dist or c5   # the default workload has no exchange step at any N (replicas)
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = os.environ.get("SVSX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    use_dist = dist.is_initialized()

    d, nq, k, R = 96, args.nq, 10, args.degree
    n = args.n or (12_500_000 if c5 else 1_000_000)
    total_shards = args.shards if c5 else 1
    if c5 and total_shards % world:
        raise SystemExit(f"--shards {total_shards} must be a multiple of the number of GPUs {world}")
    local_shards = total_shards // world if c5 else 1
    first_shard = rank * local_shards if c5 else 0   # default workload: every rank holds shard 0 = the whole index (replicas)

    # ---- shards of a (total_shards x n) database: independent seeds per shard, global ids = local + shard * n -----------------
    # c5: the same batch on every rank (every shard sees every query); replicas: each rank answers its own batch
    q = deep_like_device(nq, d, 24, seed=2 if (c5 or rank == 0) else 2 + 7919 * rank, device=device)
    shards = []
    t_gen = t_build = 0.0
    total_hbm = torch.cuda.get_device_properties(device).total_memory
    gstride = (R + 15) // 16 * 16
    mode = args.bound_codes
    if mode >= 2 and local_shards * n * (d * 4 + gstride * 4 + gstride * 36 + 64) > 0.75 * total_hbm:
        mode = 1  # the graph-inline records of every local shard do not fit next to the vectors: 128-byte code rows instead
        log(f"[bench] rank {rank}: {local_shards} shards x {n} vectors: graph-inline records would not fit, using 128-byte code rows")
    for s in range(local_shards):
        sid = first_shard + s
        t0 = time.time()
        x = deep_like_device(n, d, 24, seed=1 + 1000 * sid, device=device)
        torch.cuda.synchronize()
        t_gen += time.time() - t0
        cache = f"{args.graph_cache}.s{sid}.n{n}" if args.graph_cache else ""
        graph, entry, tb = build_or_load_graph(svs, x, n, d, R, local_rank, cache)
        t_build += tb
        d_entry = torch.tensor([entry], dtype=torch.int32, device=device)
        index = svs.VamanaIndex.from_device(graph, x, svs.DistanceL2(), d_entry, n, d, np.float32, device=local_rank, id_offset=sid * n)
        if mode:
            index.build_bound_codes(mode)
        shards.append(dict(x=x, graph=graph, entry=entry, d_entry=d_entry, index=index, sid=sid))
    deg_mean = float((shards[0]["graph"][:, :R] >= 0).sum(1).float().mean())
    if rank == 0:
        log(f"[bench] data {t_gen:.1f}s, graph build {t_build:.1f}s ({local_shards} shard(s)/rank, R={R}, mean degree {deg_mean:.1f})")
    sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(args.window), search_buffer_visited_set=args.visited)

    import ctypes as C

    from scalablevectorsearch_amd import _lib
    from scalablevectorsearch_amd.sharded import packed_views, slot_words

    m = nq * k
    sw = slot_words(nq, k)
    stats = torch.zeros((local_shards, nq, 6), dtype=torch.int32, device=device)
    # per local shard one packed buffer [nq*k i64 ids | nq*k f32 distances]; the search kernel writes straight into it
    shard_buf = torch.zeros((local_shards, sw), dtype=torch.int32, device=device)
    sh_d, sh_i = packed_views(shard_buf, local_shards, nq, k)
    loc_buf = torch.zeros(sw, dtype=torch.int32, device=device) if local_shards > 1 else shard_buf[0]
    loc_d, loc_i = (v[0] for v in packed_views(loc_buf, 1, nq, k))
    if multi:
        all_buf = torch.empty(world * sw, dtype=torch.int32, device=device)
        all_d, all_i = packed_views(all_buf, world, nq, k)
        out_d = torch.empty((nq, k), dtype=torch.float32, device=device)
        out_i = torch.empty((nq, k), dtype=torch.int64, device=device)
    L = _lib.lib()

    def merge(n_lists, src_d, src_i, dst_d, dst_i):
        _lib.check(L.svsx_merge_topk_strided_device(0, n_lists, nq, k, src_d.data_ptr(), src_d.stride(0) if n_lists > 1 else 0, src_i.data_ptr(),
                                                    src_i.stride(0) if n_lists > 1 else 0, dst_d.data_ptr(), dst_i.data_ptr(),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def step(with_stats=False, events=None):
        if events is not None:
            events[0].record()
        for s, sh in enumerate(shards):
            sh["index"].search_device(q, k, sp, d_ids64=sh_i[s], d_dists=sh_d[s], d_stats=stats[s] if with_stats else None)
        if events is not None:
            events[1].record()
        if local_shards > 1:
            merge(local_shards, sh_d, sh_i, loc_d, loc_i)  # shards/N-way local merge feeding the packed all-gather
        if multi:
            if use_dist and world > 1:
                dist.all_gather_into_tensor(all_buf, loc_buf)
                merge(world, all_d, all_i, out_d, out_i)
            else:
                merge(1, loc_d[None], loc_i[None], out_d, out_i)

    # ---- correctness of what is being timed: recall against exact ground truth ---------------------------------
    step(with_stats=True)
    torch.cuda.synchronize()
    # exact ground truth by this repo's exhaustive search (MFMA f16 shortlist + exact re-rank, svsx_flat_search_device), per shard
    gt_d_parts, gt_i_parts = [], []
    gt_recomputed = torch.zeros(1, dtype=torch.int32, device=device)  # queries the exactness proof sent to the exhaustive float32 pass
    gt_check = None
    for sh in shards:
        flat = svs.VamanaIndex.from_device(sh["graph"], sh["x"], svs.DistanceL2(), sh["d_entry"], n, d, np.float32, device=local_rank)
        gi, _ = flat.flat_search_device(q, k, recomputed=gt_recomputed)
        torch.cuda.synchronize()
        del flat
        if gt_check is None:  # in-line check of the ground truth itself: float64 brute force on a sample of the first shard's queries
            ns = min(256, nq)
            best = torch.full((ns, k), float("inf"), dtype=torch.float64, device=device)
            besti = torch.zeros((ns, k), dtype=torch.int64, device=device)
            qs = q[:ns].double()
            for lo in range(0, n, 1 << 18):
                xs = sh["x"][lo : lo + (1 << 18), :d].double()
                dd = (qs * qs).sum(1, keepdim=True) + (xs * xs).sum(1)[None, :] - 2.0 * (qs @ xs.T)
                cd_, ci_ = torch.topk(dd, k, dim=1, largest=False)
                allv, alli = torch.cat([best, cd_], 1), torch.cat([besti, ci_ + lo], 1)
                sel = torch.topk(allv, k, dim=1, largest=False)
                best, besti = sel.values, torch.gather(alli, 1, sel.indices)
            same = (torch.sort(besti, 1).values == torch.sort(gi[:ns], 1).values).all(1)
            gt_check = {"sample_queries": ns, "id_sets_equal_to_float64_brute_force": float(same.float().mean().item()),
                        "note": "a float64 ranking may order two neighbours whose float32 distances tie differently; a set differs only if such a pair straddles rank k"}
        gt_d_parts.append(((q[:, None, :] - sh["x"][gi]) ** 2).sum(-1))
        gt_i_parts.append(gi + sh["sid"] * n)
    cd = torch.cat(gt_d_parts, 1)
    ci = torch.cat(gt_i_parts, 1)
    if use_dist and world > 1 and c5:
        g_d = torch.empty((world,) + tuple(cd.shape), device=device)
        g_i = torch.empty((world,) + tuple(ci.shape), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(g_d.view(-1), cd.contiguous().view(-1))
        dist.all_gather_into_tensor(g_i.view(-1), ci.contiguous().view(-1))
        cd = g_d.permute(1, 0, 2).reshape(nq, -1)
        ci = g_i.permute(1, 0, 2).reshape(nq, -1)
    gt = torch.gather(ci, 1, torch.topk(cd, k, dim=1, largest=False).indices)
    answer_i = out_i if multi else loc_i
    recall = recall_at(gt, answer_i, k)
    st = stats.to(torch.int64).sum(0)  # per query, summed over this rank's shards
    exp_total, slots_total = int(st[:, 0].sum()), int(st[:, 1].sum())
    # SURVEY 8(d) algorithmic bytes of the REFERENCE traversal (visited set off, whole rows): reported, never used as the roofline fraction
    ref_alg_bytes = 4 * (exp_total + slots_total) + slots_total * d * 4 + local_shards * nq * (d * 4 + d * 4 + k * 12)
    kinfo = shards[0]["index"].last_kernel_info()  # ...codes=2/3/4> = 32- / 64- / 128-byte inline records
    record_bytes = {"codes=2>": 32, "codes=3>": 64, "codes=4>": 128, "codes=5>": 28}.get(kinfo[-8:], 32)
    lines = designed_lines(stats.to(torch.int64).reshape(-1, 6), gstride, d * 4, min(mode, 2), d, k, record_bytes)

    # ---- timed region: W warm-up steps, then exactly K steps between barrier + synchronize ----------------------------------------
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_step = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    # Clock ramp: the first launches after an idle period run 10-15 % slow (profiles/r05_notes.md section 6), and the driver's command
    # (--steps 20 --warmup 5) would end its warm-up inside that ramp.  So, IN ADDITION to --warmup and whatever its value: untimed steps until three
    # consecutive ones agree within 1 % (at least PREWARM_MIN, at most PREWARM_MAX).  Reported as `prewarm_steps`; nothing of it is timed.
    pre_adaptive = not (multi and use_dist and world > 1)  # a step with a collective in it must run the same number of times on every rank
    pre_ev = [torch.cuda.Event(enable_timing=True) for _ in range(PREWARM_MAX + 1)]
    pre_ms = []
    pre_ev[0].record()
    for i in range(PREWARM_MAX if pre_adaptive else PREWARM_MIN):
        step()
        pre_ev[i + 1].record()
        if pre_adaptive and i + 1 >= PREWARM_MIN and (i + 1) % 4 == 0:  # look at the clock only every few steps: the check itself drains the queue
            pre_ev[i + 1].synchronize()
            pre_ms = [pre_ev[j].elapsed_time(pre_ev[j + 1]) for j in range(i + 1)]
            last = pre_ms[-3:]
            if max(last) <= 1.01 * min(last):
                break
    prewarm_steps = len(pre_ms) if pre_ms else (PREWARM_MAX if pre_adaptive else PREWARM_MIN)
    if use_dist:  # every rank leaves the ramp before anybody's timed region starts (ranks may have stopped at different counts)
        dist.barrier()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev_step[0].record()
    for i in range(args.steps):
        step(events=ev[i])
        ev_step[i + 1].record()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev])) / local_shards  # per search_kernel launch
    step_ms = [ev_step[i].elapsed_time(ev_step[i + 1]) for i in range(args.steps)]

    # ---- the same batch through the host-pointer entry point, right behind the timed region (before any extra touches the process) ---------
    host_pointer = None
    if rank == 0 and world == 1 and not c5 and not args.force_dist:
        # SURVEY 8d wording: host queries in, host results out; the call the reference's harness times
        # (benchmark/include/svs-benchmark/search.h:209-219): svsx_search through the C ABI, pageable numpy arrays on both sides, 1 warm-up + 5 timed
        qh = q.cpu().numpy()
        h_ids = np.empty((nq, k), np.uint64)
        h_d = np.empty((nq, k), np.float32)
        spc = sp._c()
        aux0 = _lib.SearchAux()
        index0 = shards[0]["index"]

        def host_call():
            _lib.check(L.svsx_search(index0._h, qh.ctypes.data, 0, nq, k, C.byref(spc), h_ids.ctypes.data, None, h_d.ctypes.data, C.byref(aux0)))

        def timed_host(fn):
            """The reference's protocol is 1 warm-up + 5 timed calls (search.h:209-219).  Like the device-resident launches, the first host calls after
            the idle gap in front of them run slow (the kernels inside them end at 1 308, 1 244, 1 139, 1 114, 1 091, 1 089 us of the call on one lease:
            profiles/r06_notes.md section 10), so the same rule as for `value` applies: untimed calls until three consecutive ones agree within 1 %
            (at least 8, at most 16), THEN the 5 timed ones.  The untimed calls' times are reported too (`warmup_calls_ms`): the first five of them are
            what the reference's own 1 + 5 protocol would have recorded."""
            warm = []
            for i in range(16):
                t0h = time.perf_counter()
                fn()
                warm.append(1e3 * (time.perf_counter() - t0h))
                if i >= 7 and max(warm[-3:]) <= 1.01 * min(warm[-3:]):
                    break
            timed = []
            for _ in range(5):
                t0h = time.perf_counter()
                fn()
                timed.append(1e3 * (time.perf_counter() - t0h))
            out = stat3(timed, nq)
            out["warmup_calls_ms"] = [round(v, 4) for v in warm]
            out["protocol"] = "untimed calls until three agree within 1 % (8..16), then 5 timed: min / median / max"
            out["reference_protocol_1_plus_5_median_ms"] = float(np.median((warm + timed)[1:6]))  # calls 2..6 of this sequence
            return out, timed

        host_pointer, host_ms = timed_host(host_call)
        host_pointer["ids_equal_device_path"] = bool((h_ids.astype(np.int64) == loc_i.cpu().numpy()).all())
        dev_med = float(np.median([a.elapsed_time(b) for a, b in ev]))
        host_pointer["over_device_resident"] = dev_med / host_pointer["median_ms"]
        try:  # the same call on PAGE-LOCKED caller arrays (a host that pins its query / result buffers): the copies become truly asynchronous
            qp = torch.empty((nq, d), dtype=torch.float32).pin_memory()
            qp.copy_(q.cpu())
            ip_, dp_ = torch.empty((nq, k), dtype=torch.int64).pin_memory(), torch.empty((nq, k), dtype=torch.float32).pin_memory()

            def pinned_call():
                _lib.check(L.svsx_search(index0._h, qp.data_ptr(), 0, nq, k, C.byref(spc), ip_.data_ptr(), None, dp_.data_ptr(), C.byref(aux0)))

            pin_stats, pin_ms = timed_host(pinned_call)
            host_pointer["page_locked_caller_arrays"] = dict(pin_stats, ids_equal_device_path=bool((ip_.numpy() == loc_i.cpu().numpy()).all()),
                                                             over_device_resident=dev_med / float(np.median(pin_ms)))
            del qp, ip_, dp_
        except Exception as e:  # noqa: BLE001 (informational)
            host_pointer["page_locked_caller_arrays"] = {"error": f"{type(e).__name__}: {e}"}
        host_pointer["how"] = ("device staging and two streams kept with the index, copies straight from / to the caller's pageable arrays; two launches: the first "
                               "4 096 queries start as soon as their bytes have crossed PCIe, the rest is copied while they run (DESIGN.md section 1, "
                               "profiles/r03_notes.md section 5, profiles/r04_notes.md section 6)")


    # practical roof: device-to-device copy bandwidth on this box (read + write bytes / time)
    copy_gbs = float("nan")
    if rank == 0:
        cp_src = torch.empty(1 << 28, dtype=torch.float32, device=device)  # 1 GiB
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
    merged_qps = nq * args.steps / elapsed
    if c5:
        value, scaling, counts = merged_qps, "strong", "merged answers (queries) per second over the whole fixed-size sharded index"
    else:
        value, scaling = world * nq * args.steps / elapsed, "weak"
        counts = "queries answered per second" if world == 1 else f"queries answered per second by {world} replicas (each GPU holds the whole index and answers its own {nq}-query batch; no exchange step)"
    designed_bytes = lines["total"] * LINE / local_shards  # per search_kernel launch
    achieved = designed_bytes / (kern_ms * 1e-3) / 1e9
    wkey = f"{args.config}:n={n}:nq={nq}:window={args.window}:R={R}:visited={args.visited}:codes={mode}"
    traffic, traffic_src, pmc = measured_traffic(wkey) if world == 1 else (None, None, None)
    if c5:
        workload = f"synthetic {total_shards} x {n} x 96 float32 L2 sharded index ({local_shards} shard(s) per GPU), batch={nq}, search_window={args.window} per shard, k=10, {world}xMI355X"
    else:
        workload = f"deep-1M-like d=96 float32 L2, batch={nq}, search_window={args.window}, k=10, 1xMI355X" + (f" x {world} replicas" if world > 1 else "")
    result = {
        "metric": "QPS at recall@10>=0.9 (batch=10k); achieved HBM GB/s vs roofline",
        "value": value,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "prewarm_steps": prewarm_steps,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": workload,
            "value_counts": counts,
            "inputs": "value: queries and results resident in HBM when the timed region starts (svsx_search_device); the same batch through the host-pointer "
                      "entry point svsx_search (SURVEY 8d's wording: host queries in, host results out) is config.host_pointer, 1 warm-up + 5 timed",
            "vectors_per_shard": n,
            "shards_total": total_shards,
            "shards_per_gpu": local_shards,
            "graph": f"Vamana R={R} L={2*R} alpha=1.2 (GPU-built, mean degree {deg_mean:.1f})",
            "generator": "G(24, 0.1) unit norm, own counter-based RNG (splitmix64 hash + Box-Muller), base seed 1 + 1000*shard, query seed 2",
            "recall_at_10": recall,
            "ground_truth": {"how": "svsx_flat_search_device per shard: float16 MFMA shortlist, exact float32 re-rank, per-query proof against the float16 rounding "
                                    "bounds, exhaustive float32 pass for unproven queries (exact by construction; tests/test_gpu_flat_exact.py)",
                             "queries_recomputed_exhaustively": int(gt_recomputed.item()), **(gt_check or {})},
            "merged_answer_qps": merged_qps if c5 else value,
            "step_ms_min_median_max": [float(np.min(step_ms)), float(np.median(step_ms)), float(np.max(step_ms))],
            "step_ms_each": [round(float(v), 4) for v in step_ms],
            "expansions_per_query": exp_total / nq,
            "neighbor_slots_per_query": slots_total / nq,
            "build_seconds": t_build,
            "visited_filter": args.visited,
            "distance_evals_per_query": int(st[:, 2].sum()) / nq,
            "partial_distance_rejects_per_query": int(st[:, 4].sum()) / nq,
            "accepted_inserts_per_query": int(st[:, 3].sum()) / nq,
            "bound_codes": mode,
            "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq,
        },
        "roofline": {
            "bound": "hbm",
            "achieved": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else achieved / HBM_PEAK_GBS,
            "frac_basis": ("fabric bytes per launch from the rocprofv3 PMC passes of these kernel sources (roofline.traffic) / this run's kernel time / 8 TB/s"
                           if traffic else "designed bytes (no PMC summary matches these kernel sources): see achieved_designed"),
            "achieved_designed": achieved,
            "frac_designed": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "traffic_rate": (traffic / (kern_ms * 1e-3) / 1e9) if traffic else None,
            "frac_traffic": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "measured_copy_GBs": copy_gbs,
            "frac_of_measured_copy": achieved / copy_gbs,
            "kernel": "svsx::" + kinfo,
            "kernel_ms": kern_ms,
            "kernel_src_sha": kernel_src_sha(),
            "designed_bytes_per_launch": designed_bytes,
            "designed_lines_per_query": {kk_: v / (nq * local_shards) for kk_, v in lines.items()},
            "line_rate_Glines_per_s": lines["total"] / local_shards / (kern_ms * 1e-3) / 1e9,
            "random_line_roof_Glines_per_s": 49.0,
            "reference_algorithmic_bytes_per_launch": ref_alg_bytes / local_shards,
            "reference_algorithmic_GBs": ref_alg_bytes / local_shards / (kern_ms * 1e-3) / 1e9,
            "note": "achieved / frac = measured fabric traffic per launch (PMC) / kernel time (HIP events of this run) -- what a third party can recompute; "
                    "achieved_designed / frac_designed = bytes the kernel is DESIGNED to move (128-byte lines counted from its own per-query statistics: "
                    "adjacency blocks, rows of the candidates that reach the exact path, query, results) / kernel time.  "
                    "reference_algorithmic_* = the SURVEY 8(d) figure for the reference traversal (visited set off, whole rows for every neighbour); "
                    "the kernel proves most of those reads unnecessary, so that figure exceeds what moves and is not a roofline fraction.  "
                    "traffic = fabric bytes from the committed PMC passes of these kernel sources (null when none matches).  "
                    "Random 128-byte gathers top out at ~49 G lines/s = 6.3 TB/s on this GPU (tools/ubench/gather_bench).",
        },
    }

    # ---- the sharded search of SURVEY 8(e), measured in the same run (every rank takes part).  The line above is complete without it, so a
    # collective that never returns costs the run its exit code, not its headline: the watchdog prints the line with
    # sharded_search.failed = true and leaves with status 3 (ADVICE r3 / VERDICT r4: no silent rc 0).
    sharded = None
    if not c5 and not args.no_extras and args.c5_n >= 0:
        import threading

        n5_plan, per_plan = sharded_shard_size(args, world, total_hbm), max(1, 8 // world)
        # builds dominate: ~29 s per 12.5 M-vector shard + generation, records / code rows, exact ground truth; comm set-up and the searches on top
        dog_s = args.sharded_timeout or (240.0 + 75.0 * per_plan * n5_plan / 12_500_000)

        def give_up():
            if rank == 0:
                result["config"]["sharded_c5"] = {"error": f"timed out after {dog_s:.0f} s", "vectors_per_shard": n5_plan, "shards_per_gpu": per_plan}
                result["sharded_search"] = {"metric": "merged answers per second over a FIXED set of 8 database shards (SURVEY 8e; strong scaling in N)", "value": None,
                                            "unit": "queries/s", "n_gpus": world, "vectors_per_shard": n5_plan, "failed": True,
                                            "error": f"the sharded point did not return within {dog_s:.0f} s (watchdog); every other field of this line was measured before it started"}
                result["config"]["note"] = "emergency line: printed by the watchdog of the sharded point; informational extras (other configs, CPU leg) were not reached"
                _emit(json.dumps(result))
            os._exit(3)

        dog = threading.Timer(dog_s, give_up)
        dog.daemon = True
        dog.start()
        try:
            sh0 = dict(shards[0], n=n)
            sharded = sharded_c5_point(args, svs, q if rank == 0 else deep_like_device(nq, d, 24, seed=2, device=device), k, R, d, rank, world, local_rank, device,
                                       sh0 if rank == 0 else None, max(3, args.steps // 2), args.warmup)
        except Exception as e:  # recorded with failed = true; at N > 1 the run then exits 3 after the line
            sharded = {"error": f"{type(e).__name__}: {e}"}
            log(f"[bench] rank {rank}: sharded_c5 point failed: {e}")
        dog.cancel()
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        if world > 1 and isinstance(sharded, dict) and "error" in sharded:
            sys.exit(3)
        return

    if sharded is not None:
        result["config"]["sharded_c5"] = sharded
    single = world == 1 and not c5 and not args.force_dist
    ids = loc_i
    dists = loc_d
    index = shards[0]["index"]
    if single:
        if host_pointer is not None:
            result["config"]["host_pointer"] = host_pointer
            result["config"]["host_pointer_qps"] = host_pointer["qps_at_median"]
        result["config"]["device_resident"] = stat3([a.elapsed_time(b) for a, b in ev], nq)
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
        mstats = torch.zeros((nq, 6), dtype=torch.int32, device=device)
        index.search_device(q, k, sp_m, d_ids64=ids, d_dists=dists, d_stats=mstats)
        for _ in range(max(args.warmup, 12)):  # informational points warm up on their own: at least 12 launches whatever --warmup says
            index.search_device(q, k, sp_m, d_ids64=ids, d_dists=dists)
        torch.cuda.synchronize()
        mev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        t0 = time.perf_counter()
        for a_, b_ in mev:
            a_.record()
            index.search_device(q, k, sp_m, d_ids64=ids, d_dists=dists)
            b_.record()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        mk_ms = float(np.mean([a_.elapsed_time(b_) for a_, b_ in mev]))
        mlines = designed_lines(mstats.to(torch.int64), gstride, d * 4, min(mode, 2), d, k, record_bytes)
        mt, mt_src, _ = measured_traffic(f"{args.config}:n={n}:nq={nq}:window={lo_w}:R={R}:visited={args.visited}:codes={mode}")
        matched = {"window": lo_w, "recall_at_10": recall_at(gt, ids, k), "gpu_qps": nq * args.steps / dt, "kernel_ms": mk_ms,
                   "designed_lines_per_query": mlines["total"] / nq, "achieved_GBs": mlines["total"] * LINE / (mk_ms * 1e-3) / 1e9,
                   "frac": mlines["total"] * LINE / (mk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "traffic": mt, "traffic_source": mt_src, "frac_traffic": (mt / (mk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if mt else None}
        ids_matched = ids.clone()
        result["config"]["matched_recall_point"] = matched
        try:  # informational: the same kernel, graph and window on a 4x batch -- a launch pays ~0.2 ms of ramp and drain whatever its size
            if args.no_extras:
                raise RuntimeError("--no-extras")
            nq4 = 4 * nq
            q4 = deep_like_device(nq4, d, 24, seed=3, device=device)
            ids4 = torch.empty((nq4, k), dtype=torch.int64, device=device)
            dists4 = torch.empty((nq4, k), dtype=torch.float32, device=device)
            st4 = torch.zeros((nq4, 6), dtype=torch.int32, device=device)
            index.search_device(q4, k, sp, d_ids64=ids4, d_dists=dists4, d_stats=st4)
            for _ in range(max(args.warmup, 12)):  # informational points warm up on their own: at least 12 launches whatever --warmup says
                index.search_device(q4, k, sp, d_ids64=ids4, d_dists=dists4)
            torch.cuda.synchronize()
            e4 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(3, args.steps // 4))]
            for a_, b_ in e4:
                a_.record()
                index.search_device(q4, k, sp, d_ids64=ids4, d_dists=dists4)
                b_.record()
            torch.cuda.synchronize()
            ms4 = float(np.mean([a_.elapsed_time(b_) for a_, b_ in e4]))
            l4 = designed_lines(st4.to(torch.int64), gstride, d * 4, min(mode, 2), d, k, record_bytes)
            result["config"]["batch_x4"] = {"queries": nq4, "kernel_ms": ms4, "qps": nq4 / (ms4 * 1e-3), "designed_lines_per_query": l4["total"] / nq4,
                                            "achieved_GBs": l4["total"] * LINE / (ms4 * 1e-3) / 1e9, "frac": l4["total"] * LINE / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "note": "not the headline batch; shows the per-launch ramp/drain share of the 10 000-query figure"}
            del q4, ids4, dists4, st4
        except Exception as e:
            log(f"[bench] 4x batch point skipped: {e}")
        if not args.no_extras and mode >= 2:
            try:  # the same index with the record size the LIBRARY's rule picks (mode 2): on this set the 28-byte form of round 6 -- fewer lines, more QPS
                index.build_bound_codes(0)  # (mode 2 keeps records that exist, whatever their size)
                index.build_bound_codes(2)
                strule = torch.zeros((nq, 6), dtype=torch.int32, device=device)
                idsr = torch.empty((nq, k), dtype=torch.int64, device=device)
                index.search_device(q, k, sp, d_ids64=idsr, d_dists=dists, d_stats=strule)
                for _ in range(max(args.warmup, 12)):
                    index.search_device(q, k, sp, d_ids64=idsr, d_dists=dists)
                torch.cuda.synchronize()
                evr = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                t0 = time.perf_counter()
                for a_, b_ in evr:
                    a_.record()
                    index.search_device(q, k, sp, d_ids64=idsr, d_dists=dists)
                    b_.record()
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t0
                msr = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evr]))
                rbr = int(index.memory()["record_size"])
                lr = designed_lines(strule.to(torch.int64), gstride, d * 4, 2, d, k, rbr)
                tr, tr_src, _ = measured_traffic(f"{args.config}:n={n}:nq={nq}:window={args.window}:R={R}:visited={args.visited}:codes=2")
                index.search_device(q, k, sp_m, d_ids64=idsr, d_dists=dists)
                for _ in range(max(args.warmup, 12)):
                    index.search_device(q, k, sp_m, d_ids64=idsr, d_dists=dists)
                torch.cuda.synchronize()
                evm = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for a_, b_ in evm:
                    a_.record()
                    index.search_device(q, k, sp_m, d_ids64=idsr, d_dists=dists)
                    b_.record()
                torch.cuda.synchronize()
                msm = float(np.mean([a_.elapsed_time(b_) for a_, b_ in evm]))
                tm, tm_src, _ = measured_traffic(f"{args.config}:n={n}:nq={nq}:window={matched['window']}:R={R}:visited={args.visited}:codes=2")
                result["config"].setdefault("other_configs", {})["c2_records_rule"] = {
                    "what": "the headline's index, batch and window with the record size svsx_index_build_bound_codes(index, 2) picks by itself instead of the 32 bytes the headline "
                            "has used since round 2: fewer lines per expansion, so MORE queries per second at a LOWER fraction of the bandwidth roof",
                    "record_bytes": rbr, "kernel": "svsx::" + index.last_kernel_info(), "qps": nq * args.steps / dtr, "kernel_ms": msr,
                    "ids_equal_headline": None,
                    "designed_lines_per_query": lr["total"] / nq, "bound_code_rejects_per_query": int(strule[:, 5].sum()) / nq,
                    "traffic": tr, "traffic_source": tr_src, "frac_traffic": (tr / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
                    "matched_window": {"window": matched["window"], "kernel_ms": msm, "qps": nq / (msm * 1e-3), "traffic": tm, "traffic_source": tm_src,
                                       "frac_traffic": (tm / (msm * 1e-3) / 1e9 / HBM_PEAK_GBS) if tm else None}}
                index.search_device(q, k, sp, d_ids64=idsr, d_dists=dists)
                torch.cuda.synchronize()
                index.build_bound_codes(0)
                index.build_bound_codes(mode)
                index.search_device(q, k, sp, d_ids64=ids, d_dists=dists)
                torch.cuda.synchronize()
                result["config"]["other_configs"]["c2_records_rule"]["ids_equal_headline"] = bool((idsr == ids).all())
            except Exception as e:  # informational: never fail the line over it
                result["config"].setdefault("other_configs", {})["c2_records_rule"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench] rule-sized records point failed: {e}")
        if not args.no_extras:
            try:
                head = {"generator": "G(24, 0.1), unit norm (the headline's set)", "qps": result["value"], "kernel_ms": kern_ms, "recall_at_10": recall,
                        "record_bytes": record_bytes, "kernel": "svsx::" + kinfo, "expansions_per_query": exp_total / nq,
                        "distance_evals_per_query": int(st[:, 2].sum()) / nq, "bound_code_rejects_per_query": int(st[:, 5].sum()) / nq,
                        "reject_share": float(st[:, 5].sum()) / max(float(st[:, 2].sum()), 1.0), "designed_lines_per_query": lines["total"] / nq,
                        "frac_designed": achieved / HBM_PEAK_GBS, "frac_traffic": result["roofline"]["frac_traffic"]}
                result["config"].setdefault("other_configs", {})["c2_spectrum"] = spectrum_points(args, svs, device, local_rank, 2, k, R, head)
            except Exception as e:  # informational: never fail the line over it
                result["config"].setdefault("other_configs", {})["c2_spectrum"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench] spectrum points failed: {e}")
        if not args.no_extras:
            try:  # BASELINE config 3 as a point of the same line: sift-1M-like, 1 M x 128 float16 rows, float32 queries, inner product
                from scalablevectorsearch_amd.build import VamanaBuildParameters, build_vamana_device

                t0 = time.time()
                n3, d3 = 1_000_000, 128
                x3 = deep_like_device(n3, d3, 32, seed=31, device=device, normalize=False).half()
                q3 = deep_like_device(nq, d3, 32, seed=32, device=device, normalize=False)
                g3, e3 = build_vamana_device(x3, n3, d3, svs.DistanceIP(), VamanaBuildParameters(alpha=0.95, graph_max_degree=R, window_size=2 * R, prune_to=R - 4),
                                             data_dtype=np.float16, device=local_rank)
                i3 = svs.VamanaIndex.from_device(g3, x3, svs.DistanceIP(), torch.tensor([e3], dtype=torch.int32, device=device), n3, d3, np.float16, device=local_rank)
                i3.build_bound_codes(2)
                torch.cuda.synchronize()
                t_b3 = time.time() - t0
                ids3 = torch.empty((nq, k), dtype=torch.int64, device=device)
                dd3 = torch.empty((nq, k), dtype=torch.float32, device=device)
                st3 = torch.zeros((nq, 6), dtype=torch.int32, device=device)
                i3.search_device(q3, k, sp, d_ids64=ids3, d_dists=dd3, d_stats=st3)
                for _ in range(max(args.warmup, 12)):  # informational points warm up on their own: at least 12 launches whatever --warmup says
                    i3.search_device(q3, k, sp, d_ids64=ids3, d_dists=dd3)
                torch.cuda.synchronize()
                e3v = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for a_, b_ in e3v:
                    a_.record()
                    i3.search_device(q3, k, sp, d_ids64=ids3, d_dists=dd3)
                    b_.record()
                torch.cuda.synchronize()
                ms3 = float(np.mean([a_.elapsed_time(b_) for a_, b_ in e3v]))
                gt3, _ = i3.flat_search_device(q3, k)
                torch.cuda.synchronize()
                # the north star's dense case (float16 batched-query x candidate-block on the matrix cores): the exhaustive search of this set, timed
                f3ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
                f3cnt = torch.zeros(1, dtype=torch.int32, device=device)
                for a_, b_ in f3ev:
                    a_.record()
                    i3.flat_search_device(q3, k, recomputed=f3cnt)
                    b_.record()
                torch.cuda.synchronize()
                f3ms = float(np.median([a_.elapsed_time(b_) for a_, b_ in f3ev]))
                flat3 = {"what": "svsx_flat_search_device (FlatIndex::search, flat.h:322-460) of the same 10 000 queries against all 1 M rows: float16 MFMA "
                                 "threshold pipeline, exact float32 re-rank, per-query proof, exhaustive pass for unproven queries",
                         "ms_per_batch": f3ms, "tflops_f16_mfma_end_to_end": 2.0 * n3 * d3 * nq / (f3ms * 1e-3) / 1e12,
                         "frac_of_dense_f16_peak_2500_tflops": 2.0 * n3 * d3 * nq / (f3ms * 1e-3) / 1e12 / 2500.0,
                         "queries_recomputed_exhaustively": int(f3cnt.item()) // 7, "runs": 7, "exact_answers_per_second": nq / (f3ms * 1e-3)}
                k3 = i3.last_kernel_info()
                rb3 = {"codes=2>": 32, "codes=3>": 64, "codes=4>": 128, "codes=5>": 28}.get(k3[-8:], 32)
                l3 = designed_lines(st3.to(torch.int64), gstride, d3 * 2, 2, d3, k, rb3)
                result["config"].setdefault("other_configs", {})["c3"] = {
                    "workload": f"sift-1M-like d=128 float16 rows, float32 queries, inner product, batch={nq}, search_window={args.window}, k=10, 1xMI355X",
                    "qps": nq / (ms3 * 1e-3), "kernel_ms": ms3, "recall_at_10": recall_at(gt3, ids3, k), "kernel": "svsx::" + k3,
                    "designed_lines_per_query": l3["total"] / nq, "frac_designed": l3["total"] * LINE / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "expansions_per_query": int(st3[:, 0].sum()) / nq, "bound_code_rejects_per_query": int(st3[:, 5].sum()) / nq,
                    "distance_evals_per_query": int(st3[:, 2].sum()) / nq, "build_seconds": t_b3,
                    "graph": f"Vamana R={R} L={2*R} alpha=0.95 (built by svsx_vamana_build_device)", "flat_search": flat3}
                del x3, q3, g3, i3, ids3, dd3, st3, gt3
            except Exception as e:  # informational: never fail the line over it
                log(f"[bench] config-3 point skipped: {e}")
        if not args.no_extras:
            try:  # config 2 on the reference's regression graph (tools/benchmark_inputs/vamana/static-regression.toml: R = 128, L = 200, prune_to 124)
                t0 = time.time()
                g128, e128, _ = build_or_load_graph(svs, shards[0]["x"], n, d, 128, local_rank, "", build_window=200)
                i128 = svs.VamanaIndex.from_device(g128, shards[0]["x"], svs.DistanceL2(), torch.tensor([e128], dtype=torch.int32, device=device), n, d, np.float32, device=local_rank)
                i128.build_bound_codes(2)
                torch.cuda.synchronize()
                t_b128 = time.time() - t0
                ids128 = torch.empty((nq, k), dtype=torch.int64, device=device)
                dd128 = torch.empty((nq, k), dtype=torch.float32, device=device)
                st128 = torch.zeros((nq, 6), dtype=torch.int32, device=device)
                i128.search_device(q, k, sp, d_ids64=ids128, d_dists=dd128, d_stats=st128)
                for _ in range(max(args.warmup, 12)):  # informational points warm up on their own: at least 12 launches whatever --warmup says
                    i128.search_device(q, k, sp, d_ids64=ids128, d_dists=dd128)
                torch.cuda.synchronize()
                ev128 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for a_, b_ in ev128:
                    a_.record()
                    i128.search_device(q, k, sp, d_ids64=ids128, d_dists=dd128)
                    b_.record()
                torch.cuda.synchronize()
                ms128 = float(np.mean([a_.elapsed_time(b_) for a_, b_ in ev128]))
                k128 = i128.last_kernel_info()
                l128 = designed_lines(st128.to(torch.int64), 128, d * 4, 2, d, k, {"codes=2>": 32, "codes=3>": 64, "codes=4>": 128, "codes=5>": 28}.get(k128[-8:], 32))
                result["config"].setdefault("other_configs", {})["c2_regression_graph_r128"] = {
                    "workload": f"deep-1M-like d=96 float32 L2, batch={nq}, search_window={args.window}, k=10, graph R=128 L=200 prune_to 124 alpha 1.2 (static-regression.toml), 1xMI355X",
                    "qps": nq / (ms128 * 1e-3), "kernel_ms": ms128, "recall_at_10": recall_at(gt, ids128, k), "kernel": "svsx::" + k128,
                    "mean_degree": float((g128[:, :128] >= 0).sum(1).float().mean()), "expansions_per_query": int(st128[:, 0].sum()) / nq,
                    "designed_lines_per_query": l128["total"] / nq, "frac_designed": l128["total"] * LINE / (ms128 * 1e-3) / 1e9 / HBM_PEAK_GBS, "build_seconds": t_b128}
                del g128, i128, ids128, dd128, st128
            except Exception as e:  # informational: never fail the line over it
                log(f"[bench] config-2 R=128 point skipped: {e}")
            try:
                result["config"].setdefault("other_configs", {})["regression_uint8_r128"] = regression_uint8_point(args, svs, device, local_rank, k)
            except Exception as e:  # informational: never fail the line over it
                result["config"].setdefault("other_configs", {})["regression_uint8_r128"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench] uint8 regression point failed: {e}")
            torch.cuda.empty_cache()
            try:
                if args.c4_n > 0:
                    result["config"].setdefault("other_configs", {})["c4"] = config4_point(args, svs, device, local_rank, k, t_run0)
            except Exception as e:  # informational: never fail the line over it
                result["config"].setdefault("other_configs", {})["c4"] = {"error": f"{type(e).__name__}: {e}"}
                log(f"[bench] config-4 point failed: {e}")
            torch.cuda.empty_cache()
        index.search_device(q, k, sp, d_ids64=ids, d_dists=dists)  # restore window-128 results for the parity sample
        torch.cuda.synchronize()
        if not args.no_cpu_baseline:
            sh = shards[0]
            result["cpu_baseline"] = cpu_baseline(sh["x"], q, sh["graph"], sh["entry"], R, k, args.window, ids)
            cb = result["cpu_baseline"]
            cb["gpu_over_cpu"] = result["value"] / cb["value"]
            # both numbers next to each other: the measured ratio uses the CPUs the cgroup grants; a fully used host under LINEAR scaling
            # (an upper bound for a memory-bound graph search) would shrink it by cores_present / cores
            cb["gpu_over_full_host_if_linear"] = cb["gpu_over_cpu"] * cb["cores"] / max(cb["cores_present"], 1)
            cb["full_host_note"] = (f"MEASURED: {cb['cores']} threads (the container's CPU quota) of {cb['cores_present']} logical CPUs.  gpu_over_full_host_if_linear is an "
                                    "EXTRAPOLATION (the measured rate scaled linearly to every CPU of the host), not a measurement: the north star's >= 10x "
                                    "over the host CPU is established only against the CPUs this container may use")
            cm = cpu_baseline(sh["x"], q, sh["graph"], sh["entry"], R, k, matched["window"], ids_matched)
            matched["cpu_qps"] = cm["value"]
            matched["cpu_cores"] = cm["cores"]
            matched["gpu_over_cpu"] = matched["gpu_qps"] / cm["value"]
            matched["gpu_over_cpu_note"] = (f"CPU leg uses {cm['cores']} threads (cgroup quota) of {cm['cores_present']} logical CPUs present; "
                                            "a fully used host would be proportionally faster")
            matched["ids_identical_on_sample"] = "identical on sample: True" in cm["sample"]
            matched["gpu_over_full_host_if_linear"] = matched["gpu_over_cpu"] * cm["cores"] / max(cm["cores_present"], 1)
            try:  # how the host leg really scales with its threads: both operating points
                cb["thread_scaling"] = cpu_scaling(sh["x"], q, sh["graph"], sh["entry"], R, k, [args.window, matched["window"]])
                if cb["thread_scaling"]:
                    eff = cb["thread_scaling"][str(args.window)]["efficiency_at_max_threads"]
                    cb["thread_scaling"]["note"] = (f"parallel efficiency of the host leg at its {cm['cores']} usable threads: {eff:.2f} at window {args.window}; "
                                                    "gpu_over_full_host_if_linear assumes 1.0 for the CPUs the cgroup withholds, so it is a LOWER bound of the real ratio")
            except Exception as e:  # noqa: BLE001
                log(f"[bench] cpu scaling curve skipped: {e}")
    if sharded is not None:  # the multi-GPU data path (per-shard search -> ONE RCCL all-gather -> merge) as a top-level metric of its own
        ok = isinstance(sharded, dict) and "merged_answers_per_s" in sharded
        result["sharded_search"] = {"metric": "merged answers per second over a FIXED set of 8 database shards (SURVEY 8e; strong scaling in N)",
                                    "value": sharded.get("merged_answers_per_s") if ok else None, "unit": "queries/s", "n_gpus": world,
                                    "vectors_per_shard": sharded.get("vectors_per_shard") if ok else None, "failed": not ok,
                                    "error": None if ok else (sharded.get("error") or sharded.get("skipped") or "unknown")}
    _emit(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()
    if world > 1 and sharded is not None and result["sharded_search"]["failed"] and "skipped" not in (sharded or {}):
        sys.exit(3)  # at N > 1 the sharded search IS the multi-GPU path: its failure fails the run, loudly, after the line has been printed


if __name__ == "__main__":
    main()
