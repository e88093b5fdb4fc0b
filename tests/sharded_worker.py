"""Worker of tests/test_gpu_multi.py: one process per GPU (torch.distributed.run), RCCL communicator inside libsvsx (svsx_sharded_*).
Every rank builds its shards of a fixed global shard list, all ranks search the same batch, every rank checks the merged answer against
oracle searches of ALL shards merged by the stated rule (ties -> lower shard, then position) and writes ok / the failure to --out.<rank>."""
import argparse
import os
import sys
import traceback

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--shards", type=int, default=4)
    ap.add_argument("--nq", type=int, default=8200)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    try:
        import scalablevectorsearch_amd as svs
        from conftest import deep_like
        from oracle.oracle import L2, Oracle, merge_topk_ref
        from scalablevectorsearch_amd.sharded import NativeShardedSearcher

        o = Oracle()
        S, n, d, k = a.shards, 1500, 96, 10
        per = S // world
        base = (1 << 32) + 4242  # global ids beyond 32 bits
        q = deep_like(a.nq, d, 24, seed=77)
        q[1] = q[0]
        want_d, want_i, mine = [], [], []
        for s in range(S):
            x = deep_like(n, d, 24, seed=500 + (s % 3))  # shards 0 and 3 hold the same vectors: distances tie ACROSS ranks
            g, e = o.build_vamana(L2, x, R=24, L=40)
            oi, od = o.search(L2, x, g, e, q, k, 32)
            want_i.append(oi.astype(np.int64) + s * base)
            want_d.append(od)
            if s // per == rank:
                idx = svs.VamanaIndex(g, x, svs.DistanceL2(), e, device=local, id_offset=s * base)
                if s % 2:
                    idx.build_bound_codes(2)
                mine.append(idx)
        wd, wi = merge_topk_ref(L2, np.stack(want_d), np.stack(want_i), k)
        ns = NativeShardedSearcher(mine)
        info = ns.info()
        assert info["world"] == world and info["rank"] == rank and info["communicator"] and info["local_shards"] == per, info
        sp = svs.VamanaSearchParameters(svs.SearchBufferConfig(32))
        dq = torch.from_numpy(q).cuda()
        for rows in (a.nq, 3, 8192, 1):  # two chunks, one chunk, odd nq * k, the staging shrinking and growing between calls
            md, mi = ns.search(dq[:rows], k, sp)
            torch.cuda.synchronize()
            assert (mi.cpu().numpy() == wi[:rows]).all(), f"ids differ at {rows} rows"
            assert (md.cpu().numpy().view(np.uint32) == wd[:rows].view(np.uint32)).all(), f"distances differ at {rows} rows"
        ns.close()
        open(f"{a.out}.{rank}", "w").write("ok")
    except Exception:  # noqa: BLE001
        open(f"{a.out}.{rank}", "w").write(traceback.format_exc())
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
