"""The N > 1 path on real hardware: svsx_sharded_* with an RCCL communicator of TWO ranks (one process per GPU).  Skipped on a one-GPU box
(the driver's 8-GPU lease and any multi-GPU box run it); the single-GPU form of the same pipeline -- chunked exchange, concurrent local
shards, a one-rank communicator -- runs everywhere in test_gpu_configs.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch

    return torch.cuda.device_count()


@pytest.mark.parametrize("world,shards", [(2, 2), (2, 4)])
def test_sharded_search_on_two_gpus_matches_oracle(tmp_path, world, shards):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_gpus()}")
    out = str(tmp_path / "result")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29533",
           os.path.join(ROOT, "tests", "sharded_worker.py"), "--out", out, "--shards", str(shards)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    for r in range(world):
        f = f"{out}.{r}"
        assert os.path.exists(f), f"rank {r} wrote nothing\n{p.stderr[-3000:]}"
        assert open(f).read() == "ok", open(f).read()
