"""The collective calls of the N-rank path (moondream_amd/dist.py, parallel.py) executed over RCCL on the one-GPU box.

Two ranks cannot share one GPU under RCCL ("duplicate GPU"), and no multi-GPU box is available to this repository's GPU
runs, so the N-rank path is covered twice from different sides: world_size 2 over gloo on CPU (tests/test_dist_cpu.py: the
sharding, ordering and gather logic), and here ONE rank over ``backend="nccl"`` with MOONDREAM_DIST_SINGLE_RANK_GROUP=1 -- the
same calls with device tensors through ProcessGroupNCCL / RCCL: object broadcast, the flat uint8 weight broadcast, int32 /
int64 / float64 all-reduces, the id gather on its own stream, gather_object, barrier."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

from moondream_amd import dist as mdist

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env[mdist.SINGLE_RANK_GROUP_ENV] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_data_parallel_engine_over_rccl_single_rank_group(tmp_path):
    from test_dist_cpu import _engine_worker, _free_port, _write_stub_checkpoint

    assert torch.cuda.is_available()
    mp.spawn(_engine_worker, args=(1, _free_port(), str(tmp_path), _write_stub_checkpoint(tmp_path), "nccl", "cuda:0", True), nprocs=1, join=True)
    assert (tmp_path / "engine_ok").exists()


def test_bench_selftest_over_rccl_single_rank_group():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--selftest-dist"], env=_clean_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["items"] == 4
    assert line["weights_broadcast"]["equal_to_local_copy_on_every_rank"] is True


def test_bench_under_torchrun_one_rank_over_rccl():
    """The driver's multi-GPU command line with N = 1 and the one-rank group: the REAL bench step (tiny model, 2 steps) with the
    RCCL weight broadcast (verified against the local copy), the per-step id gather on its own stream, the barrier and the
    max-over-ranks around the timed region."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(mdist.free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--model", "tiny", "--batch", "4",
           "--tokens", "8", "--steps", "2", "--warmup", "2", "--no-vqa-leg", "--no-strict-leg", "--no-detect13-leg", "--no-dedup-leg",
           "--no-fp8-leg", "--no-fp8-full-leg", "--no-cpu-baseline", "--no-second-oracle", "--latency-runs", "0"]
    r = subprocess.run(cmd, env=_clean_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, text=True)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["value"] > 0
    assert line["weights_broadcast"]["equal_to_local_copy_on_every_rank"] is True and line["weights_broadcast"]["bytes"] > 0
