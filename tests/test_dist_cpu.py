"""The N>1 path (shard -> per-rank work -> gather) on CPU with gloo, world_size 2."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from moondream_amd import dist as mdist


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                seen += list(mdist.shard_range(n, r, world))
            assert seen == list(range(n))
            sizes = [len(mdist.shard_range(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = mdist.init_from_env("gloo")
    assert (r, w) == (rank, world)
    # 1. weight broadcast from rank 0
    full = {"a.weight": torch.arange(12, dtype=torch.float32).reshape(3, 4).to(torch.bfloat16), "b": torch.tensor([7], dtype=torch.int32), "c": torch.randn(5, 5, generator=torch.Generator().manual_seed(1)).to(torch.bfloat16)}
    tpl = mdist.state_dict_template(full)
    got = mdist.broadcast_state_dict(full if rank == 0 else None, tpl, "cpu")
    for k in full:
        assert torch.equal(got[k], full[k]), k
    # 2. shard 7 "images", fake per-rank decode (ids derive from the image index), gather on rank 0
    mine = mdist.shard_range(7, rank, world)
    local = torch.tensor([[i * 10 + t for t in range(4)] for i in mine], dtype=torch.int32).reshape(len(mine), 4)
    blocks = mdist.gather_token_ids(local)
    blocks_known = mdist.gather_token_ids(local, n_total=7)  # block sizes from shard_range: no size exchange
    t = mdist.max_over_ranks(float(rank + 1), "cpu")
    assert t == float(world)
    assert mdist.ranks_seen("cpu") == world
    per_rank = mdist.gather_floats(10.0 + rank, "cpu")
    assert per_rank == ([10.0 + r for r in range(world)] if rank == 0 else None)
    mdist.barrier()
    if rank == 0:
        allids = torch.cat(blocks, 0)
        assert allids.tolist() == [[i * 10 + t for t in range(4)] for i in range(7)]
        assert torch.equal(torch.cat(blocks_known, 0), allids)
        open(os.path.join(out_dir, "ok"), "w").write("1")
    else:
        assert blocks is None and blocks_known is None  # a gather to rank 0, not an all-gather
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").exists()


def test_bench_self_launches_n2_without_torchrun():
    """`python bench.py --gpus 2` with no launcher in front of it (what the driver runs) must
    re-exec itself as 2 ranks under torch.distributed.run and print ONE JSON line with n_gpus 2;
    --selftest-dist limits the run to the launch + collective plumbing (gloo on this CPU-only box)."""
    import json
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-dist"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["items"] == 7 and d["max_rank"] == 1.0 and d["ranks_seen"] == 2
    assert len(d["per_rank_ms_per_step"]) == 2  # one clock per rank reaches rank 0's JSON line
    assert d["weights_broadcast"]["equal_to_local_copy_on_every_rank"] is True and d["weights_broadcast"]["bytes"] > 0


def test_bench_self_launch_propagates_a_failing_rank():
    """A rank that dies must make `python bench.py --gpus 2` exit non-zero with that rank's traceback on stderr
    (the driver only sees the parent's exit code)."""
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["MD_SELFTEST_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--selftest-dist"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode != 0
    assert "rank 1 asked to fail" in r.stderr and "a rank of the 2-process launch failed" in r.stderr


def test_relaunch_is_a_noop_inside_a_rank_and_for_one_gpu(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert mdist.relaunch_under_torchrun(1, "bench.py", []) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert mdist.relaunch_under_torchrun(2, "bench.py", []) is None
