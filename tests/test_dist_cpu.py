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


# ------------------------------------------------------------------ DataParallelEngine (moondream_amd/parallel.py), stub model, gloo
class _StubModel:
    """Stands in for MoondreamModel: outputs are a function of (item, a broadcast weight) so that a wrong shard, a wrong
    order or a missing weight broadcast changes them."""

    def __init__(self, config, sd, device, **kw):
        self.k = int(sd["k"].item())
        self.kw = kw

    def batch_generate_ids(self, images, prompts, max_tokens=4, ignore_eos=False):
        return [[self.k + im + p[0] + t for t in range(1 + im % 3)][:max_tokens] for im, p in zip(images, prompts)]  # ragged

    def batch_generate_ids_pipelined(self, batches, max_tokens=4, ignore_eos=False):
        # EOS-truncated like MoondreamModel._collect with ignore_eos=False: ragged on a rank, and the longest row differs by rank
        for images, prompts in batches:
            yield [[self.k + im + t for t in range(max_tokens if ignore_eos else 1 + im % max_tokens)] for im in images]

    def batch_caption(self, images, length="normal", settings=None):
        return [f"{length}:{self.k + im}" for im in images]

    def batch_query(self, images, questions, settings=None):
        return [f"{q}?{self.k + im}" for im, q in zip(images, questions)]

    def batch_detect(self, images, objects, settings=None):
        return [{"objects": [{"x_min": float(im), "o": o}] * (im % 2)} for im, o in zip(images, objects)]

    def batch_point(self, images, objects, settings=None):
        return [{"points": [{"x": float(im)}]} for im in images]


def _engine_worker(rank, world, port, out_dir, weights_path, backend="gloo", device="cpu", single_rank_group=False):
    from moondream_amd.config import get_config
    from moondream_amd.parallel import DataParallelEngine

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    if single_rank_group:  # ONE rank that still builds its process group and runs every collective of the N-rank path
        os.environ[mdist.SINGLE_RANK_GROUP_ENV] = "1"
    cfg = get_config("tiny")
    # rank 0 alone reads the checkpoint FILE; the others learn names / shapes / dtypes and the bytes from the broadcast
    eng = DataParallelEngine(cfg, weights_file=weights_path if rank == 0 else None, backend=backend, device=torch.device(device),
                             model_factory=_StubModel, max_batch=4)
    assert dist.is_initialized() and dist.get_backend() == backend and eng._collective
    assert (eng.rank, eng.world) == (rank, world) and eng.model.k == 1000 and eng.model.kw == {"max_batch": 4}
    assert eng.weights_report["bytes"] > 0
    n = 7
    images = list(range(n))                      # the GLOBAL list, identical on every rank
    prompts = [[10 * i] for i in range(n)]
    want_ids = [[1000 + i + 10 * i + t for t in range(1 + i % 3)] for i in range(n)]
    got = eng.batch_generate_ids(images, prompts, max_tokens=4)
    mine = list(eng.shard(n))
    got_local = eng.batch_generate_ids([images[i] for i in mine], [prompts[i] for i in mine], max_tokens=4, local=True)
    caps = eng.batch_caption(images, length="short")
    ans = eng.batch_query(images, [f"q{i}" for i in range(n)])
    det = eng.batch_detect(images, ["cat"] * n)
    pts = eng.batch_point(images, ["cat"] * n)
    steps = list(eng.batch_generate_ids_pipelined((([images[i] for i in mine], [prompts[i] for i in mine]) for _ in range(2)), n_total=n,
                                                  max_tokens=3, ignore_eos=True))
    # the default (stop at EOS): ragged rows on a rank, unequal longest row across ranks (rank 1's block of 3 images at
    # max_tokens 5 never reaches 5 ids) -- every block is [B_rank, max_tokens] padded with -1 (advisor, round 5)
    ragged = list(eng.batch_generate_ids_pipelined((([images[i] for i in mine], [prompts[i] for i in mine]) for _ in range(2)), n_total=n,
                                                   max_tokens=5))
    assert eng.ranks_seen() == world and eng.max_over_ranks(float(rank)) == float(world - 1)
    # fewer items than ranks: the last rank's shard is empty and still takes part in every collective
    one = eng.batch_generate_ids(images[:1], prompts[:1], max_tokens=4)
    one_det = eng.batch_detect(images[:1], ["cat"])
    none_at_all = eng.batch_caption([])
    assert (one, one_det, none_at_all) == (([want_ids[0]], [{"objects": []}], []) if rank == 0 else (None, None, None))
    with pytest.raises(ValueError):
        eng.batch_generate_ids(images, prompts[:-1])
    eng.barrier()
    if rank == 0:
        assert got == want_ids and got_local == want_ids, (got, got_local)
        assert caps == [f"short:{1000 + i}" for i in range(n)]
        assert ans == [f"q{i}?{1000 + i}" for i in range(n)]
        assert det == [{"objects": [{"x_min": float(i), "o": "cat"}] * (i % 2)} for i in range(n)]
        assert pts == [{"points": [{"x": float(i)}]} for i in range(n)]
        for blocks in steps:
            assert torch.cat(blocks, 0).tolist() == [[1000 + i + t for t in range(3)] for i in range(n)]
        from moondream_amd.parallel import strip_id_padding
        for blocks in ragged:
            assert all(b.shape[1] == 5 and b.dtype == torch.int32 for b in blocks)
            assert strip_id_padding(blocks) == [[1000 + i + t for t in range(1 + i % 5)] for i in range(n)]
        open(os.path.join(out_dir, "engine_ok"), "w").write("1")
    else:
        assert got is None and got_local is None and caps is None and det is None and steps == [None, None] and ragged == [None, None]
    eng.close()


def test_data_parallel_engine_world_2_gloo(tmp_path):
    """The product-level DP runner: rank 0 loads the checkpoint file, flat broadcast, per-rank lockstep engine over
    shard_range blocks, id gather / object gather on rank 0 -- with a stub model, over gloo, world_size 2."""
    port = _free_port()
    mp.spawn(_engine_worker, args=(2, port, str(tmp_path), _write_stub_checkpoint(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "engine_ok").exists()


def _write_stub_checkpoint(tmp_path):
    weights_path = str(tmp_path / "stub.pt")
    torch.save({"k": torch.tensor([1000], dtype=torch.int32), "w": torch.arange(6, dtype=torch.float32).to(torch.bfloat16)}, weights_path)
    return weights_path


def test_data_parallel_engine_single_rank_group_runs_every_collective(tmp_path):
    """MOONDREAM_DIST_SINGLE_RANK_GROUP=1: ONE rank builds its process group and goes through the N-rank code path -- the
    object broadcast of the template, the flat weight broadcast, the all-reduces, the id gather (own stream on a GPU), the
    object gather, the barrier.  Here over gloo; tests/test_model_gpu.py runs the same worker over RCCL on the GPU box,
    which is how the RCCL calls of dist.py / parallel.py get executed where only one GPU exists."""
    port = _free_port()
    mp.spawn(_engine_worker, args=(1, port, str(tmp_path), _write_stub_checkpoint(tmp_path), "gloo", "cpu", True), nprocs=1, join=True)
    assert (tmp_path / "engine_ok").exists()


def test_bench_selftest_single_rank_group():
    """`bench.py --gpus 1 --selftest-dist` under MOONDREAM_DIST_SINGLE_RANK_GROUP=1: the bench's own plumbing
    (engine weight path with verification, id gather, per-rank floats) over a one-rank process group."""
    import json
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env[mdist.SINGLE_RANK_GROUP_ENV] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--selftest-dist"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["items"] == 4
    assert line["weights_broadcast"]["equal_to_local_copy_on_every_rank"] is True


def test_data_parallel_engine_single_process_needs_no_process_group(monkeypatch):
    from moondream_amd.config import get_config
    from moondream_amd.parallel import DataParallelEngine

    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", mdist.SINGLE_RANK_GROUP_ENV):
        monkeypatch.delenv(k, raising=False)
    eng = DataParallelEngine(get_config("tiny"), state_dict={"k": torch.tensor([5], dtype=torch.int32)}, device=torch.device("cpu"),
                             model_factory=_StubModel)
    assert eng.world == 1 and not dist.is_initialized()
    assert eng.batch_generate_ids([0, 1, 2], [[0], [0], [0]], max_tokens=2) == [[5], [6, 7], [7, 8]]
    assert eng.batch_caption([3]) == ["normal:8"]
    with pytest.raises(ValueError):
        DataParallelEngine(get_config("tiny"), device=torch.device("cpu"), model_factory=_StubModel)


def test_bench_selftest_world_8_gloo():
    """The driver's 8-GPU command line in a dry run (review, round 5): `bench.py --gpus 8 --selftest-dist` launches itself
    as 8 ranks (gloo here), every rank binds its share of the cores, builds the checkpoint, takes the flat broadcast and
    verifies it, gathers ids / floats; the line carries ranks_seen, per-rank values, the broadcast's bytes and seconds and
    every rank's core share -- disjoint across ranks."""
    import json
    import subprocess

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "8", "--selftest-dist"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8 and line["items"] == 25
    assert line["per_rank_ms_per_step"] == [r + 0.5 for r in range(8)]
    wb = line["weights_broadcast"]
    assert wb["equal_to_local_copy_on_every_rank"] is True and wb["bytes"] > 0 and wb["seconds"] >= 0
    binds = line["cpu_binding"]
    assert len(binds) == 8
    n_cpu = len(os.sched_getaffinity(0))
    if n_cpu >= 8:
        assert all(b["bound"] for b in binds)
        spans = sorted((b["first"], b["last"]) for b in binds)
        assert all(spans[i][1] < spans[i + 1][0] for i in range(7)), spans   # disjoint shares
        assert sum(b["cpus"] for b in binds) == n_cpu


def test_plan_rank_cpus_splits_numa_nodes_between_their_gpus():
    """dist.plan_rank_cpus: 8 ranks, GPUs 0-3 on node 0 (CPUs 0-63, 128-191), GPUs 4-7 on node 1 -- every rank gets a quarter
    of ITS node, shares are disjoint and cover the host; unknown topology -> even slices; a restricted mask is respected;
    and the derived rendezvous port is a function of the job, equal on every rank."""
    node0 = list(range(0, 64)) + list(range(128, 192))
    node1 = list(range(64, 128)) + list(range(192, 256))
    by_rank = [node0] * 4 + [node1] * 4
    shares = [mdist.plan_rank_cpus(r, 8, range(256), by_rank) for r in range(8)]
    assert all(len(s) == 32 for s in shares)
    assert sorted(c for s in shares for c in s) == list(range(256))
    assert all(set(shares[r]) <= set(node0) for r in range(4)) and all(set(shares[r]) <= set(node1) for r in range(4, 8))
    flat = [mdist.plan_rank_cpus(r, 8, range(16), [[]] * 8) for r in range(8)]
    assert flat == [[2 * r, 2 * r + 1] for r in range(8)]
    masked = [mdist.plan_rank_cpus(r, 2, [4, 5, 6, 7], [node0, node0]) for r in range(2)]
    assert masked == [[4, 5], [6, 7]]
    assert mdist.plan_rank_cpus(1, 4, [3], [[]] * 4) == [3]           # fewer cores than ranks: never an empty mask
    a = mdist.default_master_port()
    assert 20000 <= a < 50000 and a == mdist.default_master_port()
