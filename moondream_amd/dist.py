"""Image-level data parallelism: one process per GPU, RCCL over xGMI.

The path shards by independent units (images): every image's vision pass, KV
slab and decode loop is independent of every other image's (the reference's
batch entry point is a plain loop, hf_moondream.py:99-103).  So there is no
collective on the data path.  RCCL is used for exactly two things
(SURVEY.md section 8e):

  * once, at start-up: broadcast the checkpoint from rank 0 (one flat bf16
    buffer -> one large collective, sized for the per-link xGMI bandwidth
    instead of ~600 small ones);
  * once per batch: gather the int32 token ids (a few KB) on rank 0.

``backend="nccl"`` is RCCL on ROCm; the same code runs over ``gloo`` on CPU
tensors, which is how tests/test_dist_cpu.py covers it with world_size 2.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


SINGLE_RANK_GROUP_ENV = "MOONDREAM_DIST_SINGLE_RANK_GROUP"


def single_rank_group() -> bool:
    """``MOONDREAM_DIST_SINGLE_RANK_GROUP=1``: a job of ONE rank still creates its process group and runs every
    collective of the N-rank path (weight broadcast, id gather, the all-reduces) over it.  Nothing to gain at run time --
    it exists so that the exact collective calls of the N-rank path can be executed over RCCL on a one-GPU box
    (tests/test_model_gpu.py, ``bench.py --gpus 1`` under torchrun); without it one rank means no process group at all."""
    return os.environ.get(SINGLE_RANK_GROUP_ENV, "0") not in ("", "0")


def collectives_on() -> bool:
    """True when the N-rank code path applies: a process group exists and has more than one rank (or one rank and
    ``single_rank_group()``)."""
    return dist.is_initialized() and (dist.get_world_size() > 1 or single_rank_group())


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torchrun environment; initialises the
    default process group when WORLD_SIZE > 1 (or at WORLD_SIZE == 1 under ``single_rank_group()``)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(default_master_port()) if world > 1 else str(free_port()))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def default_master_port() -> int:
    """MASTER_PORT when a launcher set WORLD_SIZE > 1 but no port (mpirun / srun style; torchrun always sets one).  Every
    rank must arrive at the SAME number without talking, and two jobs of one host should not both land on 29500: the port
    is derived from what the ranks of one job share and other jobs do not -- the launcher's job / run id if there is one,
    else the parent process id (the launcher's) and the user id."""
    import zlib

    for key in ("TORCHELASTIC_RUN_ID", "SLURM_JOB_ID", "PBS_JOBID", "LSB_JOBID", "OMPI_MCA_ess_base_jobid", "PMIX_NAMESPACE"):
        if os.environ.get(key):
            tag = f"{key}={os.environ[key]}"
            break
    else:
        tag = f"ppid={os.getppid()}"
    return 20000 + zlib.crc32(f"{tag}/uid={os.getuid()}".encode()) % 30000


# ------------------------------------------------------------------ one process per GPU on one host: who gets which cores
def _cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            out.extend(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_cpus(device_index: int) -> Tuple[Optional[int], List[int]]:
    """(NUMA node, its logical CPUs) of the GPU ``device_index`` from sysfs (``/sys/bus/pci/devices/<bdf>/numa_node`` and
    ``local_cpulist``), or (None, []) when the PCI address or sysfs is not available (no GPU, containers without sysfs)."""
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf}"
        with open(base + "/local_cpulist") as f:
            cpus = _cpulist(f.read())
        with open(base + "/numa_node") as f:
            node = int(f.read().strip())
        return (node if node >= 0 else None), cpus
    except Exception:
        return None, []


def plan_rank_cpus(local_rank: int, local_world: int, allowed: Sequence[int], numa_cpus_by_rank: Sequence[Sequence[int]]) -> List[int]:
    """Pure planning step of ``bind_rank_cpus`` (unit-tested on CPU).  ``numa_cpus_by_rank[r]``: CPUs local to rank r's GPU
    ([] = unknown).  The ranks whose GPUs sit on the same NUMA node split that node's allowed CPUs evenly, in rank order; a rank
    whose node is unknown (or whose share would be empty) gets an even slice of ALL allowed CPUs instead.  Never returns an
    empty list."""
    allowed = sorted(set(int(c) for c in allowed))
    mine = sorted(set(numa_cpus_by_rank[local_rank]) & set(allowed)) if local_rank < len(numa_cpus_by_rank) else []
    if mine:
        peers = [r for r in range(local_world) if r < len(numa_cpus_by_rank) and sorted(set(numa_cpus_by_rank[r]) & set(allowed)) == mine]
        share = shard_range(len(mine), peers.index(local_rank), len(peers))
        if len(share):
            return [mine[i] for i in share]
    share = shard_range(len(allowed), local_rank, max(1, local_world))
    return [allowed[i] for i in share] or allowed


def bind_rank_cpus(local_rank: int, local_world: int, use_gpu_topology: bool = True) -> dict:
    """Give this rank its own cores: 8 ranks x (a PIL tiling pool + launch threads + the pipelined engine's workers) on one
    host otherwise all float over all cores, migrate across NUMA nodes and fight for the same ones (review, round 5).  The
    process's affinity mask becomes its share of the CPUs local to its GPU's NUMA node (``plan_rank_cpus``); thread pools
    created afterwards size themselves from that mask.  ``MOONDREAM_BIND_CPUS=0`` turns it off.  Returns what was done."""
    if os.environ.get("MOONDREAM_BIND_CPUS", "1") in ("0", "") or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return {"bound": False, "reason": "one rank on this host" if local_world <= 1 else "disabled"}
    allowed = sorted(os.sched_getaffinity(0))
    by_rank: List[List[int]] = []
    nodes: List[Optional[int]] = []
    for r in range(local_world):
        node, cpus = gpu_numa_cpus(r) if (use_gpu_topology and torch.cuda.is_available() and r < torch.cuda.device_count()) else (None, [])
        by_rank.append(cpus)
        nodes.append(node)
    cpus = plan_rank_cpus(local_rank, local_world, allowed, by_rank)
    os.sched_setaffinity(0, cpus)
    torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))
    return {"bound": True, "cpus": len(cpus), "first": cpus[0], "last": cpus[-1], "numa_node": nodes[local_rank],
            "how": "share of the GPU's NUMA node" if by_rank[local_rank] else "even slice of the allowed CPUs (no GPU topology in sysfs)"}


def free_port() -> int:
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def relaunch_under_torchrun(n_procs: int, script: str, argv: Sequence[str]) -> Optional[int]:
    """``python bench.py --gpus N`` without a launcher: when N > 1 and no torchrun
    environment is present, re-run ``script`` as N ranks of ONE node under
    ``python -m torch.distributed.run`` (rendezvous on 127.0.0.1, a free port) and return
    the child's exit code; returns None when no relaunch is needed (N == 1, or this
    process already is a rank: WORLD_SIZE is set).  One process per GPU either way."""
    if n_procs <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import subprocess
    import sys

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_procs}", "--max-restarts=0",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    # the ranks inherit this process's stdout / stderr (torchrun does not redirect them), so a failing rank's
    # traceback is already on the caller's stderr; make the failure itself unmissable and non-zero
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f"{os.path.basename(script)}: a rank of the {n_procs}-process launch failed (torch.distributed.run exit code {rc}); "
              "its traceback is above", file=sys.stderr, flush=True)
        return rc if rc > 0 else 1
    return 0


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous block of items for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_state_dict(
    sd: Optional[Dict[str, torch.Tensor]],
    template: Dict[str, Tuple[Tuple[int, ...], torch.dtype]],
    device,
    src: int = 0,
) -> Dict[str, torch.Tensor]:
    """Rank ``src`` holds ``sd``; every rank returns a full copy on ``device``.

    ``template`` (name -> (shape, dtype)) is known everywhere from the config, so
    only payload bytes travel: all tensors are packed into ONE flat uint8 buffer
    and sent with a single broadcast."""
    if not collectives_on():
        assert sd is not None
        return {k: v.to(device) for k, v in sd.items()}
    names = sorted(template)
    sizes = [int(torch.empty(template[n][0], dtype=template[n][1]).numel()) * torch.empty((), dtype=template[n][1]).element_size() for n in names]
    offsets, total = [], 0
    for s in sizes:
        offsets.append(total)
        total += (s + 255) // 256 * 256
    flat = torch.empty(total, dtype=torch.uint8, device=device)
    if dist.get_rank() == src:
        assert sd is not None
        for n, o, s in zip(names, offsets, sizes):
            flat[o : o + s] = sd[n].to(device).contiguous().view(torch.uint8).reshape(-1)
    dist.broadcast(flat, src=src)
    out = {}
    for n, o, s in zip(names, offsets, sizes):
        shape, dtype = template[n]
        out[n] = flat[o : o + s].view(dtype).reshape(shape)
    return out


def state_dict_template(sd: Dict[str, torch.Tensor]) -> Dict[str, Tuple[Tuple[int, ...], torch.dtype]]:
    return {k: (tuple(v.shape), v.dtype) for k, v in sd.items()}


def gather_token_ids(local: torch.Tensor, dst: int = 0, n_total: Optional[int] = None) -> Optional[List[torch.Tensor]]:
    """local: int32 [B_local, T].  Rank ``dst`` gets the list of every rank's block
    (in rank order = image order under ``shard_range``); others get None.

    A GATHER to ``dst`` (not an all-gather: no other rank needs the ids).  Blocks may differ by one row;
    with ``n_total`` (the number of items ``shard_range`` split) every rank knows every block size and no
    size exchange is needed, otherwise one tiny all-reduce (MAX) sizes the padded blocks and the true row
    counts travel in the same gather as one extra row."""
    if not collectives_on():
        return [local]
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_total is not None:
        counts = [len(shard_range(n_total, r, world)) for r in range(world)]
        assert counts[rank] == local.shape[0], (counts, rank, local.shape)
        mx = max(counts)
    else:
        counts = None
        t = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mx = int(t[0])
    # row 0 carries this rank's row count, rows 1.. the ids (zero padded to the largest block)
    pad = torch.zeros(mx + 1, max(1, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[0, 0] = local.shape[0]
    pad[1 : 1 + local.shape[0], : local.shape[1]] = local
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r, b in enumerate(bufs):
        c = int(b[0, 0])
        if counts is not None and c != counts[r]:
            raise RuntimeError(f"rank {r} sent {c} rows, shard_range says {counts[r]}")
        out.append(b[1 : 1 + c, : local.shape[1]])
    return out


def ranks_seen(device) -> int:
    """An all-reduce (SUM) of ones: how many ranks actually took part in the collectives of this run."""
    if not collectives_on():
        return 1
    t = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t[0])


def gather_floats(value: float, device, dst: int = 0) -> Optional[List[float]]:
    """One float per rank, on rank ``dst`` (per-rank timings for the bench line); None elsewhere."""
    if not collectives_on():
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    bufs = [torch.zeros_like(t) for _ in range(dist.get_world_size())] if dist.get_rank() == dst else None
    dist.gather(t, bufs, dst=dst)
    return [float(b[0]) for b in bufs] if bufs is not None else None


def max_over_ranks(value: float, device) -> float:
    if not collectives_on():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def barrier():
    if collectives_on():
        if dist.get_backend() == "nccl":  # name the device: RCCL otherwise guesses it from the rank (and warns)
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
