"""Image-batch data parallelism as product code: one process per GPU, RCCL over xGMI.

What the reference offers for many images is a plain loop (``HfMoondream.batch_answer``, hf_moondream.py:99-103): every
image's vision pass, KV slab and decode loop is independent of every other image's.  ``DataParallelEngine`` is that loop
spread over the GPUs of one node:

  * start-up, once: rank 0 reads the checkpoint, every rank receives it as ONE flat buffer by one RCCL broadcast
    (``dist.broadcast_state_dict``: 3.85 GB at 2B, sized for the per-link xGMI bandwidth instead of ~600 small collectives);
  * per batch: rank r runs the lockstep engine of ``MoondreamModel`` over its contiguous block of the images
    (``dist.shard_range``), with NO collective on the data path;
  * end of batch: the int32 token ids (a few KB) are GATHERED on rank 0 (no other rank needs them); detect / point
    results, being ragged Python objects, travel by ``gather_object``.

Use, under ``torchrun --nproc-per-node N`` (or ``DataParallelEngine.launch`` to self-launch)::

    eng = DataParallelEngine(config, weights_file="model.safetensors")
    captions = eng.batch_caption(images)          # list on rank 0, None elsewhere; images: the same list on every rank
    ids = eng.batch_generate_ids(images, prompts, max_tokens=32)

Every method takes the GLOBAL work list on every rank and shards it itself; ``local=True`` means "these are already this
rank's items" (a loader that only materialises its own shard: ``bench.py``).  With one process (no WORLD_SIZE) everything
degenerates to the single-GPU model: no process group, no collective.
"""
from __future__ import annotations

import os

from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import dist as mdist
from .config import MoondreamConfig

StateDict = Dict[str, torch.Tensor]


def _default_model_factory(config, state_dict, device, **kw):
    from .moondream import MoondreamModel  # (needs the HIP library: imported only when a real model is built)

    return MoondreamModel(config, state_dict, device=device, **kw)


def pad_id_rows(ids: Sequence[Sequence[int]], width: int) -> torch.Tensor:
    """Ragged id lists -> int32 [len(ids), width], rows padded with -1 (host tensor)."""
    out = torch.full((len(ids), int(width)), -1, dtype=torch.int32)
    for r, row in enumerate(ids):
        n = len(row)
        if n > width:
            raise ValueError(f"sequence {r} has {n} ids, more than the block width {width}")
        if n:
            out[r, :n] = torch.as_tensor(list(row), dtype=torch.int32)
    return out


def strip_id_padding(blocks: Optional[Sequence[torch.Tensor]]) -> Optional[List[List[int]]]:
    """The blocks ``batch_generate_ids_pipelined`` yields on rank 0 -> the global ragged id lists (image order)."""
    if blocks is None:
        return None
    return [[int(t) for t in row if t >= 0] for b in blocks for row in b.tolist()]


class DataParallelEngine:
    """One instance per rank.  ``weights_file`` / ``state_dict`` are read on rank 0 only (other ranks may pass None);
    ``state_dict_fn(device)`` instead builds the checkpoint on EVERY rank at once (a synthetic or locally cached checkpoint:
    no rank idles while rank 0 reads) -- the broadcast of rank 0's copy still runs and, with ``verify_broadcast``, every rank
    compares the received bytes with its own copy, so the weight path over xGMI is exercised and checked.

    ``model_factory(config, state_dict, device, **model_kwargs)`` builds the per-rank model (default:
    ``MoondreamModel``); tests pass a stub to drive the engine over gloo without a GPU."""

    def __init__(self, config: MoondreamConfig, weights_file: Optional[str] = None, state_dict: Optional[StateDict] = None, *,
                 state_dict_fn: Optional[Callable[[torch.device], StateDict]] = None, verify_broadcast: bool = False,
                 backend: Optional[str] = None, device: Optional[torch.device] = None,
                 model_factory: Callable[..., Any] = _default_model_factory, **model_kwargs):
        self.rank, self.world, self.local_rank = mdist.init_from_env(backend)
        # the N-rank path (collectives) applies: more than one rank -- or ONE rank with MOONDREAM_DIST_SINGLE_RANK_GROUP=1, which runs
        # the same collective calls over a one-rank group (how the RCCL calls of this file are exercised on a one-GPU box)
        self._collective = mdist.collectives_on()
        if device is None:
            device = torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type == "cuda":
            torch.cuda.set_device(self.device)
        self.config = config
        self.weights_report: Optional[dict] = None
        # before any thread pool exists: this rank's share of the cores next to its GPU (dist.bind_rank_cpus)
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", self.world) or self.world)
        self.cpu_binding = mdist.bind_rank_cpus(self.local_rank, local_world, use_gpu_topology=self.device.type == "cuda")
        sd = self._distribute_weights(weights_file, state_dict, state_dict_fn, verify_broadcast)
        self.model = model_factory(config, sd, self.device, **model_kwargs)
        self._gather_stream = torch.cuda.Stream(device=self.device) if (self._collective and self.device.type == "cuda") else None

    # ------------------------------------------------------------------ launch
    @staticmethod
    def launch(n_procs: int, script: str, argv: Sequence[str]) -> Optional[int]:
        """``python script.py --gpus N`` with no launcher in front: re-execute ``script`` as N ranks of one node under
        ``torch.distributed.run`` (127.0.0.1, a free port) and return the exit code; None when this process already is a
        rank or N == 1 (``dist.relaunch_under_torchrun``)."""
        return mdist.relaunch_under_torchrun(n_procs, script, argv)

    # ------------------------------------------------------------------ weights
    def _distribute_weights(self, weights_file, state_dict, state_dict_fn, verify) -> StateDict:
        import time

        local: Optional[StateDict] = None
        if state_dict_fn is not None:
            local = state_dict_fn(self.device)                      # every rank, at once
        elif self.rank == 0:
            if state_dict is not None:
                local = state_dict
            elif weights_file is not None:
                from .weights import load_state_dict_file

                local = load_state_dict_file(weights_file)
            else:
                raise ValueError("DataParallelEngine needs weights_file, state_dict or state_dict_fn (on rank 0 at least)")
        if not self._collective:
            assert local is not None
            return {k: v.to(self.device) for k, v in local.items()}
        # (name -> shape, dtype): known to every rank that built a copy, sent from rank 0 otherwise (a few KB, once)
        if state_dict_fn is not None:
            template = mdist.state_dict_template(local)
        else:
            box = [mdist.state_dict_template(local) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, device=self.device if self.device.type == "cuda" else None)
            template = box[0]
        t0 = time.perf_counter()
        sd = mdist.broadcast_state_dict(local if self.rank == 0 else None, template, self.device, src=0)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        seconds = time.perf_counter() - t0
        n_bytes = sum(int(v.numel()) * v.element_size() for v in sd.values())
        report = {"bytes": n_bytes, "seconds": round(mdist.max_over_ranks(seconds, self.device), 3)}
        if verify and local is not None and state_dict_fn is not None:
            same = all(torch.equal(sd[k], local[k].to(self.device)) for k in local)
            ok = mdist.max_over_ranks(0.0 if same else 1.0, self.device) == 0.0
            report["equal_to_local_copy_on_every_rank"] = ok
            if not ok:
                raise RuntimeError(f"rank {self.rank}: broadcast weights differ from the locally built copy")
        self.weights_report = report
        return sd

    # ------------------------------------------------------------------ sharding / collectives
    def shard(self, n_items: int) -> range:
        """This rank's contiguous block of ``n_items`` work items (block sizes differ by at most one)."""
        return mdist.shard_range(n_items, self.rank, self.world)

    def _mine(self, items: Sequence, local: bool) -> Tuple[List, int]:
        """(this rank's items, global item count)."""
        if local:
            n_local = len(items)
            if not self._collective:
                return list(items), n_local
            t = torch.tensor([n_local], dtype=torch.int64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            n_total = int(t[0])
            if len(self.shard(n_total)) != n_local:
                raise ValueError(f"rank {self.rank}: local=True expects shard_range-sized blocks ({len(self.shard(n_total))} of "
                                 f"{n_total} items here), got {n_local}")
            return list(items), n_total
        rng = self.shard(len(items))
        return [items[i] for i in rng], len(items)

    def gather_ids(self, ids: Sequence[Sequence[int]], n_total: int) -> Optional[List[List[int]]]:
        """This rank's ragged id lists -> the global list (image order) on rank 0, None elsewhere.  One fixed-shape int32
        gather: rows are padded with -1 to the longest sequence of the job (one tiny all-reduce) and cut again on rank 0."""
        if not self._collective:
            return [list(map(int, s)) for s in ids]
        longest = torch.tensor([max((len(s) for s in ids), default=0)], dtype=torch.int64, device=self.device)
        dist.all_reduce(longest, op=dist.ReduceOp.MAX)
        width = max(1, int(longest[0]))
        host = torch.full((len(ids), width), -1, dtype=torch.int32)
        for r, s in enumerate(ids):
            if len(s):
                host[r, : len(s)] = torch.tensor(list(s), dtype=torch.int32)
        blocks = self.gather_id_blocks(host, n_total)
        if blocks is None:
            return None
        return [[int(t) for t in row if t >= 0] for b in blocks for row in b.tolist()]

    def gather_id_blocks(self, host_ids: torch.Tensor, n_total: int) -> Optional[List[torch.Tensor]]:
        """int32 [B_local, T] (host) -> every rank's block on rank 0.  On a GPU the copy and the RCCL gather run on their
        own stream: nothing of a step touches the default stream (a synchronous copy there waits for everything queued on
        the device, the next step's encode and decode included)."""
        if not self._collective:
            return [host_ids]
        if self._gather_stream is None:
            return mdist.gather_token_ids(host_ids.to(self.device), n_total=n_total)
        with torch.cuda.stream(self._gather_stream):
            dev_ids = host_ids.pin_memory().to(self.device, non_blocking=True)
            blocks = mdist.gather_token_ids(dev_ids, n_total=n_total)  # RCCL orders itself behind the current (= this) stream
            self._gather_stream.synchronize()
        return blocks

    def gather_objects(self, items: List[Any]) -> Optional[List[Any]]:
        """Ragged Python results (detect / point objects, strings) of this rank's block -> the global list on rank 0."""
        if not self._collective:
            return list(items)
        bufs = [None] * self.world if self.rank == 0 else None
        dist.gather_object(list(items), bufs, dst=0)
        if self.rank != 0:
            return None
        return [x for block in bufs for x in block]

    def barrier(self) -> None:
        mdist.barrier()

    def max_over_ranks(self, value: float) -> float:
        return mdist.max_over_ranks(value, self.device)

    def ranks_seen(self) -> int:
        return mdist.ranks_seen(self.device)

    def gather_floats(self, value: float) -> Optional[List[float]]:
        return mdist.gather_floats(value, self.device)

    def close(self) -> None:
        if self._collective and dist.is_initialized():
            dist.destroy_process_group()

    # ------------------------------------------------------------------ the batched API, sharded
    def batch_generate_ids(self, images: Sequence, prompts: Sequence[Sequence[int]], max_tokens: int = 768,
                           ignore_eos: bool = False, local: bool = False) -> Optional[List[List[int]]]:
        """``MoondreamModel.batch_generate_ids`` over all GPUs: element i of the result == what one GPU returns for
        (images[i], prompts[i]) (the engine's kernels are batch-invariant).  Result on rank 0, None elsewhere."""
        if len(images) != len(prompts):
            raise ValueError("one prompt per image")
        mine_img, n_total = self._mine(images, local)
        mine_pr, _ = (list(prompts), n_total) if local else self._mine(prompts, False)
        ids = self.model.batch_generate_ids(mine_img, mine_pr, max_tokens=max_tokens, ignore_eos=ignore_eos) if mine_img else []
        return self.gather_ids(ids, n_total)

    def batch_generate_ids_pipelined(self, local_batches: Iterable[Tuple[Sequence, Sequence[Sequence[int]]]], n_total: int,
                                     max_tokens: int = 768, ignore_eos: bool = False):
        """Generator: the two-stream pipelined engine of the model (encode of batch k+1 under the decode of batch k) over
        THIS rank's batches (``local_batches``: (images, prompt ids) per step, already this rank's block of an
        ``n_total``-image global batch); yields per step the list of every rank's int32 id block on rank 0 (rank order =
        image order), None elsewhere.  Every step's gather completes inside the step.

        A block is int32 [B_rank, max_tokens] ALWAYS: sequences that stopped at an EOS (``ignore_eos=False``) are ragged on a
        rank and differently long across ranks, and a gather needs the same shape everywhere without a size exchange per
        step -- rows are padded with -1 (never a token id) to ``max_tokens``; ``strip_id_padding`` cuts them again."""
        for ids in self.model.batch_generate_ids_pipelined(local_batches, max_tokens=max_tokens, ignore_eos=ignore_eos):
            yield self.gather_id_blocks(pad_id_rows(ids, max_tokens), n_total)

    def _strings(self, fn_name: str, images: Sequence, texts: Optional[Sequence], local: bool, **kw) -> Optional[List[Any]]:
        mine_img, _ = self._mine(images, local)
        fn = getattr(self.model, fn_name)
        if texts is None:
            out = fn(mine_img, **kw) if mine_img else []
        else:
            if len(texts) != len(images):
                raise ValueError("one text per image")
            mine_txt = list(texts) if local else self._mine(texts, False)[0]
            out = fn(mine_img, mine_txt, **kw) if mine_img else []
        return self.gather_objects(list(out))

    def batch_caption(self, images: Sequence, length: str = "normal", settings: Optional[dict] = None, local: bool = False):
        """reference: a loop of caption() (moondream.py:625-651) -> List[str] on rank 0."""
        return self._strings("batch_caption", images, None, local, length=length, settings=settings)

    def batch_query(self, images: Sequence, questions: Sequence[str], settings: Optional[dict] = None, local: bool = False):
        """reference: HfMoondream.batch_answer's loop of query() (hf_moondream.py:99-103) -> List[str] on rank 0."""
        return self._strings("batch_query", images, questions, local, settings=settings)

    def batch_detect(self, images: Sequence, objects: Sequence[str], settings: Optional[dict] = None, local: bool = False):
        """reference: a loop of detect() (moondream.py:735-781) -> List[{"objects": [...]}] on rank 0."""
        return self._strings("batch_detect", images, objects, local, settings=settings)

    def batch_detect_pipelined(self, local_batches: Iterable[Tuple[Sequence, Sequence[str]]], settings: Optional[dict] = None,
                               kind: str = "detect"):
        """Generator: ``MoondreamModel.batch_detect_pipelined`` (the next batch's host tiling under the current batch's GPU
        time) over THIS rank's batches; yields per step the global result list on rank 0 (rank order = image order), None
        elsewhere."""
        for res in self.model.batch_detect_pipelined(local_batches, settings=settings, kind=kind):
            yield self.gather_objects(list(res))

    def batch_point(self, images: Sequence, objects: Sequence[str], settings: Optional[dict] = None, local: bool = False):
        """reference: a loop of point() (moondream.py:783-829) -> List[{"points": [...]}] on rank 0."""
        return self._strings("batch_point", images, objects, local, settings=settings)
