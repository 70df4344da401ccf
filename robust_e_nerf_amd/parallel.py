"""Data parallelism over event-ray shards: one process per GPU, `torch.distributed` with backend
"nccl" (= RCCL over xGMI on ROCm), or "gloo" for the CPU tests.

The reference uses Lightning DDP (scripts/run.py:81-93): every rank draws its own i.i.d. event
batch (data/datamodule.py:85-89, seed + rank) and DDP averages the gradients of the replicated
parameters.  Here the parameters live in ONE flat buffer, so the exchange is a single all-reduce
(SUM) of 50.4 MB (+ a 16-byte scalar block); the 1/world factor is applied inside the fused Adam
kernel (`ren_adam_step(grad_scale=1/world)`), so no extra pass touches the gradient.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world == 1 or dist.is_initialized():
        return int(os.environ.get("RANK", 0)), world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    dist.init_process_group(backend=backend)
    return dist.get_rank(), world


def allreduce_sum_(buffers: Iterable[torch.Tensor], group=None, world_size: Optional[int] = None) -> None:
    """In-place SUM all-reduce of each flat gradient buffer (C1 of SURVEY 2.3)."""
    if world_size is None:
        world_size = dist.get_world_size(group) if dist.is_initialized() else 1
    if world_size <= 1:
        return
    for b in buffers:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)


# tail of the packed gradient buffer [table | MLP | pad | AUX]: everything else a step has to sum over the ranks
AUX_FLOATS = 16
AUX_SMALL, AUX_CT, AUX_TAU_HI, AUX_TAU_LO, AUX_MEAN_S = 0, 4, 5, 6, 7     # small-parameter grads [0:4], C_p ratio, tau (two floats), samples / ray


class GradSync:
    """Gradient exchange of one step: the packed buffer [table | MLP | pad | aux] in ONE all-reduce (SUM), or -- with
    `early(...)` calls from the backward pass -- the slices that are final early in their own asynchronous all-reduce,
    running on RCCL's stream beside the rest of the backward; `finish()` reduces what is left and waits for all of it.
    Replaces DDP's bucketed all-reduce (scripts/run.py:81-93)."""

    def __init__(self, group=None, world_size: Optional[int] = None, compress: Optional[str] = None, n_exact_tail: int = AUX_FLOATS):
        """compress="bf16": the parameter gradients travel as bfloat16 (half the bytes per link: SURVEY 8(e), for strong
        scaling where the all-reduce is a visible fraction of the step); the last `n_exact_tail` floats (the aux block:
        tau as a float pair, counters) always travel as float32."""
        if compress not in (None, "bf16"):
            raise NotImplementedError(f"gradient compression {compress!r}")
        self.group = group
        self.world = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.compress, self.n_exact_tail = compress, n_exact_tail
        self.pending = []
        self.done = []                                           # (start, stop) element ranges already launched
        self.collectives = 0                                     # launched in the current step (tests / DESIGN numbers)

    def _launch(self, view: torch.Tensor, exact: bool):
        if self.compress == "bf16" and not exact:
            half = view.to(torch.bfloat16)
            work = dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((work, view, half))
        else:
            self.pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, None))
        self.collectives += 1

    def early(self, buf: torch.Tensor, start: int, stop: int):
        if self.world <= 1 or stop <= start:
            return
        assert stop <= buf.numel() - self.n_exact_tail
        # a range is reduced ONCE per optimiser step: a second launch would sum the already rank-summed values again
        assert all(stop <= a or b <= start for a, b in self.done), \
            f"gradient range [{start}, {stop}) was already all-reduced in this step: {self.done}"
        self._launch(buf[start:stop], exact=False)
        self.done.append((start, stop))

    def finish(self, buf: torch.Tensor):
        """all-reduce every element range of `buf` not launched by early(), wait for everything"""
        if self.world <= 1:
            return
        n_cmp = buf.numel() - (self.n_exact_tail if self.compress else 0)      # [0, n_cmp) may be compressed
        pos = 0
        for a, b in sorted(self.done) + [(n_cmp, n_cmp)]:
            if a > pos:
                self._launch(buf[pos:a], exact=False)
            pos = max(pos, b)
        if n_cmp < buf.numel():
            self._launch(buf[n_cmp:], exact=True)
        for work, view, half in self.pending:
            work.wait()
            if half is not None:
                view.copy_(half)
        self.pending, self.done = [], []

    def reset_count(self) -> int:
        c, self.collectives = self.collectives, 0
        return c


def rank_seed(base_seed: int, rank: int) -> int:
    """Per-rank data seed (data/datamodule.py:85-89: process_seed = initial_seed + rank)."""
    return base_seed + rank


def shard_bounds(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard of a global batch (used by the 1-vs-N equivalence tests)."""
    per = (n + world - 1) // world
    return min(rank * per, n), min((rank + 1) * per, n)


def per_rank_budget(eff_ray_sample_batch_size: int, world: int) -> int:
    """train_eff_ray_sample_batch_size // num_gpus (models/robust_e_nerf.py:63-66)."""
    return eff_ray_sample_batch_size // world


def allgather_mean(value: float, group=None) -> float:
    """mean over ranks of a python scalar (C2: mean_num_samples_per_ray, robust_e_nerf.py:916-919)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(value)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    t = torch.tensor([float(value)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, group=group)
    return float(t) / dist.get_world_size(group)


def new_train_batch_size(budget_per_rank: int, render_means, gather_mean=None, accumulate_grad_batches: int = 1,
                         batch_index: int = 0):
    """RobustENeRF.update_train_batch_size (robust_e_nerf.py:907-950): mean over this step's renders of their mean
    number of samples per ray (grad / start / end, each its own n / R), mean over ranks (`gather_mean`), and -- unless
    gradient accumulation is on and this is not its second-to-last micro-batch -- budget // mean as the new per-rank
    event batch size.  -> (mean over renders and ranks, new batch size or None)."""
    means = [float(m) for m in render_means]
    mean = sum(means) / len(means)
    if gather_mean is not None:
        mean = float(gather_mean(mean))
    if accumulate_grad_batches > 1 and batch_index % accumulate_grad_batches != accumulate_grad_batches - 2:
        return mean, None
    return mean, int(budget_per_rank / mean)
