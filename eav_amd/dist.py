"""Multi-GPU: one process per GPU, RCCL over xGMI through torch.distributed.

The reference's only multi-device mechanism is single-process ``nn.DataParallel``
(EEGNet_tor.py:86-88, Transformer_Audio.py:59-60, Transformer_Vision.py:82-83):
per step it scatters the batch, re-broadcasts every parameter, gathers outputs
and reduce-adds gradients onto GPU 0.  Here each rank owns a replica and a shard
of the batch; the only exchange is one all-reduce of the flat gradient buffer per
optimiser step (mean over ranks), after which every rank applies the identical
fused Adam update - no parameter broadcast is ever needed.  BatchNorm statistics
stay per replica, which is DataParallel's behaviour too.

Subject-level sharding (42 independent per-subject trainings, SURVEY 8e level 1)
needs no collective at all: ``subjects_for_rank`` assigns subjects round-robin.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment.
    Returns (rank, world_size, local_rank).  backend: 'nccl' (= RCCL) on GPUs, 'gloo' for CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def subjects_for_rank(rank, world, n_subjects=42, first=1):
    """Static round-robin partition of the per-subject loop (for sub in range(1, 43))."""
    return [s for s in range(first, first + n_subjects) if (s - first) % world == rank]


class GradSync:
    """All-reduce (mean) of flat gradient buffers; call between backward and optimizer.step()."""

    def __init__(self, flat_grads, group=None):
        self.flat_grads = list(flat_grads)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self):
        if self.world == 1:
            return
        for g in self.flat_grads:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            g.mul_(1.0 / self.world)


def attach(trainer):
    """Give a Trainer_uni (or any trainer exposing .model with a flat gradient buffer) a gradient
    all-reduce when running under torchrun with WORLD_SIZE > 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return trainer
    model = trainer.model
    model._ensure_flat()
    trainer.grad_sync = GradSync([model._flat[1]])
    if hasattr(trainer, "use_graph"):
        trainer.use_graph = False     # the RCCL all-reduce stays outside hipGraph capture
    return trainer


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch of n_items for this rank."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)
