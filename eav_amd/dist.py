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
    """All-reduce (mean) of flat gradient buffers; call between backward and optimizer.step().

    Two ways to use it.  (1) `sync()` after the backward: one all-reduce per flat buffer (EEGNet: 0.68 MB,
    pure latency).  (2) Overlapped: hand `sync.bucket` to a model as its `grad_ready_hook`; the encoders call
    it from inside the backward as soon as a layer's contiguous slice of the flat gradient buffer is final
    (last layer first), so the RCCL transfers of ~28 MB buckets run on the communicator's stream under the
    remaining backward kernels; `sync()` then only waits for the outstanding work and reduces what is left."""

    def __init__(self, flat_grads, group=None):
        self.flat_grads = list(flat_grads)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending = []
        self._done = []          # (buffer index, lo, hi) ranges already submitted during this backward
        self.active = None       # optional {buffer index: [(lo, hi), ...]}: the only slices that carry gradients
                                 # (frozen fine-tuning phase: the classifier head), default = whole buffers

    def bucket(self, lo, hi, buffer=0):
        """Asynchronously all-reduce flat_grads[buffer][lo:hi] (elements)."""
        if self.world == 1 or hi <= lo:
            return
        view = self.flat_grads[buffer][lo:hi]
        self._pending.append((dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True), view))
        self._done.append((buffer, lo, hi))

    def __call__(self):
        if self.world == 1:
            return
        inv = 1.0 / self.world
        for work, view in self._pending:
            work.wait()
            view.mul_(inv)
        for i, g in enumerate(self.flat_grads):
            covered = sorted((lo, hi) for b, lo, hi in self._done if b == i)
            wanted = sorted(self.active[i]) if self.active is not None and i in self.active else [(0, g.numel())]
            for wlo, whi in wanted:
                pos = wlo
                for lo, hi in covered + [(whi, whi)]:
                    lo, hi = max(lo, wlo), min(hi, whi)
                    if lo > pos:                  # a gap no bucket covered
                        dist.all_reduce(g[pos:lo], op=dist.ReduceOp.SUM, group=self.group)
                        g[pos:lo].mul_(inv)
                    pos = max(pos, hi)
        self._pending, self._done = [], []

    def set_active(self, ranges, buffer=0):
        """Restrict the synchronised part of a buffer to `ranges` (None = everything)."""
        if ranges is None:
            if self.active is not None:
                self.active.pop(buffer, None)
        else:
            self.active = self.active or {}
            self.active[buffer] = [(int(a), int(b)) for a, b in ranges]


def attach(trainer):
    """Give a Trainer_uni (or any trainer exposing .model with a flat gradient buffer) a gradient
    all-reduce when running under torchrun with WORLD_SIZE > 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return trainer
    model = trainer.model
    model._ensure_flat()
    trainer.grad_sync = GradSync([model._flat[1]])
    if hasattr(model, "grad_ready_hook"):
        model.grad_ready_hook = trainer.grad_sync.bucket     # overlap the all-reduce with the backward
    if hasattr(trainer, "use_graph"):
        trainer.use_graph = False     # the RCCL all-reduce stays outside hipGraph capture
    return trainer


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch of n_items for this rank."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)
