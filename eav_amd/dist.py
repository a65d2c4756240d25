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
``subject_schedule`` composes the two levels: whole rounds of subjects run one per
rank, and the ``n_subjects mod world`` subjects that would leave most ranks idle in
a last round are each trained by a GROUP of ranks with the batch split among them
(gradient all-reduce inside the group only) - 42 subjects on 8 GPUs: 5 rounds + two
4-rank groups instead of a sixth round on two GPUs.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=False):
    """Initialise the default process group from torchrun's environment.
    Returns (rank, world_size, local_rank).  backend: 'nccl' (= RCCL) on GPUs, 'gloo' for CPU tests.
    force: create the group even for WORLD_SIZE = 1 (a one-rank RCCL communicator: every collective of the N > 1 path runs
    through the real backend on a single GPU - tests/test_rccl_gpu.py)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
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


class SubjectSchedule:
    """Static plan of the per-subject loop (Dataload_audio.py:82, EEGNet_tor.py:146, Transformer_Vision.py:136:
    `for sub in range(1, 43)`) on `world` ranks.

    solo[r]   subjects rank r trains alone, full batch, no collective (whole rounds, round-robin)
    groups    [(subject, [ranks])] for the remainder: every listed rank holds a replica and 1 / len(ranks) of each batch,
              one gradient all-reduce per step inside the group (GradSync(group=...)); ranks in no group idle for that
              tail.  A group of one rank is a plain solo training.
    """

    def __init__(self, world, n_subjects=42, first=1, hybrid=True):
        if world < 1 or n_subjects < 0:
            raise ValueError("subject_schedule: world >= 1 and n_subjects >= 0")
        self.world, self.n_subjects, self.first, self.hybrid = world, n_subjects, first, hybrid
        self.rounds, rem = divmod(n_subjects, world)
        self.solo = [[first + k * world + r for k in range(self.rounds)] for r in range(world)]
        tail = [first + self.rounds * world + j for j in range(rem)]
        gsize = (world // rem) if (rem and hybrid) else 1
        self.group_size = gsize if rem else 0
        self.groups = [(s, list(range(j * gsize, (j + 1) * gsize))) for j, s in enumerate(tail)]

    def group_of(self, rank):
        """(subject, ranks) of the group `rank` belongs to, or None."""
        for s, ranks in self.groups:
            if rank in ranks:
                return s, ranks
        return None

    def subjects_of(self, rank):
        """Every subject this rank takes part in (solo first, then its group's)."""
        g = self.group_of(rank)
        return self.solo[rank] + ([g[0]] if g else [])

    def rounds_of_work(self, group_step_cost=None):
        """Length of the plan in units of one solo training: whole rounds + the tail.  group_step_cost: time of a group
        training relative to a solo one (default: perfect scaling inside the group, 1 / group_size)."""
        if not self.groups:
            return float(self.rounds)
        cost = (1.0 / self.group_size) if group_step_cost is None else float(group_step_cost)
        return self.rounds + cost

    def ideal_speedup(self, group_step_cost=None):
        """n_subjects solo trainings on one rank against this plan."""
        r = self.rounds_of_work(group_step_cost)
        return self.n_subjects / r if r > 0 else 0.0

    def make_groups(self):
        """Create the process sub-groups (every rank must call this, in the same order - torch.distributed.new_group is
        collective) and return {subject: group} for the groups of more than one rank."""
        out = {}
        for s, ranks in self.groups:
            if len(ranks) > 1:
                out[s] = dist.new_group(ranks=ranks)
        return out


def subject_schedule(world, n_subjects=42, first=1, hybrid=True):
    return SubjectSchedule(world, n_subjects, first, hybrid)


class GradSync:
    """All-reduce of flat gradient buffers into the gradient of the GLOBAL batch; call between backward and
    optimizer.step().

    Every rank's loss is the mean over its local shard, so the global-batch gradient is sum_r (n_r / n) g_r: each
    rank scales its gradient by `weight` = n_r / n (set_batch(); default 1 / world for equal shards) and the
    collective is a plain SUM.  Two ways to use it.  (1) `sync()` after the backward: one all-reduce per flat buffer
    (EEGNet: 0.68 MB, pure latency).  (2) Overlapped: hand `sync.bucket` to a model as its `grad_ready_hook`; the
    encoders call it from inside the backward as soon as a layer's contiguous slice of the flat gradient buffer is
    final (last layer first), so the RCCL transfers of ~28 MB buckets run on the communicator's stream under the
    remaining backward kernels; `sync()` then only waits for the outstanding work and reduces what is left."""

    def __init__(self, flat_grads, group=None, force=False):
        """force: issue the collectives even in a one-rank group (exercises the backend; the sum over one rank is the
        identity, so results must be bit-equal to the unsynchronised run)."""
        self.flat_grads = list(flat_grads)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.enabled = self.world > 1 or (force and dist.is_initialized())
        self.collectives = 0     # all-reduce calls issued (bookkeeping, like bytes_reduced)
        self.weight = 1.0 / self.world
        self._pending = []
        self._done = []          # (buffer index, lo, hi) ranges already submitted during this backward
        self.active = None       # optional {buffer index: [(lo, hi), ...]}: the only slices that carry gradients
                                 # (frozen fine-tuning phase: the classifier head), default = whole buffers
        self.bytes_reduced = 0   # bookkeeping for tests / the bench report

    def set_batch(self, local_n, global_n):
        """This rank contributed local_n of the global batch's global_n items (uneven shards, ragged last batch)."""
        self.weight = float(local_n) / float(global_n)

    def _reduce(self, view, async_op):
        if self.weight != 1.0:
            view.mul_(self.weight)
        self.bytes_reduced += 4 * view.numel()
        self.collectives += 1
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def bucket(self, lo, hi, buffer=0):
        """Asynchronously all-reduce flat_grads[buffer][lo:hi] (elements)."""
        if not self.enabled or hi <= lo:
            return
        self._pending.append(self._reduce(self.flat_grads[buffer][lo:hi], True))
        self._done.append((buffer, lo, hi))

    def __call__(self):
        if not self.enabled:
            return
        for work in self._pending:
            work.wait()
        for i, g in enumerate(self.flat_grads):
            wanted = sorted(self.active[i]) if self.active is not None and i in self.active else [(0, g.numel())]
            for wlo, whi in wanted:
                # the parts of [wlo, whi) that no bucket covered, walked in order
                covered = sorted((max(lo, wlo), min(hi, whi)) for b, lo, hi in self._done if b == i)
                pos = wlo
                for lo, hi in covered:
                    if hi <= lo:              # bucket entirely outside the wanted range
                        continue
                    if lo > pos:
                        self._reduce(g[pos:lo], False)
                    pos = max(pos, hi)
                if pos < whi:
                    self._reduce(g[pos:whi], False)
        self._pending, self._done = [], []

    def set_active(self, ranges, buffer=0):
        """Restrict the synchronised part of a buffer to `ranges` (None = everything)."""
        if ranges is None:
            if self.active is not None:
                self.active.pop(buffer, None)
        else:
            self.active = self.active or {}
            self.active[buffer] = [(int(a), int(b)) for a, b in ranges]


def backend_name():
    """'nccl' (= RCCL on ROCm), 'gloo', or 'none' when no process group exists - what reports must print instead of
    assuming RCCL."""
    return dist.get_backend() if dist.is_initialized() else "none"


def attach(trainer, force=False, group=None):
    """Give a Trainer_uni (or any trainer exposing .model with a flat gradient buffer) a gradient
    all-reduce when running under torchrun with WORLD_SIZE > 1 (force: also in a one-rank group).
    group: a process sub-group (SubjectSchedule.make_groups) - the replicas of ONE subject's training."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return trainer
    model = trainer.model
    model._ensure_flat()
    trainer.grad_sync = GradSync([model._flat[1]], group=group, force=force)
    if hasattr(model, "grad_ready_hook"):
        model.grad_ready_hook = trainer.grad_sync.bucket     # overlap the all-reduce with the backward
    # hipGraph replay stays on (Trainer_uni / GraphStep): with a grad_sync the step is captured as TWO graphs - batch
    # gather + forward + loss + backward, and the fused Adam update - with the all-reduce issued eagerly between them
    return trainer


def replica_shard(n_items, batch_size, i, n):
    """Training items and batch size of member i of an n-rank group that trains ONE model through a per-replica data loader
    (tools/run_*_subjects.py): (slice over the items, per-member batch size).  Every member gets the SAME number of items
    (the n_items mod n last ones are dropped, like a DistributedSampler with drop_last) and the same batch size
    batch_size / n, hence the same number of optimiser steps - members that disagree on the step count dead-lock in the
    gradient all-reduce - and equal-weight averaging (GradSync's default 1 / n) is then exact for every batch.  A group larger
    than the batch, or one that does not divide it, would silently change the global batch: refused."""
    if n < 1 or not 0 <= i < n:
        raise ValueError("replica_shard: member index outside the group")
    if n == 1:
        return slice(0, n_items), batch_size
    if batch_size % n:
        raise ValueError(f"replica_shard: a group of {n} ranks cannot split the batch of {batch_size} evenly")
    per = n_items // n
    if per == 0:
        raise ValueError(f"replica_shard: {n_items} items for {n} ranks")
    return slice(i, i + per * n, n), batch_size // n


def shard_batch(n_items, rank, world):
    """Contiguous shard [lo, hi) of a global batch of n_items for this rank."""
    per = (n_items + world - 1) // world
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)
