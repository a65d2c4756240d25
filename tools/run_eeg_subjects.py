#!/usr/bin/env python3
"""The reference's per-subject EEG driver (CNN_torch/EEGNet_tor.py:144-181) on MI355X, subject-sharded.

    python tools/run_eeg_subjects.py [--subjects 42] [--epochs 3] [--samples 500]
    python -m torch.distributed.run --nproc-per-node 8 tools/run_eeg_subjects.py ...

Each rank trains its share of the 42 independent subjects with the reference's hyper-parameters (lr 1e-5, batch 32) on
synthetic recordings (eav_amd.synth.eeg_subject -> EAVDataSplit -> EEGNet_tor -> Trainer_uni), then the per-subject test
accuracies are gathered on rank 0.  Placement: eav_amd.dist.SubjectSchedule - whole rounds one subject per rank with no
gradient traffic (SURVEY.md section 8e level 1), the 42 mod N remaining subjects on groups of ranks (every member a replica on
every n-th training trial with 1 / n of the batch, gradients all-reduced inside the group: the reference's DataParallel
wrap, EEGNet_tor.py:86-88); --no-hybrid: plain round-robin.  With real data, replace `synthetic_subject` by DataLoadEEG(...).prepare_data().
"""
import argparse
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import dist as eav_dist, synth  # noqa: E402
from eav_amd.datasplit import EAVDataSplit  # noqa: E402
from eav_amd.eegnet import EEGNet_tor, Trainer_uni  # noqa: E402


def synthetic_subject(sub, samples):
    x, y = synth.eeg_subject(sub, 400, 30, samples)       # 400 windows of 5 s, labels 0..4
    return x, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--subjects", type=int, default=42)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--samples", type=int, default=500)
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--no-hybrid", action="store_true", help="plain round-robin: the remainder one subject per rank")
    args = ap.parse_args()
    if "EAV_FORCE_DEVICE" in os.environ:                      # several ranks on one GPU (logic runs on a 1-GPU box)
        os.environ["LOCAL_RANK"] = os.environ["EAV_FORCE_DEVICE"]
    rank, world, local = eav_dist.init_from_env(os.environ.get("EAV_DIST_BACKEND"))
    torch.cuda.set_device(local)
    sched = eav_dist.subject_schedule(world, args.subjects, hybrid=not args.no_hybrid)
    groups = sched.make_groups() if world > 1 else {}
    grp = sched.group_of(rank)
    plan = [(s, None) for s in sched.solo[rank]] + ([grp] if grp else [])
    mine = [s for s, _ in plan]
    results, t0 = {}, time.perf_counter()
    for sub, ranks in plan:
        x, y = synthetic_subject(sub, args.samples)
        tr_x, tr_y, te_x, te_y = EAVDataSplit(x, y).get_split()               # h_idx = 40 -> 200 / 200
        n = len(ranks) if ranks else 1
        # this replica's share of the training trials: equal shard lengths and batch sizes on every member, hence equal
        # step counts (eav_amd.dist.replica_shard refuses a group that does not divide the batch of 32)
        sl, bs = eav_dist.replica_shard(len(tr_x), 32, ranks.index(rank) if n > 1 else 0, n)
        tr_x, tr_y = tr_x[sl], tr_y[sl]
        data = [torch.from_numpy(tr_x).float().unsqueeze(1), tr_y, torch.from_numpy(te_x).float().unsqueeze(1), te_y]
        torch.manual_seed(sub)
        model = EEGNet_tor(nb_classes=5, D=8, F2=64, Chans=30, kernLength=300, Samples=args.samples, dropoutRate=0.5)
        trainer = Trainer_uni(model=model, data=data, lr=1e-5, batch_size=bs, num_epochs=args.epochs)
        if n > 1:
            eav_dist.attach(trainer, group=groups[sub])       # gradient all-reduce inside the group
        with contextlib.redirect_stdout(io.StringIO() if args.quiet else sys.stdout):
            trainer.train()
        model.eval()
        with torch.no_grad():
            pred = model(trainer.test_dataloader.x).argmax(dim=1)
            if n == 1 or ranks[0] == rank:                    # one report per subject
                results[sub] = float((pred == trainer.test_dataloader.y).float().mean().item())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gathered = [None] * world
    if world > 1:
        torch.distributed.all_gather_object(gathered, results)
    else:
        gathered = [results]
    if rank == 0:
        allr = {k: v for d in gathered for k, v in d.items()}
        steps = len(mine) * args.epochs * 7
        print(json.dumps({"subjects": len(allr), "world": world, "seconds": round(dt, 2), "mean_test_acc": round(float(np.mean(list(allr.values()))), 4),
                          "rank0_train_steps_per_s": round(steps / dt, 1)}))


if __name__ == "__main__":
    main()
