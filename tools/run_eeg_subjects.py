#!/usr/bin/env python3
"""The reference's per-subject EEG driver (CNN_torch/EEGNet_tor.py:144-181) on MI355X, subject-sharded.

    python tools/run_eeg_subjects.py [--subjects 42] [--epochs 3] [--samples 500]
    python -m torch.distributed.run --nproc-per-node 8 tools/run_eeg_subjects.py ...

Each rank trains its share of the 42 independent subjects (round-robin, no gradient traffic - SURVEY.md section 8e
level 1) with the reference's hyper-parameters (lr 1e-5, batch 32) on synthetic recordings
(eav_amd.synth.eeg_subject -> EAVDataSplit -> EEGNet_tor -> Trainer_uni), then the per-subject test accuracies
are gathered on rank 0.  With real data, replace `synthetic_subject` by DataLoadEEG(...).prepare_data().
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
    args = ap.parse_args()
    rank, world, local = eav_dist.init_from_env()
    torch.cuda.set_device(local)
    mine = eav_dist.subjects_for_rank(rank, world, args.subjects)
    results, t0 = {}, time.perf_counter()
    for sub in mine:
        x, y = synthetic_subject(sub, args.samples)
        tr_x, tr_y, te_x, te_y = EAVDataSplit(x, y).get_split()               # h_idx = 40 -> 200 / 200
        data = [torch.from_numpy(tr_x).float().unsqueeze(1), tr_y, torch.from_numpy(te_x).float().unsqueeze(1), te_y]
        torch.manual_seed(sub)
        model = EEGNet_tor(nb_classes=5, D=8, F2=64, Chans=30, kernLength=300, Samples=args.samples, dropoutRate=0.5)
        trainer = Trainer_uni(model=model, data=data, lr=1e-5, batch_size=32, num_epochs=args.epochs)
        with contextlib.redirect_stdout(io.StringIO() if args.quiet else sys.stdout):
            trainer.train()
        model.eval()
        with torch.no_grad():
            pred = model(trainer.test_dataloader.x).argmax(dim=1)
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
