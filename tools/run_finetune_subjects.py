"""The reference's per-subject AUDIO and VISION drivers (Dataload_audio.py:80-115, Transformer_torch/Transformer_Vision.py:
132-188) on MI355X, subject-sharded, through the public trainers.

    python tools/run_finetune_subjects.py audio  [--subjects 42] [--frozen-epochs 10] [--unfrozen-epochs 15]
    python tools/run_finetune_subjects.py vision [--subjects 42] [--frozen-epochs 10] [--unfrozen-epochs 5]
    python -m torch.distributed.run --nproc-per-node 8 tools/run_finetune_subjects.py audio ...

Every subject is one `AudioModelTrainer` / `ImageClassifierTrainer` run with the reference's hyper-parameters
(audio: batch 8, train(10, 5e-4, freeze=True) + train(15, 5e-6, freeze=False); vision: batch 128, 10 + 5 epochs) on a
synthetic subject of the reference's sizes (280 + 120 clips of 5 s at 16 kHz; 200 + 200 trials x 25 frames of 56 x 56).
eav_amd.dist.SubjectSchedule places them: whole rounds one subject per rank with no collective, the remainder on groups of
ranks - every member holds a replica, every n-th training item and 1 / n of the batch size, gradients all-reduced inside the
group (the reference's nn.DataParallel wrap, Transformer_Audio.py:59-60 / Transformer_Vision.py:82-83).  At the end ONE
all_gather of `outputs_test` (SURVEY 8e level 1); vision adds the trial vote + weighted F1 of Transformer_Vision.py:174-185.
With real data, replace `synthetic_subject` by the pickles the reference drivers read.  The model directory is an HF-format
directory (`--model-path`); without one a random-init full-size model is written to a temporary directory.
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from eav_amd import dist as eav_dist, synth  # noqa: E402


def synthetic_subject(kind, sub, small):
    if kind == "audio":
        ntr, nte = (16, 8) if small else (280, 120)
        wav = synth.normal(100 + sub, (ntr + nte, 80000), 0.0, 0.1)
        y = synth.labels(200 + sub, ntr + nte)
        return [wav[:ntr], y[:ntr], wav[ntr:], y[ntr:]]
    ntri = 8 if small else 200
    fr = synth.uniform(300 + sub, (2 * ntri, 25, 56, 56, 3), 0, 256).astype(np.uint8)
    y = synth.labels(400 + sub, 2 * ntri)
    return [fr[:ntri], y[:ntri], fr[ntri:], y[ntri:]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("kind", choices=("audio", "vision"))
    ap.add_argument("--subjects", type=int, default=42)
    ap.add_argument("--frozen-epochs", type=int, default=10)
    ap.add_argument("--unfrozen-epochs", type=int, default=None)
    ap.add_argument("--model-path", default=None)
    ap.add_argument("--small", action="store_true", help="tiny synthetic subjects (logic check)")
    ap.add_argument("--no-hybrid", action="store_true", help="plain round-robin: the remainder one subject per rank")
    ap.add_argument("--backend", default=os.environ.get("EAV_DIST_BACKEND"))
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    audio = args.kind == "audio"
    unfrozen = args.unfrozen_epochs if args.unfrozen_epochs is not None else (15 if audio else 5)
    if "EAV_FORCE_DEVICE" in os.environ:                      # several ranks on one GPU (logic runs on a 1-GPU box)
        os.environ["LOCAL_RANK"] = os.environ["EAV_FORCE_DEVICE"]
    rank, world, local = eav_dist.init_from_env(args.backend)
    torch.cuda.set_device(local)
    from eav_amd.audio import AudioModelTrainer
    from eav_amd.vision import ImageClassifierTrainer, trial_vote
    sched = eav_dist.subject_schedule(world, args.subjects, hybrid=not args.no_hybrid)
    groups = sched.make_groups() if world > 1 else {}
    tmp = tempfile.mkdtemp(prefix="eav_subjects_")
    cwd = os.getcwd()
    os.chdir(tmp)                                             # the trainers append their log files to the cwd (Q17)
    try:
        path = args.model_path
        if path is None:
            import bench
            path = bench._save_full_model_dir("ast" if audio else "vit", os.path.join(tmp, "model"))
        bs = 8 if audio else 128
        out, t0 = {}, time.perf_counter()
        mine = sched.group_of(rank)
        plan = [(s, None) for s in sched.solo[rank]] + ([mine] if mine else [])
        for sub, ranks in plan:
            data = synthetic_subject(args.kind, sub, args.small)
            n = len(ranks) if ranks else 1
            # this rank's replica sees every n-th training item: equal shard lengths and batch sizes on every member,
            # hence equal step counts (eav_amd.dist.replica_shard)
            sl, bsz = eav_dist.replica_shard(len(data[0]), bs, ranks.index(rank) if n > 1 else 0, n)
            data = [data[0][sl], data[1][sl], data[2], data[3]]
            torch.manual_seed(sub)                            # the fresh head: identical on every member of a group
            with contextlib.redirect_stdout(sys.stdout if args.verbose else io.StringIO()):
                if audio:
                    tr = AudioModelTrainer(data, model_path=path, sub=f"subject_{sub:02d}", num_classes=5,
                                           weight_decay=1e-5, lr=0.005, batch_size=bsz)
                else:
                    tr = ImageClassifierTrainer(data, model_path=path, sub=f"subject_{sub:02d}", num_labels=5, lr=5e-5,
                                                batch_size=bsz)
                if n > 1:
                    eav_dist.attach(tr, group=groups[sub])
                tr.train(epochs=args.frozen_epochs, lr=5e-4, freeze=True)
                tr.train(epochs=unfrozen, lr=5e-6, freeze=False)
            if n == 1 or ranks[0] == rank:                    # one report per subject
                out[sub] = (np.asarray(tr.outputs_test, dtype=np.float32), np.asarray(data[3]))
            del tr
            torch.cuda.empty_cache()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        gathered = [None] * world
        if world > 1:
            torch.distributed.all_gather_object(gathered, out)          # the job's only whole-world collective
        else:
            gathered = [out]
        if rank == 0:
            allr = {k: v for d in gathered for k, v in d.items()}
            accs, f1s = [], []
            for sub in sorted(allr):
                logits, te_y = allr[sub]
                if audio:
                    accs.append(float((logits.argmax(1) == te_y).mean()))
                else:
                    pred, acc, f1 = trial_vote(logits, te_y, frames_per_trial=25)
                    accs.append(float(acc))
                    f1s.append(float(f1))
            rep = {"modality": args.kind, "subjects": len(allr), "world": world, "seconds": round(dt, 2),
                   "schedule": {"rounds": sched.rounds, "groups": sched.groups},
                   "outputs_test_shape": list(next(iter(allr.values()))[0].shape),
                   "mean_test_acc": round(float(np.mean(accs)), 4)}
            if f1s:
                rep["mean_weighted_f1"] = round(float(np.mean(f1s)), 4)
            print(json.dumps(rep))
        if world > 1:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
    finally:
        os.chdir(cwd)
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
