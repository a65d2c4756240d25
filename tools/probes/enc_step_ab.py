#!/usr/bin/env python3
"""Encoder training step (AST B = 8 / ViT B = 128, split precision) under run-time settings alternated inside ONE process - the
box-to-box spread (3-4 %) is larger than most of the differences looked for.  Variants: the weight gradients' term count
(Encoder.wgrad_terms 3 | 2 | 1).  EAV_LIB_PATH selects an A/B build of the library for the whole process.
usage: enc_step_ab.py ast|vit B [reps] [terms,terms,...]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eav_amd import _lib, synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402

kind, B = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
variants = [(f"wgrad_terms={w}", 0, 0, int(w)) for w in (sys.argv[4].split(",") if len(sys.argv) > 4 else ("3", "2"))]
L = _lib.load()
model = T.Encoder(T.make_config(kind)).cuda().train()
model.precision = "split"
x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
opt, crit = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True), CrossEntropyLoss()


def step():
    opt.zero_grad()
    crit(model(x).logits, y).backward()
    opt.step()


for _ in range(4):
    step()
ts = {v[0]: [] for v in variants}
for r in range(reps):
    order = variants[r % len(variants):] + variants[:r % len(variants)]
    for name, dyn, stg, wt in order:
        model.wgrad_terms = wt
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            step()
        torch.cuda.synchronize()
        ts[name].append((time.perf_counter() - t0) / 6 * 1e3)
base = statistics.median(ts[variants[0][0]])
for name, v in ts.items():
    m = statistics.median(v)
    print(f"{kind} B={B} {os.path.basename(os.environ.get('EAV_LIB_PATH') or 'libeav_hip.so'):22s} {name:14s} {m:7.2f} ms/step  {B / m * 1e3:8.1f} samples/s  ({m / base - 1:+.1%} vs {variants[0][0]})  "
          + " ".join(f"{t:.2f}" for t in v), flush=True)
