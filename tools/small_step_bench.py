#!/usr/bin/env python3
"""Step time of EEGNet at the reference's own shape (B=32, [32,1,30,500]): launch-bound regime."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import synth  # noqa: E402
from eav_amd.eegnet import EEGNet_tor  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402

B, S = 32, 500
x, y = synth.eeg_batch(1, B, 30, S)
x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
model = EEGNet_tor(5, Samples=S).cuda().train()
opt, crit = FusedAdam(model.parameters(), lr=1e-5), CrossEntropyLoss()


def step():
    scores = model(x)
    loss = crit(scores, y)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 300
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"eager: {dt * 1e6:.1f} us/step -> {B / dt:.0f} samples/s")
from eav_amd.eegnet import GraphStep  # noqa: E402
xs, ys = x.repeat(4, 1, 1, 1), y.repeat(4)
opt2 = FusedAdam(model.parameters(), lr=1e-5, capturable=True)
gs = GraphStep(model, opt2, crit, xs, ys, B)
idx = list(range(B))
for _ in range(20):
    gs.run(idx)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    gs.run(idx)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"graph: {dt * 1e6:.1f} us/step -> {B / dt:.0f} samples/s")
