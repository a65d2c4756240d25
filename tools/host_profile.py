#!/usr/bin/env python3
"""cProfile of the host side of the encoder training step (where the launch loop's CPU time goes).
    python tools/host_profile.py ast 8"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eav_amd import synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402

kind, B = sys.argv[1], int(sys.argv[2])
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = T.Encoder(T.make_config(kind)).to(dev).train()
x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
opt = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True)
crit = CrossEntropyLoss()


def step():
    opt.zero_grad()
    crit(model(x).logits, y).backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
