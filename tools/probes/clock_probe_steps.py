#!/usr/bin/env python3
"""sclk / package power (rocm-smi, sampled from a thread) while the training steps run: EEGNet bs=64, AST B=8, ViT B=128
(three-term and one-term gradients), and the exact-fp32 encoder mode."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from clock_probe import run  # noqa: E402  (runs its own loads first when imported as a script - guarded below)
from eav_amd import synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402


def encoder(kind, B, prec="split", terms=3):
    model = T.Encoder(T.make_config(kind)).cuda().train()
    model.precision = prec
    model.grad_terms = terms
    x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
    x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    opt, crit = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True), CrossEntropyLoss()

    def step():
        opt.zero_grad()
        crit(model(x).logits, y).backward()
        opt.step()
    return step


def eegnet():
    from eav_amd.eegnet import EEGNet_tor
    m = EEGNet_tor(5, Chans=30, Samples=10000).cuda().train()
    x, y = synth.eeg_batch(1, 64, 30, 10000)
    x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    opt, crit = FusedAdam(m.parameters(), lr=1e-3), CrossEntropyLoss()

    def step():
        opt.zero_grad()
        crit(m(x), y).backward()
        opt.step()
    return step


if __name__ == "__main__":
    for name, mk in (("EEGNet bs=64 step (eager)", eegnet), ("AST B=8 step", lambda: encoder("ast", 8)),
                     ("ViT B=128 step", lambda: encoder("vit", 128)),
                     ("ViT B=128 step, one-term gradients", lambda: encoder("vit", 128, terms=1)),
                     ("ViT B=128 step, exact fp32", lambda: encoder("vit", 128, prec="fp32"))):
        f = mk()
        for _ in range(2):
            f()
        torch.cuda.synchronize()
        run(name, f, 7.0, burst=4)
        del f
        torch.cuda.empty_cache()
