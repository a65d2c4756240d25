#!/usr/bin/env python3
"""Training trajectories of the 12-layer AST / ViT under the three arithmetic settings - exact-fp32 kernels, split
(default), split with fp16-operand gradients (grad_terms = 1) - from the same initial weights on the same synthetic
batches: loss per step, and the logits on a held-out batch at the end.  Run on the GPU box.
usage: encoder_trajectory.py [steps] [lr]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 5e-5
for kind, B in (("vit", 32), ("ast", 8)):
    cfg = T.make_config(kind)
    torch.manual_seed(0)
    init = {k: v.detach().clone() for k, v in T.Encoder(cfg).state_dict().items()}
    batches = []
    for s in range(4):
        x, y = (synth.mel_batch(100 + s, B) if kind == "ast" else synth.frame_batch(100 + s, B))
        batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))
    hx, _ = (synth.mel_batch(999, B) if kind == "ast" else synth.frame_batch(999, B))
    hx = torch.from_numpy(hx).cuda()
    runs = {}
    for name, prec, terms, wt, dt in (("fp32", "fp32", 3, None, None), ("split", "split", 3, None, None),
                                      ("split, wgrad_terms=2", "split", 3, 2, None),
                                      ("split, wgrad_terms=1", "split", 3, 1, None),
                                      ("split, dgrad_terms=1", "split", 3, None, 1),
                                      ("split, grad_terms=1", "split", 1, None, None)):
        model = T.Encoder(cfg).cuda().train()
        model.load_state_dict(init)
        model.precision, model.grad_terms, model.wgrad_terms, model.dgrad_terms = prec, terms, wt, dt
        opt, crit = FusedAdam(model.parameters(), lr=lr, weight_decay=0.01, decoupled=True), CrossEntropyLoss()
        losses = []
        import time
        t_steps = []
        for i in range(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, y = batches[i % len(batches)]
            opt.zero_grad()
            loss = crit(model(x).logits, y)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            t_steps.append(time.perf_counter() - t0)
        model.eval()
        with torch.no_grad():
            held = model(hx).logits.float().cpu().numpy()
        runs[name] = (np.array(losses), held, float(np.median(t_steps[5:])) * 1e3)
        del model, opt
        torch.cuda.empty_cache()
    ref_l, ref_h, _ = runs["fp32"]
    print(f"== {kind} B={B}, {steps} AdamW steps at lr {lr:g}: loss {ref_l[0]:.4f} -> {ref_l[-1]:.4f} (exact-fp32 kernels)")
    for name in ("split", "split, wgrad_terms=2", "split, wgrad_terms=1", "split, dgrad_terms=1", "split, grad_terms=1"):
        l, h, ms = runs[name]
        print(f"   {name:22s} {ms:6.2f} ms/step  max |loss - fp32 loss| over the run {np.abs(l - ref_l).max():.2e} (final {abs(l[-1] - ref_l[-1]):.2e}); "
              f"held-out logits after training: max |diff| {np.abs(h - ref_h).max():.2e} (logit range {np.abs(ref_h).max():.2f})")
