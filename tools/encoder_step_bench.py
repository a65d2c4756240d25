#!/usr/bin/env python3
"""Step time of the AST / ViT encoders at chosen batch sizes and precision modes (run on the GPU box).
usage: encoder_step_bench.py ast|vit B [precision]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402

kind, B = sys.argv[1], int(sys.argv[2])
prec = sys.argv[3] if len(sys.argv) > 3 else "fp32"
model = T.Encoder(T.make_config(kind)).cuda().train()
model.precision = prec
if os.environ.get("NO_OVERLAP"):
    model.overlap_wgrad = False
if os.environ.get("SPLITK"):          # forced slice count of the weight-gradient products (tuning hook)
    from eav_amd import _lib
    _lib.call("eav_gemm_sp_set_splitk", int(os.environ["SPLITK"]))
if os.environ.get("SP_TILE"):
    from eav_amd import _lib
    _lib.call("eav_gemm_sp_set_tile", int(os.environ["SP_TILE"]))
if os.environ.get("FREEZE"):          # the fine-tune's first phase: classifier head only
    for k, p in model.named_parameters():
        p.requires_grad = k.startswith("classifier.")
x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
x, y = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
opt, crit = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True), CrossEntropyLoss()


def step():
    opt.zero_grad()
    crit(model(x).logits, y).backward()
    opt.step()


import contextlib
hp = os.environ.get("HIGH_PRIO")
ctx = torch.cuda.stream(torch.cuda.Stream(priority=-1)) if hp else contextlib.nullcontext()
with ctx:
    for _ in range(4):                   # (lazy allocations, first launches of every kernel, event / stream pools)
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = int(os.environ.get("STEPS", "12"))
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
if os.environ.get("SPLIT_TIMES"):     # forward / backward / optimiser separately (a synchronisation between the parts)
    tf = tb = to = 0.0
    for _ in range(n):
        opt.zero_grad()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss = crit(model(x).logits, y)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        opt.step()
        torch.cuda.synchronize(); t3 = time.perf_counter()
        tf += t1 - t0; tb += t2 - t1; to += t3 - t2
    print(f"   forward {tf / n * 1e3:.2f} ms  backward {tb / n * 1e3:.2f} ms  optimiser {to / n * 1e3:.2f} ms")
gf = {"ast": 783.1, "vit": 105.4}[kind] * B
print(f"{kind} B={B} {prec}: {dt * 1e3:.1f} ms/step, {B / dt:.1f} samples/s, {gf / dt / 1e3:.1f} TFLOP/s; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
