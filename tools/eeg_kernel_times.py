#!/usr/bin/env python3
"""Per-call times of one EEGNet train step at the bench shape via HIP events on every library call (eav_amd._lib.TRACE),
then the graph-replayed step (run on the GPU box).   python tools/eeg_kernel_times.py [train|eval]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from eav_amd import _lib  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "train"
run = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
if mode == "eval":
    run.model.eval()
for i in range(4):
    run.step(i)
run.eager_step(4)
torch.cuda.synchronize()
_lib.TRACE = {}
N = 8
for i in range(N):
    run.eager_step(4 + i)          # the graph-replayed step has no per-launch events
torch.cuda.synchronize()
trace, _lib.TRACE = _lib.TRACE, None
tot = 0.0
for k, v in sorted(trace.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in kv[1])):
    ms = sum(a.elapsed_time(b) for a, b in v) / N
    tot += ms
    print(f"{k:36s} {len(v) // N} x {ms / (len(v) // N) * 1e3:8.1f} us = {ms * 1e3:8.1f} us/step")
dt, _ = run.timed(20, 0)
print(f"sum of all library calls {tot:.3f} ms; graph-replayed step {dt / 20 * 1e3:.3f} ms ({mode})")
