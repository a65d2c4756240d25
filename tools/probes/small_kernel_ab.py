#!/usr/bin/env python3
"""In-graph cost of the EEGNet step's small kernels at the bench shape (B = 64, NF = 19 968, 5 classes): a hipGraph of 20
back-to-back launches of ONE call, replayed 20 times -> us per launch as the step's graph pays it (kernel + node gap).
Second column: the same after a 640 MB sweep (cold caches; includes the sweep's write-back tail).
Round 5: a wide-grid dense forward (column slices for every sample + a finishing launch: W read once instead of once per
sample) measured 11.0 / 16.9 us against 9.6 / 18.3 us here and NO difference in the full step (1.377 ms both, three
alternating runs on one box, tools/probes/eeg_step_ab.py) - removed.
usage: small_kernel_ab.py      (run on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eav_amd import _lib  # noqa: E402

P = _lib.ptr
_lib.load()
B, NF, NC = 64, 19968, 5
dev = "cuda"
x = torch.randn(B, NF, device=dev)
w = torch.randn(NC, NF, device=dev) * 0.01
b = torch.zeros(NC, device=dev)
probs = torch.empty(B, NC, device=dev)
dprobs = torch.randn(B, NC, device=dev)
dw, db, din = torch.empty(NC, NF, device=dev), torch.empty(NC, device=dev), torch.empty(B, NF, device=dev)


def cost(fn, n=20, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            g.replay()
        e.record()
        torch.cuda.synchronize()
    return a.elapsed_time(e) / (n * reps) * 1e3


st = _lib.stream_ptr
cases = {
    "dense_softmax_fwd (block per sample)": lambda: _lib.call("eav_dense_softmax_fwd", P(x), P(w), P(b), None, P(probs), B, NF, NC, st()),
    "dense_softmax_bwd": lambda: _lib.call("eav_dense_softmax_bwd", P(dprobs), P(probs), P(x), P(w), P(dw), P(db), P(din), B, NF, NC, st()),
    "counter_inc (floor: an empty-ish kernel)": None,
}
cnt = torch.zeros((), dtype=torch.int64, device=dev)
cases["counter_inc (floor: an empty-ish kernel)"] = lambda: _lib.call("eav_counter_inc", P(cnt), st())
trash = torch.empty(160 * 1024 * 1024, device=dev)     # 640 MB: more than L2 + the 256 MB memory-side cache


def with_trash(fn):
    def g():
        trash.add_(1.0)
        fn()
    return g


base = cost(lambda: trash.add_(1.0), n=10, reps=5)
for k, fn in cases.items():
    cold = cost(with_trash(fn), n=10, reps=5) - base
    print(f"{k:44s} {cost(fn):7.2f} us / launch in a graph (warm caches)   {cold:7.2f} us after a 640 MB sweep")
