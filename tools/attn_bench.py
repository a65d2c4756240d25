"""Fused attention kernels (eav_attn_fwd / eav_attn_bwd) at the AST and ViT shapes: time and fp32-MFMA TFLOP/s
(forward 4 N^2 d flops per head, backward 10 N^2 d: dV, dP, dQ, dK and the recomputed scores)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402

for name, B, H, N in (("ast", 8, 12, 1214), ("vit", 128, 12, 197), ("shallow", 32, 1, 488)):
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda") * 0.5
    ao, dout, dqkv = torch.empty(B * N, D, device="cuda"), torch.randn(B * N, D, device="cuda"), torch.empty(B * N, 3 * D, device="cuda")
    lse, delta = torch.empty(B * H, N, device="cuda"), torch.empty(B * H, N, device="cuda")
    st = _lib.stream_ptr()
    P = _lib.ptr

    def fwd():
        _lib.call("eav_attn_fwd", P(qkv), P(ao), P(lse), B, H, N, 64, 0.125, st)

    def bwd():
        _lib.call("eav_attn_bwd", P(qkv), P(ao), P(dout), P(lse), P(delta), P(dqkv), B, H, N, 64, 0.125, st)

    for fn, flop, label in ((fwd, 4, "fwd"), (bwd, 10, "bwd")):
        for _ in range(3):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        tf = flop * N * N * 64 * B * H / (ms * 1e-3) / 1e12
        print(f"{name:8s} {label}: {ms * 1e3:8.1f} us  {tf:6.1f} TFLOP/s ({tf / 157.3:.0%} of fp32 MFMA peak)")
