#!/usr/bin/env python3
"""Times of the split-operand attention kernels alone (forward, dQ, dK/dV) at the AST / ViT shapes, by kernel events.
usage: attn_sp_time.py [fwd|all] [ast|vit]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from attn_sp_check import P, SLOT, prep, timeit  # noqa: E402

_lib.load()
what = sys.argv[1] if len(sys.argv) > 1 else "all"
only = sys.argv[2] if len(sys.argv) > 2 else ""
for tag, B, N in (("ast B=8", 8, 1214), ("vit B=128", 128, 197)):
    if only and not tag.startswith(only):
        continue
    H, D = 12, 768
    qkv = torch.randn(B * N, 3 * D, device="cuda")
    dO = torch.randn(B * N, D, device="cuda") * 1e-3
    s_qkv, rowp, tp = prep(qkv, B, N, 3 * D, D, 7)
    s_do, dorow, dotp = prep(dO, B, N, D, D, 1)
    ao = torch.empty(B * N, D, device="cuda")
    lse = torch.empty(B * H, N, device="cuda")
    s_ds = torch.zeros(SLOT, device="cuda")
    delta = torch.empty(B * H, N, device="cuda")
    dqkv = torch.empty(B * N, 3 * D, device="cuda")
    fl = 3 * 4.0 * B * H * N * N * 64
    t_fwd = timeit(lambda: _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), None, B, H, N, 64,
                                     0.125, None), 20)
    line = f"{tag}: fwd {t_fwd * 1e3:7.1f} us ({fl / t_fwd / 1e12:.2f} PF/s fp16 MFMA issued)"
    if what == "all":
        t_bwd = timeit(lambda: _lib.call("eav_attn_bwd_sp", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do),
                                         P(s_ds), P(ao), P(dO), P(lse), P(delta), P(dqkv), None, B, H, N, 64, 0.125,
                                         None), 20)
        line += f" | bwd (dQ + dK,dV) {t_bwd * 1e3:7.1f} us ({3.5 * fl / t_bwd / 1e12:.2f} PF/s)"
    print(line)
