#!/usr/bin/env python3
"""Do the maxima the fused attention backward publishes (tensor-wide shards + 128-row block entries) cost time?
eav_attn_bwd_sp with and without an amax slot on the ViT B=128 / AST B=8 shapes."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, timeit  # noqa: E402

_lib.load()
SLOT = 4128
for name, B, H, N in (("vit B=128", 128, 12, 197), ("ast B=8", 8, 12, 1214)):
    D = 64 * H
    qkv = torch.randn(B * N, 3 * D, device="cuda")
    dO = torch.randn(B * N, D, device="cuda") * 1e-3
    Npad = _lib.plain("eav_attn_sp_npad", N)

    def prep(x, ncols, secw, tmask):
        slot = torch.zeros(SLOT, device="cuda")
        _lib.call("eav_sp_absmax", P(x), B * N, ncols, ncols, P(slot), None)
        rowp = torch.empty(B * N, 2 * ncols, dtype=torch.float16, device="cuda")
        tp = torch.empty(B, ncols // 64, 64, 2 * Npad, dtype=torch.float16, device="cuda")
        _lib.call("eav_attn_sp_prep", P(x), P(slot), P(rowp), P(tp), B, N, ncols, secw, tmask, None)
        return slot, rowp, tp
    s_qkv, rowp, tp = prep(qkv, 3 * D, D, 7)
    ao, lse = torch.empty(B * N, D, device="cuda"), torch.empty(B * H, N, device="cuda")
    _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), None, B, H, N, 64, 0.125, None)
    s_do, dorow, dotp = prep(dO, D, D, 1)
    delta, dqkv = torch.empty(B * H, N, device="cuda"), torch.empty(B * N, 3 * D, device="cuda")
    amax = torch.zeros(SLOT, device="cuda")
    out = []
    for label, am in (("no maxima", None), ("maxima", amax)):
        def f():
            s_ds = torch.zeros(SLOT, device="cuda")
            _lib.call("eav_attn_bwd_sp", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds), P(ao), P(dO), P(lse),
                      P(delta), P(dqkv), P(am), B, H, N, 64, 0.125, None)
        out.append(f"{label} {timeit(f, reps=20) * 1e3:6.1f} us")
    fa = lambda am: timeit(lambda: _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), P(am), B, H, N, 64,  # noqa: E731
                                             0.125, None), reps=20) * 1e3
    print(f"{name}: backward " + ", ".join(out) + f";  forward no maxima {fa(None):6.1f} us, maxima {fa(amax):6.1f} us")
