#!/usr/bin/env python3
"""What do the epilogue options of eav_gemm_sp_ex cost on the shapes that use them (ViT B=128 / AST B=8)?"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, timeit, kpad  # noqa: E402

_lib.load()
SLOT = 4128
for tag, M in (("vit", 25216), ("ast", 9712)):
    for name, N, K in (("o dgrad", 768, 768), ("fc2 dgrad", 3072, 768), ("qkv fwd", 2304, 768)):
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda") * 0.05
        sa, pa, _ = planes(A)
        sb, pb, _ = planes(B)
        C = torch.empty(M, N, device="cuda")
        pre = torch.randn(M, N, device="cuda")
        amax = torch.zeros(SLOT, device="cuda")
        pl = torch.zeros((M + 31) // 32 * 32, 2 * kpad(N), dtype=torch.float16, device="cuda")
        slot = torch.zeros(SLOT, device="cuda")
        slot[2048], slot[2049] = 1024.0, 1.0 / 1024.0
        part = torch.empty((M + 63) // 64, N, device="cuda")

        def run(c, am, p_out, cs, gelu, flags):
            return timeit(lambda: _lib.call("eav_gemm_sp_ex", P(pa), P(pb), P(c), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None,
                                            gelu, P(pre) if gelu else None, None, 0, 0, P(am), P(p_out), P(slot) if p_out is not None else None,
                                            P(cs), flags, None), reps=20) * 1e3
        out = [f"C only {run(C, None, None, None, 0, 0):6.1f}", f"C + maxima {run(C, amax, None, None, 0, 0):6.1f}",
               f"planes only {run(None, None, pl, None, 0, 0):6.1f}", f"planes + colsum {run(None, None, pl, part, 0, 0):6.1f}",
               f"gelu' planes + colsum {run(None, None, pl, part, 2, 0):6.1f}", f"gelu' C + maxima {run(C, amax, None, None, 2, 0):6.1f}",
               f"(256x128 form) gelu' planes + colsum {run(None, None, pl, part, 2, 2):6.1f}"]
        print(f"{tag} {name:10s} M={M} N={N} K={K} us: " + "  ".join(out))
