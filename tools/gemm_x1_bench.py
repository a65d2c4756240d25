#!/usr/bin/env python3
"""One-term products (eav_gemm_sp_x1 / eav_gemm_sp_splitk_x1: hi.hi only) beside the three-term ones on the backward
shapes of AST B=8 / ViT B=128: time, and error against float64.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, row_planes, timeit  # noqa: E402


def dgrad(name, M, N, K):
    A = torch.randn(M, K, device="cuda") * torch.exp(torch.randn(M, 1, device="cuda"))
    B = torch.randn(N, K, device="cuda") * 0.02
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    ref = A[:2048].double() @ B.double().t()
    out = []
    for fn in ("eav_gemm_sp", "eav_gemm_sp_x1"):
        f = lambda: _lib.call(fn, P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0,  # noqa: E731
                              0, None, None)
        ms = timeit(f)
        rel = ((C[:2048].double() - ref).norm() / ref.norm()).item()
        out.append(f"{fn[8:]:>6s} {ms:6.3f} ms {2.0 * M * N * K / ms / 1e9:6.0f} TF rel {rel:.1e}")
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}  " + "   ".join(out))


def wgrad(name, M, N, T):
    A = torch.randn(T, M, device="cuda") * torch.exp(torch.randn(T, 1, device="cuda"))
    B = torch.randn(T, N, device="cuda")
    sa, pa = row_planes(A)
    sb, pb = row_planes(B)
    C = torch.empty(M, N, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", M, N, T)
    ws = torch.empty(max(ns, 1) * M * N, device="cuda")
    ref = A.double().t()[:256] @ B.double()
    out = []
    for fn in ("eav_gemm_sp_splitk", "eav_gemm_sp_splitk_x1"):
        f = lambda: _lib.call(fn, P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, T, 0, None)  # noqa: E731
        ms = timeit(f)
        rel = ((C[:256].double() - ref).norm() / ref.norm()).item()
        out.append(f"{fn[12:]:>9s} {ms:6.3f} ms {2.0 * M * N * T / ms / 1e9:6.0f} TF rel {rel:.1e}")
    print(f"{name:12s} M={M:6d} N={N:5d} T={T:5d}  " + "   ".join(out))


if __name__ == "__main__":
    _lib.load()
    for tag, M in (("ast B=8", 9712), ("vit B=128", 25216)):
        print("==", tag)
        dgrad("fc2 dgrad", M, 3072, 768)
        dgrad("fc1 dgrad", M, 768, 3072)
        dgrad("o dgrad", M, 768, 768)
        dgrad("qkv dgrad", M, 768, 2304)
        wgrad("fc1 wgrad", 3072, 768, M)
        wgrad("fc2 wgrad", 768, 3072, M)
        wgrad("qkv wgrad", 2304, 768, M)
        wgrad("o wgrad", 768, 768, M)
