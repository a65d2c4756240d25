#!/usr/bin/env python3
"""Weight-gradient shapes of the AST / ViT step through eav_gemm_sp_splitk (token-contracting kernel over row planes).
Run on the GPU box; under rocprofv3 --pmc for the LDS / wait counters of gemm_sp_kernel<..., TR = true>."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from tools.gemm_sp_bench import P, row_planes, timeit  # noqa: E402


def run(name, M, N, T, reps=10):
    A = torch.randn(T, M, device="cuda")
    B = torch.randn(T, N, device="cuda")
    sa, pa = row_planes(A)
    sb, pb = row_planes(B)
    C = torch.empty(M, N, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", M, N, T)
    ws = torch.empty(max(ns, 1) * M * N, device="cuda")
    ms = timeit(lambda: _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, T, 0, None), reps)
    print(f"{name:12s} M={M:5d} N={N:5d} T={T:6d} split-K x{ns:<3d} {ms:7.3f} ms {2.0 * M * N * T / ms / 1e9:7.1f} TF", flush=True)


if __name__ == "__main__":
    _lib.load()
    reps = int(os.environ.get("REPS", "10"))
    for tag, T in (("ast", 9712), ("vit", 25216)):
        run(tag + " fc1", 3072, 768, T, reps)
        run(tag + " fc2", 768, 3072, T, reps)
        run(tag + " qkv", 2304, 768, T, reps)
        run(tag + " o", 768, 768, T, reps)
    if os.environ.get("WITH_FWD"):
        from tools.gemm_sp_bench import bench
        os.environ["TILES"] = "1"
        bench("fc2 fwd", 9712, 768, 3072)
        bench("qkv fwd", 9712, 2304, 768)
