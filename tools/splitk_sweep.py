#!/usr/bin/env python3
"""eav_gemm_sp_splitk on the weight-gradient shapes: slice-count sweep through the tuning hook eav_gemm_sp_set_splitk
(the fit behind eav_gemm_sp_splitk_plan).  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, row_planes, timeit  # noqa: E402


def sweep(name, M, N, T, fn="eav_gemm_sp_splitk"):
    A = torch.randn(T, M, device="cuda")
    B = torch.randn(T, N, device="cuda")
    sa, pa = row_planes(A)
    sb, pb = row_planes(B)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(40 * M * N, device="cuda")
    ref = None
    res = []
    plan = _lib.plain("eav_gemm_sp_splitk_plan", M, N, T)
    for xcd, nss in ((0, (2, 3, 4, 5, 6, 7, 8, 10, 14, 21)),):
        for ns in nss:
            if ns * 8 > (T + 31) // 32:
                continue
            _lib.call("eav_gemm_sp_set_splitk", ns)
            _lib.call("eav_gemm_sp_set_tile", xcd)
            ms = timeit(lambda: _lib.call(fn, P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, T, 0, None), reps=20)
            if ref is None:
                ref = C.clone()
            err = ((C - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-5, (name, xcd, ns, err)
            res.append((ms, xcd, ns))
    _lib.call("eav_gemm_sp_set_splitk", 0)
    _lib.call("eav_gemm_sp_set_tile", 0)
    best = min(res)
    print(f"{name:10s} M={M:5d} N={N:5d} T={T:6d} plan {plan}: " +
          "  ".join(f"{ns}:{ms * 1e3:4.0f}{'*' if (ms, x, ns) == best else ''}" for ms, x, ns in res))


if __name__ == "__main__":
    _lib.load()
    fn = sys.argv[1] if len(sys.argv) > 1 else "eav_gemm_sp_splitk"
    for tag, M in (("ast B=8", 9712), ("ast B=32", 4 * 9712), ("vit B=128", 25216)):
        print("==", tag, fn)
        sweep("fc1", 3072, 768, M, fn)
        sweep("fc2", 768, 3072, M, fn)
        sweep("qkv", 2304, 768, M, fn)
        sweep("o", 768, 768, M, fn)
    sweep("patch vit", 768, 768, 25088, fn)
    sweep("patch ast", 768, 256, 9696, fn)
