#!/usr/bin/env python3
"""Epilogue share of the short-K, wide-output products of an encoder layer (qkv forward, fc1 forward, fc2 data gradient;
K = 768) with the epilogues the training step really uses, against the same launches from a library built with
-DEAV_ABL=64 (gemm_sp.hip: the K loop alone, nothing written).  Build first, here:
    tools/probes/build_variant.sh noepi gemm_sp -DEAV_ABL=64
then on the GPU box:  python tools/probes/gemm_epilogue_share.py [libeav_noepi.so]   (loads both libraries itself; EAV_LIB_PATH
selects the full library - e.g. round 5's kernel behind this round's ABI, with its own -DEAV_ABL=64 build as the argument)."""
import ctypes as C
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, timeit, kpad  # noqa: E402

_lib.load()
abl = C.CDLL(os.path.join(HERE, "build", sys.argv[1] if len(sys.argv) > 1 else "libeav_noepi.so"))
v, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
sig = [v, v, v, v, v, i32, i32, i32, i32, i32, i64, i64, f32, v, i32, v, v, i32, i32, v, v, v, v, i32, v]
abl.eav_gemm_sp_ex.argtypes = sig
full = _lib.load()
SLOT = 4128
# warm the GPU first (a process's first launches run 10-20 % slow: clocks of a GPU that has been idle)
_w = torch.randn(8192, 8192, device="cuda")
for _ in range(40):
    _w @ _w
torch.cuda.synchronize()
print("product                          full us   K loop alone us   epilogue share   algorithmic TFLOP/s (full / K loop alone)")
for tag, M in (("vit", 25216), ("ast", 9712)):
    for name, N, K, gelu, bias, want_pre, want_c, want_planes, colsum, amax in (
            ("qkv fwd (planes, maxima)", 2304, 768, 0, 1, 0, 0, 1, 0, 1),
            ("fc1 fwd (gelu, pre, planes)", 3072, 768, 1, 1, 1, 0, 1, 0, 0),
            ("fc2 dgrad (gelu', planes, colsum)", 3072, 768, 2, 0, 1, 0, 1, 1, 0),
            ("o fwd (bias, residual, C)", 768, 768, 0, 1, 0, 1, 0, 0, 0),
            ("fc2 fwd (bias, residual, C)", 768, 3072, 0, 1, 0, 1, 0, 0, 0)):
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda") * 0.05
        sa, pa, _ = planes(A)
        sb, pb, _ = planes(B)
        Cm = torch.empty(M, N, device="cuda") if want_c else None
        res = torch.randn(M, N, device="cuda") if want_c else None
        pre = torch.randn(M, N, device="cuda") if want_pre else None
        bia = torch.randn(N, device="cuda") if bias else None
        am = torch.zeros(SLOT, device="cuda") if amax else None
        pl = torch.zeros((M + 31) // 32 * 32, 2 * kpad(N), dtype=torch.float16, device="cuda") if want_planes else None
        slot = torch.zeros(SLOT, device="cuda")
        slot[2048], slot[2049] = 1024.0, 1.0 / 1024.0
        part = torch.empty((M + 63) // 64, N, device="cuda") if colsum else None
        args = (P(pa), P(pb), P(Cm), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bia), gelu, P(pre), P(res), N if want_c else 0, 0,
                P(am), P(pl), P(slot) if pl is not None else None, P(part), 0, None)
        t_full = timeit(lambda: full.eav_gemm_sp_ex(*args), reps=20) * 1e3
        t_abl = timeit(lambda: abl.eav_gemm_sp_ex(*args), reps=20) * 1e3
        tf = 2.0 * M * N * K / 1e6
        print(f"{tag} {name:34s} {t_full:8.1f} {t_abl:12.1f} {1 - t_abl / t_full:14.2f}        {tf / t_full:7.0f} / {tf / t_abl:5.0f}", flush=True)
