#!/usr/bin/env python3
"""Times eav_gemm_sp_splitk (token-contracting) and eav_gemm_sp on one ViT shape with each ablation library built by
tr_ablate.sh.  Timing only - the ablated kernels compute garbage."""
import ctypes as C
import glob
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, row_planes, timeit  # noqa: E402

_lib.load()
T, M, N = 25216, 3072, 768
A = torch.randn(T, M, device="cuda")
B = torch.randn(T, N, device="cuda")
sa, pa = row_planes(A)
sb, pb = row_planes(B)
Cw = torch.empty(M, N, device="cuda")
ws = torch.empty(40 * M * N, device="cuda")
# column-contracting product of the same flop count: [T, 3072] x [768, 3072]^T
X = torch.randn(T, M, device="cuda")
W = torch.randn(N, M, device="cuda")
sx, px, _ = planes(X)
sw, pw, _ = planes(W)
Cd = torch.empty(T, N, device="cuda")
v = C.c_void_p
names = {0: "full", 1: "no frag reads", 2: "no LDS-DMA", 3: "MFMA only", 4: "no MFMA", 5: "DMA only", 6: "reads only",
         21: "DMA only, A hot", 37: "DMA only, B hot", 53: "DMA only, A and B hot", 16: "full, A hot", 48: "full, A and B hot"}
for path in sorted(glob.glob(os.path.join(HERE, "build", "libgemm_abl*.so")), key=lambda p: int(p.split("abl")[-1][:-3])):
    abl = int(path.split("abl")[-1][:-3])
    lib = C.CDLL(path)
    lib.eav_gemm_sp_splitk.argtypes = [v, v, v, v, v, v, C.c_int, C.c_int, C.c_int, C.c_int, v]
    lib.eav_gemm_sp_splitk_x1.argtypes = lib.eav_gemm_sp_splitk.argtypes
    lib.eav_gemm_sp.argtypes = [v, v, v, v, v, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_float,
                                v, C.c_int, v, v, C.c_int, C.c_int, v, v]
    lib.eav_gemm_sp_x1.argtypes = lib.eav_gemm_sp.argtypes
    lib.eav_gemm_sp_set_tile(int(os.environ.get("TILE", "0")))
    out = []
    for fn in ("eav_gemm_sp_splitk", "eav_gemm_sp_splitk_x1"):
        f = getattr(lib, fn)
        ms = timeit(lambda: f(P(pa), P(pb), P(Cw), P(ws), P(sa), P(sb), M, N, T, 0, None), reps=20)
        out.append(f"{fn[12:]} {ms * 1e3:5.0f} us")
    for fn in ("eav_gemm_sp", "eav_gemm_sp_x1"):
        f = getattr(lib, fn)
        ms = timeit(lambda: f(P(px), P(pw), P(Cd), P(sx), P(sw), T, N, M, N, 1, 0, 0, 1.0, None, 0, None, None, 0, 0, None,
                              None), reps=20)
        out.append(f"{fn[4:]} {ms * 1e3:5.0f} us")
    print(f"abl {abl:2d} {names.get(abl, ''):30s} " + "   ".join(out), flush=True)
