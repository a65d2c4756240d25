#!/usr/bin/env python3
"""One product, 12 launches - the target of counter passes (tools/probes/one_gemm_pmc.sh).
    one_gemm.py nt M N K   -> eav_gemm_sp      C[M,N] = A[M,K] B[N,K]^T
    one_gemm.py tr M N T   -> eav_gemm_sp_splitk  C[M,N] = sum_t A[t,m] B[t,n]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, row_planes  # noqa: E402

_lib.load()
if os.environ.get("SP_TILE"):
    _lib.call("eav_gemm_sp_set_tile", int(os.environ["SP_TILE"]))
mode, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if mode == "nt":
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    f = lambda: _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0,  # noqa: E731
                          0, None, None)
else:
    A = torch.randn(K, M, device="cuda")
    B = torch.randn(K, N, device="cuda")
    sa, pa = row_planes(A)
    sb, pb = row_planes(B)
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(40 * M * N, device="cuda")
    f = lambda: _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, K, 0, None)  # noqa: E731
for _ in range(12):
    f()
torch.cuda.synchronize()
