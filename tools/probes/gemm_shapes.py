#!/usr/bin/env python3
"""Every dense product of one encoder layer (forward, data gradient, weight gradient), REPS launches each, in a fixed
order - the target of per-shape counter passes (tools/probes/gemm_shapes_pmc.sh; the parser groups the gemm_sp launches
by position).   gemm_shapes.py vit|ast [batch]"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, row_planes  # noqa: E402

REPS = 6


def shapes(kind, batch=None):
    tok = (batch or 128) * 197 if kind == "vit" else (batch or 8) * 1214
    D, F = 768, 3072
    nt = [("qkv fwd", tok, 3 * D, D), ("oproj fwd", tok, D, D), ("fc1 fwd", tok, F, D), ("fc2 fwd", tok, D, F),
          ("qkv dgrad", tok, D, 3 * D), ("oproj dgrad", tok, D, D), ("fc1 dgrad", tok, D, F), ("fc2 dgrad", tok, F, D)]
    tr = [("qkv wgrad", 3 * D, D, tok), ("oproj wgrad", D, D, tok), ("fc1 wgrad", F, D, tok), ("fc2 wgrad", D, F, tok)]
    return nt, tr


def timed(f, reps=10):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


if __name__ == "__main__":
    _lib.load()
    kind = sys.argv[1]
    if os.environ.get("TIME"):          # event timings of the column-contracting products, 128 x 128 and 256 x 128 tile forms
        nt, _ = shapes(kind, int(sys.argv[2]) if len(sys.argv) > 2 else None)
        for name, M, N, K in nt:
            A = torch.randn(M, K, device="cuda")
            B = torch.randn(N, K, device="cuda")
            sa, pa, _ = planes(A)
            sb, pb, _ = planes(B)
            C = torch.empty(M, N, device="cuda")
            f = lambda: _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0,  # noqa: E731
                                  None, None, 0, 0, None, None)
            out = []
            for mode in (1, 2):
                _lib.call("eav_gemm_sp_set_tile", mode)
                out.append(timed(f))
            _lib.call("eav_gemm_sp_set_tile", 0)
            print(f"{name:12s} [{M} x {N} x {K}]  128 x 128 tiles {out[0]:7.1f} us | 256 x 128 tiles {out[1]:7.1f} us")
        sys.exit(0)
    nt, tr = shapes(kind, int(sys.argv[2]) if len(sys.argv) > 2 else None)
    only = os.environ.get("ONLY")
    for name, M, N, K in nt:
        if only and only not in name:
            continue
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda")
        sa, pa, _ = planes(A)
        sb, pb, _ = planes(B)
        C = torch.empty(M, N, device="cuda")
        torch.cuda.synchronize()
        for _ in range(REPS):
            _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0,
                      0, None, None)
        torch.cuda.synchronize()
        del A, B, C, pa, pb
    for name, M, N, T in tr:
        if only and only not in name:
            continue
        A = torch.randn(T, M, device="cuda")
        B = torch.randn(T, N, device="cuda")
        sa, pa = row_planes(A)
        sb, pb = row_planes(B)
        C = torch.empty(M, N, device="cuda")
        ws = torch.empty(40 * M * N, device="cuda")
        torch.cuda.synchronize()
        for _ in range(REPS):
            _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, T, 0, None)
        torch.cuda.synchronize()
        del A, B, C, pa, pb, ws
