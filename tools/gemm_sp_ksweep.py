#!/usr/bin/env python3
"""eav_gemm_sp time against K at fixed M, N: separates the per-tile cost (prologue, epilogue, tile-boundary bubbles) from
the per-K-tile cost of the main loop.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from tools.gemm_sp_bench import P, planes, timeit  # noqa: E402

_lib.load()
for M, N in ((25216, 2304), (9712, 2304), (8192, 8192)):
    pts = []
    for K in (128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, K, device="cuda")
        B = torch.randn(N, K, device="cuda") * 0.02
        C = torch.empty(M, N, device="cuda")
        sa, pa, _ = planes(A)
        sb, pb, _ = planes(B)
        ms = timeit(lambda: _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0,
                                      None, None, 0, 0, None, None), 20)
        pts.append((K // 32, ms * 1e3))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    rounds = tiles / 512.0
    (k0, t0), (k1, t1) = pts[2], pts[-1]
    b = (t1 - t0) / (k1 - k0)
    a = t0 - b * k0
    print(f"M={M} N={N}: {tiles} tiles = {rounds:.2f} rounds of 512;  " + "  ".join(f"K={32 * k}: {t:.1f} us" for k, t in pts))
    print(f"    fit: {a:.1f} us + {b:.3f} us per K-tile  ->  per tile-round {a / rounds:.2f} us fixed + {b / rounds:.3f} us per K-tile "
          f"(MFMA-bound: {24 * 32 * 2 / 2.4e3:.3f} us per K-tile for the two workgroups of a CU)")
