#!/usr/bin/env python3
"""layernorm_bwd variants (rows per group RG, persistent blocks) built as stand-alone libraries: timing on the ViT / AST shapes."""
import ctypes as C
import glob
import os
import re
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from gemm_sp_bench import timeit  # noqa: E402

v = C.c_void_p
for path in sorted(glob.glob(os.path.join(HERE, "build", "libtf_rg*_b*.so"))):
    rg, nb = (int(x) for x in re.findall(r"rg(\d+)_b(\d+)", path)[0])
    lib = C.CDLL(path)
    lib.eav_layernorm_bwd_amax.argtypes = [v, v, v, v, v, v, C.c_int, v, C.c_int, C.c_int, v, v]
    lib.eav_layernorm_fwd.argtypes = [v, v, v, v, v, v, C.c_int, C.c_int, C.c_float, v]
    out = []
    for M in (25216, 9712):
        D = 768
        x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
        y = torch.empty_like(x); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
        lib.eav_layernorm_fwd(x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), M, D, 1e-12, None)
        dy = torch.randn(M, D, device="cuda"); dx = torch.zeros(M, D, device="cuda")
        npart = lib.eav_layernorm_bwd_nparts(M)
        part = torch.empty(npart, 2 * D, device="cuda")
        slot = torch.zeros(4128, device="cuda")
        us = timeit(lambda: lib.eav_layernorm_bwd_amax(dy.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                       dx.data_ptr(), 1, part.data_ptr(), M, D, slot.data_ptr(), None), reps=30) * 1e3
        out.append(f"M={M}: {us:6.1f} us ({4 * M * D * 4 / us / 1e6:.2f} TB/s)")
    print(f"RG={rg:2d} blocks={nb:4d}: " + "   ".join(out), flush=True)
