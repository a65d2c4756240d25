#!/usr/bin/env python3
"""Streaming-copy variants (eav_peak_copy_variant): loads in flight per lane x grid size x non-temporal."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib
_lib.load()
n = 1 << 28
src, dst = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
names = {0: "U1", 5: "U2", 1: "U4", 2: "U4 nt", 3: "U8", 4: "U8 nt"}
for var in (0, 5, 1, 2, 3, 4):
    row = []
    for blocks in (256, 512, 768, 1024, 1280, 1536):
        f = lambda: _lib.call("eav_peak_copy_variant", src.data_ptr(), dst.data_ptr(), n, var, blocks, None)
        for _ in range(2): f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): f()
        b.record(); torch.cuda.synchronize()
        row.append(f"{blocks}: {5 * 8.0 * n / (a.elapsed_time(b) * 1e-3) / 1e12:5.2f}")
    print(f"{names[var]:6s} TB/s (read+write)  " + "  ".join(row))
