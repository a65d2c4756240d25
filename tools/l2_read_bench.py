#!/usr/bin/env python3
"""L2 -> CU read ceilings (eav_peak_l2_read): 16-byte loads into registers against LDS-DMA, by footprint and grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib
_lib.load()
src = torch.randn(64 << 20, device="cuda")   # 256 MB
sink = torch.zeros(4, device="cuda")
for fp_kb in (1024, 4096, 16384, 65536):
    for blocks in (256, 512, 1024):
        line = f"footprint {fp_kb >> 10:3d} MB, {blocks} blocks:"
        for mode, name in ((0, "VGPR"), (1, "LDS-DMA")):
            iters = 400
            f = lambda: _lib.call("eav_peak_l2_read", src.data_ptr(), fp_kb, mode, iters, blocks, sink.data_ptr(), None)
            for _ in range(2):
                f()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                f()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 5
            line += f"  {name} {blocks * 4 * iters * 8192 / ms / 1e9:6.2f} TB/s"
        print(line)
