#!/usr/bin/env python3
"""v_mfma_f32_32x32x2_f32 rate of eav_peak_mfma_f32 (4 independent chains per wave, register operands) as a function of
workgroups (x 4 waves) and chain length: how much of the 157 TFLOP/s does a SHORT launch at 2 waves per SIMD get?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from eav_amd import _lib as L  # noqa: E402

sink = torch.zeros(4, device="cuda")
for blocks in (256, 512, 1024, 2048):
    for iters in (104, 416, 4000):
        L.call("eav_peak_mfma_f32", sink.data_ptr(), blocks, iters, None)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        a.record()
        for _ in range(reps):
            L.call("eav_peak_mfma_f32", sink.data_ptr(), blocks, iters, None)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / reps
        tf = blocks * 4 * iters * 4 * 4096.0 / (ms * 1e-3) / 1e12
        print(f"blocks {blocks:5d} ({blocks * 4 / 1024:.0f} waves/SIMD) x {iters:5d} iterations: {ms * 1e3:8.1f} us  {tf:6.1f} TFLOP/s")
