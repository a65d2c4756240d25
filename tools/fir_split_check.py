"""Accuracy and speed of the split-fp16 FIR forward against the exact-fp32 MFMA kernel, both measured against a
float64 reference (torch CPU conv in double precision on a sample of rows)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib, synth  # noqa: E402

B, C, S, K = 64, 30, 10000, 300
x = torch.from_numpy(synth.normal(1, (B, C, S))).cuda()
w = torch.from_numpy(synth.uniform(2, (8, K), -0.0577, 0.0577)).cuda()
P, st = _lib.ptr, _lib.stream_ptr()
npart = _lib.plain("eav_eegnet_fir_fwd_nparts", B, C, S)
y_f32, y_sp = torch.empty(B, 8, C, S, device="cuda"), torch.empty(B, 8, C, S, device="cuda")
part = torch.empty(npart, 16, device="cuda")
sx, sw = torch.empty(2, device="cuda"), torch.empty(2, device="cuda")
pp = torch.zeros(1032, device="cuda")


def f32():
    _lib.call("eav_eegnet_fir_fwd", P(x), P(w), P(y_f32), P(part), B, C, S, K, st)


def split():
    _lib.call("eav_absmax_scale", P(x), x.numel(), 1.0, P(pp), P(sx), st)
    _lib.call("eav_absmax_scale", P(w), w.numel(), 1.0, P(pp), P(sw), st)
    _lib.call("eav_eegnet_fir_fwd_split", P(x), P(w), P(sx), P(sw), P(y_sp), P(part), B, C, S, K, st)


for fn, name in ((f32, "fp32 MFMA"), (split, "split fp16")):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 10:.3f} ms")
print("scales", sx.tolist(), sw.tolist())
# float64 reference on two batch items
xd = x[:2].double().cpu().reshape(2 * C, 1, S)
wd = w.double().cpu().reshape(8, 1, K)
ref = torch.nn.functional.conv1d(torch.nn.functional.pad(xd, ((K - 1) // 2, K - 1 - (K - 1) // 2)), wd)
ref = ref.reshape(2, C, 8, S).permute(0, 2, 1, 3)
scale = ref.abs().max().item()
for y, name in ((y_f32, "fp32 MFMA"), (y_sp, "split fp16")):
    err = (y[:2].double().cpu() - ref).abs()
    print(f"{name}: max |err| {err.max().item():.3e}  rms {err.pow(2).mean().sqrt().item():.3e}  (|y| max {scale:.3f}, "
          f"rel-to-max {err.max().item() / scale:.2e})")
print("split vs fp32 max diff", (y_sp - y_f32).abs().max().item())
