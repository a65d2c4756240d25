"""Accuracy and speed of the split-fp16 conv64 forward against the exact-fp32 MFMA kernel, both against float64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib, synth  # noqa: E402

B, T = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (64, 2500)))
padl = int(sys.argv[3]) if len(sys.argv) > 3 else 7
x = torch.from_numpy(synth.normal(1, (B, 64, T))).cuda()
w = torch.from_numpy(synth.uniform(2, (64, 64, 16), -0.03, 0.03)).cuda()
P, st = _lib.ptr, _lib.stream_ptr()
wTf, wTb = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
_lib.call("eav_conv64_prep_weights", P(w), P(wTf), P(wTb), st)
npart = _lib.plain("eav_conv64_fwd_nparts", B, T)
ya, yb = torch.empty(B, 64, T, device="cuda"), torch.empty(B, 64, T, device="cuda")
pa, pb = torch.zeros(npart, 128, device="cuda"), torch.zeros(npart, 128, device="cuda")
sx, sw, pp = torch.empty(4, device="cuda"), torch.empty(4, device="cuda"), torch.zeros(1032, device="cuda")


def f32():
    _lib.call("eav_conv64_fwd", P(x), P(wTf), P(ya), P(pa), B, T, padl, st)


def split():
    _lib.call("eav_absmax_scale", P(x), x.numel(), 1.0, P(pp), P(sx), st)
    _lib.call("eav_absmax_scale", P(w), w.numel(), 1.0, P(pp), P(sw), st)
    _lib.call("eav_conv64_fwd_split", P(x), P(wTf), P(sx), P(sw), P(yb), P(pb), B, T, padl, st)


for fn, name in ((f32, "fp32 MFMA"), (split, "split fp16 (incl. 2 absmax)")):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 10:.3f} ms")
nb = min(B, 4)
ref = torch.nn.functional.conv1d(torch.nn.functional.pad(x[:nb].double().cpu(), (padl, 15 - padl)), w.double().cpu())
for y, name in ((ya, "fp32 MFMA"), (yb, "split fp16")):
    err = (y[:nb].double().cpu() - ref).abs()
    print(f"{name}: max |err| {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e} (|y| max {ref.abs().max().item():.3f})")
print("stats diff", (pa.double().sum(0) - pb.double().sum(0)).abs().max().item(), "of", pa.double().sum(0).abs().max().item())
