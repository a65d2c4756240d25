"""Accuracy and speed of the split-fp16 conv64 weight gradient against the exact-fp32 MFMA kernel (float64 reference)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib, synth  # noqa: E402

B, T = (int(a) for a in (sys.argv[1:3] if len(sys.argv) > 2 else (64, 2500)))
x = torch.from_numpy(synth.normal(1, (B, 64, T))).cuda()
du = torch.from_numpy(synth.normal(2, (B, 64, T))).cuda() * 1e-4
P, st = _lib.ptr, _lib.stream_ptr()
npart = _lib.plain("eav_conv64_wgrad_nparts", B, T)
pa, pb = torch.empty(npart, 65536, device="cuda"), torch.empty(npart, 65536, device="cuda")
da, db = torch.empty(64, 64, 16, device="cuda"), torch.empty(64, 64, 16, device="cuda")
sx, sd, pp = torch.empty(4, device="cuda"), torch.empty(4, device="cuda"), torch.zeros(1032, device="cuda")
_lib.call("eav_absmax_scale", P(x), x.numel(), 1.0, P(pp), P(sx), st)
_lib.call("eav_absmax_scale", P(du), du.numel(), 1.0, P(pp), P(sd), st)


def f32():
    _lib.call("eav_conv64_wgrad", P(du), P(x), P(pa), B, T, 7, st)
    _lib.call("eav_reduce_partials", P(pa), npart, 65536, 65536, 1.0, P(da), st)


def split():
    _lib.call("eav_conv64_wgrad_split", P(du), P(x), P(sd), P(sx), P(pb), B, T, 7, st)
    _lib.call("eav_reduce_partials", P(pb), npart, 65536, 65536, 1.0, P(db), st)


for fn, name in ((f32, "fp32 MFMA"), (split, "split fp16")):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 10:.3f} ms (incl. reduce)")
xp = torch.nn.functional.pad(x.double(), (7, 8))
ref = torch.zeros(64, 64, 16, dtype=torch.float64, device="cuda")
for b in range(B):
    ref += torch.einsum("ot,itk->oik", du[b].double(), xp[b].unfold(-1, 16, 1))
for d, name in ((da, "fp32 MFMA"), (db, "split fp16")):
    err = (d.double() - ref).abs()
    print(f"{name}: max |err| {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e} (|dW| max {ref.abs().max().item():.3e})")
