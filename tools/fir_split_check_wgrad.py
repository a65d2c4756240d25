"""Accuracy and speed of the split-fp16 FIR weight gradient against the exact-fp32 MFMA kernel, both against a float64
reference (einsum over an unfolded double-precision window, on the GPU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib, synth  # noqa: E402

B, C, S, K = (int(a) for a in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 30, 10000, 300)))
x = torch.from_numpy(synth.normal(1, (B, C, S))).cuda()
y1 = torch.from_numpy(synth.normal(2, (B, 8, C, S))).cuda() * 0.7 + 0.1
g1 = torch.from_numpy(synth.normal(3, (B, 8, C, S))).cuda() * 1e-4
bn = torch.from_numpy(synth.uniform(4, (6, 8), 0.5, 1.5)).cuda().contiguous()    # mean, invstd, scale, shift, m1, m2
bn[4] *= 1e-5
bn[5] *= 1e-5
P, st = _lib.ptr, _lib.stream_ptr()
np32 = _lib.plain("eav_eegnet_fir_wgrad_nparts", B, C, S)
nps = _lib.plain("eav_eegnet_fir_wgrad_split_nparts", B, C, S)
p32, ps = torch.empty(np32, 8 * K, device="cuda"), torch.empty(nps, 8 * K, device="cuda")
d32, dsp = torch.empty(8, K, device="cuda"), torch.empty(8, K, device="cuda")
sx, sg, sdy = torch.empty(3, device="cuda"), torch.empty(3, device="cuda"), torch.empty(3, device="cuda")
pp = torch.zeros(1032, device="cuda")
_lib.call("eav_absmax_scale", P(x), x.numel(), 1.0, P(pp), P(sx), st)


def f32():
    _lib.call("eav_eegnet_fir_wgrad", P(x), P(y1), P(g1), P(bn), P(p32), B, C, S, K, st)
    _lib.call("eav_reduce_partials", P(p32), np32, 8 * K, 8 * K, 1.0, P(d32), st)


def split():
    _lib.call("eav_absmax_scale", P(g1), g1.numel(), 1.0, P(pp), P(sg), st)
    # dy-scale bound: max|g| enters as "max|dz| times the depthwise row norm"; a unit-norm stand-in weight makes it max|g|
    w2u = torch.zeros(64, C, device="cuda")
    w2u[:, 0] = 0.125
    _lib.call("eav_fir_dy_scale", P(bn), P(sg) + 8, 1, P(w2u), C, P(sdy), st)
    _lib.call("eav_eegnet_fir_wgrad_split", P(x), P(y1), P(g1), P(bn), P(sx), P(sdy), P(ps), B, C, S, K, st)
    _lib.call("eav_reduce_partials", P(ps), nps, 8 * K, 8 * K, 1.0, P(dsp), st)


for fn, name in ((f32, "fp32 MFMA"), (split, "split fp16 (incl. absmax pass over g1)")):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        fn()
    b.record()
    torch.cuda.synchronize()
    print(f"{name}: {a.elapsed_time(b) / 10:.3f} ms")
print("scales x", sx.tolist(), "dy", sdy.tolist())
mean, invstd, sc, _, m1, m2 = (bn[i].double().view(1, 8, 1, 1) for i in range(6))
dy = sc * (g1.double() - m1 - (y1.double() - mean) * invstd * m2)
padl = (K - 1) // 2
xp = torch.nn.functional.pad(x.double(), (padl, K - 1 - padl))
ref = torch.zeros(8, K, dtype=torch.float64, device="cuda")
for b in range(B):
    ref += torch.einsum("fcs,csk->fk", dy[b], xp[b].unfold(-1, K, 1))
scale = ref.abs().max().item()
for d, name in ((d32, "fp32 MFMA"), (dsp, "split fp16")):
    err = (d.double() - ref).abs()
    print(f"{name}: max |err| {err.max().item():.3e} rms {err.pow(2).mean().sqrt().item():.3e} (|dW| max {scale:.3e}, "
          f"rel-to-max {err.max().item() / scale:.2e})")
