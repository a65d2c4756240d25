#!/usr/bin/env python3
"""separableConv forward / data gradient at the bench shape [64,64,2500]: frequency-domain kernels
(csrc/eegnet_conv64_fft.hip) beside the direct fp32 MFMA kernel (csrc/eegnet_conv64.hip)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eav_amd import _lib as L  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    B, T = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (64, 2500)))
    torch.manual_seed(0)
    x = torch.randn(B, 64, T, device="cuda")
    w = (torch.rand(64, 64, 16, device="cuda") - 0.5) * 0.1
    P = lambda t: t.data_ptr()  # noqa: E731
    ws = torch.zeros(L.plain("eav_conv64_fft_ws_floats", B, T), device="cuda")
    pf = torch.zeros(L.plain("eav_conv64_fft_nparts", B, T), 128, device="cuda")
    pm = torch.zeros(L.plain("eav_conv64_fwd_nparts", B, T), 128, device="cuda")
    wTf, wTb = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
    L.call("eav_conv64_prep_weights", P(w), P(wTf), P(wTb), None)
    y_f, y_m = torch.empty(B, 64, T, device="cuda"), torch.empty(B, 64, T, device="cuda")
    for name, bwd, wT, padl in (("fwd  ", 0, wTf, 7), ("dgrad", 1, wTb, 8)):
        t_f = timeit(lambda: L.call("eav_conv64_fft_fwd", P(x), P(w), P(y_f), P(pf) if not bwd else None, P(ws), B, T, bwd, None))
        t_m = timeit(lambda: L.call("eav_conv64_fwd", P(x), P(wT), P(y_m), P(pm) if not bwd else None, B, T, padl, None))
        print(f"{name} fft {t_f:.3f} ms   mfma {t_m:.3f} ms   max |diff| {float((y_f - y_m).abs().max()):.2e} "
              f"(max |y| {float(y_m.abs().max()):.2f})")
    du = torch.randn(B, 64, T, device="cuda")
    L.call("eav_conv64_fft_fwd", P(x), P(w), P(y_f), None, P(ws), B, T, 0, None)        # leaves the input spectra in ws
    d_f, d_m = torch.empty(64, 64, 16, device="cuda"), torch.empty(64, 64, 16, device="cuda")
    npw = L.plain("eav_conv64_wgrad_nparts", B, T)
    pw = torch.empty(npw, 65536, device="cuda")
    t_f = timeit(lambda: L.call("eav_conv64_fft_wgrad", P(du), P(d_f), P(ws), B, T, None))

    def mf():
        L.call("eav_conv64_wgrad", P(du), P(x), P(pw), B, T, 7, None)
        L.call("eav_reduce_partials", P(pw), npw, 65536, 65536, 1.0, P(d_m), None)
    t_m = timeit(mf)
    print(f"wgrad fft {t_f:.3f} ms   mfma {t_m:.3f} ms   max |diff| / max |dW| {float((d_f - d_m).abs().max() / d_m.abs().max()):.2e}")


if __name__ == "__main__":
    main()
