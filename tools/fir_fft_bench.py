#!/usr/bin/env python3
"""firstConv forward / weight gradient at the bench shape [64,1,30,10000], K = 300: FFT kernels (csrc/eegnet_fir_fft.hip)
beside the Toeplitz-MFMA kernels (csrc/eegnet_fir.hip) - time per launch and agreement."""
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
    B, C, S, K = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (64, 30, 10000, 300)))
    torch.manual_seed(0)
    x = torch.randn(B, C, S, device="cuda")
    w = (torch.rand(8, K, device="cuda") - 0.5) * 0.2
    y_f, y_m = torch.empty(B, 8, C, S, device="cuda"), torch.empty(B, 8, C, S, device="cuda")
    pf = torch.zeros(L.plain("eav_eegnet_fir_fwd_fft_nparts", B, C, S), 16, device="cuda")
    pm = torch.zeros(L.plain("eav_eegnet_fir_fwd_nparts", B, C, S), 16, device="cuda")
    P = lambda t: t.data_ptr()  # noqa: E731
    t_f = timeit(lambda: L.call("eav_eegnet_fir_fwd_fft", P(x), None, P(w), P(y_f), P(pf), B, C, S, K, None))
    t_m = timeit(lambda: L.call("eav_eegnet_fir_fwd", P(x), P(w), P(y_m), P(pm), B, C, S, K, None))
    print(f"fwd   fft {t_f:.3f} ms   mfma {t_m:.3f} ms   max |diff| {float((y_f - y_m).abs().max()):.2e} "
          f"(max |y| {float(y_m.abs().max()):.2f}); stats diff {float((pf.sum(0) - pm.sum(0)).abs().max()):.2e}")
    g1 = torch.randn(B, 8, C, S, device="cuda")
    bn = torch.zeros(48, device="cuda")
    bn[8:24] = 1.0
    bn[32:48] = 0.01
    ws = torch.empty(L.plain("eav_eegnet_fir_wgrad_fft_ws_floats", B, C, S), device="cuda")
    npw = L.plain("eav_eegnet_fir_wgrad_nparts", B, C, S)
    part = torch.empty(npw, 8 * K, device="cuda")
    d_f, d_m = torch.empty(8, K, device="cuda"), torch.empty(8, K, device="cuda")
    for name, y1 in (("train", y_m), ("eval ", None)):
        t_f = timeit(lambda: L.call("eav_eegnet_fir_wgrad_fft", P(x), None, P(y1) if y1 is not None else None, P(g1), P(bn),
                                    P(ws), P(d_f), B, C, S, K, None))

        def mf():
            L.call("eav_eegnet_fir_wgrad", P(x), P(y1) if y1 is not None else None, P(g1), P(bn), P(part), B, C, S, K, None)
            L.call("eav_reduce_partials", P(part), npw, 8 * K, 8 * K, 1.0, P(d_m), None)
        t_m = timeit(mf)
        print(f"wgrad {name} fft {t_f:.3f} ms   mfma {t_m:.3f} ms   max |diff| / max |dW| "
              f"{float((d_f - d_m).abs().max() / d_m.abs().max()):.2e}")


if __name__ == "__main__":
    main()
