#!/usr/bin/env python3
"""What clock / power does the GPU hold under each kind of load?  Runs a load loop in this process and samples
`rocm-smi` (sclk, average power) from a thread a few times while it runs.  Loads: idle, MFMA-only peak kernel, the
split GEMM (three terms / one term), a streaming copy."""
import os
import subprocess
import sys
import threading
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes  # noqa: E402

_lib.load()


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
    except Exception as e:  # noqa: BLE001
        return f"rocm-smi failed: {e}"
    keep = [ln.strip() for ln in out.splitlines() if ("sclk" in ln or "Power" in ln or "mclk" in ln) and "GPU[0]" in ln]
    return " | ".join(k.split(":", 1)[1].strip() if ":" in k else k for k in keep)


def run(name, f, seconds=6.0, burst=20):
    samples = []
    stop = False

    def sampler():
        time.sleep(1.5)
        while not stop:
            samples.append(smi())
            time.sleep(1.0)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(burst):
            f()
        torch.cuda.synchronize()
        n += burst
    stop = True
    th.join()
    print(f"== {name}: {n} launches in {seconds:.0f} s ({seconds / n * 1e3:.3f} ms each)")
    for s in samples[:4]:
        print("   ", s)
    sys.stdout.flush()


def main():
    M, N, K = 25216, 768, 3072
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    sink = torch.zeros(64, device="cuda")
    src = torch.randn(64 << 20, device="cuda")
    dst = torch.empty_like(src)
    run("idle", lambda: time.sleep(0.01), 4.0)
    run("MFMA only (eav_peak_mfma_f16)", lambda: _lib.call("eav_peak_mfma_f16", P(sink), 1024, 4000, None))
    run("split GEMM, three terms", lambda: _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0,
                                                     None, 0, None, None, 0, 0, None, None))
    run("split GEMM, one term", lambda: _lib.call("eav_gemm_sp_x1", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0,
                                                  None, 0, None, None, 0, 0, None, None))
    run("streaming copy", lambda: _lib.call("eav_peak_copy", P(src), P(dst), src.numel(), None))


if __name__ == "__main__":
    main()
