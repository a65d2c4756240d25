#!/usr/bin/env python3
"""fc1 of the encoder MLP: GEMM (+bias, pre-activation kept, max|GELU|) followed by the GELU-applying conversion pass
versus ONE GEMM whose epilogue applies the GELU and writes the operand planes itself (eav_gemm_sp_planes); and the
LayerNorm -> conversion pair versus eav_layernorm_fwd_planes.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402
from tools.gemm_sp_bench import P, kpad, planes, timeit  # noqa: E402

_lib.load()
for tag, M in (("ast B=8", 9712), ("vit B=128", 25216)):
    N, K = 3072, 768
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.02
    bias = torch.randn(N, device="cuda") * 0.1
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    pre = torch.empty(M, N, device="cuda")
    slot = torch.zeros(4128, device="cuda")
    pl = torch.zeros((M + 31) // 32 * 32, 2 * kpad(N), dtype=torch.float16, device="cuda")

    def unfused():
        slot.zero_()
        _lib.call("eav_gemm_sp", P(pa), P(pb), P(pre), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), 3, None, None, 0, 0,
                  P(slot), None)
        _lib.call("eav_sp_convert_gelu", P(pre), M, N, N, P(slot), P(pl), None, None)

    slot2 = torch.zeros(4128, device="cuda")
    slot2[2048], slot2[2049] = 4096.0, 1 / 4096.0

    def fused():
        _lib.call("eav_gemm_sp_planes", P(pa), P(pb), None, P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), 1, P(pre), None,
                  0, 0, None, P(pl), P(slot2), None)

    def fused_nopre():
        _lib.call("eav_gemm_sp_planes", P(pa), P(pb), None, P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), 1, None, None,
                  0, 0, None, P(pl), P(slot2), None)

    def plain():
        _lib.call("eav_gemm_sp", P(pa), P(pb), P(pre), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), 0, None, None, 0, 0,
                  None, None)

    print(f"{tag} fc1 [{M},{N},{K}]: GEMM(gelu=3)+convert_gelu {timeit(unfused) * 1e3:7.1f} us | fused epilogue "
          f"{timeit(fused) * 1e3:7.1f} us | fused, no pre store {timeit(fused_nopre) * 1e3:7.1f} us | plain GEMM + bias "
          f"{timeit(plain) * 1e3:7.1f} us", flush=True)
    D = 768
    x = torch.randn(M, D, device="cuda")
    g, b = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.1
    y = torch.empty(M, D, device="cuda")
    st = torch.empty(2, M, device="cuda")
    ypl = torch.zeros((M + 31) // 32 * 32, 2 * D, dtype=torch.float16, device="cuda")

    def ln_unfused():
        slot.zero_()
        _lib.call("eav_layernorm_fwd_amax", P(x), P(g), P(b), P(y), P(st), P(st) + 4 * M, M, D, 1e-12, P(slot), None)
        _lib.call("eav_sp_convert", P(y), M, D, D, P(slot), P(ypl), None, None)

    def ln_fused():
        _lib.call("eav_layernorm_fwd_planes", P(x), P(g), P(b), None, P(ypl), P(slot2), P(st), P(st) + 4 * M, M, D, 1e-12, None)

    print(f"{tag} LayerNorm [{M},{D}]: LN + convert {timeit(ln_unfused) * 1e3:6.1f} us | LN -> planes {timeit(ln_fused) * 1e3:6.1f} us",
          flush=True)
