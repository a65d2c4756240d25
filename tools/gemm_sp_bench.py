#!/usr/bin/env python3
"""eav_gemm_sp (split-operand fp16 MFMA) on the AST / ViT shapes: accuracy against float64 beside the exact-fp32
kernel, and TFLOP/s (algorithmic 2MNK; the kernel issues 3x that on the matrix cores).  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402

P = _lib.ptr


def kpad(k):
    return (k + 31) // 32 * 32


def planes(x, want=True, wantT=False):
    """x [R,C] fp32 device -> (slot, planes [R, 2*Cp] f16 | None, planesT [C, 2*Rp] f16 | None)"""
    R, C = x.shape
    slot = torch.zeros(4128, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, x.stride(0), P(slot), None)
    d = torch.empty(R, 2 * kpad(C), dtype=torch.float16, device="cuda") if want else None
    dT = torch.empty(C, 2 * kpad(R), dtype=torch.float16, device="cuda") if wantT else None
    _lib.call("eav_sp_convert", P(x), R, C, x.stride(0), P(slot), P(d), P(dT), None)
    return slot, d, dT


def row_planes(x):
    """row planes with the rows zero-padded to a multiple of 32 (operands of the token-contracting eav_gemm_sp_splitk)"""
    R, C = x.shape
    slot = torch.zeros(4128, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, x.stride(0), P(slot), None)
    d = torch.zeros((R + 31) // 32 * 32, 2 * kpad(C), dtype=torch.float16, device="cuda")
    _lib.call("eav_sp_convert", P(x), R, C, x.stride(0), P(slot), P(d), None, None)
    return slot, d


def timeit(f, reps=10):
    for _ in range(3):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def check(M, N, K, scaleA=1.0, scaleB=0.02, wide=False):
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda") * scaleA
    B = torch.randn(N, K, device="cuda") * scaleB
    if wide:   # rows of very different magnitude (gradient-like tensors)
        A *= torch.exp(torch.randn(M, 1, device="cuda") * 4)
    ref = A.double() @ B.double().t()
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0, 0,
              None, None)
    C32 = torch.empty(M, N, device="cuda")
    _lib.call("eav_gemm_f32", P(A), P(B), P(C32), M, N, K, K, K, N, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, None,
              None, 0, 0, None)
    den = (A.double().abs() @ B.double().abs().t())
    e_sp = ((C.double() - ref).abs() / den).max().item()
    e_32 = ((C32.double() - ref).abs() / den).max().item()
    r_sp = ((C.double() - ref).norm() / ref.norm()).item()
    r_32 = ((C32.double() - ref).norm() / ref.norm()).item()
    rows_ok = (((C.double() - ref).abs() / den).max(dim=1).values <= 2 * e_32).double().mean().item()
    print(f"check M={M} N={N} K={K} wide={wide}: max err / sum|a||b|  split {e_sp:.2e}  fp32 {e_32:.2e};  "
          f"rel Frobenius split {r_sp:.2e} fp32 {r_32:.2e};  rows within 2x the fp32 kernel's worst element: {100 * rows_ok:.1f} %")
    return e_sp, e_32


def check_wgrad(Mtok, N, K):
    """dW[N,K] = dY^T X through the transposed planes + split-K."""
    torch.manual_seed(7)
    dY = torch.randn(Mtok, N, device="cuda") * 1e-3
    X = torch.randn(Mtok, K, device="cuda")
    ref = dY.double().t() @ X.double()
    sa, paT = row_planes(dY)
    sb, pbT = row_planes(X)
    C = torch.empty(N, K, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", N, K, Mtok)
    ws = torch.empty(ns * N * K, device="cuda")
    _lib.call("eav_gemm_sp_splitk", P(paT), P(pbT), P(C), P(ws), P(sa), P(sb), N, K, Mtok, 0, None)
    den = dY.double().abs().t() @ X.double().abs()
    e = ((C.double() - ref).abs() / den).max().item()
    print(f"check wgrad tokens={Mtok} N={N} K={K} (split-K x{ns}): max err / sum|a||b| {e:.2e}")


def bench(name, M, N, K, epi=False):
    """tile codes (eav_gemm_sp_set_tile): 1 = 128x128 two-accumulator planes, 2 = 256x128, +4 = single-accumulator planes, +8 = one workgroup per tile instead of persistent workgroups (the operand planes are
    converted in the matching format)."""
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.02
    C = torch.empty(M, N, device="cuda")
    bias = torch.randn(N, device="cuda") if epi else None
    pre = torch.empty(M, N, device="cuda") if epi else None
    ref = None
    out = []
    for tile in [int(v) for v in os.environ.get('TILES', '1,9').split(',')]:
        _lib.call("eav_gemm_sp_set_tile", tile)
        sa, pa, _ = planes(A)
        sb, pb, _ = planes(B)
        ms = timeit(lambda: _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0,
                                      P(bias), 1 if epi else 0, P(pre), None, 0, 0, None, None))
        if ref is None:
            ref = C.clone()
        err = ((C - ref).abs().max() / ref.abs().max()).item()
        out.append(f"tile{tile}: {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF (d {err:.0e})")
    _lib.call("eav_gemm_sp_set_tile", 0)
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:6d}  " + "   ".join(out))


def bench_splitk(name, M, N, K):
    """weight-gradient shape: C[M,N] = sum over K tokens of A[t,m] B[t,n], row planes of A [K,M], B [K,N]"""
    A = torch.randn(K, M, device="cuda")
    B = torch.randn(K, N, device="cuda")
    sa, pa = row_planes(A)
    sb, pb = row_planes(B)
    C = torch.empty(M, N, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", M, N, K)
    ws = torch.empty(max(ns, 1) * M * N, device="cuda")
    ms = timeit(lambda: _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), M, N, K, 0, None))
    print(f"{name:28s} M={M:6d} N={N:5d} K={K:6d}  split-K x{ns:<3d} {ms:7.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TF")


def bench_convert(R, C):
    x = torch.randn(R, C, device="cuda")
    slot = torch.zeros(4128, device="cuda")
    d = torch.empty(R, 2 * kpad(C), dtype=torch.float16, device="cuda")
    dT = torch.empty(C, 2 * kpad(R), dtype=torch.float16, device="cuda")
    gb = R * C * 4 / 1e9
    for swap in (0,):
        ms0 = timeit(lambda: _lib.call("eav_sp_absmax", P(x), R, C, C, P(slot), None))
        ms1 = timeit(lambda: _lib.call("eav_sp_convert", P(x), R, C, C, P(slot), P(d), None, None))
        ms2 = timeit(lambda: _lib.call("eav_sp_convert", P(x), R, C, C, P(slot), P(d), P(dT), None))
        print(f"convert [{R},{C}]: absmax {ms0 * 1e3:6.1f} us ({gb / ms0:5.2f} TB/s)  planes {ms1 * 1e3:6.1f} us "
              f"({2 * gb / ms1:5.2f} TB/s)  planes+T {ms2 * 1e3:6.1f} us ({3 * gb / ms2:5.2f} TB/s)")
    _lib.call("eav_gemm_sp_set_tile", 0)


def checks():
    check(512, 384, 768)
    check(1000, 200, 100)           # ragged everything
    check(9712, 768, 3072)
    check(2048, 768, 768, wide=True)
    check(300, 130, 40, scaleA=1e-6, scaleB=1e4)
    check(9712, 2304, 768)          # 1368 tiles: persistent workgroups, several tiles each
    check_wgrad(9712, 768, 256)
    check_wgrad(1214, 200, 136)


if __name__ == "__main__":
    _lib.load()
    checks()
    print("-- single-accumulator mode (lo not lifted by 2^11)")
    _lib.call("eav_gemm_sp_set_tile", 4)
    for a in ((512, 384, 768), (1000, 200, 100), (9712, 768, 3072), (9712, 2304, 768)):
        check(*a)
    _lib.call("eav_gemm_sp_set_tile", 0)
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(0)
    for tag, M in (("ast B=8", 9712), ("vit B=128", 25216)):
        print("==", tag)
        bench("qkv fwd", M, 2304, 768)
        bench("fc1 fwd (+bias,gelu,pre)", M, 3072, 768, epi=True)
        bench("fc1 fwd", M, 3072, 768)
        bench("fc2 fwd / fc1 dgrad", M, 768, 3072)
        bench("o fwd", M, 768, 768)
        bench_splitk("fc1 wgrad", 3072, 768, M)
        bench_splitk("fc2 wgrad", 768, 3072, M)
        bench_splitk("qkv wgrad", 2304, 768, M)
        bench_splitk("o wgrad", 768, 768, M)
        bench_convert(M, 768)
        bench_convert(M, 3072)
    bench("square 4096", 4096, 4096, 4096)
    bench("square 8192", 8192, 8192, 8192)
