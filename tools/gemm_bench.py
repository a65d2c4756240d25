#!/usr/bin/env python3
"""Micro-benchmark of eav_gemm_f32 on the shapes the AST / ViT schedules issue (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402


KERNEL = "eav_gemm_f32"


def run(name, M, N, K, tA, tB, batch=1, heads=1, strides=None, reps=10):
    lda = (M if tA else K)
    ldb = (N if tB else K)
    lda, ldb = (lda + 3) // 4 * 4, (ldb + 3) // 4 * 4
    A = torch.randn(batch, (K if tA else M), lda, device="cuda")
    B = torch.randn(batch, (K if tB else N), ldb, device="cuda")
    C = torch.empty(batch, M, N, device="cuda")
    sA = (A.stride(0) * heads, A.stride(0)) if batch > 1 else (0, 0)
    sB = (B.stride(0) * heads, B.stride(0)) if batch > 1 else (0, 0)
    sC = (C.stride(0) * heads, C.stride(0)) if batch > 1 else (0, 0)

    def call():
        _lib.call(KERNEL, A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, N, tA, tB, batch, heads,
                  sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], 1.0, None, 0, None, None, 0, 0, None)
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    tf = 2.0 * M * N * K * batch / ms / 1e9
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} tA={tA} tB={tB} batch={batch:5d}  {ms:8.3f} ms  {tf:7.1f} TFLOP/s")
    return ms


def run_splitk(name, M, N, K, reps=10):
    A = torch.randn(K, M, device="cuda")
    B = torch.randn(K, N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    ns = _lib.plain("eav_gemm_f32_splitk_plan", M, N, K)
    ws = torch.empty(ns * M * N, device="cuda")

    def call():
        _lib.call(KERNEL + "_splitk", A.data_ptr(), B.data_ptr(), C.data_ptr(), ws.data_ptr(), M, N, K, M, N, 1, 1, None)
    for _ in range(3):
        call()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        call()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:34s} M={M:6d} N={N:5d} K={K:6d} split-K x{ns:<3d}              {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    _lib.load()
    if len(sys.argv) > 1 and sys.argv[1] == "bf16":
        KERNEL = "eav_gemm_bf16"
    print("kernel:", KERNEL)
    for tag, M in (("ast B=8", 9712), ("vit B=128", 25216)):
        print("==", tag)
        run("qkv fwd (NT)", M, 2304, 768, 0, 0)
        run("fc1 fwd (NT)", M, 3072, 768, 0, 0)
        run("fc2 fwd (NT)", M, 768, 3072, 0, 0)
        run("o fwd (NT)", M, 768, 768, 0, 0)
        run("fc2 dgrad (NN)", M, 3072, 768, 0, 1)
        run("fc1 dgrad (NN)", M, 768, 3072, 0, 1)
        run_splitk("fc1 wgrad (TN)", 3072, 768, M)
        run_splitk("fc2 wgrad (TN)", 768, 3072, M)
        run_splitk("qkv wgrad (TN)", 2304, 768, M)
        run_splitk("o wgrad (TN)", 768, 768, M)
    print("== attention ast B=8 (96 heads, N=1214)")
    run("QK^T", 1214, 1214, 64, 0, 0, batch=96, heads=12)
    run("PV", 1214, 64, 1214, 0, 1, batch=96, heads=12)
    run("dV = P^T dO", 1214, 64, 1214, 1, 1, batch=96, heads=12)
    print("== attention vit B=128 (1536 heads, N=197)")
    run("QK^T", 197, 197, 64, 0, 0, batch=1536, heads=12)
    run("PV", 197, 64, 197, 0, 1, batch=1536, heads=12)
    run("dV = P^T dO", 197, 64, 197, 1, 1, batch=1536, heads=12)

    print("== fused attention forward (fp32 MFMA)")
    for tag, Bn, N in (("ast B=8", 8, 1214), ("vit B=128", 128, 197)):
        H, D = 12, 768
        qkv = torch.randn(Bn * N, 3 * D, device="cuda")
        ao = torch.empty(Bn * N, D, device="cuda")
        lse = torch.empty(Bn * H, N, device="cuda")
        f = lambda: _lib.call("eav_attn_fwd", qkv.data_ptr(), ao.data_ptr(), lse.data_ptr(), Bn, H, N, 64, 0.125, None)  # noqa: E731
        for _ in range(3):
            f()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            f()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"{tag:12s} fwd {ms:8.3f} ms  {4.0 * Bn * H * N * N * 64 / ms / 1e9:7.1f} TFLOP/s")
        dO, dq, delta = torch.randn(Bn * N, D, device="cuda"), torch.empty(Bn * N, 3 * D, device="cuda"), torch.empty(Bn * H, N, device="cuda")
        g = lambda: _lib.call("eav_attn_bwd", qkv.data_ptr(), ao.data_ptr(), dO.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), Bn, H, N, 64, 0.125, None)  # noqa: E731
        for _ in range(3):
            g()
        a.record()
        for _ in range(10):
            g()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"{tag:12s} bwd {ms:8.3f} ms  {10.0 * Bn * H * N * N * 64 / ms / 1e9:7.1f} TFLOP/s (5 algorithmic products)")
