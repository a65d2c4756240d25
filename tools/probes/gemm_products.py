#!/usr/bin/env python3
"""The split GEMM on the products of an encoder layer with the epilogues the training step uses - ViT B = 128 (25216 token
rows) and AST B = 8 (9712): median of 5 x 20 launches per product, the token-contracting weight gradients on three / two /
one terms.  One library per process (EAV_LIB_PATH selects an A/B build: tools/probes/build_variant.sh), run on ONE box:
    for so in "" tools/probes/build/libeav_r05gemm.so; do EAV_LIB_PATH=$so python3 tools/probes/gemm_products.py; done
(Round 6 also ran it with the tile walk switched at run time - static ids against tiles popped from per-XCD counters behind a
staggered start; that experiment lost and its code is gone: profiles/r06_gemm_sched_ab.txt, commit 63a5f0e.)"""
import os
import statistics
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from eav_amd import _lib  # noqa: E402
from gemm_sp_bench import P, planes, row_planes, timeit, kpad  # noqa: E402

L = _lib.load()
SLOT = 4128
which = sys.argv[1] if len(sys.argv) > 1 else "both"
VARIANTS = [(os.path.basename(os.environ.get("EAV_LIB_PATH") or "libeav_hip.so")[6:-3] or "hip", 0, 0)]
print("product".ljust(44) + "".join(f"{n:>10s}" for n, _, _ in VARIANTS) + "   (us per launch; TFLOP/s of the best)")
tot = {}
for tag, M in (("vit", 25216), ("ast", 9712)):
    if which not in ("both", tag):
        continue
    for shared in (0, 1):
        for name, N, K, gelu, bias, want_pre, want_c, want_planes, colsum, amax in (
                ("qkv fwd (planes, maxima)", 2304, 768, 0, 1, 0, 0, 1, 0, 1),
                ("fc1 fwd (gelu, pre, planes)", 3072, 768, 1, 1, 1, 0, 1, 0, 0),
                ("fc2 dgrad (gelu', planes, colsum)", 3072, 768, 2, 0, 1, 0, 1, 1, 0),
                ("o fwd (bias, residual, C)", 768, 768, 0, 1, 0, 1, 0, 0, 0),
                ("fc2 fwd (bias, residual, C)", 768, 3072, 0, 1, 0, 1, 0, 0, 0),
                ("fc1 dgrad (C, maxima)", 768, 3072, 0, 0, 0, 1, 0, 0, 1),
                ("qkv dgrad (C, maxima)", 768, 2304, 0, 0, 0, 1, 0, 0, 1)):
            if shared and "fwd" in name:
                continue              # (the 256 x 128 form is the backward's: EAV_GEMM_SHARED_GPU)
            torch.manual_seed(N + K)
            A = torch.randn(M, K, device="cuda")
            B = torch.randn(N, K, device="cuda") * 0.05
            sa, pa, _ = planes(A)
            sb, pb, _ = planes(B)
            Cm = torch.empty(M, N, device="cuda") if want_c else None
            res = torch.randn(M, N, device="cuda") if (want_c and bias) else None
            pre = torch.randn(M, N, device="cuda") if want_pre else None
            bia = torch.randn(N, device="cuda") if bias else None
            am = torch.zeros(SLOT, device="cuda") if amax else None
            pl = torch.zeros((M + 31) // 32 * 32, 2 * kpad(N), dtype=torch.float16, device="cuda") if want_planes else None
            slot = torch.zeros(SLOT, device="cuda")
            slot[2048], slot[2049] = 1024.0, 1.0 / 1024.0
            part = torch.empty((M + 63) // 64, N, device="cuda") if colsum else None
            args = (P(pa), P(pb), P(Cm), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bia), gelu, P(pre), P(res), N if res is not None else 0,
                    0, P(am), P(pl), P(slot) if pl is not None else None, P(part), 2 if shared else 0, None)
            ts = {n: [] for n, _, _ in VARIANTS}
            outs = {}
            for rep in range(5):
                order = VARIANTS[rep % len(VARIANTS):] + VARIANTS[:rep % len(VARIANTS)]
                for n, dyn, stg in order:
                    ts[n].append(timeit(lambda: _lib.call("eav_gemm_sp_ex", *args), reps=20) * 1e3)
                    if rep == 0:
                        torch.cuda.synchronize()
                        outs[n] = (Cm.clone() if Cm is not None else None, pl.clone() if pl is not None else None)
            ref = outs[VARIANTS[0][0]]
            for n, o in outs.items():           # the tile walk must not change a bit of the result
                for a, b in zip(o, ref):
                    assert (a is None and b is None) or torch.equal(a, b), (tag, name, n)
            med = {n: statistics.median(v) for n, v in ts.items()}
            best = min(med, key=med.get)
            label = f"{tag} {'256x128 ' if shared else ''}{name}"
            print(label.ljust(44) + "".join(f"{med[n]:10.1f}" for n, _, _ in VARIANTS)
                  + f"   {2.0 * M * N * K / 1e6 / med[best]:5.0f} ({best})", flush=True)
            for n in med:
                tot[(tag, n)] = tot.get((tag, n), 0.0) + med[n]
    print(f"{tag} sum".ljust(44) + "".join(f"{tot[(tag, n)]:10.1f}" for n, _, _ in VARIANTS), flush=True)
    # weight gradients (token-contracting, split-K): three terms against two (hi_grad.hi_act + lo_grad.hi_act)
    for name, n1, n2 in (("fc2 wgrad", 768, 3072), ("fc1 wgrad", 3072, 768), ("o wgrad", 768, 768), ("qkv wgrad", 2304, 768)):
        torch.manual_seed(n1)
        G = torch.randn(M, n1, device="cuda") * 1e-3
        X = torch.randn(M, n2, device="cuda")
        sg, pg = row_planes(G)
        sx, px = row_planes(X)
        W = torch.empty(n1, n2, device="cuda")
        ns = _lib.plain("eav_gemm_sp_splitk_plan", n1, n2, M)
        ws = torch.empty(max(ns, 1) * n1 * n2, device="cuda")
        t = {}
        for fn in ("eav_gemm_sp_splitk", "eav_gemm_sp_splitk_x2", "eav_gemm_sp_splitk_x1"):
            t[fn] = statistics.median(timeit(lambda: _lib.call(fn, P(pg), P(px), P(W), P(ws), P(sg), P(sx), n1, n2, M, 0, None),
                                             reps=20) * 1e3 for _ in range(3))
        tf = 2.0 * M * n1 * n2 / 1e6
        print(f"{tag} {name} ({ns} slices)".ljust(44) + "  ".join(f"{k[12:] or 'x3':>10s} {v:7.1f} us {tf / v:5.0f} TF" for k, v in t.items()),
              flush=True)
