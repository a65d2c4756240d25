#!/usr/bin/env python3
"""Split-operand fused attention (eav_attn_*_sp) against a float64 reference, beside the exact-fp32 kernels, and its
speed on the AST / ViT shapes.  Run on the GPU box.  usage: attn_sp_check.py [check]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib  # noqa: E402

P = _lib.ptr
SLOT = 4128


def prep(x, B, N, ncols, secw, tmask, amax=True):
    Npad = _lib.plain("eav_attn_sp_npad", N)
    slot = torch.zeros(SLOT, device="cuda")
    if amax:
        _lib.call("eav_sp_absmax", P(x), B * N, ncols, ncols, P(slot), None)
    rowp = torch.empty(B * N, 2 * ncols, dtype=torch.float16, device="cuda")
    tp = torch.zeros(B, ncols // 64, 64, 2 * Npad, dtype=torch.float16, device="cuda")
    _lib.call("eav_attn_sp_prep", P(x), P(slot), P(rowp), P(tp), B, N, ncols, secw, tmask, None)
    return slot, rowp, tp


def reference(qkv, dO, B, H, N):
    D = H * 64
    x = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)   # [3,B,H,N,64]
    q, k, v = x[0], x[1], x[2]
    s = (q @ k.transpose(-1, -2)) * 0.125
    p = torch.softmax(s, -1)
    o = p @ v                                                                                    # [B,H,N,64]
    out = o.permute(0, 2, 1, 3).reshape(B * N, D)
    out.backward(dO.double())
    dqkv = x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * D)
    lse = torch.logsumexp(s, -1).reshape(B * H, N)
    return out.detach(), dqkv, lse.detach()


def run_sp(qkv, dO, B, H, N):
    D = H * 64
    s_qkv, rowp, tp = prep(qkv, B, N, 3 * D, D, 7)
    ao = torch.empty(B * N, D, device="cuda")
    lse = torch.empty(B * H, N, device="cuda")
    _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), None, B, H, N, 64, 0.125, None)
    s_do, dorow, dotp = prep(dO, B, N, D, D, 1)
    s_ds = torch.zeros(SLOT, device="cuda")
    delta = torch.empty(B * H, N, device="cuda")
    dqkv = torch.empty(B * N, 3 * D, device="cuda")
    _lib.call("eav_attn_bwd_sp", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds), P(ao), P(dO), P(lse),
              P(delta), P(dqkv), None, B, H, N, 64, 0.125, None)
    return ao, dqkv, lse


def run_f32(qkv, dO, B, H, N):
    D = H * 64
    ao = torch.empty(B * N, D, device="cuda")
    lse = torch.empty(B * H, N, device="cuda")
    _lib.call("eav_attn_fwd", P(qkv), P(ao), P(lse), B, H, N, 64, 0.125, None)
    delta = torch.empty(B * H, N, device="cuda")
    dqkv = torch.empty(B * N, 3 * D, device="cuda")
    _lib.call("eav_attn_bwd", P(qkv), P(ao), P(dO), P(lse), P(delta), P(dqkv), B, H, N, 64, 0.125, None)
    return ao, dqkv, lse


def check(B, H, N, qscale=1.0, gscale=1e-3, spike=False):
    torch.manual_seed(B * 1000 + N)
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda") * qscale
    if spike:   # one key strongly aligned with one query: the running maximum jumps inside a tile
        qkv[5, D:D + 64] = qkv[3, :64] * 6.0
    dO = torch.randn(B * N, D, device="cuda") * gscale
    ro, rg, rl = reference(qkv, dO, B, H, N)
    out = {}
    for name, f in (("split", run_sp), ("fp32", run_f32)):
        ao, dqkv, lse = f(qkv, dO, B, H, N)
        eo = ((ao.double() - ro).abs().max() / ro.abs().max()).item()
        el = (lse.double() - rl).abs().max().item()
        secs = []
        for i, nm in enumerate("QKV"):
            a, r = dqkv[:, i * D:(i + 1) * D].double(), rg[:, i * D:(i + 1) * D]
            secs.append(f"d{nm} {((a - r).abs().max() / r.abs().max()).item():.2e}")
        out[name] = f"O {eo:.2e}  lse {el:.2e}  " + "  ".join(secs)
    print(f"B={B} H={H} N={N} qscale={qscale} spike={spike}: max err / max|ref|")
    for k, v in out.items():
        print(f"   {k:6s} {v}")


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


def bench(tag, B, N, H=12):
    D = H * 64
    qkv = torch.randn(B * N, 3 * D, device="cuda")
    dO = torch.randn(B * N, D, device="cuda") * 1e-3
    s_qkv, rowp, tp = prep(qkv, B, N, 3 * D, D, 7)
    s_do, dorow, dotp = prep(dO, B, N, D, D, 1)
    ao = torch.empty(B * N, D, device="cuda")
    lse = torch.empty(B * H, N, device="cuda")
    s_ds = torch.zeros(SLOT, device="cuda")
    delta = torch.empty(B * H, N, device="cuda")
    dqkv = torch.empty(B * N, 3 * D, device="cuda")
    fl = 4.0 * B * H * N * N * 64
    t_prep = timeit(lambda: _lib.call("eav_attn_sp_prep", P(qkv), P(s_qkv), P(rowp), P(tp), B, N, 3 * D, D, 7, None))
    t_fwd = timeit(lambda: _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), None, B, H, N, 64,
                                     0.125, None))
    t_bwd = timeit(lambda: _lib.call("eav_attn_bwd_sp", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds),
                                     P(ao), P(dO), P(lse), P(delta), P(dqkv), None, B, H, N, 64, 0.125, None))
    f_fwd = timeit(lambda: _lib.call("eav_attn_fwd", P(qkv), P(ao), P(lse), B, H, N, 64, 0.125, None))
    f_bwd = timeit(lambda: _lib.call("eav_attn_bwd", P(qkv), P(ao), P(dO), P(lse), P(delta), P(dqkv), B, H, N, 64,
                                     0.125, None))
    print(f"{tag}: prep(qkv) {t_prep * 1e3:.0f} us | fwd split {t_fwd:.3f} ms ({fl / t_fwd / 1e9:.0f} TF) fp32 "
          f"{f_fwd:.3f} ms ({fl / f_fwd / 1e9:.0f} TF) | bwd split {t_bwd:.3f} ms ({2.5 * fl / t_bwd / 1e9:.0f} TF) fp32 "
          f"{f_bwd:.3f} ms ({2.5 * fl / f_bwd / 1e9:.0f} TF)")


if __name__ == "__main__":
    _lib.load()
    check(1, 2, 64)
    check(2, 3, 197)
    check(1, 2, 1214)
    check(2, 2, 300, qscale=3.0, spike=True)
    check(1, 1, 33)
    if len(sys.argv) > 1 and sys.argv[1] == "check":
        sys.exit(0)
    bench("ast B=8 ", 8, 1214)
    bench("vit B=128", 128, 197)
    _lib.call("eav_attn_sp_set_nw4_above", 512)
    bench("vit B=128 (2-wave blocks)", 128, 197)
    _lib.call("eav_attn_sp_set_nw4_above", 128)
    bench("shallow B=32", 32, 488, H=1)
