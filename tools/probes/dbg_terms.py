#!/usr/bin/env python3
"""debug: (1) the [300,130,40] two-term kernel case, (2) fused_dact on / off gradient gap per tensor for wgrad_terms 3 and 2"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from eav_amd import _lib, transformer as T
from eav_amd.optim import CrossEntropyLoss, FusedAdam
import tests.test_split_kernels_gpu as K
import tests.test_transformer_model_gpu as Mt
P = _lib.ptr
for (Tn, n1, n2) in ((300, 132, 40), (512, 256, 136)):
    torch.manual_seed(Tn + n1)
    G = torch.randn(Tn, n1, device="cuda") * 1e-3 * (1 + torch.arange(n1, device="cuda") % 7)
    X = torch.randn(Tn, n2, device="cuda") * (1 + torch.arange(n2, device="cuda") % 3)
    sg, pg = K.row_planes(G); sx, px = K.row_planes(X)
    ns = _lib.plain("eav_gemm_sp_splitk_plan", n1, n2, Tn)
    ws = torch.empty(max(ns, 1) * n1 * n2, device="cuda")
    ref = G.double().t() @ X.double()
    den = G.double().abs().t() @ X.double().abs()
    n2p = K.kpad(n2); Tp = px.numel() // (2 * n2p)
    xhi = px.view(Tp, n2p // 8, 2, 8)[:Tn, :, 0, :].reshape(Tn, n2p)[:, :n2].double() * sx[2049].double()
    ref_hi = G.double().t() @ xhi
    for fn in ("eav_gemm_sp_splitk", "eav_gemm_sp_splitk_x2", "eav_gemm_sp_splitk_x1"):
        W = torch.empty(n1, n2, device="cuda")
        _lib.call(fn, P(pg), P(px), P(W), P(ws), P(sg), P(sx), n1, n2, Tn, 0, None)
        print(Tn, n1, n2, fn, "rel err", ((W.double() - ref).norm() / ref.norm()).item(), "max/den", ((W.double() - ref).abs() / den).max().item(),
              "vs hi-ref max/den", ((W.double() - ref_hi).abs() / den).max().item())
for kind in ("vit", "ast"):
    cfg = T.make_config(kind, hidden=128, layers=2, heads=2, ff=256)
    W = Mt._weights(kind, 13, 0.08, hidden=128, layers=2, heads=2, ff=256)
    x, y = Mt._batch(kind, cfg, 6, 4)
    xd, yd = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    def run(fused, terms, second=True):
        model = T.Encoder(cfg, W).cuda().train()
        model.precision, model.fused_dact, model.wgrad_terms = "split", fused, terms
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.01, decoupled=True)
        opt.zero_grad()
        CrossEntropyLoss()(model(xd).logits, yd).backward()
        if second:
            opt.step()
            model.eval()
            with torch.no_grad():
                model(xd)
            model.train()
            opt.zero_grad()
            CrossEntropyLoss()(model(xd).logits, yd).backward()
        torch.cuda.synchronize()
        return {k: p.grad.clone() for k, p in model.named_parameters()}
    for second in (False, True):
        for terms in (3, 2):
            g1, g0 = run(True, terms, second), run(False, terms, second)
            worst = sorted(((float((g1[k] - g0[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-30), k) for k in g1 if not k.endswith("k_proj.bias")), reverse=True)[:3]
            print(kind, "second step" if second else "first step", "wgrad_terms", terms, "worst fused-vs-unfused gaps:", [(f"{a:.2e}", k.split("layers.")[-1]) for a, k in worst])
        a, b = run(True, 3, second), run(True, 2, second)
        worst = sorted(((float((a[k] - b[k]).abs().max()) / (float(a[k].abs().max()) + 1e-30), k) for k in a if not k.endswith("k_proj.bias")), reverse=True)[:3]
        print(kind, "second step" if second else "first step", "three vs two terms:", [(f"{v:.2e}", k.split("layers.")[-1]) for v, k in worst])
