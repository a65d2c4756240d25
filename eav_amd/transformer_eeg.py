"""ShallowConvNet + 12-layer single-head transformer and TrainerUni on MI355X: the class API of
Transformer_torch/Transformer_EEG.py over libeav_hip.so.

    ShallowConvNet(nb_classes, chans=30, samples=500, dropout=0.5, num_layers=12)                       (:109-130)
        __call__(x[B,1,30,500]) -> softmax probabilities [B,nb_classes]                                 (:132-148)
    TrainerUni(model, data, lr=1e-3, batch_size=32, epochs=10, subject=0, device=None)                  (:151-181)
        .train() / .validate()                                                                          (:183-219)
    PatchEmbedding / MultiHeadAttention / FeedForwardBlock / TransformerLayer: parameter containers with the
        reference's attribute names, so ``state_dict()`` keys and the default initialisation stream are identical.

Arithmetic (all in hand-written gfx950 kernels, no CPU path):
  * conv(1->40,(1,13)) and the 40 per-filter Linear(30->1) are one fused kernel that never materialises the
    [B,40,30,488] conv output (csrc/shallow_tf.hip);
  * q/k/v/FFN projections: fp32-MFMA GEMMs; attention: the fused flash-style kernels of the AST/ViT encoders with the
    40-wide single head zero-padded to their 64-wide tile (the pad columns of the q/k/v buffers stay exactly zero) -
    split-operand fp16 MFMA at fp32-grade accuracy by default, exact-fp32 MFMA with attention_precision = "fp32";
  * LayerNorm, ReLU+Dropout, Dropout+residual, BatchNorm -> square -> AvgPool(35,7) -> log -> Dropout: HBM-bound kernels.
The reference's trainer cannot be constructed as shipped (`_loader` lacks `self`, :175); this one works and otherwise
keeps its behaviour: CrossEntropyLoss on the softmax output, Adam, fc max-norm 0.5 after every step, the result line
appended to ``eeg_results_new_shallow_.txt`` after the last epoch.
"""
from __future__ import annotations

import math
import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib
from .eegnet import DeviceLoader, GraphStep, cached_workspace
from .optim import CrossEntropyLoss, FusedAdam, flatten_parameters

NF, KC, POOL, STRIDE, HD = 40, 13, 35, 7, 64     # filters / conv taps / pool window / pool stride / attention tile
SLOT = 4128                                      # floats per operand-scale slot (EAV_SP_SLOT, include/eav_hip.h)


class PatchEmbedding(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, qkv_dim: int):
        super().__init__()
        assert embed_dim % num_heads == 0, "embed_dim must be divisible by num_heads"
        self.embed_dim, self.num_heads, self.qkv_dim = embed_dim, num_heads, qkv_dim
        self.value_proj = nn.ModuleList([nn.Linear(30, 1, bias=False) for _ in range(40)])       # :24-26


class MultiHeadAttention(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, qkv_dim: int):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.embed_dim, self.num_heads, self.head_dim = embed_dim, num_heads, embed_dim // num_heads
        self.W_q = nn.Linear(self.head_dim, qkv_dim, bias=False)                                  # :46-48
        self.W_k = nn.Linear(self.head_dim, qkv_dim, bias=False)
        self.W_v = nn.Linear(self.head_dim, qkv_dim, bias=False)


class FeedForwardBlock(nn.Module):
    def __init__(self, embed_dim: int, expansion: int = 4, drop_p: float = 0.5):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(embed_dim, embed_dim * expansion), nn.ReLU(), nn.Dropout(drop_p),
                                 nn.Linear(embed_dim * expansion, embed_dim))                     # :82-87


class TransformerLayer(nn.Module):
    def __init__(self, embed_dim: int, num_heads: int, qkv_dim: int, drop_p: float = 0.5):
        super().__init__()
        self.attn = MultiHeadAttention(embed_dim, num_heads, qkv_dim)
        self.ffn = FeedForwardBlock(embed_dim, drop_p=drop_p)
        self.norm1 = nn.LayerNorm(embed_dim)
        self.norm2 = nn.LayerNorm(embed_dim)
        self.dropout = nn.Dropout(drop_p)


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        ctx.model = model
        ctx.token = model._launch_forward(x)
        return model._ws.probs.clone()

    @staticmethod
    def backward(ctx, dprobs):
        return (None, None, *ctx.model._launch_backward(dprobs.contiguous(), ctx.token))


class ShallowConvNet(nn.Module):
    def __init__(self, nb_classes: int, chans: int = 30, samples: int = 500, dropout: float = 0.5,
                 num_layers: int = 12):
        super().__init__()
        # same sub-modules in the same construction order as the reference (:120-130)
        self.conv = nn.Conv2d(1, NF, (1, KC), bias=False)
        self.pool = nn.AvgPool2d((1, POOL), stride=(1, STRIDE))
        self.dropout = nn.Dropout(dropout)
        self.bn = nn.BatchNorm2d(NF)
        self.embedding = PatchEmbedding(embed_dim=NF, num_heads=1, qkv_dim=NF)
        self.transformer = nn.ModuleList([TransformerLayer(NF, 1, NF, dropout) for _ in range(num_layers)])
        self.fc = nn.Linear(2600, nb_classes, bias=False)
        # the reference hard-codes Linear(30,1) and Linear(2600,nb): only 30 channels and 65 pooled frames fit
        self.chans, self.samples, self.num_layers, self.nb_classes = chans, samples, num_layers, nb_classes
        self.drop_p = float(dropout)
        if not 1 <= nb_classes <= 16:
            raise NotImplementedError("eav_amd.ShallowConvNet: nb_classes <= 16")
        self._ws = None
        self._flat = None
        self._token = 0
        self._saved = None
        self.dropout_seed = 0x5A110EED
        self._dropout_masks = None     # tests: list of uint8 keep masks in the reference's call order
        self._fwd_counter = None
        # "split": softmax(QK^T)V and its backward on the fp16 matrix cores with fp16 hi + lo operand planes
        # (fp32-grade, eav_attn_*_sp); "fp32": the exact-fp32 MFMA kernels
        self.attention_precision = os.environ.get("EAV_SHALLOW_ATTENTION", "split")

    # ------------------------------------------------------------------ plumbing
    def _ensure_flat(self):
        p0 = self.conv.weight
        if self._flat is None or self._flat[0].device != p0.device or getattr(p0, "_eav_flat", None) is None \
                or p0.data_ptr() != self._flat[0].data_ptr():
            # W_q / W_k / W_v [40,40] become rows 0..39 of three consecutive zero-padded [64,40] blocks: one [192,40]
            # operand for a single fused q/k/v GEMM whose output columns match the 64-wide attention tile
            pads = {f"transformer.{l}.attn.W_{n}.weight": (HD - NF) * NF for l in range(self.num_layers) for n in "qkv"}
            self._flat = flatten_parameters(self, pad_after=pads)
            self._names = [n for n, _ in self.named_parameters()]

    def set_dropout_masks(self, masks):
        self._dropout_masks = masks

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _lib.EavError("eav_amd.ShallowConvNet runs on an MI355X only: move the model and the input to the "
                                "ROCm device (there is no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != 30 or (x.shape[3] - KC + 1 - POOL) // STRIDE + 1 != 65:
            raise ValueError(f"expected input [B,1,30,S] with 65 pooled frames (S in 495..501), got {tuple(x.shape)}")
        if self.conv.weight.device != x.device:
            raise _lib.EavError("model and input are on different devices")
        self._ensure_flat()
        return _Fn.apply(x.contiguous().float(), self, *self.parameters())

    # ------------------------------------------------------------------ kernels
    def _alloc(self, B, S, dev):
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)  # noqa: E731
        T, L = S - KC + 1, self.num_layers
        M = B * T
        ws = SimpleNamespace(key=(B, S), B=B, S=S, T=T, M=M)
        ws.u = f(B, NF, S)
        ws.h = [f(M, NF) for _ in range(L + 1)]
        ws.qkv = [f(M, 3 * HD) for _ in range(L)]       # pad columns 40..63 of q, k, v are written as exact zeros
        ws.ao = [z(M, HD) for _ in range(L)]
        ws.lse = [f(B, T) for _ in range(L)]
        ws.a = [f(M, NF) for _ in range(L)]
        ws.x1 = [f(M, NF) for _ in range(L)]
        ws.f1 = [f(M, 4 * NF) for _ in range(L)]
        ws.f2 = [f(M, NF) for _ in range(L)]
        ws.st = [f(4, M) for _ in range(L)]              # mean1, rstd1, mean2, rstd2
        ws.y = f(M, NF)
        ws.bn = f(6 * NF)
        ws.np_cs = _lib.plain("eav_colstats_nparts", M)
        ws.part_cs = f(ws.np_cs, 2 * NF)
        ws.pooled, ws.feat, ws.dfeat = f(B, NF * 65), f(B, NF * 65), f(B, NF * 65)
        ws.probs = f(B, self.nb_classes)
        ws.zero_bias, ws.dbias = z(16), f(16)
        # backward
        ws.g, ws.dh, ws.dy, ws.da = f(M, NF), f(M, NF), f(M, NF), f(M, NF)
        ws.df2, ws.df1 = f(M, NF), f(M, 4 * NF)
        ws.dao, ws.dqkv, ws.delta = z(M, HD), f(M, 3 * HD), f(B, T)
        # split-operand attention (attention_precision == "split"): fp16 hi + lo planes of q | k | v rows and of their
        # transposes per layer (kept for the backward), the same for d(attention out); one operand-scale slot each
        h16 = lambda *s: torch.zeros(*s, dtype=torch.float16, device=dev)  # noqa: E731
        npad = _lib.plain("eav_attn_sp_npad", T)
        ws.qkvrow = [h16(M, 2 * 3 * HD) for _ in range(L)]
        ws.dorow = h16(M, 2 * HD)
        ws.fslots, ws.bslots = z(L, SLOT), z(2 * L, SLOT)
        ws.part_bn = f(B, 2 * NF)
        ws.np_ln = _lib.plain("eav_layernorm_bwd_nparts", M)
        ws.part_ln = f(ws.np_ln, 2 * NF)
        ws.np_col = _lib.plain("eav_colsum_nparts", M)
        ws.part_col = f(ws.np_col, 4 * NF)
        shapes = [(NF, 4 * NF, M), (4 * NF, NF, M), (3 * HD, NF, M)]
        ws.splitk = f(max(_lib.plain("eav_gemm_f32_splitk_plan", m, n, k) * m * n for m, n, k in shapes))
        ws.np_e = _lib.plain("eav_shallow_embed_nparts", B, S)
        ws.part_ec, ws.part_ev, ws.dwv = f(ws.np_e, NF * KC), f(ws.np_e, NF * 30), f(NF, 30)
        return ws

    def _split_attention(self):
        if self.attention_precision not in ("split", "fp32"):
            raise ValueError(f"attention_precision must be 'split' or 'fp32', got {self.attention_precision!r}")
        return self.attention_precision == "split"

    def _seed(self, layer, site):
        return self.dropout_seed + ((3 * layer + site + 1) << 40)

    def _gemm(self, A, Bm, C, M, N, K, lda, ldb, ldc, tB=0, bias=None, acc=0):
        _lib.call("eav_gemm_f32", A, Bm, C, M, N, K, lda, ldb, ldc, 0, tB, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, bias, 0, None,
                  None, 0, acc, self._st)

    def _wgrad(self, A, Bm, C, M, N, K, lda, ldb):
        """C[M,N] = A^T.B for A stored [K,M] (lda), B stored [K,N] (ldb): contraction over the tokens."""
        _lib.call("eav_gemm_f32_splitk", A, Bm, C, _lib.ptr(self._ws.splitk), M, N, K, lda, ldb, 1, 1, self._st)

    def _launch_forward(self, x):
        L, P = _lib.call, _lib.ptr
        st = self._st = _lib.stream_ptr()
        B, S = x.shape[0], x.shape[3]
        # one workspace per batch size, never freed: captured hipGraphs hold its raw pointers, and the zero pad columns
        # of ws.ao / ws.dao must survive from step to step (see EEGNet_tor._workspace)
        wkey = (B, S, str(x.device))
        if not hasattr(self, "_wss"):
            self._wss = {}
        ws = self._ws = cached_workspace(self._wss, wkey, lambda: self._alloc(B, S, x.device))
        T, M = ws.T, ws.M
        n = dict(self.named_parameters())
        w = lambda k: P(n[k])  # noqa: E731
        training = bool(self.training)
        drop = self.drop_p if training else 0.0
        masks = list(self._dropout_masks) if (training and self._dropout_masks is not None) else None
        self._token += 1
        cnt = None
        if drop > 0.0 and masks is None:
            if self._fwd_counter is None or self._fwd_counter.device != x.device:
                self._fwd_counter = torch.zeros((), dtype=torch.int64, device=x.device)
            L("eav_counter_inc", P(self._fwd_counter), st)
            cnt = P(self._fwd_counter)
        mk = (lambda i: P(masks[i])) if masks is not None else (lambda i: None)
        scale = 1.0 / math.sqrt(NF)
        split = self._split_attention()
        if split:
            ws.fslots.zero_()

        L("eav_shallow_embed_fwd", P(x), w("conv.weight"), w("embedding.value_proj.0.weight"), 32, P(ws.u), P(ws.h[0]),
          B, 30, S, NF, KC, st)
        for l in range(self.num_layers):
            p = f"transformer.{l}."
            hin, qkv, stp = P(ws.h[l]), P(ws.qkv[l]), P(ws.st[l])
            self._gemm(hin, w(p + "attn.W_q.weight"), qkv, M, 3 * HD, NF, NF, NF, 3 * HD)            # :62-64, fused
            if split:                                                                              # :66-69
                s_qkv = P(ws.fslots) + 4 * SLOT * l
                L("eav_sp_absmax", qkv, M, 3 * HD, 3 * HD, s_qkv, st)
                L("eav_attn_sp_prep", qkv, s_qkv, P(ws.qkvrow[l]), None, B, T, 3 * HD, HD, 0, st)
                L("eav_attn_fwd_sp", P(ws.qkvrow[l]), None, s_qkv, P(ws.ao[l]), P(ws.lse[l]), None, B, 1, T, HD,
                  scale, st)
            else:
                L("eav_attn_fwd", qkv, P(ws.ao[l]), P(ws.lse[l]), B, 1, T, HD, scale, st)
            L("eav_add_strided", P(ws.ao[l]), HD, qkv + 8 * HD, 3 * HD, P(ws.a[l]), NF, M, NF, st)    # out + res, :76
            L("eav_layernorm_fwd", P(ws.a[l]), w(p + "norm1.weight"), w(p + "norm1.bias"), P(ws.y), stp, stp + 4 * M,
              M, NF, 1e-5, st)
            L("eav_dropout_add", P(ws.y), hin, P(ws.x1[l]), M * NF, drop, self._seed(l, 0), mk(3 * l), cnt, st)  # :103
            self._gemm(P(ws.x1[l]), w(p + "ffn.net.0.weight"), P(ws.f1[l]), M, 4 * NF, NF, NF, NF, 4 * NF,
                       bias=w(p + "ffn.net.0.bias"))
            L("eav_relu_dropout", P(ws.f1[l]), M * 4 * NF, drop, self._seed(l, 1), mk(3 * l + 1), cnt, st)   # :84-85
            self._gemm(P(ws.f1[l]), w(p + "ffn.net.3.weight"), P(ws.f2[l]), M, NF, 4 * NF, 4 * NF, 4 * NF, NF,
                       bias=w(p + "ffn.net.3.bias"))
            L("eav_layernorm_fwd", P(ws.f2[l]), w(p + "norm2.weight"), w(p + "norm2.bias"), P(ws.y), stp + 8 * M,
              stp + 12 * M, M, NF, 1e-5, st)
            L("eav_dropout_add", P(ws.y), P(ws.x1[l]), P(ws.h[l + 1]), M * NF, drop, self._seed(l, 2), mk(3 * l + 2),
              cnt, st)                                                                            # :104
        # head (:135-146): BatchNorm over (batch, time) per feature
        hl, b0 = P(ws.h[self.num_layers]), P(ws.bn)
        L("eav_colstats", hl, P(ws.part_cs), M, NF, NF, st)
        L("eav_bn_finalize", P(ws.part_cs), ws.np_cs, NF, float(M), w("bn.weight"), w("bn.bias"),
          P(self.bn.running_mean), P(self.bn.running_var), int(training), float(self.bn.momentum), float(self.bn.eps),
          b0, b0 + 4 * NF, b0 + 8 * NF, b0 + 12 * NF, st)
        if training:
            self.bn.num_batches_tracked += 1
        hs = self._seed(self.num_layers, 0)
        L("eav_sqpool_log_fwd", hl, b0, P(ws.pooled), P(ws.feat), B, T, NF, 65, POOL, STRIDE, 1e-7, 1e4, drop, hs,
          mk(3 * self.num_layers), cnt, st)
        L("eav_dense_softmax_fwd", P(ws.feat), w("fc.weight"), P(ws.zero_bias), None, P(ws.probs), B, NF * 65,
          self.nb_classes, st)
        self._saved = (self._token, x, training, drop, masks, cnt, ws, split)
        return self._token

    def _launch_backward(self, dprobs, token):
        if self._saved is None or self._saved[0] != token:
            raise _lib.EavError("ShallowConvNet.backward: the activations of this forward were overwritten by a later "
                                "forward (one outstanding forward per backward)")
        L, P = _lib.call, _lib.ptr
        st = self._st = _lib.stream_ptr()
        _, x, training, drop, masks, cnt, ws, split = self._saved
        if split:
            ws.bslots.zero_()
        self._ws = ws
        B, S, T, M = ws.B, ws.S, ws.T, ws.M
        flat, gflat, offs = self._flat
        n = dict(self.named_parameters())
        w = lambda k: P(n[k])  # noqa: E731
        gp = lambda k: P(gflat) + 4 * offs[k][0]  # noqa: E731
        mk = (lambda i: P(masks[i])) if masks is not None else (lambda i: None)
        scale = 1.0 / math.sqrt(NF)
        dh, dy, da, df1, df2, dqkv = P(ws.dh), P(ws.dy), P(ws.da), P(ws.df1), P(ws.df2), P(ws.dqkv)

        def reduce(part, nparts, stride, nn_, out):
            L("eav_reduce_partials", part, nparts, stride, nn_, 1.0, out, st)

        def bias_grad(dyp, N, out):
            L("eav_colsum", dyp, P(ws.part_col), M, N, N, st)
            reduce(P(ws.part_col), ws.np_col, N, N, out)

        def ln_bwd(dyp, xin, gk, bk, mean, rstd, dx):
            L("eav_layernorm_bwd", dyp, xin, w(gk), mean, rstd, dx, 0, P(ws.part_ln), M, NF, st)
            if gp(bk) == gp(gk) + 4 * NF:      # weight and bias gradients are neighbours in the flat buffer: one launch
                reduce(P(ws.part_ln), ws.np_ln, 2 * NF, 2 * NF, gp(gk))
            else:
                reduce(P(ws.part_ln), ws.np_ln, 2 * NF, NF, gp(gk))
                reduce(P(ws.part_ln) + 4 * NF, ws.np_ln, 2 * NF, NF, gp(bk))

        # fc + softmax, then log <- pool <- square <- BatchNorm
        L("eav_dense_softmax_bwd", P(dprobs), P(ws.probs), P(ws.feat), w("fc.weight"), gp("fc.weight"), P(ws.dbias),
          P(ws.dfeat), B, NF * 65, self.nb_classes, st)
        hl, b0 = P(ws.h[self.num_layers]), P(ws.bn)
        L("eav_sqpool_log_bwd", P(ws.dfeat), P(ws.pooled), hl, b0, P(ws.g), P(ws.part_bn), B, T, NF, 65, POOL, STRIDE,
          1e-7, 1e4, drop, self._seed(self.num_layers, 0), mk(3 * self.num_layers), cnt, st)
        L("eav_bn_bwd_finalize", P(ws.part_bn), B, NF, float(M), int(training), gp("bn.weight"), gp("bn.bias"),
          b0 + 16 * NF, b0 + 20 * NF, st)
        L("eav_bn_rows_bwd", P(ws.g), hl, b0, dh, M, NF, st)
        for l in reversed(range(self.num_layers)):
            p = f"transformer.{l}."
            hin, qkv, stp = P(ws.h[l]), P(ws.qkv[l]), P(ws.st[l])
            # h[l+1] = x1 + Dropout(LN2(f2)):  dh stays the gradient w.r.t. x1, the branch goes through LN2
            L("eav_dropout_add", dh, None, dy, M * NF, drop, self._seed(l, 2), mk(3 * l + 2), cnt, st)
            ln_bwd(dy, P(ws.f2[l]), p + "norm2.weight", p + "norm2.bias", stp + 8 * M, stp + 12 * M, df2)
            self._wgrad(df2, P(ws.f1[l]), gp(p + "ffn.net.3.weight"), NF, 4 * NF, M, NF, 4 * NF)
            bias_grad(df2, NF, gp(p + "ffn.net.3.bias"))
            self._gemm(df2, w(p + "ffn.net.3.weight"), df1, M, 4 * NF, NF, NF, 4 * NF, 4 * NF, tB=1)
            L("eav_relu_dropout_bwd", df1, P(ws.f1[l]), M * 4 * NF, drop, st)
            self._wgrad(df1, P(ws.x1[l]), gp(p + "ffn.net.0.weight"), 4 * NF, NF, M, 4 * NF, NF)
            bias_grad(df1, 4 * NF, gp(p + "ffn.net.0.bias"))
            self._gemm(df1, w(p + "ffn.net.0.weight"), dh, M, NF, 4 * NF, 4 * NF, NF, NF, tB=1, acc=1)
            # x1 = h[l] + Dropout(LN1(a)),  a = attention + V
            L("eav_dropout_add", dh, None, dy, M * NF, drop, self._seed(l, 0), mk(3 * l), cnt, st)
            ln_bwd(dy, P(ws.a[l]), p + "norm1.weight", p + "norm1.bias", stp, stp + 4 * M, da)
            L("eav_add_strided", da, NF, None, 0, P(ws.dao), HD, M, NF, st)
            if split:
                s_qkv, s_do, s_ds = P(ws.fslots) + 4 * SLOT * l, P(ws.bslots) + 8 * SLOT * l, P(ws.bslots) + 8 * SLOT * l \
                    + 4 * SLOT
                L("eav_sp_absmax", P(ws.dao), M, HD, HD, s_do, st)
                L("eav_attn_sp_prep", P(ws.dao), s_do, P(ws.dorow), None, B, T, HD, HD, 0, st)
                L("eav_attn_bwd_sp", P(ws.qkvrow[l]), None, P(ws.dorow), None, s_qkv, s_do, s_ds,
                  P(ws.ao[l]), P(ws.dao), P(ws.lse[l]), P(ws.delta), dqkv, None, B, 1, T, HD, scale, st)
            else:
                L("eav_attn_bwd", qkv, P(ws.ao[l]), P(ws.dao), P(ws.lse[l]), P(ws.delta), dqkv, B, 1, T, HD, scale, st)
            L("eav_add_strided", dqkv + 8 * HD, 3 * HD, da, NF, dqkv + 8 * HD, 3 * HD, M, NF, st)   # the "+ V" branch
            k = p + "attn.W_q.weight"          # the padded [192,40] block that starts at W_q (see _ensure_flat)
            self._wgrad(dqkv, hin, gp(k), 3 * HD, NF, M, 3 * HD, NF)
            self._gemm(dqkv, w(k), dh, M, NF, 3 * HD, 3 * HD, NF, NF, tB=1, acc=1)
        # conv taps and channel projections (the input needs no gradient)
        L("eav_shallow_embed_bwd", dh, P(x), P(ws.u), w("conv.weight"), P(ws.part_ec), P(ws.part_ev), B, 30, S, NF, KC,
          st)
        reduce(P(ws.part_ec), ws.np_e, NF * KC, NF * KC, gp("conv.weight"))
        reduce(P(ws.part_ev), ws.np_e, NF * 30, NF * 30, P(ws.dwv))
        # the 40 Linear(30,1) weights sit 32 floats apart in the flat buffer (16-byte alignment of every tensor)
        assert offs["embedding.value_proj.1.weight"][0] - offs["embedding.value_proj.0.weight"][0] == 32
        L("eav_add_strided", P(ws.dwv), 30, None, 0, gp("embedding.value_proj.0.weight"), 32, NF, 30, st)
        return [gflat[offs[k][0]:offs[k][0] + offs[k][1]].view(n[k].shape) if n[k].requires_grad else None
                for k in self._names]


class TrainerUni:
    def __init__(self, model, data, lr=1e-3, batch_size=32, epochs=10, subject=0, device=None):
        self.device = torch.device(device or ("cuda" if torch.cuda.is_available() else "cpu"))
        if self.device.type != "cuda":
            raise _lib.EavError("eav_amd.TrainerUni needs an MI355X (torch device 'cuda' on ROCm); no CPU fallback")
        tr_x, tr_y, te_x, te_y = data
        self.batch_size = batch_size
        self.train_loader = self._loader(tr_x, tr_y, batch_size, True)
        self.test_loader = self._loader(te_x, te_y, batch_size, False)
        self.model = model.to(self.device)
        self.criterion = CrossEntropyLoss()                                                      # :172
        self.optimizer = FusedAdam(self.model.parameters(), lr=lr, capturable=True)              # :173
        self.epochs = epochs
        self.subject = subject
        self.grad_sync = None      # set by eav_amd.dist.attach(trainer) under torchrun
        self.use_graph = True
        self._graph = None

    def _loader(self, x, y, batch_size, shuffle):
        return DeviceLoader(x, y, batch_size, shuffle, self.device)

    def _max_norm(self):
        """model.fc.weight <- renorm(p=2, dim=0, maxnorm=0.5) after every optimiser step (:195-199)."""
        w = self.model.fc.weight
        _lib.call("eav_renorm_rows", _lib.ptr(w), w.shape[0], w.shape[1], 0.5, _lib.stream_ptr())

    def train(self):
        dl = self.train_loader
        for epoch in range(self.epochs):
            self.model.train()
            for idx in dl.index_batches():
                if self.use_graph and len(idx) == self.batch_size:
                    if self._graph is None:
                        self._graph = GraphStep(self.model, self.optimizer, self.criterion, dl.x, dl.y, len(idx),
                                                self.grad_sync, post_step=self._max_norm)
                    self._graph.run(idx)
                    continue
                x, y = dl.gather(idx)
                out = self.model(x)
                loss = self.criterion(out, y)
                self.optimizer.zero_grad()
                loss.backward()
                if self.grad_sync is not None:
                    self.grad_sync()
                self.optimizer.step()
                self._max_norm()
            self.criterion.check()        # labels outside [0, classes) seen by any step of this epoch raise here
            acc = self.validate()
            if epoch == self.epochs - 1:
                with open("eeg_results_new_shallow_.txt", "a") as f:                             # :203-205
                    f.write(f"Subject {self.subject} | Accuracy: {acc:.4f}\n")

    def validate(self):
        self.model.eval()
        correct, total = 0, 0
        with torch.no_grad():
            for x, y in self.test_loader:
                preds = self.model(x).argmax(dim=1)
                correct += (preds == y).sum().item()
                total += y.size(0)
        acc = correct / total
        print(f"Validation Accuracy: {acc:.4f}")
        return acc
