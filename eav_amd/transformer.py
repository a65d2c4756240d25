"""AST / ViT classification encoders on MI355X: forward, backward and weight hand-off.

The reference never defines these models - it instantiates Hugging Face classes
(`AutoModelForAudioClassification.from_pretrained`, Transformer_Audio.py:22;
`AutoModelForImageClassification.from_pretrained`, Transformer_Vision.py:29) and trains them
with `loss.backward()`.  This module restates that arithmetic as an explicit schedule of
libeav_hip.so kernels, keeps the HF 5.x parameter names so `state_dict()` / safetensors
checkpoints interchange (HF 4.x names are accepted on load), and exposes the pieces the
reference trainers touch: `model(x).logits`, `model.classifier`, `model.parameters()`,
`train()/eval()`, `.to(device)`.

Two arithmetic paths behind `Encoder.precision` (env EAV_ENCODER_PRECISION), same parity bounds:
  "split" (default)  every dense projection and the fused attention on the fp16 matrix cores with
                     fp16 hi + lo operand planes, three MFMAs per product, fp32 accumulation -
                     fp32-grade (gemm_sp.hip, attention_sp.hip).  Weight-gradient
                     GEMMs, the final bias / LayerNorm gradient reductions and the refresh of the
                     weight planes run on a side HIP stream.
  "fp32"             the same schedule on the exact-fp32 MFMA (gemm_f32.hip, attention.hip).

Everything is kept resident (288 GB HBM): what a backward needs is saved, nothing is recomputed
(split: the fp16 planes of the layer inputs of every GEMM and of q | k | v, the pre-activations of
the MLP and the attention outputs / log-sum-exps; the fused attention never materialises the
[B*H, N, N] probabilities - only the reduced test configurations with head_dim != 64 do).
"""
from __future__ import annotations

import contextlib
import json
import os
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .optim import flatten_parameters


DEFAULT_PRECISION = "split"

# ----------------------------------------------------------------------------- configuration
def make_config(kind, hidden=768, layers=12, heads=12, ff=3072, eps=1e-12, num_labels=5, patch=16, mel=128,
                frames=1024, fstride=10, tstride=10, image=224, channels=3):
    if kind == "ast":
        ny, nx = (mel - patch) // fstride + 1, (frames - patch) // tstride + 1
        geo = dict(C=1, H=mel, W=frames, sy=fstride, sx=tstride, transposed=1)
        nextra, prefix = 2, "audio_spectrogram_transformer"
    elif kind == "vit":
        ny = nx = image // patch
        geo = dict(C=channels, H=image, W=image, sy=patch, sx=patch, transposed=0)
        nextra, prefix = 1, "vit"
    else:
        raise ValueError(kind)
    return SimpleNamespace(kind=kind, hidden=hidden, layers=layers, heads=heads, ff=ff, eps=eps,
                           num_labels=num_labels, patch=patch, ny=ny, nx=nx, npatch=ny * nx, nextra=nextra,
                           ntok=ny * nx + nextra, prefix=prefix, kp=geo["C"] * patch * patch, **geo)


def config_from_hf(cfg_json: dict):
    mt = cfg_json.get("model_type", "")
    common = dict(hidden=cfg_json.get("hidden_size", 768), layers=cfg_json.get("num_hidden_layers", 12),
                  heads=cfg_json.get("num_attention_heads", 12), ff=cfg_json.get("intermediate_size", 3072),
                  eps=cfg_json.get("layer_norm_eps", 1e-12), patch=cfg_json.get("patch_size", 16),
                  num_labels=len(cfg_json["id2label"]) if "id2label" in cfg_json else cfg_json.get("num_labels", 2))
    if cfg_json.get("hidden_act", "gelu") != "gelu":
        raise NotImplementedError("only the exact erf GELU is implemented")
    if mt == "audio-spectrogram-transformer":
        return make_config("ast", mel=cfg_json.get("num_mel_bins", 128), frames=cfg_json.get("max_length", 1024),
                           fstride=cfg_json.get("frequency_stride", 10), tstride=cfg_json.get("time_stride", 10),
                           **common)
    if mt == "vit":
        return make_config("vit", image=cfg_json.get("image_size", 224), channels=cfg_json.get("num_channels", 3),
                           **common)
    raise NotImplementedError(f"model_type {mt!r}")


def param_shapes(cfg):
    """Ordered {HF 5.x key: shape}.  Order = flat-buffer order: q,k,v weights (and biases) adjacent so the
    three projections run as one [3D, D] GEMM."""
    d, ff, p = cfg.hidden, cfg.ff, cfg.prefix
    s = {f"{p}.embeddings.cls_token": (1, 1, d)}
    if cfg.kind == "ast":
        s[f"{p}.embeddings.distillation_token"] = (1, 1, d)
    s[f"{p}.embeddings.position_embeddings"] = (1, cfg.ntok, d)
    s[f"{p}.embeddings.patch_embeddings.projection.weight"] = (d, cfg.C, cfg.patch, cfg.patch)
    s[f"{p}.embeddings.patch_embeddings.projection.bias"] = (d,)
    for i in range(cfg.layers):
        L = f"{p}.layers.{i}"
        for n in ("q", "k", "v"):
            s[f"{L}.attention.{n}_proj.weight"] = (d, d)
        for n in ("q", "k", "v"):
            s[f"{L}.attention.{n}_proj.bias"] = (d,)
        s[f"{L}.attention.o_proj.weight"] = (d, d)
        s[f"{L}.attention.o_proj.bias"] = (d,)
        for n in ("layernorm_before", "layernorm_after"):
            s[f"{L}.{n}.weight"] = (d,)
            s[f"{L}.{n}.bias"] = (d,)
        s[f"{L}.mlp.fc1.weight"] = (ff, d)
        s[f"{L}.mlp.fc1.bias"] = (ff,)
        s[f"{L}.mlp.fc2.weight"] = (d, ff)
        s[f"{L}.mlp.fc2.bias"] = (d,)
    s[f"{p}.layernorm.weight"] = (d,)
    s[f"{p}.layernorm.bias"] = (d,)
    if cfg.kind == "ast":
        s["classifier.layernorm.weight"] = (d,)
        s["classifier.layernorm.bias"] = (d,)
        s["classifier.dense.weight"] = (cfg.num_labels, d)
        s["classifier.dense.bias"] = (cfg.num_labels,)
    else:
        s["classifier.weight"] = (cfg.num_labels, d)
        s["classifier.bias"] = (cfg.num_labels,)
    return s


_HF4 = [  # (HF 4.x fragment, HF 5.x fragment)
    (".encoder.layer.", ".layers."), (".attention.attention.query.", ".attention.q_proj."),
    (".attention.attention.key.", ".attention.k_proj."), (".attention.attention.value.", ".attention.v_proj."),
    (".attention.output.dense.", ".attention.o_proj."), (".intermediate.dense.", ".mlp.fc1."),
    (".output.dense.", ".mlp.fc2."),
]


def normalise_key(k):
    for a, b in _HF4:
        k = k.replace(a, b)
    return k


class _Node(nn.Module):
    """Anonymous container so that dotted HF names become real sub-modules (model.classifier.dense ...)."""


def _set_param(root, dotted, value):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            m.add_module(p, _Node())
        m = getattr(m, p)
    m.register_parameter(parts[-1], nn.Parameter(value))


class _Out:
    def __init__(self, logits, loss=None):
        self.logits, self.loss = logits, loss


class _EncFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        ctx.model = model
        ctx.token = model._launch_forward(x)
        return model._ws.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        return (None, None, *ctx.model._launch_backward(dlogits.contiguous(), ctx.token))


class _HeadFn(torch.autograd.Function):
    """The classification head alone on cached backbone features (Encoder.head): the same kernels, in the same order and
    with the same arguments, as the tail of _launch_forward / the start of _launch_backward."""

    @staticmethod
    def forward(ctx, feat, model, *params):
        ctx.model, ctx.B = model, feat.shape[0]
        hw = model._head_ws(feat.shape[0], feat.device)
        hw.feat.copy_(feat)
        model._main = torch.cuda.current_stream()
        model._st = model._main.cuda_stream
        model._head_forward(hw.feat, hw, feat.shape[0])
        hw.token = model._token = model._token + 1
        ctx.token = hw.token
        return hw.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        model = ctx.model
        hw = model._head_ws(ctx.B, dlogits.device)
        if hw.token != ctx.token:
            raise _lib.EavError("Encoder.head backward: activations were overwritten by a later head forward")
        model._main = torch.cuda.current_stream()
        model._st = model._main.cuda_stream
        model._head_backward(dlogits.contiguous(), hw.feat, hw, ctx.B, False)
        offs, gflat = model._flat[2], model._flat[1]
        out = []
        for k in model._names:
            p = model._pmap[k]
            trained = p.requires_grad and k.startswith("classifier.")
            out.append(gflat[offs[k][0]:offs[k][0] + offs[k][1]].view(p.shape) if trained else None)
        return (None, None, *out)


class Encoder(nn.Module):
    """ASTForAudioClassification / ViTForImageClassification (5-class head) on the HIP kernels."""

    def __init__(self, cfg, weights=None):
        super().__init__()
        self.cfg = cfg
        shapes = param_shapes(cfg)
        for k, shp in shapes.items():
            if weights is not None and k in weights:
                v = torch.as_tensor(np.asarray(weights[k]), dtype=torch.float32).reshape(shp).clone()
            elif k.endswith("layernorm.weight") or k.endswith("layernorm_before.weight") or k.endswith("layernorm_after.weight"):
                v = torch.ones(shp)
            elif k.endswith(".bias") or "token" in k or "position_embeddings" in k:
                v = torch.zeros(shp)
            else:
                v = torch.randn(shp) * 0.02          # HF initializer_range
            _set_param(self, k, v)
        self._names = list(shapes)
        self._ws = None
        self._ws_cache = {}           # batch size -> workspace (a ragged last batch must not evict the full-batch one)
        self._hws = {}                # batch size -> head-only workspace (Encoder.head)
        self._flat = None
        self._token = 0
        self._saved = None
        self.kernel_events = None
        # GEMM / attention operand precision: "split" (default: fp32-grade on the fp16 matrix cores - every operand as
        # fp16 hi + lo planes, three MFMAs per product, csrc/gemm_sp.hip + attention_sp.hip; measured against float64
        # it is not worse than the exact-fp32 kernels and it passes the same parity bounds), "fp32" (exact-fp32 MFMA),
        # "bf16" (bf16 MFMA operands everywhere, fp32 accumulate: ~5e-3 logit drift) or "bf16_bwd" (fp32 forward -
        # logits unchanged - and bf16 operands for the backward products only).  EAV_ENCODER_PRECISION overrides.
        self.precision = os.environ.get("EAV_ENCODER_PRECISION", DEFAULT_PRECISION)
        # split mode: weight-gradient GEMMs on a side stream (see _wgrad_sp).  "auto" (default): two streams from 8192 token
        # rows per step on, one stream below - at the per-rank batches of a data-parallel group (ViT B = 16: 3152 rows) the
        # ~100 events / waits of the two-stream schedule cost more host time than the overlap returns (20.1 -> 13.1 ms;
        # AST B = 4 21.1 -> 16.8 ms; ViT B = 128 49.2 two streams / 51.3 one: tools/encoder_graph_step.py).  True / False pin it.
        self.overlap_wgrad = "auto"
        # split mode, backward GEMMs only (data and weight gradients): 3 = the fp32-grade three-term product (default),
        # 1 = the hi.hi term alone - operands rounded to fp16 under the planes' scales (11-bit mantissas; fp32
        # accumulation), i.e. classic fp16 mixed-precision gradients: 3e-4 relative gradient error instead of 3e-7, a
        # third of the backward's matrix work.  The forward - the logits - always runs on three terms.
        self.grad_terms = int(os.environ.get("EAV_GRAD_TERMS", "3"))
        # per-kind overrides of grad_terms (None: follow it): the weight-gradient products (their rounding stays in that
        # tensor's update) and the data-gradient products (their rounding travels down the layers) priced separately -
        # tools/encoder_trajectory.py, profiles/r05_term_budget.txt, r06_term_budget.txt.  wgrad_terms = 2 (round 6, opt-in):
        # hi_grad.hi_act + lo_grad.hi_act - the ACTIVATION operand rounded to fp16 (11-bit mantissa, random signs over >= 1576
        # tokens), the gradient operand and the accumulation at full split precision, the producers' a-priori gradient planes
        # still in use.  Weight-gradient error 2-4e-4 of the tensor's maximum (three terms: 1e-7); -20 % on the fc1 / fc2
        # weight gradients, -3.2 % (ViT B = 128) / -2.6 % (AST B = 8) on the step.  40 AdamW steps move the held-out logits by
        # 3.9e-5 (ViT) / 2.3e-5 (AST) at the reference's learning rate 5e-6, but by 5.8e-4 / 1.5e-4 at 5e-5 (three terms:
        # 3.6e-5 / 4.3e-5) - AdamW's early updates are ~lr x sign(g), so gradient rounding that flips near-zero elements is
        # amplified; outside the 3e-4 this repository holds trajectories to, hence not the default.
        self.wgrad_terms = int(os.environ["EAV_WGRAD_TERMS"]) if os.environ.get("EAV_WGRAD_TERMS") else None
        self.dgrad_terms = int(os.environ["EAV_DGRAD_TERMS"]) if os.environ.get("EAV_DGRAD_TERMS") else None
        # the same switch for the forward products (comparison only: 16-bit matrix operands everywhere - the logits then
        # move by a few 1e-3, outside north_star's bound; bench.py reports the leg beside the literal bf16 one)
        self.fwd_terms = int(os.environ.get("EAV_FWD_TERMS", "3"))
        # split mode: LayerNorm / fc1 write their consumers' operand planes (a-priori scales); EAV_FUSED_PLANES=0 for A/B runs
        self.fused_planes = os.environ.get("EAV_FUSED_PLANES", "1") != "0"
        # split mode, backward: fc2's data-gradient GEMM writes the planes of dact (and the partials of fc1's bias gradient)
        # itself, scaled by a bound of |dact| known before the launch - no fp32 dact, no conversion pass; EAV_FUSED_DACT=0
        self.fused_dact = os.environ.get("EAV_FUSED_DACT", "1") != "0"
        # split mode, forward: the fused attention writes its output as the o-proj planes itself (EAV_FUSED_AO=0 for A/B runs)
        self.fused_ao = os.environ.get("EAV_FUSED_AO", "1") != "0"
        # split mode, forward: the fused q/k/v projection writes the attention kernels' row planes itself (scale from a bound of
        # |qkv|), the per-head transposes are made from those planes - no fp32 qkv tensor (EAV_FUSED_QKV=0 for A/B runs)
        self.fused_qkv = os.environ.get("EAV_FUSED_QKV", "1") != "0"
        # split mode, backward: the attention backward writes dqkv as the planes of the q/k/v projection's gradient products
        # itself (scale from a rigorous bound of |dqkv|, eav_attn_dqkv_bound) and leaves the bias-gradient partials - no fp32
        # dqkv, no conversion pass (EAV_FUSED_DQKV=0 for A/B runs)
        self.fused_dqkv = os.environ.get("EAV_FUSED_DQKV", "1") != "0"
        # split mode, backward: the LayerNorm backward kernels write the residual-stream gradient they produce as operand
        # planes as well (scale from a rigorous bound, eav_layernorm_bwd_bound) and leave the bias-gradient partials of the
        # linear layer that consumes it - two conversion passes per layer disappear (EAV_FUSED_DH=0 for A/B runs)
        self.fused_dh = os.environ.get("EAV_FUSED_DH", "1") != "0"
        self._side, self._aux, self._wgrad_done, self._wready, self._wnorm_ready = None, None, {}, {}, None
        self._part_busy, self._ring_pos = {}, {}
        self._wplanes = None          # split mode: {weight key: (planes, planes of the transpose, slot index)}
        self._wplanes_key = None
        self._phase = "fwd"
        self._scales_all = False
        # multi-GPU: called as hook(lo, hi) from inside the backward whenever flat_grad[lo:hi] is final
        self.grad_ready_hook = None
        if cfg.hidden % cfg.heads or (cfg.hidden // cfg.heads) % 4 or cfg.hidden % 4 or cfg.hidden > 1024:
            raise NotImplementedError("hidden size must be <= 1024, a multiple of 4, head_dim a multiple of 4")
        if cfg.ntok > 2048 or cfg.num_labels > 16:
            raise NotImplementedError("at most 2048 tokens and 16 classes")

    # ------------------------------------------------------------------ loading
    @classmethod
    def from_pretrained(cls, model_path):
        """HF directory: config.json + model.safetensors (pytorch_model.bin accepted)."""
        cfg = config_from_hf(json.load(open(os.path.join(model_path, "config.json"))))
        st = os.path.join(model_path, "model.safetensors")
        if os.path.exists(st):
            from safetensors.numpy import load_file
            raw = load_file(st)
        elif os.path.exists(os.path.join(model_path, "pytorch_model.bin")):
            raw = {k: v.numpy() for k, v in torch.load(os.path.join(model_path, "pytorch_model.bin"), map_location="cpu").items()}
        else:
            raise OSError(f"no model.safetensors / pytorch_model.bin under {model_path}")
        w = {normalise_key(k): v for k, v in raw.items()}
        shapes = param_shapes(cfg)
        missing = [k for k in shapes if k not in w]
        if missing:
            raise KeyError(f"checkpoint lacks {missing[:4]} ...")
        return cls(cfg, w)

    def reset_head(self, weight, bias):
        """Replace the classification Linear (the reference does `classifier.dense = nn.Linear(768, n)`,
        Transformer_Audio.py:24 / `classifier = nn.Linear(...)`, Transformer_Vision.py:30)."""
        head = self.classifier.dense if self.cfg.kind == "ast" else self.classifier
        dev = head.weight.device
        head.weight = nn.Parameter(torch.as_tensor(weight, dtype=torch.float32).clone().to(dev))
        head.bias = nn.Parameter(torch.as_tensor(bias, dtype=torch.float32).clone().to(dev))
        self.cfg.num_labels = head.weight.shape[0]
        self._flat = None
        self._ws = None
        self._ws_cache, self._hws = {}, {}

    def head_parameters(self):
        return list(self.classifier.parameters())

    # Anything that rewrites parameters wholesale drops the cached fp16 weight planes of the split path (the version key
    # in _refresh_weight_planes would catch these too; this makes it independent of how torch implements them).
    def invalidate_weight_planes(self):
        """Call after writing parameters in a way torch cannot see (`p.data.mul_(...)`, raw-pointer kernels): `.data`
        writes move neither the parameter's nor the flat buffer's version counter."""
        self._wplanes_key = None

    def load_state_dict(self, *args, **kwargs):
        self._wplanes_key = None
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        self._wplanes_key = None
        self._wplanes = None
        self._ws = None
        self._ws_cache, self._hws = {}, {}
        return super()._apply(fn, *args, **kwargs)

    def head_grad_ranges(self):
        """[(lo, hi)] element ranges of the classifier's gradients in the flat gradient buffer."""
        self._ensure_flat()
        offs = self._flat[2]
        return [(offs[k][0], offs[k][0] + offs[k][1]) for k in self._names if k.startswith("classifier.")]

    # ------------------------------------------------------------------ plumbing
    def _ensure_flat(self):
        p0 = next(self.parameters())
        ok = (self._flat is not None and getattr(p0, "_eav_flat", None) is not None
              and p0._eav_flat[0] is self._flat[0] and p0.data_ptr() == self._flat[0].data_ptr()
              and all(getattr(p, "_eav_flat", (None,))[0] is self._flat[0] for p in self.parameters()))
        if not ok:
            self._flat = flatten_parameters(self, order=self._names)
            self._pmap = dict(self.named_parameters())

    def forward(self, x=None, labels=None, pixel_values=None, input_values=None):
        x = x if x is not None else (pixel_values if pixel_values is not None else input_values)
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _lib.EavError("eav_amd encoders run on an MI355X only (no CPU fallback): move model and input to "
                                "the ROCm device")
        c = self.cfg
        want = (c.W, c.H) if c.kind == "ast" else (c.C, c.H, c.W)
        if tuple(x.shape[1:]) != want:
            raise ValueError(f"expected input [B,{','.join(map(str, want))}], got {tuple(x.shape)}")
        x = x.contiguous().float()
        self._ensure_flat()
        self._want_full = torch.is_grad_enabled() and any(
            p.requires_grad for k, p in self._pmap.items() if not k.startswith("classifier."))
        logits = _EncFn.apply(x, self, *[self._pmap[k] for k in self._names])
        loss = None
        if labels is not None:
            from .optim import CrossEntropyLoss
            loss = CrossEntropyLoss()(logits, labels)
        return _Out(logits, loss)

    # ------------------------------------------------------------------ kernel schedule
    def _call(self, name, *args):
        ev = self.kernel_events
        if ev is not None and name in ev:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.call(name, *args)
            b.record()
            ev[name].append((a, b))
        else:
            _lib.call(name, *args)

    def _gemm(self, A, B, C, M, N, K, lda, ldb, ldc, tA=0, tB=0, batch=1, heads=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
              alpha=1.0, bias=None, gelu=0, pre=None, resid=None, ldr=0, acc=0):
        self._call(self._gemm_name(), A, B, C, M, N, K, lda, ldb, ldc, tA, tB, batch, heads, sA[0], sA[1], sB[0],
                   sB[1], sC[0], sC[1], float(alpha), bias, gelu, pre, resid, ldr, acc, self._st)

    def _fused_attention(self):
        """Flash-style fused attention kernels exist for head_dim 64 (AST, ViT-B); other head sizes take the
        materialised-score path (GEMM + softmax kernels).  `use_fused_attention = False` forces the latter."""
        return getattr(self, "use_fused_attention", True) and self.cfg.hidden // self.cfg.heads == 64

    def _gemm_name(self):
        p = self.precision
        if p not in ("fp32", "split", "bf16", "bf16_bwd"):
            raise ValueError(f"unknown precision {p!r}")
        low = p == "bf16" or (p == "bf16_bwd" and self._phase == "bwd")
        return "eav_gemm_bf16" if low else "eav_gemm_f32"

    def _alloc(self, B, dev, full_backward):
        c = self.cfg
        D, FF, N, H, Lr = c.hidden, c.ff, c.ntok, c.heads, c.layers
        M = B * N
        ldn = (N + 3) // 4 * 4
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        sp = self.precision == "split"
        ws = SimpleNamespace(B=B, M=M, ldn=ldn, full=full_backward, sp=sp)
        nsave = Lr if full_backward else 1
        ws.col = f(B * c.npatch, c.kp)
        ws.hs = [f(M, D) for _ in range(Lr + 1)] if full_backward else [f(M, D), f(M, D)]
        ws.y1 = [f(M, D) for _ in range(1 if sp else nsave)]     # split mode keeps planes, not fp32 copies
        ws.fused = self._fused_attention()
        # split mode + fused attention: the backward reads the fp16 planes of qkv, the fp32 tensor is transient
        ws.qkv = [f(M, 3 * D) for _ in range(1 if (sp and ws.fused) else nsave)]
        if sp:
            self._alloc_split(ws, dev, nsave)
        if ws.fused:      # flash-style kernels: only the log-sum-exp per (image, head, query) is kept
            ws.lse = [f(B * H, N) for _ in range(nsave)]
            ws.delta = f(B * H, N)
        else:
            ws.P = [torch.zeros(B * H, N, ldn, dtype=torch.float32, device=dev) for _ in range(nsave)]
        ws.ao = [f(M, D) for _ in range(nsave)]
        ws.hmid = [f(M, D) for _ in range(nsave)]
        ws.y2 = [f(M, D) for _ in range(1 if sp else nsave)]
        ws.pre = [f(M, FF) for _ in range(nsave)]
        ws.act = [f(M, FF) for _ in range(1 if sp else nsave)]
        ws.st = [f(4, M) for _ in range(nsave)]           # mean1, rstd1, mean2, rstd2
        R = B * c.nextra
        ws.rows, ws.seqr, ws.stf = f(R, D), f(R, D), f(2, R)
        ws.pooled, ws.hl, ws.sth = f(B, D), f(B, D), f(2, B)
        ws.logits = f(B, c.num_labels)
        if full_backward:
            ws.dh, ws.dy, ws.dao = f(M, D), f(M, D), f(M, D)
            ws.dact, ws.dqkv = f(M, FF), f(M, 3 * D)
            if not ws.fused:
                ws.dP = torch.zeros(B * H, N, ldn, dtype=torch.float32, device=dev)
            ws.demb = f(B * c.npatch, D)
            ws.np_ln = _lib.plain("eav_layernorm_bwd_nparts", M)
            ws.part_ln = f(ws.np_ln, 2 * D)
            ws.part_ln_pool = [ws.part_ln, f(ws.np_ln, 2 * D)]     # split path: see _part_buf
            ws.part_ln3_pool = [f(ws.np_ln, 3 * D) for _ in range(4)]   # ... with the bias-gradient section (fused_dh)
            ws.np_cs = _lib.plain("eav_colsum_nparts", M)
            ws.part_cs = f(ws.np_cs, max(FF, 3 * D))
            shapes = [(D, FF, M), (FF, D, M), (3 * D, D, M), (D, D, M), (D, c.kp, B * c.npatch)]
            plan = "eav_gemm_sp_splitk_plan" if sp else "eav_gemm_f32_splitk_plan"
            ws.splitk = f(max(_lib.plain(plan, m, n, k) * m * n for m, n, k in shapes))
        ws.drows, ws.dseqr = f(R, D), f(R, D)
        ws.dpooled, ws.dhl = f(B, D), f(B, D)
        ws.np_lnr = _lib.plain("eav_layernorm_bwd_nparts", R)
        ws.part_lnr = f(ws.np_lnr, 2 * D)
        return ws

    # ------------------------------------------------------------------ split-operand (fp16 hi/lo planes) plumbing
    SLOT = 4128   # floats per scale slot (include/eav_hip.h EAV_SP_SLOT)
    FS, BS = 5, 8   # slots per layer: forward y1, qkv, ao, y2, act; backward dh(fc2), dact, dh(o), dao, dS, dqkv, dy(fc1), dy(qkv)

    def _alloc_split(self, ws, dev, nsave):
        c = self.cfg
        D, FF, Lr, M = c.hidden, c.ff, c.layers, ws.M
        MP = ws.B * c.npatch
        kp = lambda k: _lib.plain("eav_sp_kpad", k)  # noqa: E731
        # Row planes only: the weight-gradient products contract over the ROWS (tokens) of the same planes the forward /
        # data-gradient products read (transposing LDS reads in gemm_sp.hip), so no transposed copy of any activation or
        # gradient exists.  Rows are padded to a multiple of 32 with zeros (contracted like real tokens; the conversion
        # never writes them).
        h = lambda r, k: torch.zeros((r + 31) // 32 * 32, 2 * kp(k), dtype=torch.float16, device=dev)  # noqa: E731
        full = ws.full
        ws.colp = h(MP, c.kp)
        ws.y1p = [h(M, D) for _ in range(nsave)]
        ws.aop = [h(M, D) for _ in range(nsave)]
        ws.y2p = [h(M, D) for _ in range(nsave)]
        ws.actp = [h(M, FF) for _ in range(nsave)]
        ws.fslots = torch.zeros(1 + self.FS * Lr, self.SLOT, dtype=torch.float32, device=dev)
        H, N = c.heads, c.ntok
        Npad = _lib.plain("eav_attn_sp_npad", N)
        if ws.fused:   # attention operands: row planes of qkv and per-head transposed planes (csrc/attention_sp.hip)
            ws.qkvrow = [torch.empty(M, 6 * D, dtype=torch.float16, device=dev) for _ in range(nsave)]
        if full:
            # backward operands: one set, reused by every layer.  dh is converted twice per layer (fc2 and o_proj stage)
            # and its weight gradients run on the side stream, so the two uses must not share one buffer
            ws.dhp, ws.dhp2 = h(M, D), h(M, D)
            ws.dactp, ws.dqkvp = h(M, FF), h(M, 3 * D)
            ws.dembp = h(MP, D)
            ws.np_cs2 = _lib.plain("eav_sp_convert_colsum_nparts", M)
            ws.part_cs2 = torch.empty(ws.np_cs2, max(FF, 3 * D), dtype=torch.float32, device=dev)
            ws.part_cs2_pool = [ws.part_cs2] + [torch.empty_like(ws.part_cs2) for _ in range(3)]
            ws.np_attn = ws.B * ((c.ntok + 31) // 32)          # bias-gradient partials of the attention backward: one row per 32-token tile
            ws.part_attn_pool = [torch.zeros(ws.np_attn, 3 * D, dtype=torch.float32, device=dev) for _ in range(4)]
            ws.bslots = torch.zeros(2 + self.BS * Lr, self.SLOT, dtype=torch.float32, device=dev)
            if ws.fused:
                ws.dorow = torch.empty(M, 2 * D, dtype=torch.float16, device=dev)

    def _weight_keys(self):
        """[(cache key, parameter name of the [out, in] matrix, out, in)] of every GEMM weight."""
        c = self.cfg
        keys = [("patch", f"{c.prefix}.embeddings.patch_embeddings.projection.weight", c.hidden, c.kp)]
        for i in range(c.layers):
            L = f"{c.prefix}.layers.{i}"
            keys += [(f"qkv{i}", f"{L}.attention.q_proj.weight", 3 * c.hidden, c.hidden),
                     (f"o{i}", f"{L}.attention.o_proj.weight", c.hidden, c.hidden),
                     (f"fc1{i}", f"{L}.mlp.fc1.weight", c.ff, c.hidden),
                     (f"fc2{i}", f"{L}.mlp.fc2.weight", c.hidden, c.ff)]
        return keys

    def _refresh_weight_planes(self, dev, need_T):
        """(Re)build the fp16 hi/lo planes of the GEMM weights (and of their transposes, for the data-gradient
        products) that changed: FusedAdam records the byte ranges it updated (`_eav_dirty`), any torch in-place write
        to the flat buffer (load_state_dict, .copy_) bumps its version and invalidates everything."""
        flat = self._flat[0]
        # Invalidation key: flatten_parameters rebinds p.data to views of the flat buffer, and a rebound .data has its
        # OWN version counter - load_state_dict, p.copy_/add_ under no_grad and torch.optim optimisers bump the
        # parameters' versions, never the flat buffer's.  So the key is the sum of the GEMM weights' versions (host-side,
        # ~50 attribute reads) plus the flat buffer's own version (flat.copy_ / flat.zero_ style writes);
        # load_state_dict / _apply also drop the cache outright.  FusedAdam writes through raw pointers (no version
        # moves): it reports the byte ranges it updated instead (`_eav_dirty`, recorded only because this model asked
        # for it through `_eav_track_dirty`).
        flat._eav_track_dirty = True
        key = (flat.data_ptr(), flat._version, sum(self._pmap[pn]._version for _, pn, _, _ in self._weight_keys()))
        dirty = getattr(flat, "_eav_dirty", [])
        flat._eav_dirty = []
        have_T = self._wplanes is not None and self._wplanes["_T"]
        keys = self._weight_keys()
        kp = lambda k: _lib.plain("eav_sp_kpad", k)  # noqa: E731
        everything = self._wplanes is None or self._wplanes["_dev"] != dev or (need_T and not have_T) \
            or self._wplanes_key != key
        if everything and (self._wplanes is None or self._wplanes["_dev"] != dev or (need_T and not have_T)):
            wp = {"_T": need_T, "_dev": dev,
                  "_slots": torch.zeros(len(keys), self.SLOT, dtype=torch.float32, device=dev),
                  # max_n ||W1_n||_2 per layer: input of the a-priori scale of the MLP activation (eav_tf_forward_scales)
                  "_wnorm_fc1": torch.zeros(self.cfg.layers, dtype=torch.float32, device=dev),
                  # max_n ||Wqkv_n||_2 per layer: bound of the fused q/k/v projection's output
                  "_wnorm_qkv": torch.zeros(self.cfg.layers, dtype=torch.float32, device=dev),
                  # max_j ||W2[:, j]||_2 per layer: bound of the MLP hidden-state gradient (eav_sp_bound_scale)
                  "_wcolnorm_fc2": torch.zeros(self.cfg.layers, dtype=torch.float32, device=dev)}
            for n, (k, _, out, inn) in enumerate(keys):
                wp[k] = (torch.empty(out, 2 * kp(inn), dtype=torch.float16, device=dev),
                         torch.empty(inn, 2 * kp(out), dtype=torch.float16, device=dev) if need_T else None, n)
            self._wplanes = wp
        wp = self._wplanes
        stale = []
        for k, pname, out, inn in keys:
            src = _lib.ptr(self._pmap[pname])
            if everything or any(lo < src + 4 * out * inn and src < hi for lo, hi in dirty):
                stale.append((k, src, out, inn))
        # After an optimiser step every matrix is stale: ~100 small launches (max|w| + conversion per matrix).  They go to
        # the side stream - idle during the forward - in layer order, one event per matrix; the main stream waits for a
        # matrix's event right before the first GEMM that reads its planes (_wp), so only the patch projection's
        # conversion is ever on the critical path.
        side = self._side_stream(dev) if (stale and self._two_streams() and self.kernel_events is None) else None
        self._wready = {}
        self._wnorm_ready = None
        if side is not None:
            start = torch.cuda.Event()
            start.record()                      # the weights are final (the optimiser ran on this stream)
            side.wait_event(start)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            st = _lib.stream_ptr()
            if len(stale) == len(keys):
                wp["_slots"].zero_()
                wp["_wnorm_fc1"].zero_()
                wp["_wcolnorm_fc2"].zero_()
                wp["_wnorm_qkv"].zero_()
                # everything is stale (the state after an optimiser step): the whole table in two launches
                jobs = wp.get("_jobs")
                if jobs is None or wp["_jobs_key"] != key[0]:
                    rows = []
                    for k, src, out, inn in stale:
                        pl, plT, n = wp[k]
                        rows.append([src, _lib.ptr(pl), _lib.ptr(plT) or 0,
                                     wp["_slots"].data_ptr() + 4 * self.SLOT * n, out | (inn << 32)])     # EavPlaneJob
                    jobs = wp["_jobs"] = torch.tensor(rows, dtype=torch.int64).to(dev)
                    wp["_jobs_key"] = key[0]
                # the row / column norms behind the a-priori scales FIRST (their own event: the forward's scales wait for
                # nothing else) and in one launch for the whole table (36 launches of ~10 us stood between the optimiser step
                # and the first scales of the next forward)
                njobs = wp.get("_njobs")
                if njobs is None or wp["_njobs_key"] != (key[0], wp["_T"]):
                    rows, mb = [], 1
                    for k, src, out, inn in stale:
                        if k.startswith("fc1"):
                            rows.append([src, inn, wp["_wnorm_fc1"].data_ptr() + 4 * int(k[3:]), out | (inn << 32), 0])
                            mb = max(mb, min((out + 3) // 4, 128))
                        elif k.startswith("fc2") and wp["_T"]:
                            # (keyed on the planes HAVING transposes, not on this call's need_T: a no_grad forward right after
                            # an optimiser step refreshes everything with need_T = False, and the next training step finds
                            # nothing stale - its backward must still see the norms of the CURRENT weights)
                            rows.append([src, inn, wp["_wcolnorm_fc2"].data_ptr() + 4 * int(k[3:]), out | (inn << 32), 1])
                            mb = max(mb, (inn + 63) // 64)
                        elif k.startswith("qkv"):
                            rows.append([src, inn, wp["_wnorm_qkv"].data_ptr() + 4 * int(k[3:]), out | (inn << 32), 0])
                            mb = max(mb, min((out + 3) // 4, 128))
                    njobs = wp["_njobs"] = (torch.tensor(rows, dtype=torch.int64).to(dev), len(rows), mb)
                    wp["_njobs_key"] = (key[0], wp["_T"])
                if njobs[1]:
                    _lib.call("eav_norm_max_multi", _lib.ptr(njobs[0]), njobs[1], njobs[2], st)
                if side is not None:
                    self._wnorm_ready = torch.cuda.Event()
                    self._wnorm_ready.record(side)
                _lib.call("eav_sp_refresh_planes", _lib.ptr(jobs), len(stale), max(o for _, _, o, _ in stale),
                          max(i for _, _, _, i in stale), st)
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._wready = {k: ev for k, _, _, _ in stale}
                stale = []
            # some matrices are stale (a partial update): the row / column norms first - the a-priori scales of EVERY layer
            # are computed by one launch before layer 0 and wait for ONE event (`_wnorm_ready`), not for the conversions -
            # then max|w| + conversion per matrix with one event each (the main stream waits for a matrix's planes right
            # before the first GEMM that reads them: _wp)
            for k, src, out, inn in stale:
                li = int(k[3:]) if k[:3] in ("fc1", "fc2", "qkv") else -1
                if k.startswith("fc1"):
                    wp["_wnorm_fc1"][li].zero_()
                    _lib.call("eav_rownorm_max", src, out, inn, inn, wp["_wnorm_fc1"].data_ptr() + 4 * li, st)
                elif k.startswith("fc2") and wp["_T"]:
                    wp["_wcolnorm_fc2"][li].zero_()
                    _lib.call("eav_colnorm_max", src, out, inn, inn, wp["_wcolnorm_fc2"].data_ptr() + 4 * li, st)
                elif k.startswith("qkv"):
                    wp["_wnorm_qkv"][li].zero_()
                    _lib.call("eav_rownorm_max", src, out, inn, inn, wp["_wnorm_qkv"].data_ptr() + 4 * li, st)
            if stale and side is not None:
                self._wnorm_ready = torch.cuda.Event()
                self._wnorm_ready.record(side)
            for k, src, out, inn in stale:
                pl, plT, n = wp[k]
                wp["_slots"][n].zero_()
                slot = wp["_slots"].data_ptr() + 4 * self.SLOT * n
                _lib.call("eav_sp_absmax", src, out, inn, inn, slot, st)
                _lib.call("eav_sp_convert", src, out, inn, inn, slot, _lib.ptr(pl), _lib.ptr(plT), st)
                if side is not None:
                    ev = torch.cuda.Event()
                    ev.record(side)
                    self._wready[k] = ev
        self._wplanes_key = key

    def _terms(self, kind):
        """MFMA terms of the backward products of `kind` ("dgrad" | "wgrad"): 3 = fp32-grade, 1 = hi.hi only; wgrad also 2 =
        hi.hi + lo_grad.hi_act (the activation operand rounded to fp16, the gradient operand at full split precision -
        the producers' a-priori gradient planes stay on)."""
        t = self.wgrad_terms if kind == "wgrad" else self.dgrad_terms
        return self.grad_terms if t is None else int(t)

    def _bwd_three_terms(self):
        """Every backward product on three terms: the producers may then write gradient planes under loose a-priori
        bounds (a hi.hi-only product needs the tight measured scale of its operands)."""
        return self._terms("dgrad") != 1 and self._terms("wgrad") != 1

    def _two_streams(self):
        """Whether this step runs its weight gradients / final reductions beside the main stream (overlap_wgrad)."""
        o = self.overlap_wgrad
        if o == "auto":
            ws = getattr(self, "_ws", None)
            return ws is not None and ws.B * self.cfg.ntok >= 8192
        return bool(o)

    def _cur_stream(self):
        """The stream the current launch sequence runs on (cached by _launch_forward / _launch_backward / the head functions:
        torch.cuda.current_stream() costs tens of microseconds per call, and the step waits on ~50 events)."""
        m = getattr(self, "_main", None)
        return m if m is not None else torch.cuda.current_stream()

    def _wp(self, key, transposed=False):
        ev = self._wready.pop(key, None)
        if ev is not None:
            self._cur_stream().wait_event(ev)
        pl, plT, n = self._wplanes[key]
        return _lib.ptr(plT if transposed else pl), self._wplanes["_slots"].data_ptr() + 4 * self.SLOT * n

    def _to_planes(self, src, R, C, ld, slot, dst, amax_done=False):
        """fp32 [R, C] -> row planes (one set serves the products that contract over columns AND those over rows)."""
        if not amax_done:
            self._call("eav_sp_absmax", src, R, C, ld, slot, self._st)
        self._call("eav_sp_convert", src, R, C, ld, slot, _lib.ptr(dst), None, self._st)

    def _part_buf(self, pool):
        """Next buffer of a small ring of partial-sum buffers.  The final reductions of bias / LayerNorm parameter gradients
        are gradient OUTPUTS nothing downstream reads, so they run on the side stream (_reduce_async); the main stream
        waits for the reduction that last read a buffer only when the ring comes round to it (a layer later)."""
        ws = self._ws
        ring = getattr(ws, pool)
        i = self._ring_pos.get(pool, -1) + 1
        self._ring_pos[pool] = i = i % len(ring)
        ev = self._part_busy.pop(ring[i].data_ptr(), None)
        if ev is not None:
            self._cur_stream().wait_event(ev)
        return ring[i]

    def _reduce_async(self, buf, off_bytes, nparts, stride, n, out):
        """Final fixed-order reduction of a partial-sum buffer into a gradient nothing downstream reads: off the main stream.
        It gets its OWN stream (not the weight-gradient stream): a 24-block kernel queued in order between persistent GEMMs
        waits for a CU whose LDS is not taken by two GEMM workgroups, and held the weight gradients behind it back by
        ~130 us per launch at ViT B=128 (9.6 ms of side-stream time per step)."""
        if not (self._two_streams() and self.kernel_events is None):
            self._call("eav_reduce_partials", _lib.ptr(buf) + off_bytes, nparts, stride, n, 1.0, out, self._st)
            return
        if self._aux is None or self._aux.device != buf.device:
            self._aux = torch.cuda.Stream(device=buf.device)
        aux = self._aux
        ready = torch.cuda.Event()
        ready.record()
        aux.wait_event(ready)
        _lib.call("eav_reduce_partials", _lib.ptr(buf) + off_bytes, nparts, stride, n, 1.0, out, aux.cuda_stream)
        done = torch.cuda.Event()
        done.record(aux)
        self._part_busy[buf.data_ptr()] = done

    def _to_planes_bias(self, src, R, C, slot, dst, bias_grad):
        """Conversion pass that also produces the bias gradient (column sums of src) - src's max|x| is already in slot."""
        ws = self._ws
        self._before_overwrite(dst)
        part = self._part_buf("part_cs2_pool")
        self._call("eav_sp_convert_colsum", src, R, C, C, slot, _lib.ptr(dst), None, _lib.ptr(part), self._st)
        self._reduce_async(part, 0, ws.np_cs2, C, C, bias_grad)

    def _gemm_sp(self, A, slotA, B, slotB, C, M, N, K, ldc, batch=1, sA=0, sC=0, alpha=1.0, bias=None, gelu=0,
                 pre=None, resid=None, ldr=0, acc=0, amax=None, blockmax=True):
        """C[M,N] = epilogue(alpha A[M,K] . B[N,K]^T) on planes."""
        # flags: backward products with grad_terms = 1 run on the hi.hi term alone (see the class attribute); the backward's
        # data gradients share the GPU with the side stream's weight gradients (EAV_GEMM_SHARED_GPU: see csrc/gemm_sp.hip)
        bwd = self._phase == "bwd"
        flags = (1 if (self._terms("dgrad") if bwd else self.fwd_terms) == 1 else 0) | (2 if bwd and self._two_streams() else 0) \
            | (0 if blockmax else 8)
        self._call("eav_gemm_sp_ex", A, B, C, slotA, slotB, M, N, K, ldc, batch, sA, sC, float(alpha), bias, gelu, pre,
                   resid, ldr, acc, amax, None, None, None, flags, self._st)

    def _wgrad_sp(self, AT, slotA, BT, slotB, C, M, N, K):
        """C[M,N] = sum over the K tokens of A[t,m] B[t,n]: ROW planes of A [K,M] and B [K,N] (the contraction runs over the
        rows: eav_gemm_sp_splitk reads the fragments with transposing LDS loads); split-K.

        Weight gradients are off the critical path of the backward (nothing downstream reads them before the optimiser),
        so they run on a side HIP stream: their MFMA work fills the matrix pipe while the main stream is in its
        HBM- / VALU-bound stretches (operand conversions, LayerNorm / GELU backward, the attention backward) and in the
        tails of its own GEMMs.  Ordering: the side stream waits for the event recorded after the conversion that
        produced A; the main stream waits for a weight gradient only before it overwrites that gradient's A planes
        (one layer later) and at the end of the backward."""
        name = {1: "eav_gemm_sp_splitk_x1", 2: "eav_gemm_sp_splitk_x2"}.get(self._terms("wgrad"), "eav_gemm_sp_splitk")
        if not self._two_streams() or (self.kernel_events is not None and name in self.kernel_events):
            self._call(name, _lib.ptr(AT), _lib.ptr(BT), C, _lib.ptr(self._ws.splitk), slotA, slotB, M,
                       N, K, 0, self._st)
            return
        self._side_stream(AT.device)
        ready = torch.cuda.Event()
        ready.record()
        self._side.wait_event(ready)
        _lib.call(name, _lib.ptr(AT), _lib.ptr(BT), C, _lib.ptr(self._ws.splitk), slotA, slotB, M, N, K,
                  0, self._side.cuda_stream)
        done = torch.cuda.Event()
        done.record(self._side)
        self._wgrad_done[AT.data_ptr()] = done

    def _side_stream(self, dev):
        if self._side is None or self._side.device != dev:
            self._side = torch.cuda.Stream(device=dev)
        return self._side

    def _before_overwrite(self, buf):
        """Main stream: the weight gradient that still reads `buf` (launched a layer ago on the side stream) must be
        finished before the next conversion overwrites it."""
        ev = self._wgrad_done.pop(buf.data_ptr(), None) if buf is not None else None
        if ev is not None:
            self._cur_stream().wait_event(ev)

    def _join_wgrads(self):
        if self._side is not None and self._wgrad_done:
            self._cur_stream().wait_stream(self._side)
            self._wgrad_done.clear()
        if self._aux is not None and self._part_busy:
            self._cur_stream().wait_stream(self._aux)
            self._part_busy.clear()

    # ------------------------------------------------------------------ classification head (shared by the full path
    # and by Encoder.head on cached features: same kernels, same arguments, hence bit-equal logits and gradients)
    def _head_forward(self, feat, hw, B):
        """feat [B, D] = the classifier's input (AST: mean of the cls / distillation rows after the final LayerNorm,
        HF modeling_audio_spectrogram_transformer.py ASTMLPHead; ViT: the cls row after the final LayerNorm)."""
        c, P, L, st = self.cfg, _lib.ptr, self._call, self._st
        D = c.hidden
        w = lambda k: P(self._pmap[k])  # noqa: E731
        if c.kind == "ast":
            sh = P(hw.sth)
            L("eav_layernorm_fwd", P(feat), w("classifier.layernorm.weight"), w("classifier.layernorm.bias"),
              P(hw.hl), sh, sh + 4 * B, B, D, c.eps, st)
            L("eav_dense_softmax_fwd", P(hw.hl), w("classifier.dense.weight"), w("classifier.dense.bias"),
              P(hw.logits), None, B, D, c.num_labels, st)
        else:
            L("eav_dense_softmax_fwd", P(feat), w("classifier.weight"), w("classifier.bias"), P(hw.logits), None,
              B, D, c.num_labels, st)

    def _head_backward(self, dlogits, feat, hw, B, need_dfeat):
        """Head gradients into the flat gradient buffer; d loss / d feat into hw.dpooled (AST) / hw.dseqr (ViT)."""
        c, P, L, st = self.cfg, _lib.ptr, self._call, self._st
        D = c.hidden
        gflat, offs = self._flat[1], self._flat[2]
        gp = lambda k: gflat.data_ptr() + 4 * offs[k][0]  # noqa: E731
        w = lambda k: P(self._pmap[k])  # noqa: E731
        if c.kind == "ast":
            L("eav_dense_softmax_bwd", P(dlogits), None, P(hw.hl), w("classifier.dense.weight"),
              gp("classifier.dense.weight"), gp("classifier.dense.bias"), P(hw.dhl), B, D, c.num_labels, st)
            sh = P(hw.sth)
            L("eav_layernorm_bwd", P(hw.dhl), P(feat), w("classifier.layernorm.weight"), sh, sh + 4 * B,
              P(hw.dpooled), 0, P(hw.part_lnr), B, D, st)
            npb = _lib.plain("eav_layernorm_bwd_nparts", B)
            self._reduce(hw.part_lnr, npb, 2 * D, D, gp("classifier.layernorm.weight"))
            L("eav_reduce_partials", P(hw.part_lnr) + 4 * D, npb, 2 * D, D, 1.0, gp("classifier.layernorm.bias"), st)
        else:
            L("eav_dense_softmax_bwd", P(dlogits), None, P(feat), w("classifier.weight"), gp("classifier.weight"),
              gp("classifier.bias"), P(hw.dseqr), B, D, c.num_labels, st)

    def _head_ws(self, B, dev):
        hw = self._hws.get(B)
        if hw is None or hw.feat.device != dev or hw.logits.shape[1] != self.cfg.num_labels:
            c, D = self.cfg, self.cfg.hidden
            f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
            hw = self._hws[B] = SimpleNamespace(
                feat=f(B, D), hl=f(B, D), sth=f(2, B), logits=f(B, c.num_labels), dhl=f(B, D), dpooled=f(B, D),
                dseqr=f(B, D), part_lnr=f(_lib.plain("eav_layernorm_bwd_nparts", B * c.nextra), 2 * D), token=-1)
        return hw

    def last_features(self):
        """The classifier's input of the most recent forward ([B, hidden], a copy): constant per sample while the backbone
        is frozen (every dropout of the reference checkpoints is 0.0), which is what the trainers' frozen-phase feature
        cache stores (finetune.FineTuneBase)."""
        ws = self._ws
        if ws is None:
            raise _lib.EavError("Encoder.last_features: no forward has run")
        return (ws.pooled if self.cfg.kind == "ast" else ws.seqr[:ws.B]).clone()

    def head(self, feat):
        """Classifier on backbone features [B, hidden] (from last_features): logits with the head's autograd graph."""
        if not isinstance(feat, torch.Tensor) or not feat.is_cuda or feat.dim() != 2 or feat.shape[1] != self.cfg.hidden:
            raise _lib.EavError("Encoder.head: features must be a [batch, hidden] tensor on the ROCm device")
        self._ensure_flat()
        return _Out(_HeadFn.apply(feat.contiguous().float(), self, *[self._pmap[k] for k in self._names]))

    def _launch_forward(self, x):
        c = self.cfg
        P, L = _lib.ptr, self._call
        self._main = torch.cuda.current_stream()      # (looked up once per launch sequence: ~50 event waits per step use it)
        self._st = st = self._main.cuda_stream
        self._phase = "fwd"
        B = x.shape[0]
        D, FF, N, H = c.hidden, c.ff, c.ntok, c.heads
        hd = D // H
        pm = self._pmap
        full = self._want_full
        ws = self._ws
        sp = self.precision == "split"
        ok = lambda v: (v is not None and v.B == B and v.hs[0].device == x.device and (v.full or not full)  # noqa: E731
                        and v.fused == self._fused_attention() and v.sp == sp)
        if not ok(ws):
            # one workspace per batch size (at most three: the full batch, a ragged last batch, an evaluation batch) - the
            # 5000 % 128 = 8 frames at the end of every vision epoch must not free and re-zero the 19 GB of the B = 128 one
            ws = self._ws_cache.get(B)
            if not ok(ws):
                if len(self._ws_cache) >= 3:
                    self._ws_cache.pop(next(k for k in self._ws_cache if k != B))
                    torch.cuda.empty_cache()
                ws = self._ws_cache[B] = self._alloc(B, x.device, full)
            self._ws = ws
        if sp:
            self._refresh_weight_planes(x.device, full)
            ws.fslots.zero_()
            fslot = lambda n: ws.fslots.data_ptr() + 4 * self.SLOT * n  # noqa: E731
            kpb = lambda k: 4 * _lib.plain("eav_sp_kpad", k)            # noqa: E731  plane row stride in bytes
        M, ldn = ws.M, ws.ldn
        pre = c.prefix
        w = lambda k: P(pm[k])  # noqa: E731
        # patch embedding: im2col rows x projection weight -> token rows [nextra:], then cls/dist + positions
        L("eav_im2col", P(x), P(ws.col), B, c.C, c.H, c.W, c.patch, c.sy, c.sx, c.transposed, st)
        h0 = ws.hs[0]
        if sp:
            self._to_planes(P(ws.col), B * c.npatch, c.kp, c.kp, fslot(0), ws.colp)
            wpl, wsl = self._wp("patch")
            self._gemm_sp(P(ws.colp), fslot(0), wpl, wsl, P(h0) + 4 * c.nextra * D, c.npatch, D, c.kp, D, batch=B,
                          sA=c.npatch * kpb(c.kp), sC=N * D,
                          bias=w(f"{pre}.embeddings.patch_embeddings.projection.bias"))
        else:
            self._gemm(P(ws.col), w(f"{pre}.embeddings.patch_embeddings.projection.weight"),
                       P(h0) + 4 * c.nextra * D, c.npatch, D, c.kp, c.kp, c.kp, D, batch=B,
                       sA=(c.npatch * c.kp, 0), sC=(N * D, 0),
                       bias=w(f"{pre}.embeddings.patch_embeddings.projection.bias"))
        L("eav_embed_finish", P(h0), w(f"{pre}.embeddings.cls_token"),
          w(f"{pre}.embeddings.distillation_token") if c.kind == "ast" else None,
          w(f"{pre}.embeddings.position_embeddings"), B, N, D, c.nextra, st)
        scale = hd ** -0.5
        for i in range(c.layers):
            j = i if ws.full else 0
            hin = ws.hs[i] if ws.full else ws.hs[i & 1]
            hout = ws.hs[i + 1] if ws.full else ws.hs[(i + 1) & 1]
            Lk = f"{pre}.layers.{i}"
            stp = P(ws.st[j])
            if sp:
                if i == 0:
                    self._forward_scales(fslot)
                self._layer_forward_split(i, j, hin, hout, Lk, stp, fslot, scale)
                continue
            L("eav_layernorm_fwd", P(hin), w(f"{Lk}.layernorm_before.weight"), w(f"{Lk}.layernorm_before.bias"),
              P(ws.y1[j]), stp, stp + 4 * M, M, D, c.eps, st)
            qkv = P(ws.qkv[j])
            self._gemm(P(ws.y1[j]), w(f"{Lk}.attention.q_proj.weight"), qkv, M, 3 * D, D, D, D, 3 * D,
                       bias=w(f"{Lk}.attention.q_proj.bias"))
            if ws.fused:
                L("eav_attn_fwd", qkv, P(ws.ao[j]), P(ws.lse[j]), B, H, N, hd, scale, st)
            else:
                Pm = P(ws.P[j])
                self._gemm(qkv, qkv + 4 * D, Pm, N, N, hd, 3 * D, 3 * D, ldn, batch=B * H, heads=H,
                           sA=(N * 3 * D, hd), sB=(N * 3 * D, hd), sC=(H * N * ldn, N * ldn), alpha=scale)
                L("eav_softmax_fwd", Pm, B * H * N, N, ldn, st)
                self._gemm(Pm, qkv + 8 * D, P(ws.ao[j]), N, hd, N, ldn, 3 * D, D, tB=1, batch=B * H, heads=H,
                           sA=(H * N * ldn, N * ldn), sB=(N * 3 * D, hd), sC=(N * D, hd))
            self._gemm(P(ws.ao[j]), w(f"{Lk}.attention.o_proj.weight"), P(ws.hmid[j]), M, D, D, D, D, D,
                       bias=w(f"{Lk}.attention.o_proj.bias"), resid=P(hin), ldr=D)
            L("eav_layernorm_fwd", P(ws.hmid[j]), w(f"{Lk}.layernorm_after.weight"), w(f"{Lk}.layernorm_after.bias"),
              P(ws.y2[j]), stp + 8 * M, stp + 12 * M, M, D, c.eps, st)
            self._gemm(P(ws.y2[j]), w(f"{Lk}.mlp.fc1.weight"), P(ws.act[j]), M, FF, D, D, D, FF,
                       bias=w(f"{Lk}.mlp.fc1.bias"), gelu=1, pre=P(ws.pre[j]))
            self._gemm(P(ws.act[j]), w(f"{Lk}.mlp.fc2.weight"), P(hout), M, D, FF, FF, FF, D,
                       bias=w(f"{Lk}.mlp.fc2.bias"), resid=P(ws.hmid[j]), ldr=D)
        hlast = ws.hs[c.layers] if ws.full else ws.hs[c.layers & 1]
        R = B * c.nextra
        L("eav_token_rows", P(hlast), P(ws.rows), B, N, D, c.nextra, 0, st)
        sf = P(ws.stf)
        L("eav_layernorm_fwd", P(ws.rows), w(f"{pre}.layernorm.weight"), w(f"{pre}.layernorm.bias"), P(ws.seqr), sf,
          sf + 4 * R, R, D, c.eps, st)
        if c.kind == "ast":
            L("eav_pair_mean", P(ws.seqr), P(ws.pooled), B, D, 0, st)
        self._head_forward(ws.pooled if c.kind == "ast" else ws.seqr, ws, B)
        self._token += 1
        self._saved = (self._token, x, full, None)
        return self._token

    def _forward_scales(self, fslot):
        """A-priori operand scales (rigorous bounds: eav_tf_forward_scales_qkv) of y1, qkv, y2, act of EVERY layer in one
        launch - the flat parameter buffer lays the layers out identically.  Falls back to one launch per layer (inside
        _layer_forward_split) if it does not."""
        c, ws = self.cfg, self._ws
        self._scales_all = False
        if not (self.fused_planes and c.hidden % 8 == 0 and c.ff % 8 == 0):
            return
        offs, pre = self._flat[2], c.prefix
        first = lambda i: offs[f"{pre}.layers.{i}.attention.q_proj.weight"][0]  # noqa: E731
        names = ("layernorm_before.weight", "layernorm_before.bias", "layernorm_after.weight", "layernorm_after.bias",
                 "mlp.fc1.bias", "attention.q_proj.bias")
        rel = [tuple(offs[f"{pre}.layers.{i}.{n}"][0] - first(i) for n in names) for i in range(c.layers)]
        stride = first(1) - first(0) if c.layers > 1 else 0
        if any(r != rel[0] for r in rel) or any(first(i) - first(0) != i * stride for i in range(c.layers)):
            return
        if self._wnorm_ready is not None:              # the row norms come from the side-stream weight refresh
            self._cur_stream().wait_event(self._wnorm_ready)
        o = rel[0]
        qkvp = self.fused_qkv and ws.fused
        self._call("eav_tf_forward_scales_qkv", _lib.ptr(self._flat[0]) + 4 * first(0), stride, c.layers, o[0], o[1], o[2],
                   o[3], o[4], o[5], c.hidden, c.ff, self._wplanes["_wnorm_fc1"].data_ptr(),
                   self._wplanes["_wnorm_qkv"].data_ptr(), fslot(1), self.FS * self.SLOT, 0, 3, 4, 1 if qkvp else -1,
                   self._st)
        self._scales_all = True

    def _layer_forward_split(self, i, j, hin, hout, Lk, stp, fslot, scale):
        """One encoder layer with every projection on the split-operand GEMM; the attention core stays on the fp32
        kernels.  LayerNorm / attention / GELU outputs are converted to planes once (plus the transposed planes when
        a backward will follow); only the planes are kept per layer."""
        c, ws = self.cfg, self._ws
        P, L, st = _lib.ptr, self._call, self._st
        D, FF, N, H, M = c.hidden, c.ff, c.ntok, c.heads, ws.M
        hd = D // H
        w = lambda k: P(self._pmap[k])  # noqa: E731
        s_y1, s_qkv, s_ao, s_y2, s_act = (fslot(1 + self.FS * i + k) for k in range(5))
        y, ao, act = ws.y1[0], ws.ao[j], ws.act[0]
        # Producers write the operand planes themselves where a rigorous bound of the tensor exists BEFORE it is computed
        # (LayerNorm outputs, the MLP's GELU output: eav_tf_forward_scales) - no fp32 copy, no measured maximum, no
        # conversion pass for y1, y2, act.  The attention output keeps the measured scale (its kernel emits max|O|).
        fusedp = self.fused_planes and D % 8 == 0 and FF % 8 == 0
        if fusedp:
            self._wp(f"fc1{i}")        # (waits for the side-stream refresh of this layer's fc1 planes / row norms)
            offs = self._flat[2]
            base = offs[f"{Lk}.attention.q_proj.weight"][0]
            o = lambda k: offs[f"{Lk}.{k}"][0] - base  # noqa: E731
            qkvp = self.fused_qkv and ws.fused
            if not self._scales_all:
                L("eav_tf_forward_scales_qkv", P(self._flat[0]) + 4 * base, 0, 1, o("layernorm_before.weight"),
                  o("layernorm_before.bias"), o("layernorm_after.weight"), o("layernorm_after.bias"), o("mlp.fc1.bias"),
                  o("attention.q_proj.bias"), D, FF, self._wplanes["_wnorm_fc1"].data_ptr() + 4 * i,
                  self._wplanes["_wnorm_qkv"].data_ptr() + 4 * i, s_y1, 0, 0, 3, 4, 1 if qkvp else -1, st)
            L("eav_layernorm_fwd_planes", P(hin), w(f"{Lk}.layernorm_before.weight"), w(f"{Lk}.layernorm_before.bias"),
              None, P(ws.y1p[j]), s_y1, stp, stp + 4 * M, M, D, c.eps, st)
        else:
            L("eav_layernorm_fwd_amax", P(hin), w(f"{Lk}.layernorm_before.weight"), w(f"{Lk}.layernorm_before.bias"),
              P(y), stp, stp + 4 * M, M, D, c.eps, s_y1, st)
            self._to_planes(P(y), M, D, D, s_y1, ws.y1p[j], amax_done=True)
        qkv = P(ws.qkv[0 if ws.fused else j])
        wpl, wsl = self._wp(f"qkv{i}")
        qkvp = fusedp and self.fused_qkv and ws.fused
        if qkvp:
            # the projection writes the row planes of Q | K | V itself (lo without the 2^11 lift: the attention kernels' format,
            # scale = the bound eav_tf_forward_scales_qkv put into s_qkv); the per-head transposes (V^T for the forward; Q^T,
            # K^T for the backward) are a pure fp16 transposition of those planes
            L("eav_gemm_sp_ex", P(ws.y1p[j]), wpl, None, s_y1, wsl, M, 3 * D, D, 3 * D, 1, 0, 0, 1.0,
              w(f"{Lk}.attention.q_proj.bias"), 0, None, None, 0, 0, s_qkv, P(ws.qkvrow[j]), s_qkv, None,
              4 | 8 | (1 if self.fwd_terms == 1 else 0), st)      # (+ the MEASURED max|qkv| into the slot's shards: the
            #                                                         backward's bound of |dqkv| uses it, eav_attn_dqkv_bound)
        else:
            self._gemm_sp(P(ws.y1p[j]), s_y1, wpl, wsl, qkv, M, 3 * D, D, 3 * D, bias=w(f"{Lk}.attention.q_proj.bias"),
                          amax=s_qkv if ws.fused else None)
        if ws.fused:
            if not qkvp:
                # row planes of Q | K | V and the per-head transposes (V^T for the forward; Q^T, K^T for the backward)
                L("eav_attn_sp_prep", qkv, s_qkv, P(ws.qkvrow[j]), None, ws.B, N, 3 * D, D, 0, st)
            if fusedp and self.fused_ao:
                # the attention output leaves as the planes of the o-proj products (scale: qkv's own, |O| <= max|V|); its
                # fp32 copy is written only when a backward will read it
                # (... and only by the unfused gradient flow: the fused one forms delta = dO . O from these planes)
                # (recorded on the workspace: the backward forms delta from ws.aop ONLY if THIS kernel wrote them - its
                # planes carry one tensor-wide scale; eav_sp_convert's planes below have per-row-block boosts)
                ws.delta_from_planes = self.fused_dqkv and self._bwd_three_terms()
                need_ao = ws.full and not ws.delta_from_planes
                L("eav_attn_fwd_sp_planes", P(ws.qkvrow[j]), None, s_qkv, P(ao) if need_ao else None,
                  P(ws.lse[j]), None, P(ws.aop[j]), s_ao, ws.B, H, N, hd, scale, st)
            else:
                ws.delta_from_planes = False
                L("eav_attn_fwd_sp", P(ws.qkvrow[j]), None, s_qkv, P(ao), P(ws.lse[j]), s_ao, ws.B, H, N, hd,
                  scale, st)
        else:
            ldn = ws.ldn
            Pm = P(ws.P[j])
            self._gemm_f32(qkv, qkv + 4 * D, Pm, N, N, hd, 3 * D, 3 * D, ldn, batch=ws.B * H, heads=H,
                           sA=(N * 3 * D, hd), sB=(N * 3 * D, hd), sC=(H * N * ldn, N * ldn), alpha=scale)
            L("eav_softmax_fwd", Pm, ws.B * H * N, N, ldn, st)
            self._gemm_f32(Pm, qkv + 8 * D, P(ao), N, hd, N, ldn, 3 * D, D, tB=1, batch=ws.B * H, heads=H,
                           sA=(H * N * ldn, N * ldn), sB=(N * 3 * D, hd), sC=(N * D, hd))
        if not (ws.fused and fusedp and self.fused_ao):
            self._to_planes(P(ao), M, D, D, s_ao, ws.aop[j], amax_done=ws.fused)
        wpl, wsl = self._wp(f"o{i}")
        self._gemm_sp(P(ws.aop[j]), s_ao, wpl, wsl, P(ws.hmid[j]), M, D, D, D, bias=w(f"{Lk}.attention.o_proj.bias"),
                      resid=P(hin), ldr=D)
        wpl, wsl = None, None
        if fusedp:
            L("eav_layernorm_fwd_planes", P(ws.hmid[j]), w(f"{Lk}.layernorm_after.weight"),
              w(f"{Lk}.layernorm_after.bias"), None, P(ws.y2p[j]), s_y2, stp + 8 * M, stp + 12 * M, M, D, c.eps, st)
            wpl, wsl = self._wp(f"fc1{i}")
            # fc1: bias + erf-GELU in the epilogue; the pre-activation is kept (fp32) for the backward only, the
            # activation leaves as planes - it never exists in fp32
            self._call("eav_gemm_sp_ex", P(ws.y2p[j]), wpl, None, s_y2, wsl, M, FF, D, FF, 1, 0, 0, 1.0,
                       w(f"{Lk}.mlp.fc1.bias"), 1, P(ws.pre[j]) if ws.full else None, None, 0, 0, None, P(ws.actp[j]),
                       s_act, None, 1 if self.fwd_terms == 1 else 0, st)
        else:
            L("eav_layernorm_fwd_amax", P(ws.hmid[j]), w(f"{Lk}.layernorm_after.weight"),
              w(f"{Lk}.layernorm_after.bias"), P(y), stp + 8 * M, stp + 12 * M, M, D, c.eps, s_y2, st)
            self._to_planes(P(y), M, D, D, s_y2, ws.y2p[j], amax_done=True)
            wpl, wsl = self._wp(f"fc1{i}")
            # fc1 stores the PRE-activation only (kept per layer for the backward; a scratch buffer in the frozen phase)
            # and max|GELU|; the conversion applies the GELU while it splits - the activation never exists in fp32
            pre = P(ws.pre[j]) if ws.full else P(act)
            self._gemm_sp(P(ws.y2p[j]), s_y2, wpl, wsl, pre, M, FF, D, FF, bias=w(f"{Lk}.mlp.fc1.bias"), gelu=3,
                          amax=s_act)
            self._call("eav_sp_convert_gelu", pre, M, FF, FF, s_act, P(ws.actp[j]), None, self._st)
        wpl, wsl = self._wp(f"fc2{i}")
        self._gemm_sp(P(ws.actp[j]), s_act, wpl, wsl, P(hout), M, D, FF, D, bias=w(f"{Lk}.mlp.fc2.bias"),
                      resid=P(ws.hmid[j]), ldr=D)

    def _gemm_f32(self, A, B, C, M, N, K, lda, ldb, ldc, tA=0, tB=0, batch=1, heads=1, sA=(0, 0), sB=(0, 0),
                  sC=(0, 0), alpha=1.0):
        self._call("eav_gemm_f32", A, B, C, M, N, K, lda, ldb, ldc, tA, tB, batch, heads, sA[0], sA[1], sB[0], sB[1],
                   sC[0], sC[1], float(alpha), None, 0, None, None, 0, 0, self._st)

    def _layer_backward_split(self, i, Lk, stp, gp, bslot, fslot, scale):
        """Backward of one layer: dh (gradient w.r.t. the layer output, fp32) in ws.dh on entry, gradient w.r.t. the
        layer input on exit.  Every weight gradient is a split-K GEMM over the transposed planes; data gradients use
        the planes of the transposed weights."""
        c, ws = self.cfg, self._ws
        P, L, st = _lib.ptr, self._call, self._st
        D, FF, N, H, M = c.hidden, c.ff, c.ntok, c.heads, ws.M
        hd = D // H
        w = lambda k: P(self._pmap[k])  # noqa: E731
        dh, dy, dao, dact, dqkv = P(ws.dh), P(ws.dy), P(ws.dao), P(ws.dact), P(ws.dqkv)
        s_y1, s_qkv, s_ao, s_y2, s_act = (fslot(1 + self.FS * i + k) for k in range(5))
        b_dh2, b_dact, b_dh1, b_dao, b_ds, b_dqkv, b_dy2, b_dy1 = (bslot(1 + self.BS * i + k) for k in range(8))
        fdh = self.fused_dh and self._bwd_three_terms()      # (hi.hi-only gradient products need the tight measured scales)
        # fc2: h_out = h_mid + act.W2^T + b2.  max|dh| is already in b_dh2 (left there by the producer of dh); every
        # conversion pass also yields the bias gradient of its tensor.  (fused_dh: the layer above's LayerNorm backward
        # already wrote these planes and the bias-gradient partials - only the top layer's dh comes from the head)
        if not (fdh and i < c.layers - 1):
            self._to_planes_bias(dh, M, D, b_dh2, ws.dhp, gp(f"{Lk}.mlp.fc2.bias"))
        self._wgrad_sp(ws.dhp, b_dh2, ws.actp[i], s_act, gp(f"{Lk}.mlp.fc2.weight"), D, FF, M)
        wpl, wsl = self._wp(f"fc2{i}", transposed=True)
        if self.fused_dact and FF % 8 == 0:
            # data gradient through fc2 and the GELU in one pass, result straight into the planes of dact: its scale comes
            # from |dact| = |(dh W2) gelu'(pre)| <= 1.13 sqrt(D) max|dh| max_j ||W2[:, j]||_2 (max|dh| is in b_dh2, the column
            # norms are refreshed with the weight planes); the epilogue also leaves fc1's bias-gradient partials
            L("eav_sp_bound_scale", b_dact, b_dh2, self._wplanes["_wcolnorm_fc2"].data_ptr() + 4 * i,
              1.13 * float(np.sqrt(D)), st)
            self._before_overwrite(ws.dactp)
            part = self._part_buf("part_cs2_pool")
            flags = (1 if self._terms("dgrad") == 1 else 0) | (2 if self._two_streams() else 0)
            L("eav_gemm_sp_ex", P(ws.dhp), wpl, None, b_dh2, wsl, M, FF, D, FF, 1, 0, 0, 1.0, None, 2, P(ws.pre[i]), None, 0,
              0, None, P(ws.dactp), b_dact, P(part), flags, st)
            self._reduce_async(part, 0, ws.np_cs2, FF, FF, gp(f"{Lk}.mlp.fc1.bias"))
        else:
            # ... the epilogue multiplies by gelu'(pre) and emits max|dact|; one conversion pass (planes + bias gradient)
            self._gemm_sp(P(ws.dhp), b_dh2, wpl, wsl, dact, M, FF, D, FF, gelu=2, pre=P(ws.pre[i]), amax=b_dact)
            self._to_planes_bias(dact, M, FF, b_dact, ws.dactp, gp(f"{Lk}.mlp.fc1.bias"))
        # fc1
        self._wgrad_sp(ws.dactp, b_dact, ws.y2p[i], s_y2, gp(f"{Lk}.mlp.fc1.weight"), FF, D, M)
        wpl, wsl = self._wp(f"fc1{i}", transposed=True)
        if fdh:
            self._gemm_sp(P(ws.dactp), b_dact, wpl, wsl, dy, M, D, FF, D, amax=b_dy2, blockmax=False)
            self._ln_bwd_planes(dy, P(ws.hmid[i]), w(f"{Lk}.layernorm_after.weight"), stp + 8 * M, stp + 12 * M, dh, b_dh1,
                                b_dh2, b_dy2, ws.dhp2, gp(f"{Lk}.layernorm_after.weight"), gp(f"{Lk}.layernorm_after.bias"),
                                gp(f"{Lk}.attention.o_proj.bias"), s_y2)
        else:
            self._gemm_sp(P(ws.dactp), b_dact, wpl, wsl, dy, M, D, FF, D)
            part = self._part_buf("part_ln_pool")
            L("eav_layernorm_bwd_amax", dy, P(ws.hmid[i]), w(f"{Lk}.layernorm_after.weight"), stp + 8 * M, stp + 12 * M, dh,
              1, P(part), M, D, b_dh1, st)
            self._reduce_ln(part, gp(f"{Lk}.layernorm_after.weight"), gp(f"{Lk}.layernorm_after.bias"))
            # o_proj
            self._to_planes_bias(dh, M, D, b_dh1, ws.dhp2, gp(f"{Lk}.attention.o_proj.bias"))
        self._wgrad_sp(ws.dhp2, b_dh1, ws.aop[i], s_ao, gp(f"{Lk}.attention.o_proj.weight"), D, D, M)
        wpl, wsl = self._wp(f"o{i}", transposed=True)
        # (dao goes to the attention operand preparation: one scale per tensor, no row-block maxima needed)
        self._gemm_sp(P(ws.dhp2), b_dh1, wpl, wsl, dao, M, D, D, D, amax=b_dao if ws.fused else None, blockmax=False)
        # attention core
        if ws.fused:
            L("eav_attn_sp_prep", dao, b_dao, P(ws.dorow), None, ws.B, N, D, D, 0, st)
            # dqkv as planes straight from the attention backward, delta = dO . O from the planes of O - only when the
            # forward of THIS step wrote those planes with eav_attn_fwd_sp_planes (EAV_FUSED_AO=0 / EAV_FUSED_PLANES=0 runs
            # convert a fp32 O with per-row-block boosts the planes-delta kernel does not read); hi.hi-only gradient
            # products need the tight measured scale
            fused_bwd = bool(getattr(ws, "delta_from_planes", False)) and self.fused_dqkv and self._bwd_three_terms()
            if fused_bwd:
                self._before_overwrite(ws.dqkvp)
                part = self._part_buf("part_attn_pool")
                # (delta = dO . O from the planes of dO and of the attention output: no fp32 attention output in the step)
                L("eav_attn_bwd_sp_planes", P(ws.qkvrow[i]), None, P(ws.dorow), None, s_qkv, b_dao, b_ds,
                  None, None, P(ws.lse[i]), P(ws.delta), None, None, P(ws.dqkvp), b_dqkv, P(part), P(ws.aop[i]), s_ao,
                  ws.B, H, N, hd, scale, st)
                self._reduce_async(part, 0, ws.np_attn, 3 * D, 3 * D, gp(f"{Lk}.attention.q_proj.bias"))
            else:
                L("eav_attn_bwd_sp", P(ws.qkvrow[i]), None, P(ws.dorow), None, s_qkv, b_dao, b_ds,
                  P(ws.ao[i]), dao, P(ws.lse[i]), P(ws.delta), dqkv, b_dqkv, ws.B, H, N, hd, scale, st)
        else:
            qkv = P(ws.qkv[i])
            ldn = ws.ldn
            Pm, dP = P(ws.P[i]), P(ws.dP)
            sP, sQ, sO = (H * N * ldn, N * ldn), (N * 3 * D, hd), (N * D, hd)
            g = self._gemm_f32
            g(Pm, dao, dqkv + 8 * D, N, hd, N, ldn, D, 3 * D, tA=1, tB=1, batch=ws.B * H, heads=H, sA=sP, sB=sO, sC=sQ)
            g(dao, qkv + 8 * D, dP, N, N, hd, D, 3 * D, ldn, batch=ws.B * H, heads=H, sA=sO, sB=sQ, sC=sP)
            L("eav_softmax_bwd", Pm, dP, ws.B * H * N, N, ldn, st)
            g(dP, qkv + 4 * D, dqkv, N, hd, N, ldn, 3 * D, 3 * D, tB=1, batch=ws.B * H, heads=H, sA=sP, sB=sQ, sC=sQ,
              alpha=scale)
            g(dP, qkv, dqkv + 4 * D, N, hd, N, ldn, 3 * D, 3 * D, tA=1, tB=1, batch=ws.B * H, heads=H, sA=sP, sB=sQ,
              sC=sQ, alpha=scale)
        # fused q/k/v projection
        if not ws.fused:
            self._call("eav_sp_absmax", dqkv, M, 3 * D, 3 * D, b_dqkv, st)
        if not (ws.fused and fused_bwd):
            self._to_planes_bias(dqkv, M, 3 * D, b_dqkv, ws.dqkvp, gp(f"{Lk}.attention.q_proj.bias"))
        self._wgrad_sp(ws.dqkvp, b_dqkv, ws.y1p[i], s_y1, gp(f"{Lk}.attention.q_proj.weight"), 3 * D, D, M)
        wpl, wsl = self._wp(f"qkv{i}", transposed=True)
        # the gradient w.r.t. this layer's input is the next (lower) layer's dh: leave its max in that layer's slot
        if fdh and i > 0:
            self._gemm_sp(P(ws.dqkvp), b_dqkv, wpl, wsl, dy, M, D, 3 * D, D, amax=b_dy1, blockmax=False)
            below = Lk.rsplit(".", 1)[0] + f".{i - 1}"
            self._ln_bwd_planes(dy, P(ws.hs[i]), w(f"{Lk}.layernorm_before.weight"), stp, stp + 4 * M, dh,
                                bslot(1 + self.BS * (i - 1)), b_dh1, b_dy1, ws.dhp, gp(f"{Lk}.layernorm_before.weight"),
                                gp(f"{Lk}.layernorm_before.bias"), gp(f"{below}.mlp.fc2.bias"), s_y1)
        else:
            self._gemm_sp(P(ws.dqkvp), b_dqkv, wpl, wsl, dy, M, D, 3 * D, D)
            part = self._part_buf("part_ln_pool")
            L("eav_layernorm_bwd_amax", dy, P(ws.hs[i]), w(f"{Lk}.layernorm_before.weight"), stp, stp + 4 * M, dh, 1,
              P(part), M, D, bslot(1 + self.BS * (i - 1)) if i > 0 else bslot(0), st)
            self._reduce_ln(part, gp(f"{Lk}.layernorm_before.weight"), gp(f"{Lk}.layernorm_before.bias"))

    def _ln_bwd_planes(self, dy, x, gamma, mean, rstd, dh, slot_out, slot_old, slot_dy, planes, g_gamma, g_beta, g_bias,
                       slot_fwd):
        """LayerNorm backward, accumulated into dh, whose result ALSO leaves as the operand planes of the next gradient
        products (scale: the bound of eav_layernorm_bwd_bound - measured max|dh| before, max|dy|, max|gamma|, the forward's max
        rstd - formed inside the launch), with the bias-gradient partials of the linear layer that consumes dh: no conversion pass."""
        ws, D, M = self._ws, self.cfg.hidden, self._ws.M
        L, P = self._call, _lib.ptr
        self._before_overwrite(planes)
        part = self._part_buf("part_ln3_pool")
        L("eav_layernorm_bwd_planes", dy, x, gamma, mean, rstd, dh, 1, P(part), M, D, slot_out, P(planes), slot_old, slot_dy,
          slot_fwd, self._st)
        if g_beta == g_gamma + 4 * D:
            self._reduce_async(part, 0, ws.np_ln, 3 * D, 2 * D, g_gamma)
        else:
            self._reduce_async(part, 0, ws.np_ln, 3 * D, D, g_gamma)
            self._reduce_async(part, 4 * D, ws.np_ln, 3 * D, D, g_beta)
        self._reduce_async(part, 8 * D, ws.np_ln, 3 * D, D, g_bias)

    def _wgrad(self, A, B, C, M, N, K, lda, ldb):
        """C[M,N] = A^T.B for A stored [K,M], B stored [K,N] (weight gradient: contraction over tokens)."""
        self._call(self._gemm_name() + "_splitk", A, B, C, _lib.ptr(self._ws.splitk), M, N, K, lda, ldb, 1, 1, self._st)

    def _reduce_ln(self, part, gw, gb):
        """LayerNorm weight / bias gradients from `part` ([np_ln][2 D]: dgamma | dbeta partials): one launch when the
        two gradients are neighbours in the flat buffer (they are: weight, then bias, D floats each)."""
        ws, D = self._ws, self.cfg.hidden
        if gb == gw + 4 * D:
            self._reduce_async(part, 0, ws.np_ln, 2 * D, 2 * D, gw)
        else:
            self._reduce(part, ws.np_ln, 2 * D, D, gw)
            self._call("eav_reduce_partials", _lib.ptr(part) + 4 * D, ws.np_ln, 2 * D, D, 1.0, gb, self._st)

    def _reduce(self, part, nparts, stride, n, out):
        self._call("eav_reduce_partials", _lib.ptr(part), nparts, stride, n, 1.0, out, self._st)

    def _bias_grad(self, dy_ptr, M, N, ld, out):
        ws = self._ws
        self._call("eav_colsum", dy_ptr, _lib.ptr(ws.part_cs), M, N, ld, self._st)
        self._call("eav_reduce_partials", _lib.ptr(ws.part_cs), ws.np_cs, N, N, 1.0, out, self._st)

    def _launch_backward(self, dlogits, token):
        if self._saved is None or self._saved[0] != token:
            raise _lib.EavError("Encoder.backward: activations were overwritten by a later forward")
        c = self.cfg
        P, L = _lib.ptr, self._call
        self._main = torch.cuda.current_stream()
        st = self._st = self._main.cuda_stream
        self._phase = "bwd"
        _, x, full, _ = self._saved
        ws = self._ws
        B, M, ldn = ws.B, ws.M, ws.ldn
        D, FF, N, H = c.hidden, c.ff, c.ntok, c.heads
        hd = D // H
        pm = self._pmap
        flat, gflat, offs = self._flat
        gp = lambda k: gflat.data_ptr() + 4 * offs[k][0]  # noqa: E731
        w = lambda k: P(pm[k])  # noqa: E731
        pre = c.prefix
        R = B * c.nextra
        # ---- head
        self._head_backward(dlogits, ws.pooled if c.kind == "ast" else ws.seqr, ws, B, full)
        if c.kind == "ast" and full:
            L("eav_pair_mean", P(ws.dseqr), P(ws.dpooled), B, D, 1, st)
        if full:
            sf = P(ws.stf)
            L("eav_layernorm_bwd", P(ws.dseqr), P(ws.rows), w(f"{pre}.layernorm.weight"), sf, sf + 4 * R,
              P(ws.drows), 0, P(ws.part_lnr), R, D, st)
            self._reduce(ws.part_lnr, ws.np_lnr, 2 * D, D, gp(f"{pre}.layernorm.weight"))
            L("eav_reduce_partials", P(ws.part_lnr) + 4 * D, ws.np_lnr, 2 * D, D, 1.0, gp(f"{pre}.layernorm.bias"), st)
            ws.dh.zero_()
            L("eav_token_rows", P(ws.dh), P(ws.drows), B, N, D, c.nextra, 1, st)
            scale = hd ** -0.5
            dh, dy, dao, dact, dqkv = P(ws.dh), P(ws.dy), P(ws.dao), P(ws.dact), P(ws.dqkv)
            sp = ws.sp
            if sp:
                ws.bslots.zero_()
                bslot = lambda n: ws.bslots.data_ptr() + 4 * self.SLOT * n  # noqa: E731
                fslot = lambda n: ws.fslots.data_ptr() + 4 * self.SLOT * n  # noqa: E731
                # dh of the top layer comes from the head (token-row scatter): one pass for its maximum; every other dh
                # gets its maximum from the LayerNorm backward that produces it
                self._call("eav_sp_absmax", P(ws.dh), M, D, D, bslot(1 + self.BS * (c.layers - 1)), st)
            for i in reversed(range(c.layers)):
                Lk = f"{pre}.layers.{i}"
                stp = P(ws.st[i])
                if sp:
                    self._layer_backward_split(i, Lk, stp, gp, bslot, fslot, scale)
                    if self.grad_ready_hook is not None:
                        self._join_wgrads()         # the layer's weight gradients must be complete before they travel
                        lo = offs[f"{Lk}.attention.q_proj.weight"][0]
                        hi = offs[f"{Lk}.mlp.fc2.bias"][0] + offs[f"{Lk}.mlp.fc2.bias"][1]
                        self.grad_ready_hook(lo, hi)
                    continue
                # fc2: h_out = h_mid + act.W2^T + b2
                self._wgrad(dh, P(ws.act[i]), gp(f"{Lk}.mlp.fc2.weight"), D, FF, M, D, FF)
                self._bias_grad(dh, M, D, D, gp(f"{Lk}.mlp.fc2.bias"))
                self._gemm(dh, w(f"{Lk}.mlp.fc2.weight"), dact, M, FF, D, D, FF, FF, tB=1)
                L("eav_gelu_bwd", dact, P(ws.pre[i]), M * FF, st)
                # fc1
                self._wgrad(dact, P(ws.y2[i]), gp(f"{Lk}.mlp.fc1.weight"), FF, D, M, FF, D)
                self._bias_grad(dact, M, FF, FF, gp(f"{Lk}.mlp.fc1.bias"))
                self._gemm(dact, w(f"{Lk}.mlp.fc1.weight"), dy, M, D, FF, FF, D, D, tB=1)
                # layernorm_after: dh (now gradient w.r.t. h_mid) += LN backward
                L("eav_layernorm_bwd", dy, P(ws.hmid[i]), w(f"{Lk}.layernorm_after.weight"), stp + 8 * M,
                  stp + 12 * M, dh, 1, P(ws.part_ln), M, D, st)
                self._reduce(ws.part_ln, ws.np_ln, 2 * D, D, gp(f"{Lk}.layernorm_after.weight"))
                L("eav_reduce_partials", P(ws.part_ln) + 4 * D, ws.np_ln, 2 * D, D, 1.0,
                  gp(f"{Lk}.layernorm_after.bias"), st)
                # o_proj
                self._wgrad(dh, P(ws.ao[i]), gp(f"{Lk}.attention.o_proj.weight"), D, D, M, D, D)
                self._bias_grad(dh, M, D, D, gp(f"{Lk}.attention.o_proj.bias"))
                self._gemm(dh, w(f"{Lk}.attention.o_proj.weight"), dao, M, D, D, D, D, D, tB=1)
                # attention core
                qkv = P(ws.qkv[i])
                if ws.fused:
                    L("eav_attn_bwd", qkv, P(ws.ao[i]), dao, P(ws.lse[i]), P(ws.delta), dqkv, B, H, N, hd, scale, st)
                else:     # materialised scores, batched over (image, head)
                    Pm, dP = P(ws.P[i]), P(ws.dP)
                    sP, sQ, sO = (H * N * ldn, N * ldn), (N * 3 * D, hd), (N * D, hd)
                    self._gemm(Pm, dao, dqkv + 8 * D, N, hd, N, ldn, D, 3 * D, tA=1, tB=1, batch=B * H, heads=H,
                               sA=sP, sB=sO, sC=sQ)                                            # dV = P^T dO
                    self._gemm(dao, qkv + 8 * D, dP, N, N, hd, D, 3 * D, ldn, batch=B * H, heads=H,
                               sA=sO, sB=sQ, sC=sP)                                            # dP = dO V^T
                    L("eav_softmax_bwd", Pm, dP, B * H * N, N, ldn, st)
                    self._gemm(dP, qkv + 4 * D, dqkv, N, hd, N, ldn, 3 * D, 3 * D, tB=1, batch=B * H, heads=H,
                               sA=sP, sB=sQ, sC=sQ, alpha=scale)                               # dQ = s dS K
                    self._gemm(dP, qkv, dqkv + 4 * D, N, hd, N, ldn, 3 * D, 3 * D, tA=1, tB=1, batch=B * H, heads=H,
                               sA=sP, sB=sQ, sC=sQ, alpha=scale)                               # dK = s dS^T Q
                # fused q/k/v projection
                self._wgrad(dqkv, P(ws.y1[i]), gp(f"{Lk}.attention.q_proj.weight"), 3 * D, D, M, 3 * D, D)
                self._bias_grad(dqkv, M, 3 * D, 3 * D, gp(f"{Lk}.attention.q_proj.bias"))
                self._gemm(dqkv, w(f"{Lk}.attention.q_proj.weight"), dy, M, D, 3 * D, 3 * D, D, D, tB=1)
                L("eav_layernorm_bwd", dy, P(ws.hs[i]), w(f"{Lk}.layernorm_before.weight"), stp, stp + 4 * M, dh, 1,
                  P(ws.part_ln), M, D, st)
                self._reduce(ws.part_ln, ws.np_ln, 2 * D, D, gp(f"{Lk}.layernorm_before.weight"))
                L("eav_reduce_partials", P(ws.part_ln) + 4 * D, ws.np_ln, 2 * D, D, 1.0,
                  gp(f"{Lk}.layernorm_before.bias"), st)
                if self.grad_ready_hook is not None:   # layer i's parameters are one contiguous slice
                    lo = offs[f"{Lk}.attention.q_proj.weight"][0]
                    hi = offs[f"{Lk}.mlp.fc2.bias"][0] + offs[f"{Lk}.mlp.fc2.bias"][1]
                    self.grad_ready_hook(lo, hi)
            # embeddings
            L("eav_embed_bwd", dh, gp(f"{pre}.embeddings.position_embeddings"), P(ws.demb), B, N, D, c.nextra, st)
            gpos = gflat[offs[f"{pre}.embeddings.position_embeddings"][0]:]
            gflat[offs[f"{pre}.embeddings.cls_token"][0]:][:D].copy_(gpos[:D])
            if c.kind == "ast":
                gflat[offs[f"{pre}.embeddings.distillation_token"][0]:][:D].copy_(gpos[D:2 * D])
            MP = B * c.npatch
            if sp:
                # demb (the patch rows of dh) gets its own maximum pass: the slot layer 0's LayerNorm backward filled is
                # indexed by the rows of dh (cls / distillation rows included), and the per-row-block maxima must be the
                # converted tensor's own
                s_demb = bslot(1 + self.BS * c.layers)
                self._to_planes(P(ws.demb), MP, D, D, s_demb, ws.dembp)
                self._wgrad_sp(ws.dembp, s_demb, ws.colp, fslot(0),
                               gp(f"{pre}.embeddings.patch_embeddings.projection.weight"), D, c.kp, MP)
            else:
                self._wgrad(P(ws.demb), P(ws.col), gp(f"{pre}.embeddings.patch_embeddings.projection.weight"), D, c.kp,
                            MP, D, c.kp)
            self._call("eav_colsum", P(ws.demb), P(ws.part_cs), MP, D, D, st)
            self._call("eav_reduce_partials", P(ws.part_cs), _lib.plain("eav_colsum_nparts", MP), D, D, 1.0,
                       gp(f"{pre}.embeddings.patch_embeddings.projection.bias"), st)
        self._join_wgrads()          # side-stream weight gradients complete before autograd / the optimiser see them
        out = []
        for k in self._names:
            p = pm[k]
            trained = p.requires_grad and (full or k.startswith("classifier."))
            out.append(gflat[offs[k][0]:offs[k][0] + offs[k][1]].view(p.shape) if trained else None)
        return out


def ASTForAudioClassification(cfg=None, weights=None):
    return Encoder(cfg or make_config("ast"), weights)


def ViTForImageClassification(cfg=None, weights=None):
    return Encoder(cfg or make_config("vit"), weights)
