"""Oracle: ShallowConvNet + 12-layer single-head transformer of Transformer_torch/Transformer_EEG.py, restated op by
op (fp32, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Pinned against the imported reference (Transformer_EEG.py imports
unmodified; only its trainer needs the `self` shim of SURVEY.md section 8f row 4) by tests/golden/shallow_tf_*.npz.

Reference semantics reproduced:
  * conv(1,40,(1,13),valid,bias=False) -> per-filter Linear(30,1,bias=False) over channels -> tokens [B,T,40]  :117,:28-35
  * 12 x { a = softmax(QK^T/sqrt(40)) V + V (one head, bias-free q/k/v, :42-76);  x = x + Drop(LN1(a));
           x = x + Drop(LN2(W2 Drop(ReLU(W1 x + b1)) + b2)) }                                                  :92-106
  * BatchNorm2d(40) on [B,40,1,T] -> square -> AvgPool(1,35)/7 -> log(clamp(1e-7,1e4)) -> Dropout -> flatten ->
    Linear(2600,nb,bias=False) -> softmax                                                                       :132-148
  * trainer: CrossEntropyLoss on the softmax output (double softmax), Adam(lr), and after every step
    fc.weight <- renorm(p=2, dim=0, maxnorm=0.5)                                                               :170-199
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .eegnet_oracle import _batchnorm, adam_step_, ce_on_probs, renorm_rows_

NF, KC, POOL, STRIDE, FF = 40, 13, 35, 7, 160


def param_names(num_layers=12):
    names = ["conv.weight", "bn.weight", "bn.bias"] + [f"embedding.value_proj.{i}.weight" for i in range(NF)]
    for l in range(num_layers):
        p = f"transformer.{l}."
        names += [p + "attn.W_q.weight", p + "attn.W_k.weight", p + "attn.W_v.weight", p + "ffn.net.0.weight",
                  p + "ffn.net.0.bias", p + "ffn.net.3.weight", p + "ffn.net.3.bias", p + "norm1.weight",
                  p + "norm1.bias", p + "norm2.weight", p + "norm2.bias"]
    return names + ["fc.weight"]


BUFFER_NAMES = ["bn.running_mean", "bn.running_var"]


def _drop(h, masks, drop_p):
    """nn.Dropout with an explicit 0/1 keep mask popped from `masks` (None: dropout disabled)."""
    if masks is None:
        return h
    return h * masks.pop(0).to(h.dtype).view(h.shape) / (1.0 - drop_p)


def forward(P, Bf, x, training, masks=None, drop_p=0.5, num_layers=12):
    """x [B,1,30,S] -> softmax probabilities [B,nb].  masks: list of keep masks in the reference's call order (per
    layer: after norm1 [B,T,40], inside the FFN [B,T,160], after norm2 [B,T,40]; then the head [B,40,65]); consumed."""
    masks = list(masks) if (training and masks is not None) else None
    h = F.conv2d(x, P["conv.weight"])                                                                 # :117
    v = torch.cat([F.linear(h[:, i].permute(0, 2, 1), P[f"embedding.value_proj.{i}.weight"]) for i in range(NF)],
                  dim=-1)                                                                             # :28-35
    for l in range(num_layers):
        p = f"transformer.{l}."
        q, k, val = (F.linear(v, P[p + f"attn.W_{n}.weight"]) for n in "qkv")                          # :62-64
        a = torch.softmax(q @ k.transpose(-1, -2) / (NF ** 0.5), dim=-1) @ val + val                   # :66-76
        v = v + _drop(F.layer_norm(a, (NF,), P[p + "norm1.weight"], P[p + "norm1.bias"]), masks, drop_p)   # :103
        f = F.relu(F.linear(v, P[p + "ffn.net.0.weight"], P[p + "ffn.net.0.bias"]))                    # :83-84
        f = F.linear(_drop(f, masks, drop_p), P[p + "ffn.net.3.weight"], P[p + "ffn.net.3.bias"])     # :85-86
        v = v + _drop(F.layer_norm(f, (NF,), P[p + "norm2.weight"], P[p + "norm2.bias"]), masks, drop_p)   # :104
    h = v.permute(0, 2, 1).unsqueeze(2)                                                               # :135
    h = _batchnorm(h, P["bn.weight"], P["bn.bias"], Bf["bn.running_mean"], Bf["bn.running_var"], training)   # :136
    h = F.avg_pool2d(torch.square(h), (1, POOL), stride=(1, STRIDE))                                  # :138-139
    h = torch.log(torch.clamp(h, 1e-7, 1e4)).squeeze(2)                                               # :140-142
    h = _drop(h, masks, drop_p).flatten(1)                                                            # :143-144
    return torch.softmax(F.linear(h, P["fc.weight"]), dim=1)                                          # :146


class Stepper:
    """forward + CE(probs) + backward + Adam + fc max-norm: the loop body of TrainerUni.train (:186-199)."""

    def __init__(self, P, Bf, lr, drop_p=0.5, num_layers=12):
        self.P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        self.Bf = {k: v.clone() for k, v in Bf.items()}
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.t = 0
        self.lr, self.drop_p, self.num_layers = lr, drop_p, num_layers

    def step(self, x, y, training=True, masks=None):
        for p in self.P.values():
            p.grad = None
        probs = forward(self.P, self.Bf, x, training, masks, self.drop_p, self.num_layers)
        loss = ce_on_probs(probs, y)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in self.P.items()}
        self.t += 1
        with torch.no_grad():
            for k, p in self.P.items():
                adam_step_(p, p.grad, self.m[k], self.v[k], self.t, self.lr)
            renorm_rows_(self.P["fc.weight"].data, 0.5)                                               # :196-199
        return probs.detach(), loss.detach(), grads
