"""Oracle: the canonical EEGNet of CNN_torch/CNN_EEG.py restated op by op (fp32, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Pinned against the imported reference
(CNN_torch/CNN_EEG.py, imports unmodified) by tests/golden/cnn_eeg_*.npz.

Reference semantics reproduced:
  * block1: Conv2d(1,F1,(1,K),'same') -> BN -> depthwise Conv2d(F1,D*F1,(Chans,1),groups=F1) -> BN -> ELU ->
    AvgPool(1,4) -> Dropout                                                   CNN_EEG.py:20-30
  * block2: depthwise Conv2d(C2,C2,(1,16),'same',groups=C2) -> pointwise Conv2d(C2,F2,1) -> BN -> ELU ->
    AvgPool(1,8) -> Dropout                                                   CNN_EEG.py:33-42
  * flatten -> Linear -> LOGITS (no softmax; CrossEntropyLoss on logits)      CNN_EEG.py:57-67,88
  * 'same' padding with an even kernel: (k-1)//2 left, the rest right; norm_rate is accepted and unused (:13)
  * Adam(lr), betas (0.9, 0.999), eps 1e-8                                    CNN_EEG.py:89
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .eegnet_oracle import _batchnorm, _same_pad, adam_step_

PARAM_NAMES = [
    "block1.0.weight", "block1.1.weight", "block1.1.bias", "block1.2.weight", "block1.3.weight", "block1.3.bias",
    "block2.0.weight", "block2.1.weight", "block2.2.weight", "block2.2.bias", "classifier.weight", "classifier.bias",
]
BUFFER_NAMES = [
    "block1.1.running_mean", "block1.1.running_var", "block1.3.running_mean", "block1.3.running_var",
    "block2.2.running_mean", "block2.2.running_var",
]


def forward(P, Bf, x, training, masks=None, drop_p=0.5):
    """x [B,Chans,Samples] or [B,1,Chans,Samples] -> logits [B,nb].  masks: optional pair of 0/1 keep masks
    ([B,C2,1,S//4], [B,F2,1,S//32]) for train-mode dropout; None = dropout disabled."""
    if x.dim() == 3:
        x = x.unsqueeze(1)                                                                   # :61-62
    w1 = P["block1.0.weight"]
    F1 = w1.shape[0]
    h = F.conv2d(_same_pad(x, w1.shape[-1]), w1)                                             # :22
    h = _batchnorm(h, P["block1.1.weight"], P["block1.1.bias"], Bf["block1.1.running_mean"],
                   Bf["block1.1.running_var"], training)                                     # :23
    h = F.conv2d(h, P["block1.2.weight"], groups=F1)                                         # :25
    h = _batchnorm(h, P["block1.3.weight"], P["block1.3.bias"], Bf["block1.3.running_mean"],
                   Bf["block1.3.running_var"], training)                                     # :26
    h = F.avg_pool2d(F.elu(h), (1, 4))                                                       # :27-28
    if training and masks is not None:
        h = h * masks[0] / (1.0 - drop_p)                                                    # :29
    wdw = P["block2.0.weight"]
    h = F.conv2d(_same_pad(h, wdw.shape[-1]), wdw, groups=wdw.shape[0])                      # :35
    h = F.conv2d(h, P["block2.1.weight"])                                                    # :37
    h = _batchnorm(h, P["block2.2.weight"], P["block2.2.bias"], Bf["block2.2.running_mean"],
                   Bf["block2.2.running_var"], training)                                     # :38
    h = F.avg_pool2d(F.elu(h), (1, 8))                                                       # :39-40
    if training and masks is not None:
        h = h * masks[1] / (1.0 - drop_p)                                                    # :41
    return F.linear(h.flatten(1), P["classifier.weight"], P["classifier.bias"])              # :66-67


class Stepper:
    """forward + CrossEntropyLoss(logits) + backward + Adam: the body of EEGNetTrainer.train_epoch (:95-108)."""

    def __init__(self, P, Bf, lr, drop_p=0.5):
        self.P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        self.Bf = {k: v.clone() for k, v in Bf.items()}
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.t = 0
        self.lr, self.drop_p = lr, drop_p

    def step(self, x, y, training=True, masks=None):
        for p in self.P.values():
            p.grad = None
        logits = forward(self.P, self.Bf, x, training, masks, self.drop_p)
        loss = F.cross_entropy(logits, y)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in self.P.items()}
        self.t += 1
        with torch.no_grad():
            for k, p in self.P.items():
                adam_step_(p, p.grad, self.m[k], self.v[k], self.t, self.lr)
        return logits.detach(), loss.detach(), grads
