"""Oracle: EEGNet_tor forward / train step restated op by op (fp32, CPU).

TEST INFRASTRUCTURE - see oracle/__init__.py.  Pinned against the imported,
shimmed reference (CNN_torch/EEGNet_tor.py) by tests/golden/eegnet_*.npz.

Reference semantics reproduced:
  * forward chain                       EEGNet_tor.py:50-67
  * 'same' padding K=300 -> 149 left / 150 right; K=16 -> 7 / 8   (:24,:37)
  * max-norm renorm of depthwiseConv / dense weights AFTER the forward and
    BEFORE the backward (intended meaning of the hooks at :33-34,:47-48; Q1,Q2)
  * the model returns softmax probabilities and the trainer feeds them to
    CrossEntropyLoss, i.e. a double softmax (:44,:66,:81,:105; Q3)
  * Adam(lr), betas (0.9,0.999), eps 1e-8, no weight decay (:82)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

PARAM_NAMES = [
    "firstConv.weight", "firstBN.weight", "firstBN.bias",
    "depthwiseConv.weight", "depthwiseBN.weight", "depthwiseBN.bias",
    "separableConv.weight", "separableBN.weight", "separableBN.bias",
    "dense.weight", "dense.bias",
]
BUFFER_NAMES = [
    "firstBN.running_mean", "firstBN.running_var",
    "depthwiseBN.running_mean", "depthwiseBN.running_var",
    "separableBN.running_mean", "separableBN.running_var",
]


def _batchnorm(x, w, b, rm, rv, training, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d (EEGNet_tor.py:25,29,38): biased variance normalises,
    unbiased variance feeds the running estimate, momentum 0.1, eps 1e-5."""
    if training:
        dims = (0, 2, 3)
        n = x.numel() // x.shape[1]
        mean = x.mean(dims)
        var = ((x - mean[None, :, None, None]) ** 2).mean(dims)
        with torch.no_grad():
            rm.mul_(1 - momentum).add_(momentum * mean)
            rv.mul_(1 - momentum).add_(momentum * var * (n / max(n - 1, 1)))
    else:
        mean, var = rm, rv
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    return xh * w[None, :, None, None] + b[None, :, None, None]


def _same_pad(x, k):
    """torch padding='same' for an even kernel: (k-1)//2 left, k-1-(k-1)//2 right."""
    left = (k - 1) // 2
    return F.pad(x, (left, k - 1 - left))


def renorm_rows_(w, maxnorm):
    """Tensor.renorm_(p=2, dim=0, maxnorm): rows with L2 norm > maxnorm are
    scaled by maxnorm / (norm + 1e-7)  (EEGNet_tor.py:34,48)."""
    flat = w.view(w.shape[0], -1)
    norms = flat.norm(2, dim=1)
    scale = torch.where(norms > maxnorm, maxnorm / (norms + 1e-7), torch.ones_like(norms))
    flat.mul_(scale[:, None])
    return w


def forward(P, Bf, x, training, masks=None, drop_p=0.5, norm_rate=1.0, apply_renorm=True):
    """x [B,1,C,S] -> probs [B,nb].  P / Bf: dicts keyed as PARAM_NAMES /
    BUFFER_NAMES.  masks: optional (m1 [B,64,1,S//4], m2 [B,F2,1,S//32]) 0/1
    keep masks (train-mode dropout; None = dropout disabled)."""
    if x.dim() == 3:
        x = x.unsqueeze(1)
    w1 = P["firstConv.weight"]
    F1 = w1.shape[0]
    h = F.conv2d(_same_pad(x, w1.shape[-1]), w1)                              # :51
    h = _batchnorm(h, P["firstBN.weight"], P["firstBN.bias"],
                   Bf["firstBN.running_mean"], Bf["firstBN.running_var"], training)   # :52
    h = F.elu(h)                                                              # :53
    h = F.conv2d(h, P["depthwiseConv.weight"], groups=F1)                     # :54
    h = _batchnorm(h, P["depthwiseBN.weight"], P["depthwiseBN.bias"],
                   Bf["depthwiseBN.running_mean"], Bf["depthwiseBN.running_var"], training)  # :55
    h = F.elu(h)                                                              # :56
    h = F.avg_pool2d(h, (1, 4))                                               # :57
    if training and masks is not None:
        h = h * masks[0] / (1.0 - drop_p)                                     # :58
    w3 = P["separableConv.weight"]
    h = F.conv2d(_same_pad(h, w3.shape[-1]), w3)                              # :59
    h = _batchnorm(h, P["separableBN.weight"], P["separableBN.bias"],
                   Bf["separableBN.running_mean"], Bf["separableBN.running_var"], training)  # :60
    h = F.elu(h)                                                              # :61
    h = F.avg_pool2d(h, (1, 8))                                               # :62
    if training and masks is not None:
        h = h * masks[1] / (1.0 - drop_p)                                     # :63
    h = h.flatten(1)                                                          # :64
    logits = F.linear(h, P["dense.weight"], P["dense.bias"])                  # :65
    probs = torch.softmax(logits, dim=1)                                      # :66
    if apply_renorm:                                                          # hooks :33-34,:47-48
        with torch.no_grad():
            renorm_rows_(P["depthwiseConv.weight"].data, norm_rate)
            renorm_rows_(P["dense.weight"].data, norm_rate)
    return probs


def ce_on_probs(probs, y):
    """nn.CrossEntropyLoss applied to the model's softmax output (:81,:105)."""
    return F.nll_loss(torch.log_softmax(probs, dim=1), y)


def adam_step_(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, decoupled=False):
    """torch.optim.Adam / AdamW single-tensor update (step = count AFTER increment)."""
    if decoupled and weight_decay != 0.0:
        p.mul_(1 - lr * weight_decay)
    elif weight_decay != 0.0:
        g = g + weight_decay * p
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


class Stepper:
    """fwd + CE(probs) + bwd + Adam, the body of Trainer_uni.train (:104-110)."""

    def __init__(self, P, Bf, lr, drop_p=0.5, norm_rate=1.0):
        self.P = {k: v.clone().requires_grad_(True) for k, v in P.items()}
        self.Bf = {k: v.clone() for k, v in Bf.items()}
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.t = 0
        self.lr, self.drop_p, self.norm_rate = lr, drop_p, norm_rate

    def step(self, x, y, training=True, masks=None):
        for p in self.P.values():
            p.grad = None
        probs = forward(self.P, self.Bf, x, training, masks, self.drop_p, self.norm_rate)
        loss = ce_on_probs(probs, y)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in self.P.items()}
        self.t += 1
        with torch.no_grad():
            for k, p in self.P.items():
                adam_step_(p, p.grad, self.m[k], self.v[k], self.t, self.lr)
        return probs.detach(), loss.detach(), grads
