"""Shared two-phase fine-tuning machinery of the audio and vision trainers.

Both reference trainers (Transformer_Audio.py:44-103, Transformer_Vision.py:61-129) run the same schedule: set
the learning rate of ONE AdamW that spans both phases (Q10/Q11), freeze everything but `model.classifier` or
unfreeze all, loop epochs of (train batches -> eval batches), and keep the test logits of the last unfrozen
epoch in `outputs_test` (Q15).  They differ only in bookkeeping (how accuracy is averaged, what is printed and
logged).  This module holds the common part on top of eav_amd.transformer.Encoder; the two public classes in
audio.py / vision.py keep the reference's constructors, method signatures and console output.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .eegnet import DeviceLoader
from .optim import CrossEntropyLoss, FusedAdam
from .transformer import Encoder


def require_gpu(who):
    dev = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if dev.type != "cuda":
        raise _lib.EavError(f"eav_amd.{who} needs an MI355X (no CPU fallback)")
    return dev


class FineTuneBase:
    """Owns model, optimiser, loaders; subclasses provide the reference-specific reporting."""

    def _build(self, model_path, n_classes, lr, device):
        self.device = device
        self.model = Encoder.from_pretrained(model_path)
        in_features = self.model.cfg.hidden
        fresh = torch.nn.Linear(in_features, n_classes)       # torch-default init, drawn from the torch RNG
        self.model.reset_head(fresh.weight.detach(), fresh.bias.detach())
        self.model.to(device)
        self.initial_lr = lr
        # AdamW with torch's default weight decay 0.01: the reference never forwards its own argument (Q10)
        self.optimizer = FusedAdam(self.model.parameters(), lr=lr, weight_decay=0.01, decoupled=True)
        self.loss_fn = CrossEntropyLoss()
        self.grad_sync = None

    def _loader(self, x, y, shuffle):
        return DeviceLoader(x, y, self.batch_size, shuffle, self.device)

    def _enter_phase(self, lr, freeze):
        lr = self.initial_lr if lr is None else lr
        for group in self.optimizer.param_groups:
            group['lr'] = lr
        trainable_head = {id(p) for p in self.model.classifier.parameters()}
        for p in self.model.parameters():
            p.requires_grad = (not freeze) or (id(p) in trainable_head)
        if self.grad_sync is not None:       # frozen phase: only the head's gradients cross the xGMI links
            self.grad_sync.set_active(self.model.head_grad_ranges() if freeze else None)
        return lr

    def _train_one_epoch(self, after_batch=None):
        """Returns (#correct on device, #seen).  One optimiser step per batch; nothing is read back per step."""
        self.model.train()
        correct = torch.zeros((), dtype=torch.long, device=self.device)
        seen, nb = 0, len(self.train_dataloader)
        for k, (xb, tb) in enumerate(self.train_dataloader, start=1):
            self.optimizer.zero_grad()
            logits = self.model(xb).logits
            self.loss_fn(logits, tb).backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.optimizer.step()
            correct += (logits.argmax(dim=-1) == tb).sum()
            seen += tb.size(0)
            if after_batch is not None:
                after_batch(k, nb)
        self.loss_fn.check()            # labels outside [0, classes) seen by any step of this epoch raise here
        return correct, seen

    def _evaluate(self):
        """Test pass: list of (logits numpy [b, classes], #correct, b) per batch."""
        self.model.eval()
        rows = []
        with torch.no_grad():
            for xb, tb in self.test_dataloader:
                logits = self.model(xb).logits
                rows.append((logits.detach().cpu().numpy(), int((logits.argmax(dim=-1) == tb).sum().item()), tb.size(0)))
        return rows

    def _keep_outputs(self, rows, is_last_epoch, freeze):
        if is_last_epoch and not freeze:
            self.outputs_test = np.concatenate([r[0] for r in rows], axis=0)
