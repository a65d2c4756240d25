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
        self._begin_phase_cache(freeze)
        return lr

    # Frozen-phase feature cache.  With `freeze=True` the backbone and its inputs are constant and every dropout of the
    # reference checkpoints is 0.0 (Pre_trained_models/ast-finetuned-audioset/config.json: hidden_dropout_prob 0.0,
    # attention_probs_dropout_prob 0.0; ViTConfig defaults the same), so the classifier's input of a given sample is the
    # same in every frozen epoch (Transformer_Audio.py:44-56, Transformer_Vision.py:61-77 re-run the whole backbone 10
    # times per subject).  The first frozen epoch of a train() call runs the backbone once per sample and keeps those
    # [N, hidden] features in HBM; the following epochs run only the head's forward / CE / backward / AdamW and the head on
    # the cached test features.  Same kernels on the same values: in "fp32" precision outputs, head weights and optimiser
    # state are bit-equal to the uncached run; in "split" precision the patch planes' scale depends on which samples share
    # a batch, so the two runs agree to rounding (~1e-6) instead.  `cache_frozen_features = False` restores the literal
    # schedule.  A trainer-level saving: it never enters bench.py's step metric.
    cache_frozen_features = True

    def _begin_phase_cache(self, freeze):
        self._feat_cache = None
        if freeze and self.cache_frozen_features and self.grad_sync is None:
            hid = self.model.cfg.hidden
            self._feat_cache = {"train": torch.empty(len(self.train_dataloader.dataset), hid, device=self.device),
                                "test": torch.empty(len(self.test_dataloader.dataset), hid, device=self.device),
                                "have_train": False, "have_test": False}

    def _train_one_epoch(self, after_batch=None):
        """Returns (#correct on device, #seen).  One optimiser step per batch; nothing is read back per step."""
        self.model.train()
        correct = torch.zeros((), dtype=torch.long, device=self.device)
        dl = self.train_dataloader
        seen, nb = 0, len(dl)
        fc = getattr(self, "_feat_cache", None)
        for k, idx in enumerate(dl.index_batches(), start=1):
            self.optimizer.zero_grad()
            if fc is not None and fc["have_train"]:       # cached features: only the labels of the batch are gathered
                tb = dl.gather_labels(idx)
                logits = self.model.head(fc["train"][self._index(idx)]).logits
            else:
                xb, tb = dl.gather(idx)
                logits = self.model(xb).logits
                if fc is not None:
                    fc["train"][self._index(idx)] = self.model.last_features()
            self.loss_fn(logits, tb).backward()
            if self.grad_sync is not None:
                self.grad_sync()
            self.optimizer.step()
            correct += (logits.argmax(dim=-1) == tb).sum()
            seen += tb.size(0)
            if after_batch is not None:
                after_batch(k, nb)
        if fc is not None:
            fc["have_train"] = True
        self.loss_fn.check()            # labels outside [0, classes) seen by any step of this epoch raise here
        return correct, seen

    def _index(self, idx):
        return torch.as_tensor(idx, dtype=torch.long, device=self.device)

    def _evaluate(self):
        """Test pass: list of (logits numpy [b, classes], #correct, b) per batch.  Logits and per-batch hit counts stay on
        the device until the pass is over - ONE device-to-host copy per epoch (the reference synchronises twice per
        batch: Transformer_Audio.py:91-96, Transformer_Vision.py:111-116)."""
        self.model.eval()
        dl = self.test_dataloader
        n, nb = len(dl.dataset), len(dl)
        fc = getattr(self, "_feat_cache", None)
        all_logits = torch.empty(n, self.model.cfg.num_labels, device=self.device)
        hits = torch.zeros(nb, dtype=torch.long, device=self.device)
        spans, pos = [], 0
        with torch.no_grad():
            for k, idx in enumerate(dl.index_batches()):
                if fc is not None and fc["have_test"]:
                    tb = dl.gather_labels(idx)
                    logits = self.model.head(fc["test"][pos:pos + len(idx)]).logits
                else:
                    xb, tb = dl.gather(idx)
                    logits = self.model(xb).logits
                    if fc is not None:
                        fc["test"][pos:pos + len(idx)] = self.model.last_features()
                all_logits[pos:pos + len(idx)] = logits
                hits[k] = (logits.argmax(dim=-1) == tb).sum()
                spans.append((pos, len(idx)))
                pos += len(idx)
        if fc is not None:
            fc["have_test"] = True
        host_logits, host_hits = all_logits.cpu().numpy(), hits.cpu().tolist()      # the epoch's only read-back
        return [(host_logits[a:a + b], int(h), b) for (a, b), h in zip(spans, host_hits)]

    def _keep_outputs(self, rows, is_last_epoch, freeze):
        if is_last_epoch and not freeze:
            self.outputs_test = np.concatenate([r[0] for r in rows], axis=0)
