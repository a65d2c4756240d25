"""AudioModelTrainer on MI355X - the reference's class, same constructor and train() signature.

Mirrors Transformer_torch/Transformer_Audio.py:9-103 line for line at the API level:
    AudioModelTrainer(DATA, model_path, sub='', num_classes=5, weight_decay=1e-5, lr=0.001, batch_size=128)
        .train(epochs=20, lr=None, freeze=True)      attribute: outputs_test  (float32 [N_test, 5])
The model behind it is eav_amd.transformer.Encoder (HIP kernels) instead of the Hugging Face
ASTForAudioClassification; the log-mel front-end stays the reference's own call of the HF
ASTFeatureExtractor on the host ("next" row 1).  Reference quirks kept: the
weight_decay argument is ignored and AdamW's default 0.01 applies (Q10); one optimiser spans the
frozen and unfrozen phases (Q11); outputs_test is set only on the last unfrozen epoch (Q15); one line
per epoch is appended to training_performance_audio.txt in the cwd (Q17).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .eegnet import DeviceLoader
from .optim import CrossEntropyLoss, FusedAdam
from .transformer import Encoder


class AudioModelTrainer:
    def __init__(self, DATA, model_path, sub='', num_classes=5, weight_decay=1e-5, lr=0.001, batch_size=128):
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.device.type != "cuda":
            raise _lib.EavError("eav_amd.AudioModelTrainer needs an MI355X (no CPU fallback)")
        self.tr, self.tr_y, self.te, self.te_y = DATA
        self.tr_x = self._feature_extract(self.tr)
        self.te_x = self._feature_extract(self.te)

        self.sub = sub
        self.batch_size = batch_size

        self.train_dataloader = self._prepare_dataloader(self.tr_x, self.tr_y, shuffle=True)
        self.test_dataloader = self._prepare_dataloader(self.te_x, self.te_y, shuffle=False)

        self.model = Encoder.from_pretrained(model_path)                      # :22
        # :24 - fresh torch-default-initialised head (draws from the torch RNG like the reference)
        fresh = torch.nn.Linear(self.model.classifier.dense.weight.shape[1], num_classes)
        self.model.reset_head(fresh.weight.detach(), fresh.bias.detach())
        self.model = self.model.to(self.device)

        self.initial_lr = lr
        self.optimizer = FusedAdam(self.model.parameters(), lr=self.initial_lr, weight_decay=0.01, decoupled=True)  # :30
        self.loss_fn = CrossEntropyLoss()
        self.grad_sync = None

    def _prepare_dataloader(self, x, y, shuffle=False):
        return DeviceLoader(x, y, self.batch_size, shuffle, self.device)

    def _feature_extract(self, x):
        """Reference (:38-42): ASTFeatureExtractor()(x, sampling_rate=16000, padding='max_length') on the host.
        Equal-length clips go through the HIP log-mel kernel (same recipe, float64 up to the log)."""
        if isinstance(x, torch.Tensor) and x.dim() == 3:
            return x                                    # already [N,1024,128] input_values
        arr = np.asarray(x)
        if arr.ndim == 2 and arr.dtype != object and arr.shape[1] >= 400:
            from .preprocess import waveforms_to_input_values
            return waveforms_to_input_values(arr, device=self.device).cpu()
        from transformers import ASTFeatureExtractor   # ragged input: the reference's own host call
        feature_extractor = ASTFeatureExtractor()
        ft = feature_extractor(x, sampling_rate=16000, padding='max_length', return_tensors='pt')
        return ft['input_values']

    def train(self, epochs=20, lr=None, freeze=True):
        lr = lr if lr is not None else self.initial_lr
        if lr is not None:
            for param_group in self.optimizer.param_groups:
                param_group['lr'] = lr
        for param in self.model.parameters():
            param.requires_grad = not freeze
        for param in self.model.classifier.parameters():
            param.requires_grad = True
        if self.grad_sync is not None:     # frozen phase: only the head's gradients cross the xGMI links
            self.grad_sync.set_active(self.model.head_grad_ranges() if freeze else None)

        for epoch in range(epochs):
            self.model.train()
            correct_dev = torch.zeros((), dtype=torch.long, device=self.device)
            train_total = 0
            for batch_idx, (x, t) in enumerate(self.train_dataloader, start=1):
                self.optimizer.zero_grad()
                logits = self.model(x).logits
                loss = self.loss_fn(logits, t)
                loss.backward()
                if self.grad_sync is not None:
                    self.grad_sync()
                self.optimizer.step()
                correct_dev += (logits.argmax(dim=-1) == t).sum()       # read once per epoch (no per-step sync)
                train_total += t.size(0)
            train_accuracy = int(correct_dev.item()) / train_total

            self.model.eval()
            correct, total = 0, 0
            outputs_batch = []
            with torch.no_grad():
                for x, t in self.test_dataloader:
                    logits = self.model(x).logits
                    correct += (logits.argmax(dim=-1) == t).sum().item()
                    total += t.size(0)
                    outputs_batch.append(logits.detach().cpu().numpy())
                test_accuracy = correct / total
            if epoch == epochs - 1 and not freeze:
                self.outputs_test = np.concatenate(outputs_batch, axis=0)

            print(f"Epoch {epoch + 1}/{epochs}, Training Accuracy: {train_accuracy * 100:.2f}%, Test Accuracy: {test_accuracy * 100:.2f}%")
            with open('training_performance_audio.txt', 'a') as f:
                f.write(f"{self.sub}, Epoch {epoch + 1}, Test Accuracy: {test_accuracy * 100:.2f}%\n")
