"""AudioModelTrainer on MI355X - the reference's class, same constructor and train() signature.

API of Transformer_torch/Transformer_Audio.py:9-103:
    AudioModelTrainer(DATA, model_path, sub='', num_classes=5, weight_decay=1e-5, lr=0.001, batch_size=128)
        .train(epochs=20, lr=None, freeze=True)      attribute: outputs_test  (float32 [N_test, 5])
The model is eav_amd.transformer.Encoder (HIP kernels) instead of the Hugging Face ASTForAudioClassification;
equal-length clips go through the HIP log-mel front-end, ragged ones through the reference's own host call of
ASTFeatureExtractor.  Kept quirks: `weight_decay` is accepted and ignored (Q10); one optimiser spans both phases
(Q11); outputs_test only after the last unfrozen epoch (Q15); one line per epoch appended to
training_performance_audio.txt in the cwd (Q17).
"""
from __future__ import annotations

import numpy as np
import torch

from .finetune import FineTuneBase, require_gpu


class AudioModelTrainer(FineTuneBase):
    def __init__(self, DATA, model_path, sub='', num_classes=5, weight_decay=1e-5, lr=0.001, batch_size=128):
        device = require_gpu("AudioModelTrainer")
        self.device = device
        self.tr, self.tr_y, self.te, self.te_y = DATA
        self.tr_x, self.te_x = self._feature_extract(self.tr), self._feature_extract(self.te)
        self.sub, self.batch_size = sub, batch_size
        self.train_dataloader = self._prepare_dataloader(self.tr_x, self.tr_y, shuffle=True)
        self.test_dataloader = self._prepare_dataloader(self.te_x, self.te_y, shuffle=False)
        self._build(model_path, num_classes, lr, device)          # :22-31

    def _prepare_dataloader(self, x, y, shuffle=False):
        return self._loader(x, y, shuffle)

    def _feature_extract(self, x):
        """:38-42 - log-mel features [N, 1024, 128]."""
        if isinstance(x, torch.Tensor) and x.dim() == 3:
            return x
        arr = np.asarray(x)
        if arr.ndim == 2 and arr.dtype != object and arr.shape[1] >= 400:
            from .preprocess import waveforms_to_input_values
            return waveforms_to_input_values(arr, device=self.device).cpu()
        from transformers import ASTFeatureExtractor
        return ASTFeatureExtractor()(x, sampling_rate=16000, padding='max_length', return_tensors='pt')['input_values']

    def train(self, epochs=20, lr=None, freeze=True):
        self._enter_phase(lr, freeze)
        for epoch in range(epochs):
            correct, seen = self._train_one_epoch()
            train_accuracy = int(correct.item()) / seen
            rows = self._evaluate()
            test_accuracy = sum(r[1] for r in rows) / sum(r[2] for r in rows)      # sample-weighted (:92-97)
            self._keep_outputs(rows, epoch == epochs - 1, freeze)
            print(f"Epoch {epoch + 1}/{epochs}, Training Accuracy: {train_accuracy * 100:.2f}%, Test Accuracy: {test_accuracy * 100:.2f}%")
            with open('training_performance_audio.txt', 'a') as f:
                f.write(f"{self.sub}, Epoch {epoch + 1}, Test Accuracy: {test_accuracy * 100:.2f}%\n")
