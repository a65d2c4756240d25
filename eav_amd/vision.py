"""ImageClassifierTrainer on MI355X - the reference's class, same constructor and train() signature.

API of Transformer_torch/Transformer_Vision.py:8-129:
    ImageClassifierTrainer(DATA, model_path, sub='', num_labels=5, lr=5e-5, batch_size=128)
        .train(epochs=3, lr=None, freeze=True, log=False)     attribute: outputs_test
    calculate_accuracy(outputs, labels)
plus the trial-level vote of the driver block (:174-185) as `trial_vote`.  The ViT is
eav_amd.transformer.Encoder; uniform uint8 frames are pre-processed by one HIP kernel (bit-identical to the HF
processor), anything else by the reference's host route; processed frames live in HBM (Q13).  Kept quirks:
test accuracy is the mean of per-batch accuracies (Q14), outputs_test only after the last unfrozen epoch (Q15).
"""
from __future__ import annotations

import numpy as np
import torch

from .finetune import FineTuneBase, require_gpu


def calculate_accuracy(outputs, labels):
    _, predicted = torch.max(outputs.logits, 1)
    return (predicted == labels).sum().item() / labels.size(0)


def trial_vote(outputs_test, labels, frames_per_trial=25):
    """mean of the per-frame logits over each trial -> argmax; accuracy and weighted F1 (:174-185)."""
    from sklearn.metrics import f1_score
    per_trial = np.reshape(outputs_test, (-1, frames_per_trial, outputs_test.shape[-1]), 'C').mean(1)
    pred = np.argmax(per_trial, axis=1)
    return pred, float(np.mean(pred == labels)), float(f1_score(labels, pred, average='weighted'))


def _load_processor(model_path):
    try:
        from transformers import AutoImageProcessor
        return AutoImageProcessor.from_pretrained(model_path)
    except Exception:   # images without torchvision: the PIL implementation of the same processor (shim S5)
        from transformers.models.vit.image_processing_pil_vit import ViTImageProcessorPil
        return ViTImageProcessorPil.from_pretrained(model_path)


class ImageClassifierTrainer(FineTuneBase):
    def __init__(self, DATA, model_path, sub='', num_labels=5, lr=5e-5, batch_size=128):
        device = require_gpu("ImageClassifierTrainer")
        self.tr_x, self.tr_y, self.te_x, self.te_y = DATA
        self.model_path, self.num_labels, self.batch_size, self.sub = model_path, num_labels, batch_size, sub
        self.frame_per_sample = np.shape(self.tr_x)[1]
        self.test_prediction = list()
        self.processor = _load_processor(model_path)                          # :28
        self._build(model_path, num_labels, lr, device)                       # :29-36
        self.model.num_labels = num_labels                                    # :31
        print("Image preprocessing..")
        self.train_dataloader = self._prepare_dataloader(self.tr_x, self.tr_y, shuffle=True)[0]
        self.test_dataloader = self._prepare_dataloader(self.te_x, self.te_y, shuffle=False)[0]
        print("Ended..")

    def _prepare_dataloader(self, x, y, shuffle=True):
        processed_x = self.preprocess_images(x)
        y_repeated = torch.from_numpy(np.repeat(y, self.frame_per_sample)).long()
        c, hw = self.model.cfg.C, self.model.cfg.H
        return self._loader(processed_x.view(-1, c, hw, hw), y_repeated, shuffle), processed_x, y_repeated

    def preprocess_images(self, image_list):
        """:52-59 - per-frame processor + one stack().to(device) in the reference."""
        p = self.processor
        arr = image_list if isinstance(image_list, np.ndarray) else np.asarray(image_list)
        flags = [getattr(p, k, False) for k in ("do_resize", "do_rescale", "do_normalize")]
        plain = all(flags) and not getattr(p, "do_center_crop", False) and int(getattr(p, "resample", 2)) == 2
        if plain and getattr(p, "size", None) is not None and arr.dtype == np.uint8 and arr.ndim == 5 and arr.shape[-1] == 3:
            from .preprocess import frames_to_pixel_values
            sz = p.size
            hw = (int(sz["height"]), int(sz["width"])) if isinstance(sz, dict) else (int(sz.height), int(sz.width))
            return frames_to_pixel_values(arr.reshape(-1, *arr.shape[2:]), hw, p.image_mean, p.image_std,
                                          p.rescale_factor, self.device)
        frames = [self.processor(images=img, return_tensors="pt").pixel_values.squeeze() for s in image_list for img in s]
        return torch.stack(frames).to(self.device)

    def train(self, epochs=3, lr=None, freeze=True, log=False):
        lr = self._enter_phase(lr, freeze)
        print(f"Training with {'frozen' if freeze else 'unfrozen'} feature layers at lr={lr}")
        for epoch in range(epochs):
            self._train_one_epoch(after_batch=lambda k, nb: print(f'batch ({k}/{nb})'))
            rows = self._evaluate()
            self._keep_outputs(rows, epoch == epochs - 1, freeze)
            avg_accuracy = sum(r[1] / r[2] for r in rows) / len(rows)          # mean of batch accuracies (Q14)
            print(f"Epoch {epoch + 1}, Test Accuracy: {avg_accuracy * 100:.2f}%")
            if log:
                with open('training_performance.txt', 'a') as f:
                    f.write(f"{self.sub}, Epoch {epoch + 1}, Test Accuracy: {avg_accuracy * 100:.2f}%\n")
