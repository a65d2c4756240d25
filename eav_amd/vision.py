"""ImageClassifierTrainer on MI355X - the reference's class, same constructor and train() signature.

Mirrors Transformer_torch/Transformer_Vision.py:8-129:
    ImageClassifierTrainer(DATA, model_path, sub='', num_labels=5, lr=5e-5, batch_size=128)
        .train(epochs=3, lr=None, freeze=True, log=False)     attribute: outputs_test
    calculate_accuracy(outputs, labels)
plus the trial-level vote of the driver block (:174-185) as `trial_vote`.  The ViT behind it is
eav_amd.transformer.Encoder; frame pre-processing stays the reference's per-frame call of the HF image
processor on the host (SURVEY.md section 8f "next" row 2) and the processed frames live in HBM (Q13).
Kept quirks: per-batch-mean test accuracy (Q14), outputs_test only on the last unfrozen epoch (Q15),
AdamW default weight decay 0.01 (Q10), one optimiser across phases (Q11).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .eegnet import DeviceLoader
from .optim import FusedAdam
from .transformer import Encoder


def calculate_accuracy(outputs, labels):
    _, predicted = torch.max(outputs.logits, 1)
    correct = (predicted == labels).sum().item()
    return correct / labels.size(0)


def trial_vote(outputs_test, labels, frames_per_trial=25):
    """mean of the per-frame logits over each trial -> argmax; accuracy and weighted F1 (:174-185)."""
    from sklearn.metrics import f1_score
    a = np.reshape(outputs_test, (-1, frames_per_trial, outputs_test.shape[-1]), 'C')
    pred = np.argmax(np.mean(a, 1), axis=1)
    return pred, float(np.mean(pred == labels)), float(f1_score(labels, pred, average='weighted'))


def _load_processor(model_path):
    try:
        from transformers import AutoImageProcessor
        return AutoImageProcessor.from_pretrained(model_path)
    except Exception:
        # torchvision-free images: the PIL implementation of the same processor (SURVEY shim S5)
        from transformers.models.vit.image_processing_pil_vit import ViTImageProcessorPil
        return ViTImageProcessorPil.from_pretrained(model_path)


class ImageClassifierTrainer:
    def __init__(self, DATA, model_path, sub='', num_labels=5, lr=5e-5, batch_size=128):
        self.tr_x, self.tr_y, self.te_x, self.te_y = DATA
        self.model_path = model_path
        self.num_labels = num_labels
        self.initial_lr = lr
        self.batch_size = batch_size
        self.frame_per_sample = np.shape(self.tr_x)[1]
        self.sub = sub
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if self.device.type != "cuda":
            raise _lib.EavError("eav_amd.ImageClassifierTrainer needs an MI355X (no CPU fallback)")
        self.test_prediction = list()

        self.processor = _load_processor(model_path)                           # :28
        self.model = Encoder.from_pretrained(model_path)                       # :29
        fresh = torch.nn.Linear(self.model.cfg.hidden, self.num_labels)        # :30
        self.model.reset_head(fresh.weight.detach(), fresh.bias.detach())
        self.model.num_labels = self.num_labels                                # :31
        self.model.to(self.device)

        self.optimizer = FusedAdam(self.model.parameters(), lr=self.initial_lr, weight_decay=0.01, decoupled=True)  # :36
        self.grad_sync = None

        print("Image preprocessing..")
        self.train_dataloader = self._prepare_dataloader(self.tr_x, self.tr_y, shuffle=True)[0]
        self.test_dataloader = self._prepare_dataloader(self.te_x, self.te_y, shuffle=False)[0]
        print("Ended..")

    def _prepare_dataloader(self, x, y, shuffle=True):
        processed_x = self.preprocess_images(x)
        y_repeated = torch.from_numpy(np.repeat(y, self.frame_per_sample)).long()
        size = self.model.cfg.H
        loader = DeviceLoader(processed_x.view(-1, self.model.cfg.C, size, size), y_repeated, self.batch_size, shuffle,
                              self.device)
        return loader, processed_x, y_repeated

    def preprocess_images(self, image_list):
        """Reference: a Python loop calling the HF processor per frame, then one stack().to(device)
        (:52-59).  Uniform uint8 HWC frames with the standard resize/rescale/normalize recipe go through
        one HIP kernel (bit-identical result); anything else takes the reference's host route."""
        p = self.processor
        arr = np.asarray(image_list) if not isinstance(image_list, np.ndarray) else image_list
        std_recipe = (getattr(p, "do_resize", False) and getattr(p, "do_rescale", False)
                      and getattr(p, "do_normalize", False) and not getattr(p, "do_center_crop", False)
                      and int(getattr(p, "resample", 2)) == 2 and getattr(p, "size", None) is not None)
        if std_recipe and arr.dtype == np.uint8 and arr.ndim == 5 and arr.shape[-1] == 3:
            from .preprocess import frames_to_pixel_values
            sz = p.size
            hw = (int(sz["height"]), int(sz["width"])) if isinstance(sz, dict) else (int(sz.height), int(sz.width))
            return frames_to_pixel_values(arr.reshape(-1, *arr.shape[2:]), hw, p.image_mean, p.image_std,
                                          p.rescale_factor, self.device)
        pixel_values_list = []
        for img_set in image_list:
            for img in img_set:
                processed = self.processor(images=img, return_tensors="pt")
                pixel_values_list.append(processed.pixel_values.squeeze())
        return torch.stack(pixel_values_list).to(self.device)

    def train(self, epochs=3, lr=None, freeze=True, log=False):
        lr = lr if lr is not None else self.initial_lr
        for param_group in self.optimizer.param_groups:
            param_group['lr'] = lr
        for param in self.model.parameters():
            param.requires_grad = not freeze
        for param in self.model.classifier.parameters():
            param.requires_grad = True
        if self.grad_sync is not None:     # frozen phase: only the head's gradients cross the xGMI links
            self.grad_sync.set_active(self.model.head_grad_ranges() if freeze else None)

        print(f"Training with {'frozen' if freeze else 'unfrozen'} feature layers at lr={lr}")

        for epoch in range(epochs):
            self.model.train()
            total_batches = len(self.train_dataloader)
            for batch_idx, (pixel_values, labels) in enumerate(self.train_dataloader, start=1):
                self.optimizer.zero_grad()
                outputs = self.model(pixel_values=pixel_values, labels=labels)
                loss = outputs.loss
                loss.backward()
                if self.grad_sync is not None:
                    self.grad_sync()
                self.optimizer.step()
                print(f'batch ({batch_idx}/{total_batches})')

            self.model.eval()
            total_accuracy = 0
            outputs_batch = []
            with torch.no_grad():
                for pixel_values, labels in self.test_dataloader:
                    outputs = self.model(pixel_values)
                    total_accuracy += calculate_accuracy(outputs, labels)
                    outputs_batch.append(outputs.logits.detach().cpu().numpy())

            if epoch == epochs - 1 and not freeze:
                self.outputs_test = np.concatenate(outputs_batch, axis=0)

            avg_accuracy = total_accuracy / len(self.test_dataloader)
            print(f"Epoch {epoch + 1}, Test Accuracy: {avg_accuracy * 100:.2f}%")
            if log:
                with open('training_performance.txt', 'a') as f:
                    f.write(f"{self.sub}, Epoch {epoch + 1}, Test Accuracy: {avg_accuracy * 100:.2f}%\n")
