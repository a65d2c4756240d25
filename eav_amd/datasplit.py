"""EAVDataSplit - the per-class ordered train/test splitter.

Mirrors the reference class one-for-one (EAV_datasplit.py:7-58): same
constructor, ``get_split(h_idx=40)`` and ``get_loaders()``.  It is pure
int64/numpy index work and is parity-pinned bit-exactly by
tests/test_datasplit.py against index vectors captured from the imported
reference (tests/golden/datasplit_*.npz).
"""
from __future__ import annotations

import numpy as np


class EAVDataSplit:
    def __init__(self, x, y, batch_size=32):
        # EAV_datasplit.py:8-11
        self.x = np.array(x)
        self.y = np.array(y)
        self.batch_size = batch_size

    def _split_features_labels(self):
        # EAV_datasplit.py:12-24 - order-preserving selection per class 0..4
        features, labels = [], []
        for class_idx in range(5):
            sel = np.where(self.y == class_idx)
            features.append(self.x[sel])
            labels.append(self.y[sel])
        return features, labels

    def split_indices(self, h_idx=40):
        """(train_idx, test_idx) int64 - the index form of get_split (1-D y)."""
        per_class = [np.flatnonzero(self.y == c) for c in range(5)]
        tr = np.concatenate([p[:h_idx] for p in per_class]).astype(np.int64)
        te = np.concatenate([p[h_idx:] for p in per_class]).astype(np.int64)
        return tr, te

    def get_split(self, h_idx=40):
        # EAV_datasplit.py:26-40
        features, labels = self._split_features_labels()
        train_features = np.concatenate([f[:h_idx] for f in features], axis=0)
        test_features = np.concatenate([f[h_idx:] for f in features], axis=0)
        train_labels = np.concatenate([l[:h_idx] for l in labels], axis=0)
        test_labels = np.concatenate([l[h_idx:] for l in labels], axis=0)
        return np.squeeze(train_features), train_labels, np.squeeze(test_features), test_labels

    def get_loaders(self):
        # EAV_datasplit.py:42-58
        import torch
        from torch.utils.data import DataLoader, TensorDataset

        tr_x, tr_y, te_x, te_y = self.get_split()
        tr_x = torch.Tensor(np.squeeze(tr_x))
        te_x = torch.Tensor(np.squeeze(te_x))
        tr_y = torch.Tensor(tr_y).long()
        te_y = torch.Tensor(te_y).long()
        loader_train = DataLoader(TensorDataset(tr_x, tr_y), batch_size=self.batch_size, shuffle=True)
        loader_test = DataLoader(TensorDataset(te_x, te_y), batch_size=self.batch_size, shuffle=False)
        return loader_train, loader_test
