"""Canonical EEGNet + EEGNetTrainer on MI355X: the class API of CNN_torch/CNN_EEG.py over libeav_hip.so.

    EEGNet(nb_classes, Chans=64, Samples=128, dropoutRate=0.5, kernLength=64, F1=8, D=2, F2=16, norm_rate=0.25) (:12-13)
        __call__(x[B,Chans,Samples] or [B,1,Chans,Samples]) -> logits [B,nb_classes]                  (:58-67)
    EEGNetTrainer(model, train_dataset, val_dataset, batch_size=32, epochs=100, lr=0.001)             (:75)
        .train_epoch() / .validate_epoch() / .train() / .predict(dataset=None)                         (:91-162)

The module owns the same ``block1`` / ``block2`` / ``classifier`` sub-modules, so ``state_dict()`` keys match the
reference's, and the constructor repeats the reference's shape probe (a train-mode dummy forward on the host,
:48-54), which leaves ``running_var = 0.9`` / ``num_batches_tracked = 1`` in every BatchNorm and advances the torch
RNG by the two dropout draws - a freshly constructed model is therefore in the reference's state.  All training and
inference arithmetic is in hand-written gfx950 kernels (csrc/eegnet_canon.hip, eegnet_block.hip, head_optim.hip);
there is no CPU path: calling the model with a host tensor raises.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .eegnet import DeviceLoader, GraphStep, cached_workspace
from .optim import CrossEntropyLoss, FusedAdam, flatten_parameters

_PARAM_ORDER = [
    "block1.0.weight", "block1.1.weight", "block1.1.bias", "block1.2.weight", "block1.3.weight", "block1.3.bias",
    "block2.0.weight", "block2.1.weight", "block2.2.weight", "block2.2.bias", "classifier.weight", "classifier.bias",
]


class _Workspace:
    """Device buffers for one (B, Chans, Samples) problem size (all fp32)."""

    def __init__(self, m, B, dev):
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        C, S, F1, C2, F2, K, K2 = m.Chans, m.Samples, m.F1, m.F1 * m.D, m.F2, m.kernLength, m.K2
        self.key = (B, C, S)
        T2, T3 = S // 4, S // 4 // 8
        self.T2, self.T3, self.NF = T2, T3, F2 * T3
        self.y1, self.g1 = f(B, F1, C, S), f(B, F1, C, S)
        self.z2, self.dz2 = f(B, C2, S), f(B, C2, S)
        self.a2, self.da2 = f(B, C2, T2), f(B, C2, T2)
        self.d3, self.dd3 = f(B, C2, T2), f(B, C2, T2)
        self.z3, self.dz3 = f(B, F2, T2), f(B, F2, T2)
        self.a3, self.da3 = f(B, F2 * T3), f(B, F2 * T3)
        self.logits = f(B, m.nb_classes)
        self.bn1, self.bn2, self.bn3 = f(6 * F1), f(6 * C2), f(6 * F2)
        self.np_t = _lib.plain("eav_tconv_fwd_nparts", B, C, S, F1, K)
        self.part_t = f(self.np_t, 2 * F1)
        self.np_s = _lib.plain("eav_spatial_nparts", B, S)
        self.part_s = f(self.np_s, 2 * C2)
        self.np_c = _lib.plain("eav_sepconv_fwd_nparts", B, T2)
        self.part_c = f(self.np_c, 2 * F2)
        self.part_pb = f(B, 2 * max(C2, F2))
        self.part_sst = f(self.np_s, 2 * F1)
        self.part_sw = f(self.np_s, C2 * C)
        self.np_tw = _lib.plain("eav_tconv_wgrad_nparts", B, C, S, F1, K)
        self.part_tw = f(self.np_tw, F1 * K)
        self.np_pw = _lib.plain("eav_pointwise_bwd_nparts", B, T2)
        self.part_pw = f(self.np_pw, F2 * C2)
        self.part_dw = f(B, C2 * K2)


class _EEGNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        ctx.model = model
        ctx.token = model._launch_forward(x)
        return model._ws.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        grads = ctx.model._launch_backward(dlogits.contiguous(), ctx.token)
        return (None, None, *grads)


class EEGNet(nn.Module):
    K2 = 16   # taps of the depthwise temporal conv of block2 (CNN_EEG.py:35)

    def __init__(self, nb_classes, Chans=64, Samples=128, dropoutRate=0.5, kernLength=64, F1=8, D=2, F2=16,
                 norm_rate=0.25):
        super().__init__()
        if not (1 <= F1 <= 16 and 1 <= D <= 8 and D * F1 <= 64 and 1 <= F2 <= 64 and 1 <= kernLength <= 1024
                and 1 <= Chans <= 256 and 1 <= nb_classes <= 16 and Samples >= 32):
            raise NotImplementedError("eav_amd.EEGNet: the gfx950 kernels cover F1<=16, D<=8, D*F1<=64, F2<=64, "
                                      "kernLength<=1024, Chans<=256, nb_classes<=16, Samples>=32")
        self.Chans, self.Samples = Chans, Samples
        # the reference's modules at the reference's indices (:20-42): identical state_dict keys and default init
        self.block1 = nn.Sequential(
            nn.Conv2d(1, F1, (1, kernLength), padding='same', bias=False),
            nn.BatchNorm2d(F1),
            nn.Conv2d(F1, D * F1, (Chans, 1), groups=F1, bias=False),
            nn.BatchNorm2d(D * F1),
            nn.ELU(),
            nn.AvgPool2d((1, 4)),
            nn.Dropout(dropoutRate),
        )
        self.block2 = nn.Sequential(
            nn.Conv2d(D * F1, D * F1, (1, self.K2), padding='same', groups=D * F1, bias=False),
            nn.Conv2d(D * F1, F2, (1, 1), bias=False),
            nn.BatchNorm2d(F2),
            nn.ELU(),
            nn.AvgPool2d((1, 8)),
            nn.Dropout(dropoutRate),
        )
        self.flatten = nn.Flatten()
        # the reference sizes the classifier with a train-mode dummy forward (:48-54); repeated on the host so that the
        # BatchNorm buffers and the torch RNG end up exactly where the reference's constructor leaves them
        with torch.no_grad():
            n_flatten = self.flatten(self.block2(self.block1(torch.zeros(1, 1, Chans, Samples)))).shape[1]
        assert n_flatten == F2 * (Samples // 4 // 8)
        self.classifier = nn.Linear(n_flatten, nb_classes)

        self.nb_classes, self.kernLength, self.F1, self.D, self.F2 = nb_classes, kernLength, F1, D, F2
        self.dropoutRate, self.norm_rate = float(dropoutRate), norm_rate      # norm_rate: accepted, unused (:13)
        self._ws = None
        self._flat = None
        self._token = 0
        self._saved = None
        self.dropout_seed = 0x0CA2EED
        self._dropout_masks = None
        self._fwd_counter = None

    # ------------------------------------------------------------------ plumbing
    def _ensure_flat(self):
        p0 = self.block1[0].weight
        if self._flat is None or self._flat[0].device != p0.device or getattr(p0, "_eav_flat", None) is None \
                or p0.data_ptr() != self._flat[0].data_ptr():
            assert list(dict(self.named_parameters())) == _PARAM_ORDER
            self._flat = flatten_parameters(self)

    def _params(self):
        n = dict(self.named_parameters())
        return [n[k] for k in _PARAM_ORDER]

    def set_dropout_masks(self, masks):
        """Testing hook: explicit uint8 keep-masks ([B,C2,S/4], [B,F2,S/32]) instead of the generator."""
        self._dropout_masks = masks

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _lib.EavError("eav_amd.EEGNet runs on an MI355X only: move the model and the input to the ROCm "
                                "device (there is no CPU fallback)")
        if x.dim() == 3:
            x = x.unsqueeze(1)                                                           # :61-62
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.Chans or x.shape[3] != self.Samples:
            raise ValueError(f"expected input [B,{self.Chans},{self.Samples}], got {tuple(x.shape)}")
        if self.block1[0].weight.device != x.device:
            raise _lib.EavError("model and input are on different devices")
        self._ensure_flat()
        return _EEGNetFn.apply(x.contiguous().float(), self, *self._params())

    # ------------------------------------------------------------------ kernels
    def _launch_forward(self, x):
        L, P, st = _lib.call, _lib.ptr, _lib.stream_ptr()
        B, C, S, K, K2 = x.shape[0], self.Chans, self.Samples, self.kernLength, self.K2
        F1, D, F2, C2 = self.F1, self.D, self.F2, self.F1 * self.D
        # one workspace per batch size, never freed: captured hipGraphs hold its raw pointers (see EEGNet_tor._workspace)
        wkey = (B, C, S, str(x.device))
        if not hasattr(self, "_wss"):
            self._wss = {}
        ws = self._ws = cached_workspace(self._wss, wkey, lambda: _Workspace(self, B, x.device))
        training = bool(self.training)
        w1, g1w, g1b, wd, g2w, g2b, wdw, wp, g3w, g3b, wc, bc = [P(p) for p in self._params()]
        drop = self.dropoutRate if training else 0.0
        masks = self._dropout_masks if training else None
        self._token += 1
        seed1, seed2 = self.dropout_seed, self.dropout_seed + 1
        cnt = None
        if drop > 0.0 and masks is None:    # device-resident dropout counter: graph replays draw fresh masks
            if self._fwd_counter is None or self._fwd_counter.device != x.device:
                self._fwd_counter = torch.zeros((), dtype=torch.int64, device=x.device)
            L("eav_counter_inc", P(self._fwd_counter), st)
            cnt = P(self._fwd_counter)
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None

        def bnfin(part, nparts, nch, count, gw, gb, bn, buf):
            b0 = P(buf)
            L("eav_bn_finalize", P(part), nparts, nch, float(count), gw, gb, P(bn.running_mean), P(bn.running_var),
              int(training), float(bn.momentum), float(bn.eps), b0, b0 + 4 * nch, b0 + 8 * nch, b0 + 12 * nch, st)
            if training:
                bn.num_batches_tracked += 1

        L("eav_tconv_fwd", P(x), w1, P(ws.y1), P(ws.part_t), B, C, S, F1, K, st)
        bnfin(ws.part_t, ws.np_t, F1, B * C * S, g1w, g1b, self.block1[1], ws.bn1)
        L("eav_spatial_fwd", P(ws.y1), P(ws.bn1), wd, P(ws.z2), P(ws.part_s), B, C, S, F1, D, 0, st)
        bnfin(ws.part_s, ws.np_s, C2, B * S, g2w, g2b, self.block1[3], ws.bn2)
        L("eav_bn_elu_pool_fwd", P(ws.z2), P(ws.bn2), P(ws.a2), B, C2, S, 4, drop, seed1, m1, cnt, st)
        L("eav_sepconv_fwd", P(ws.a2), wdw, wp, P(ws.d3), P(ws.z3), P(ws.part_c), B, C2, F2, ws.T2, K2, st)
        bnfin(ws.part_c, ws.np_c, F2, B * ws.T2, g3w, g3b, self.block2[2], ws.bn3)
        L("eav_bn_elu_pool_fwd", P(ws.z3), P(ws.bn3), P(ws.a3), B, F2, ws.T2, 8, drop, seed2, m2, cnt, st)
        L("eav_dense_softmax_fwd", P(ws.a3), wc, bc, P(ws.logits), None, B, ws.NF, self.nb_classes, st)
        self._saved = (self._token, x, training, drop, seed1, seed2, masks, cnt, ws)
        return self._token

    def _launch_backward(self, dlogits, token):
        if self._saved is None or self._saved[0] != token:
            raise _lib.EavError("EEGNet.backward: the activations of this forward were overwritten by a later forward "
                                "(one outstanding forward per backward)")
        L, P, st = _lib.call, _lib.ptr, _lib.stream_ptr()
        _, x, training, drop, seed1, seed2, masks, cnt, ws = self._saved
        B, C, S, K, K2 = x.shape[0], self.Chans, self.Samples, self.kernLength, self.K2
        F1, D, F2, C2, T2, NF = self.F1, self.D, self.F2, self.F1 * self.D, ws.T2, ws.NF
        flat, gflat, offs = self._flat
        g = {k: gflat[offs[k][0]:offs[k][0] + offs[k][1]] for k in _PARAM_ORDER}
        named = dict(self.named_parameters())
        wd, wdw, wp, wc = (P(named[k]) for k in ("block1.2.weight", "block2.0.weight", "block2.1.weight",
                                                 "classifier.weight"))
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None
        tr = int(training)

        L("eav_dense_softmax_bwd", P(dlogits), None, P(ws.a3), wc, P(g["classifier.weight"]),
          P(g["classifier.bias"]), P(ws.da3), B, NF, self.nb_classes, st)
        # block2 tail: Dropout <- AvgPool8 <- ELU <- BatchNorm
        b3 = P(ws.bn3)
        L("eav_bn_elu_pool_bwd_reduce", P(ws.da3), P(ws.z3), b3, P(ws.part_pb), B, F2, T2, 8, drop, seed2, m2, cnt, st)
        L("eav_bn_bwd_finalize", P(ws.part_pb), B, F2, float(B * T2), tr, P(g["block2.2.weight"]),
          P(g["block2.2.bias"]), b3 + 16 * F2, b3 + 20 * F2, st)
        L("eav_bn_elu_pool_bwd_apply", P(ws.da3), P(ws.z3), b3, b3 + 16 * F2, P(ws.dz3), B, F2, T2, 8, drop, seed2,
          m2, cnt, st)
        # pointwise and depthwise temporal convs
        L("eav_pointwise_bwd", P(ws.dz3), P(ws.d3), wp, P(ws.dd3), P(ws.part_pw), B, C2, F2, T2, st)
        L("eav_reduce_partials", P(ws.part_pw), ws.np_pw, F2 * C2, F2 * C2, 1.0, P(g["block2.1.weight"]), st)
        L("eav_dwt_bwd", P(ws.dd3), P(ws.a2), wdw, P(ws.da2), P(ws.part_dw), B, C2, T2, K2, st)
        L("eav_reduce_partials", P(ws.part_dw), B, C2 * K2, C2 * K2, 1.0, P(g["block2.0.weight"]), st)
        # block1 tail: Dropout <- AvgPool4 <- ELU <- BatchNorm
        b2 = P(ws.bn2)
        L("eav_bn_elu_pool_bwd_reduce", P(ws.da2), P(ws.z2), b2, P(ws.part_pb), B, C2, S, 4, drop, seed1, m1, cnt, st)
        L("eav_bn_bwd_finalize", P(ws.part_pb), B, C2, float(B * S), tr, P(g["block1.3.weight"]),
          P(g["block1.3.bias"]), b2 + 16 * C2, b2 + 20 * C2, st)
        L("eav_bn_elu_pool_bwd_apply", P(ws.da2), P(ws.z2), b2, b2 + 16 * C2, P(ws.dz2), B, C2, S, 4, drop, seed1, m1,
          cnt, st)
        # depthwise spatial conv <- BatchNorm <- temporal conv
        b1 = P(ws.bn1)
        L("eav_spatial_bwd", P(ws.y1), P(ws.dz2), b1, wd, P(ws.g1), P(ws.part_sst), P(ws.part_sw), B, C, S, F1, D, 0, st)
        L("eav_reduce_partials", P(ws.part_sw), ws.np_s, C2 * C, C2 * C, 1.0, P(g["block1.2.weight"]), st)
        L("eav_bn_bwd_finalize", P(ws.part_sst), ws.np_s, F1, float(B * C * S), tr, P(g["block1.1.weight"]),
          P(g["block1.1.bias"]), b1 + 16 * F1, b1 + 20 * F1, st)
        L("eav_tconv_wgrad", P(x), P(ws.y1), P(ws.g1), b1, P(ws.part_tw), B, C, S, F1, K, st)
        L("eav_reduce_partials", P(ws.part_tw), ws.np_tw, F1 * K, F1 * K, 1.0, P(g["block1.0.weight"]), st)
        return [g[k].view(named[k].shape) if named[k].requires_grad else None for k in _PARAM_ORDER]


class EEGNetTrainer:
    """CNN_EEG.py:70-162.  The datasets are ``TensorDataset(x, y)``; both splits are moved to HBM once and batches
    are assembled there (eav_amd.eegnet.DeviceLoader: same samplers and torch RNG consumption as DataLoader)."""

    def __init__(self, model, train_dataset, val_dataset, batch_size=32, epochs=100, lr=0.001):
        if not torch.cuda.is_available():
            raise _lib.EavError("eav_amd.EEGNetTrainer needs an MI355X (torch device 'cuda' on ROCm); no CPU fallback")
        self.device = torch.device("cuda")
        print(f"Using device: {self.device}")
        self.model = model.to(self.device)
        self.epochs = epochs
        self.batch_size = batch_size
        self.train_loader = DeviceLoader(*train_dataset.tensors, batch_size, True, self.device)
        self.test_loader = DeviceLoader(*val_dataset.tensors, batch_size, False, self.device)
        self.criterion = CrossEntropyLoss()                                          # :88
        self.optimizer = FusedAdam(model.parameters(), lr=lr, capturable=True)       # :89
        self.grad_sync = None     # set by eav_amd.dist.attach(trainer) under torchrun
        self.use_graph = True
        self._graph = None

    def train_epoch(self):
        self.model.train()
        running_loss = torch.zeros((), dtype=torch.float32, device=self.device)
        dl = self.train_loader
        batches = dl.index_batches()
        for idx in batches:
            if self.use_graph and len(idx) == self.batch_size:
                if self._graph is None:
                    self._graph = GraphStep(self.model, self.optimizer, self.criterion, dl.x, dl.y, len(idx),
                                            self.grad_sync)
                _, loss = self._graph.run(idx)
            else:
                inputs, labels = dl.gather(idx)
                self.optimizer.zero_grad()
                outputs = self.model(inputs)
                loss = self.criterion(outputs, labels)
                loss.backward()
                if self.grad_sync is not None:
                    self.grad_sync()
                self.optimizer.step()
                loss = loss.detach()
            running_loss += loss          # accumulated on the device: one host read per epoch, not per step (:106)
        self.criterion.check()            # labels outside [0, classes) seen by any step of this epoch raise here
        return running_loss.item() / len(batches)

    def validate_epoch(self):
        self.model.eval()
        val_loss, correct, total = 0.0, 0, 0
        with torch.no_grad():
            for inputs, labels in self.test_loader:
                outputs = self.model(inputs)
                val_loss += self.criterion(outputs, labels).item()
                correct += (outputs.argmax(1) == labels).sum().item()
                total += labels.size(0)
        return val_loss / len(self.test_loader), 100 * correct / total

    def train(self):
        print(f"Starting training for {self.epochs} epochs...")
        for epoch in range(self.epochs):
            train_loss = self.train_epoch()
            val_loss, accuracy = self.validate_epoch()
            print(f'Epoch {epoch + 1}/{self.epochs} | Train Loss: {train_loss:.4f} | Val Loss: {val_loss:.4f} | '
                  f'Val Acc: {accuracy:.2f}%')

    def predict(self, dataset=None):
        loader = self.test_loader
        if dataset is not None:
            loader = DeviceLoader(*dataset.tensors, 32, False, self.device)
        predictions = []
        self.model.eval()
        with torch.no_grad():
            for inputs, _ in loader:
                predictions.extend(self.model(inputs).argmax(1).cpu().tolist())
        return predictions
