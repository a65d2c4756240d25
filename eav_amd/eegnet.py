"""EEGNet_tor + Trainer_uni on MI355X: the reference's class API over libeav_hip.so.

Mirrors CNN_torch/EEGNet_tor.py one-for-one at the Python boundary:

    EEGNet_tor(nb_classes, Chans=30, Samples=500, dropoutRate=0.5, kernLength=300,
               F1=8, D=8, F2=64, norm_rate=1.0, dropoutType='Dropout')      (:16-17)
        __call__(x[B,1,Chans,Samples]) -> softmax probabilities [B,nb_classes]   (:50-67)
    Trainer_uni(model, data, lr=1e-4, batch_size=32, num_epochs=10, device=None) (:70)
        .train() / .validate()                                                   (:96,:118)

The module owns the same sub-modules, so ``state_dict()`` keys and the default
initialisation stream (torch RNG) are those of the reference; the arithmetic of
forward and backward is entirely in hand-written gfx950 kernels (eav_amd/csrc).
Reference behaviour kept on purpose: max-norm renorm after the
forward and before the backward (Q1/Q2), softmax output fed to CrossEntropyLoss
(Q3), ``model.train()`` called once so that epochs >= 2 train in eval mode (Q4).

There is no CPU path: calling the model with a non-device tensor raises.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import _lib
from .optim import CrossEntropyLoss, FusedAdam, flatten_parameters, unit_gradient

_PARAM_ORDER = [
    "firstConv.weight", "firstBN.weight", "firstBN.bias",
    "depthwiseConv.weight", "depthwiseBN.weight", "depthwiseBN.bias",
    "separableConv.weight", "separableBN.weight", "separableBN.bias",
    "dense.weight", "dense.bias",
]


class _Workspace:
    """Device buffers for one (B, Chans, Samples) problem size (all fp32)."""

    def __init__(self, B, C, S, klen, nb, dev):
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        self.key = (B, C, S)
        T2, T3 = S // 4, S // 4 // 8
        self.T2, self.T3, self.NF = T2, T3, 64 * T3
        nchunk = (S + 1023) // 1024
        self.nchunk = nchunk
        self.y1, self.g1 = f(B, 8, C, S), f(B, 8, C, S)
        self.z = f(B, 64, S)
        self.p2, self.dp2 = f(B, 64, T2), f(B, 64, T2)
        self.u3, self.du3 = f(B, 64, T2), f(B, 64, T2)
        self.p3, self.dp3 = f(B, 64 * T3), f(B, 64 * T3)
        self.bn1, self.bn2, self.bn3 = f(6 * 8), f(6 * 64), f(6 * 64)
        self.wTf, self.wTb = f(1024, 64), f(1024, 64)
        self.np_fir = _lib.plain("eav_eegnet_fir_fwd_nparts", B, C, S)
        self.np_fir_fft = _lib.plain("eav_eegnet_fir_fwd_fft_nparts", B, C, S)
        self.part_fir = f(max(self.np_fir, self.np_fir_fft), 16)
        self.fft_ws = None         # spectrum partials of the FFT weight gradient, allocated on first use
        self.part_dw = f(B * nchunk, 128)
        self.np_c3 = _lib.plain("eav_conv64_fwd_nparts", B, T2)
        self.np_c3_fft = _lib.plain("eav_conv64_fft_nparts", B, T2)
        self.part_c3 = f(max(self.np_c3, self.np_c3_fft), 128)
        self.c64_ws = None         # spectra workspace of the frequency-domain separableConv, allocated on first use
        self.part_pb = f(B, 128)
        self.part_dst = f(B * nchunk, 16)
        self.part_dw2 = f(B * nchunk, 64 * C)
        self.np_fw = _lib.plain("eav_eegnet_fir_wgrad_nparts", B, C, S)
        self.part_fw = f(self.np_fw, 8 * klen)
        self.np_cw = _lib.plain("eav_conv64_wgrad_nparts", B, T2)
        self.part_cw = f(self.np_cw, 64 * 1024)


class _GenericWorkspace:
    """Device buffers of the run-time-parametrised path (EEGNet_tor._generic)."""

    def __init__(self, m, B, dev):
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)  # noqa: E731
        C, S, F1, C2, F2, K = m.Chans, m.Samples, m.F1, m.F1 * m.D, m.F2, m.kernLength
        T2, T3 = S // 4, S // 4 // 8
        self.T2, self.T3, self.NF = T2, T3, F2 * T3
        self.y1, self.g1 = f(B, F1, C, S), f(B, F1, C, S)
        self.z, self.dz = f(B, C2, S), f(B, C2, S)
        self.p2, self.dp2 = f(B, C2, T2), f(B, C2, T2)
        self.u3, self.du3 = f(B, F2, T2), f(B, F2, T2)
        self.p3, self.dp3 = f(B, F2 * T3), f(B, F2 * T3)
        self.bn1, self.bn2, self.bn3 = f(6 * F1), f(6 * C2), f(6 * F2)
        self.np_t = _lib.plain("eav_tconv_fwd_nparts", B, C, S, F1, K)
        self.part_t = f(self.np_t, 2 * F1)
        self.np_s = _lib.plain("eav_spatial_nparts", B, S)
        self.part_s = f(self.np_s, 2 * C2)
        self.np_c = _lib.plain("eav_dconv_fwd_nparts", B, T2)
        self.part_c = f(self.np_c, 2 * F2)
        self.part_pb = f(B, 2 * max(C2, F2))
        self.part_sst = f(self.np_s, 2 * F1)
        self.part_sw = f(self.np_s, C2 * C)
        self.np_tw = _lib.plain("eav_tconv_wgrad_nparts", B, C, S, F1, K)
        self.part_tw = f(self.np_tw, F1 * K)
        self.part_cw = f(B, F2 * C2 * 16)


def cached_workspace(cache, key, make, keep_unpinned=2):
    """Workspace cache shared by the EEG models.  A captured hipGraph (GraphStep) has the raw device pointers of the
    workspace it was captured with baked in, so THOSE workspaces (marked `pinned` by GraphStep after capture) live as long
    as the model; eager sizes - ragged last batches, validation, user-chosen inference batches - share `keep_unpinned`
    replaceable slots (most recently used first), so varying batch sizes do not accumulate multi-GB workspaces."""
    ws = cache.get(key)
    if ws is None:
        if not torch.cuda.is_current_stream_capturing():
            loose = [k for k, w in cache.items() if not getattr(w, "pinned", False)]
            for k in loose[:max(0, len(loose) - (keep_unpinned - 1))]:     # dict order = insertion / last-use order
                del cache[k]
        ws = make()
    else:
        del cache[key]          # re-insert: most recently used last
    cache[key] = ws
    return ws


class IndexedBatch:
    """A batch addressed in place: samples `idx` (device int64 [B]) of an HBM-resident data set `data` [N,1,C,S].  The FIR
    kernels - the only readers of the network input - take the index vector, so no gathered copy of the batch is made."""

    def __init__(self, data, idx):
        self.data, self.idx = data, idx
        self.shape = (idx.numel(),) + tuple(data.shape[1:])
        self.device = data.device


class _EEGNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, model, *params):
        ctx.model = model
        ctx.token = model._launch_forward(x)
        # the probabilities tensor this forward wrote: no copy kernel.  (A detached alias, not the saved object itself:
        # returning the very tensor that the model also keeps for its backward crashes hipGraph capture in torch 2.10.)
        # The alias shares its version counter with the saved tensor: an in-place edit of the returned scores before
        # backward (clamp_, += eps ...) would silently corrupt dense_softmax_bwd's input - checked in backward.
        ctx.probs_version = model._saved[-1]._version
        return model._saved[-1].detach()

    @staticmethod
    def backward(ctx, dprobs):
        saved = ctx.model._saved
        if saved is not None and saved[0] == ctx.token and saved[-1]._version != ctx.probs_version:
            raise _lib.EavError("EEGNet_tor: the scores returned by forward() were modified in place before backward(); "
                                "they alias the probabilities the backward reads - clone() them first")
        grads = ctx.model._launch_backward(dprobs.contiguous(), ctx.token)
        return (None, None, *grads)


class EEGNet_tor(nn.Module):
    def __init__(self, nb_classes, Chans=30, Samples=500, dropoutRate=0.5, kernLength=300, F1=8, D=8, F2=64,
                 norm_rate=1.0, dropoutType='Dropout'):
        super().__init__()
        # The reference configuration (F1=8, D=8, F2=64, kernLength<=300, Chans<=32: EEGNet_tor.py:159) runs the specialised
        # fp32-MFMA kernels; every other width the reference constructor accepts (:16-17) takes the run-time-parametrised
        # kernels of csrc/eegnet_canon.hip (`_generic`), whose LDS tiles bound it at the sizes below.
        self._generic = not (F1 == 8 and D == 8 and F2 == 64 and 1 <= kernLength <= 300 and 1 <= Chans <= 32)
        if not (1 <= F1 <= 16 and 1 <= D <= 8 and F1 * D <= 64 and 1 <= F2 <= 64 and 1 <= kernLength <= 1024
                and 1 <= Chans <= 256 and 1 <= nb_classes <= 16 and Samples >= 32):
            raise NotImplementedError("eav_amd.EEGNet_tor: the gfx950 kernels cover F1<=16, D<=8, F1*D<=64, F2<=64, "
                                      "kernLength<=1024, Chans<=256, nb_classes<=16, Samples>=32")
        # same sub-modules in the same construction order as the reference (:21-48): identical
        # state_dict keys and identical consumption of the torch RNG by the default initialisers
        self.dropout = nn.Dropout(dropoutRate) if dropoutType == 'Dropout' else nn.Dropout2d(dropoutRate)
        self.firstConv = nn.Conv2d(1, F1, (1, kernLength), padding='same', bias=False)
        self.firstBN = nn.BatchNorm2d(F1)
        self.elu = nn.ELU()
        self.depthwiseConv = nn.Conv2d(F1, F1 * D, (Chans, 1), groups=F1, padding=0, bias=False)
        self.depthwiseBN = nn.BatchNorm2d(F1 * D)
        self.depthwisePool = nn.AvgPool2d((1, 4))
        self.separableConv = nn.Conv2d(F1 * D, F2, (1, 16), padding='same', bias=False)
        self.separableBN = nn.BatchNorm2d(F2)
        self.separablePool = nn.AvgPool2d((1, 8))
        self.flatten = nn.Flatten()
        self.dense = nn.Linear(F2 * (Samples // 4 // 8), nb_classes)
        self.softmax = nn.Softmax(dim=1)

        self.nb_classes, self.Chans, self.Samples, self.kernLength = nb_classes, Chans, Samples, kernLength
        self.F1, self.D, self.F2 = F1, D, F2
        self.norm_rate, self.dropoutRate = float(norm_rate), float(dropoutRate)
        # any other dropoutType is nn.Dropout2d in the reference (:21): one keep decision per (sample, channel) map - the
        # kernels take it as a negative probability (eav_hip.h)
        self.spatial_dropout = dropoutType != 'Dropout'
        self._ws = None
        self._wss = {}                         # workspaces by (B, Chans, Samples, device): see _workspace()
        self._flat = None
        self._token = 0
        self._saved = None
        self.dropout_seed = 0x0EA5EED          # base seed of the counter-based dropout generator
        self._dropout_masks = None             # tests: (mask1 uint8 [B,64,S/4], mask2 uint8 [B,64,S/32])
        self.apply_max_norm = True
        self.kernel_events = None              # bench: {kernel name: [(start_event, end_event), ...]}
        # (the round-2 "split" mode - FIR / separableConv products on the fp16 matrix cores with two-piece operands - was
        # retired in round 5: with the FFT FIR and the frequency-domain separableConv it was the slower path)
        self.fir_precision = "fp32"
        # How the exact-fp32 firstConv and its weight gradient are evaluated.  "fft": overlap-save blocks of 1024-point
        # FFTs (csrc/eegnet_fir_fft.hip: ~350 flops per output sample for the 8 filters together instead of 4800 - the two
        # kernels become HBM-bound); "mfma": the Toeplitz GEMMs on the fp32 matrix cores (csrc/eegnet_fir.hip); "auto"
        # (default): FFT for recordings of at least two 704-sample blocks and kernels of <= 321 taps, MFMA for short epochs
        # (the reference's own [B,1,30,500], where one FFT block would be mostly padding).  EAV_FIR_ALGO overrides.
        self.fir_algo = os.environ.get("EAV_FIR_ALGO", "auto")
        self.conv_algo = os.environ.get("EAV_CONV_ALGO", "auto")      # separableConv: see _use_conv_fft
        self._fwd_counter = None               # device uint64: number of training forwards (dropout stream)
        self._infer = False                    # set per call: no-grad eval-mode forward
        # GraphStep: (optimiser step counter, labels, idx, targets, batch) - raw pointers for eav_step_begin, taken by the next
        # forward's counter launch (left in place if that forward has none to merge them into)
        self._step_begin = None
        # validate()'s forward with block 1 as ONE kernel (eav_eegnet_block1_infer: y1 / z never written).  Opt-in: at the
        # benchmark shape [64,1,30,10000] it measures 1.35 ms against 1.30 ms for the training kernels in eval mode - with its
        # 128 z accumulators per lane it runs one wave per SIMD and cannot hide its VALU tail behind another wave's MFMAs
        self.fused_eval = False

    # ------------------------------------------------------------------ plumbing
    def _use_fft(self):
        if self.fir_algo not in ("auto", "fft", "mfma"):
            raise ValueError(f"fir_algo {self.fir_algo!r}: expected 'auto', 'fft' or 'mfma'")
        if self.fir_algo == "mfma" or self.kernLength > _lib.plain("eav_eegnet_fir_fft_max_taps"):
            return False
        return self.fir_algo == "fft" or self.Samples >= 1408

    def _use_conv_fft(self, B):
        """separableConv (forward, data and weight gradient) in the frequency domain (csrc/eegnet_conv64_fft.hip: per-bin
        [128 x 128] GEMMs instead of a 1024-deep contraction, 6x fewer multiply-adds) when the batch is large enough to
        amortise its extra launches; the direct fp32-MFMA kernels otherwise (the reference's own [32,1,30,500]).
        `conv_algo` / EAV_CONV_ALGO: "auto" (default), "fft", "mfma"."""
        if self.conv_algo not in ("auto", "fft", "mfma"):
            raise ValueError(f"conv_algo {self.conv_algo!r}: expected 'auto', 'fft' or 'mfma'")
        if self.conv_algo == "mfma":
            return False
        return self.conv_algo == "fft" or B * (self.Samples // 4) >= 40000

    def _ensure_flat(self):
        p0 = self.firstConv.weight
        if self._flat is None or self._flat[0].device != p0.device or getattr(p0, "_eav_flat", None) is None \
                or p0.data_ptr() != self._flat[0].data_ptr():
            ordered = dict(self.named_parameters())
            assert list(ordered) == _PARAM_ORDER, list(ordered)
            self._flat = flatten_parameters(self)

    def _params(self):
        n = dict(self.named_parameters())
        return [n[k] for k in _PARAM_ORDER]

    def set_dropout_masks(self, masks):
        """Testing hook: explicit uint8 keep-masks instead of the counter-based generator."""
        self._dropout_masks = masks

    def forward(self, x):
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise _lib.EavError("eav_amd.EEGNet_tor runs on an MI355X only: move the model and the input to the "
                                "ROCm device (there is no CPU fallback)")
        if x.dim() == 3:
            x = x.unsqueeze(1)
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != self.Chans or x.shape[3] != self.Samples:
            raise ValueError(f"expected input [B,1,{self.Chans},{self.Samples}], got {tuple(x.shape)}")
        if self.firstConv.weight.device != x.device:
            raise _lib.EavError("model and input are on different devices")
        x = x.contiguous().float()
        self._ensure_flat()
        # no-grad evaluation (validate(), EEGNet_tor.py:118-135): block 1 runs as one fused kernel, see _launch_forward
        self._infer = (not torch.is_grad_enabled()) and (not self.training)
        return _EEGNetFn.apply(x, self, *self._params())

    def forward_indexed(self, data, idx):
        """forward(data[idx]) without materialising data[idx] (Trainer_uni's per-step batch assembly,
        EEGNet_tor.py:100-101): `data` [N,1,Chans,Samples] fp32 contiguous on the device, `idx` device int64 [B]."""
        if data.dim() != 4 or data.shape[1] != 1 or data.shape[2] != self.Chans or data.shape[3] != self.Samples or \
                not data.is_cuda or data.dtype != torch.float32 or not data.is_contiguous():
            raise ValueError(f"expected a contiguous fp32 device array [N,1,{self.Chans},{self.Samples}]")
        if idx.dtype != torch.int64 or idx.device != data.device or idx.dim() != 1:
            raise ValueError("idx must be a 1-D int64 tensor on the data's device")
        if self._generic:
            out = torch.empty((idx.numel(),) + tuple(data.shape[1:]), dtype=torch.float32, device=data.device)
            _lib.call("eav_gather_rows", data.data_ptr(), idx.data_ptr(), out.data_ptr(), idx.numel(), data[0].numel(),
                      _lib.stream_ptr())
            return self.forward(out)
        if self.firstConv.weight.device != data.device:
            raise _lib.EavError("model and input are on different devices")
        self._ensure_flat()
        self._infer = (not torch.is_grad_enabled()) and (not self.training)
        return _EEGNetFn.apply(IndexedBatch(data, idx), self, *self._params())

    # ------------------------------------------------------------------ kernels
    def _call(self, name, *args):
        """_lib.call, optionally bracketed by HIP events on the launch stream (bench.py's live
        per-kernel timing of the dominant kernels)."""
        ev = self.kernel_events
        if ev is not None and name in ev:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _lib.call(name, *args)
            b.record()
            ev[name].append((a, b))
        else:
            _lib.call(name, *args)

    def _workspace(self, B, dev):
        """One workspace per problem size; the ones a hipGraph was captured with are pinned for the life of the model,
        the others share a small replaceable set (cached_workspace)."""
        key = (B, self.Chans, self.Samples, str(dev))
        return cached_workspace(self._wss, key, lambda: _Workspace(B, self.Chans, self.Samples, self.kernLength,
                                                                   self.nb_classes, dev))

    def _launch_forward(self, x):
        if self._generic:
            return self._launch_forward_generic(x)
        L, P, st = self._call, _lib.ptr, _lib.stream_ptr()
        B, C, S, K, nb = x.shape[0], self.Chans, self.Samples, self.kernLength, self.nb_classes
        ws = self._ws = self._workspace(B, x.device)
        training = bool(self.training)
        w1, g1w, g1b, w2, g2w, g2b, w3, g3w, g3b, wd, bd = [P(p) for p in self._params()]
        bn1, bn2, bn3 = self.firstBN, self.depthwiseBN, self.separableBN
        drop = self.dropoutRate if training else 0.0
        if self.spatial_dropout:
            drop = -drop
        masks = self._dropout_masks if training else None
        self._token += 1
        # dropout stream: effective seed = base + 2 * (device-resident count of training forwards) - no host
        # argument changes from step to step, so the whole step can be replayed from a hipGraph
        seed1, seed2 = self.dropout_seed, self.dropout_seed + 1
        cnt = None
        if drop != 0.0 and masks is None:
            if self._fwd_counter is None or self._fwd_counter.device != x.device:
                self._fwd_counter = torch.zeros((), dtype=torch.int64, device=x.device)
            cnt = P(self._fwd_counter)
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None

        def bnfin(part, nparts, nch, count, gw, gb, bn, buf):
            b0 = P(buf)
            L("eav_bn_finalize", P(part), nparts, nch, float(count), gw, gb, P(bn.running_mean), P(bn.running_var),
              int(training), float(bn.momentum), float(bn.eps), b0, b0 + 4 * nch, b0 + 8 * nch, b0 + 12 * nch, st)

        if self.fir_precision != "fp32":
            raise ValueError(f"fir_precision {self.fir_precision!r}: only 'fp32' exists (the split mode was retired)")
        counters = [cnt] + ([P(bn.num_batches_tracked) for bn in (bn1, bn2, bn3)] if training else [None] * 3)
        begin, self._step_begin = self._step_begin, None
        if not self._use_conv_fft(B):    # (the frequency-domain separableConv takes the weight tensor as it is)
            # per-step prologue, one launch: the transposed separableConv weights of the direct kernels + the dropout step
            # counter and the three BatchNorm step counters (nn.BatchNorm2d's num_batches_tracked)
            L("eav_eegnet_step_prologue", w3, P(ws.wTf), P(ws.wTb), *counters, st)
            self._step_begin = begin     # (not taken: GraphStep issues its own launches)
        elif begin is not None:
            # ... and, in a captured step (GraphStep), the optimiser's step count and the gather of the batch's labels with them
            L("eav_step_begin", *counters, *begin, st)
        elif cnt is not None or training:
            # library kernels, no torch op inside a captured step
            L("eav_counter_inc4", *counters, st)
        np_fir = ws.np_fir
        infer = self.fused_eval and self._infer and not training and S % 4 == 0
        if infer:
            # validate(): x -> block-1 output in ONE kernel (FIR -> firstBN -> ELU -> depthwiseConv -> depthwiseBN -> ELU ->
            # AvgPool4; BatchNorms on running statistics): y1 (614 MB at [64,1,30,10000]) and z are never written
            bnfin(ws.part_fir, ws.np_fir, 8, B * C * S, g1w, g1b, bn1, ws.bn1)          # eval mode: running statistics only
            bnfin(ws.part_dw, B * ws.nchunk, 64, B * S, g2w, g2b, bn2, ws.bn2)
            if isinstance(x, IndexedBatch):
                L("eav_eegnet_block1_infer", P(x.data), P(x.idx), w1, P(ws.bn1), w2, P(ws.bn2), P(ws.p2), B, C, S, K, st)
            else:
                L("eav_eegnet_block1_infer", P(x), None, w1, P(ws.bn1), w2, P(ws.bn2), P(ws.p2), B, C, S, K, st)
        elif self._use_fft():
            np_fir = ws.np_fir_fft
            pf = P(ws.part_fir) if training else None      # eval mode: firstBN needs no batch statistics
            if isinstance(x, IndexedBatch):
                L("eav_eegnet_fir_fwd_fft", P(x.data), P(x.idx), w1, P(ws.y1), pf, B, C, S, K, st)
            else:
                L("eav_eegnet_fir_fwd_fft", P(x), None, w1, P(ws.y1), pf, B, C, S, K, st)
        else:
            if isinstance(x, IndexedBatch):
                L("eav_eegnet_fir_fwd_indexed", P(x.data), P(x.idx), w1, P(ws.y1), P(ws.part_fir), B, C, S, K, st)
            else:
                L("eav_eegnet_fir_fwd", P(x), w1, P(ws.y1), P(ws.part_fir), B, C, S, K, st)
        if not infer:
            bnfin(ws.part_fir, np_fir, 8, B * C * S, g1w, g1b, bn1, ws.bn1)
            if not training and drop == 0.0 and m1 is None and S % 4 == 0:
                # eval-mode step (what 349 of the reference's 350 epochs run, Q4): depthwiseBN's scale / shift come from the
                # running statistics, i.e. they are known BEFORE the depthwise pass - which then leaves the pooled block-1
                # output too (z is still written: the backward forms dz from it); one launch and one pass over z less
                bnfin(ws.part_dw, B * ws.nchunk, 64, B * S, g2w, g2b, bn2, ws.bn2)
                L("eav_eegnet_dw_fwd_pool_eval", P(ws.y1), P(ws.bn1), w2, P(ws.z), P(ws.part_dw), P(ws.bn2), P(ws.p2), B, C, S,
                  st)
            else:
                L("eav_eegnet_dw_fwd", P(ws.y1), P(ws.bn1), w2, P(ws.z), P(ws.part_dw), B, C, S, st)
                bnfin(ws.part_dw, B * ws.nchunk, 64, B * S, g2w, g2b, bn2, ws.bn2)
                L("eav_bn_elu_pool_fwd", P(ws.z), P(ws.bn2), P(ws.p2), B, 64, S, 4, drop, seed1, m1, cnt, st)
        np_c3 = ws.np_c3
        if self._use_conv_fft(B):
            if ws.c64_ws is None:
                ws.c64_ws = torch.zeros(_lib.plain("eav_conv64_fft_ws_floats", B, ws.T2), dtype=torch.float32,
                                        device=ws.y1.device)
            np_c3 = ws.np_c3_fft
            L("eav_conv64_fft_fwd", P(ws.p2), w3, P(ws.u3), P(ws.part_c3), P(ws.c64_ws), B, ws.T2, 0, st)
        else:
            L("eav_conv64_fwd", P(ws.p2), P(ws.wTf), P(ws.u3), P(ws.part_c3), B, ws.T2, 7, st)
        bnfin(ws.part_c3, np_c3, 64, B * ws.T2, g3w, g3b, bn3, ws.bn3)
        L("eav_bn_elu_pool_fwd", P(ws.u3), P(ws.bn3), P(ws.p3), B, 64, ws.T2, 8, drop, seed2, m2, cnt, st)
        probs = torch.empty(B, nb, dtype=torch.float32, device=x.device)   # fresh per forward: returned, kept for backward
        L("eav_dense_softmax_fwd", P(ws.p3), wd, bd, None, P(probs), B, ws.NF, nb, st)
        if self.apply_max_norm:  # the forward hooks of the reference (:33-34, :47-48), intended meaning: one launch
            L("eav_renorm_rows2", w2, 64, C, wd, nb, ws.NF, self.norm_rate, st)
        self._saved = (self._token, x, training, drop, seed1, seed2, masks, cnt, ws, probs)
        return self._token

    # ------------------------------------------------------------------ generic widths (csrc/eegnet_canon.hip)
    def _launch_forward_generic(self, x):
        """EEGNet_tor.forward (:50-67) for any F1 / D / F2 / kernLength / Chans the reference constructor accepts, on the
        run-time-parametrised kernels: eav_tconv_* (firstConv), eav_spatial_* with the ELU flag (firstBN -> ELU ->
        depthwiseConv), eav_dconv_* (the dense "separableConv"), the shared BN -> ELU -> pool -> dropout and classifier
        kernels.  Same quirks as the specialised path: max-norm after the forward (Q1/Q2), softmax output (Q3)."""
        L, P, st = self._call, _lib.ptr, _lib.stream_ptr()
        B, C, S, K, nb = x.shape[0], self.Chans, self.Samples, self.kernLength, self.nb_classes
        F1, D, F2 = self.F1, self.D, self.F2
        C2 = F1 * D
        key = ("generic", B, C, S, str(x.device))
        ws = self._ws = cached_workspace(self._wss, key, lambda: _GenericWorkspace(self, B, x.device))
        training = bool(self.training)
        w1, g1w, g1b, w2, g2w, g2b, w3, g3w, g3b, wd, bd = [P(p) for p in self._params()]
        bn1, bn2, bn3 = self.firstBN, self.depthwiseBN, self.separableBN
        drop = self.dropoutRate if training else 0.0
        if self.spatial_dropout:
            drop = -drop
        masks = self._dropout_masks if training else None
        self._token += 1
        seed1, seed2 = self.dropout_seed, self.dropout_seed + 1
        cnt = None
        if drop != 0.0 and masks is None:
            if self._fwd_counter is None or self._fwd_counter.device != x.device:
                self._fwd_counter = torch.zeros((), dtype=torch.int64, device=x.device)
            cnt = P(self._fwd_counter)
        if cnt is not None or training:
            L("eav_counter_inc4", cnt, *([P(bn.num_batches_tracked) for bn in (bn1, bn2, bn3)] if training else [None] * 3),
              st)
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None

        def bnfin(part, nparts, nch, count, gw, gb, bn, buf):
            b0 = P(buf)
            L("eav_bn_finalize", P(part), nparts, nch, float(count), gw, gb, P(bn.running_mean), P(bn.running_var),
              int(training), float(bn.momentum), float(bn.eps), b0, b0 + 4 * nch, b0 + 8 * nch, b0 + 12 * nch, st)

        L("eav_tconv_fwd", P(x), w1, P(ws.y1), P(ws.part_t), B, C, S, F1, K, st)                       # :51
        bnfin(ws.part_t, ws.np_t, F1, B * C * S, g1w, g1b, bn1, ws.bn1)                                 # :52
        L("eav_spatial_fwd", P(ws.y1), P(ws.bn1), w2, P(ws.z), P(ws.part_s), B, C, S, F1, D, 1, st)     # :53-54
        bnfin(ws.part_s, ws.np_s, C2, B * S, g2w, g2b, bn2, ws.bn2)                                     # :55
        L("eav_bn_elu_pool_fwd", P(ws.z), P(ws.bn2), P(ws.p2), B, C2, S, 4, drop, seed1, m1, cnt, st)   # :56-58
        L("eav_dconv_fwd", P(ws.p2), w3, P(ws.u3), P(ws.part_c), B, C2, F2, ws.T2, 16, 0, st)           # :59
        bnfin(ws.part_c, ws.np_c, F2, B * ws.T2, g3w, g3b, bn3, ws.bn3)                                 # :60
        L("eav_bn_elu_pool_fwd", P(ws.u3), P(ws.bn3), P(ws.p3), B, F2, ws.T2, 8, drop, seed2, m2, cnt, st)   # :61-63
        probs = torch.empty(B, nb, dtype=torch.float32, device=x.device)
        L("eav_dense_softmax_fwd", P(ws.p3), wd, bd, None, P(probs), B, ws.NF, nb, st)                  # :64-66
        if self.apply_max_norm:
            L("eav_renorm_rows", w2, C2, C, self.norm_rate, st)
            L("eav_renorm_rows", wd, nb, ws.NF, self.norm_rate, st)
        self._saved = (self._token, x, training, drop, seed1, seed2, masks, cnt, False, ws, probs)
        return self._token

    def _launch_backward_generic(self, dprobs):
        L, P, st = self._call, _lib.ptr, _lib.stream_ptr()
        _, x, training, drop, seed1, seed2, masks, cnt, _, ws, probs = self._saved
        B, C, S, K, nb = x.shape[0], self.Chans, self.Samples, self.kernLength, self.nb_classes
        F1, D, F2 = self.F1, self.D, self.F2
        C2, T2, NF = F1 * D, ws.T2, ws.NF
        flat, gflat, offs = self._flat
        g = {k: gflat[offs[k][0]:offs[k][0] + offs[k][1]] for k in _PARAM_ORDER}
        w2, w3, wd = P(self.depthwiseConv.weight), P(self.separableConv.weight), P(self.dense.weight)
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None
        tr = int(training)
        L("eav_dense_softmax_bwd", P(dprobs), P(probs), P(ws.p3), wd, P(g["dense.weight"]), P(g["dense.bias"]),
          P(ws.dp3), B, NF, nb, st)
        b3 = P(ws.bn3)
        L("eav_bn_elu_pool_bwd_reduce", P(ws.dp3), P(ws.u3), b3, P(ws.part_pb), B, F2, T2, 8, drop, seed2, m2, cnt, st)
        L("eav_bn_bwd_finalize", P(ws.part_pb), B, F2, float(B * T2), tr, P(g["separableBN.weight"]),
          P(g["separableBN.bias"]), b3 + 16 * F2, b3 + 20 * F2, st)
        L("eav_bn_elu_pool_bwd_apply", P(ws.dp3), P(ws.u3), b3, b3 + 16 * F2, P(ws.du3), B, F2, T2, 8, drop, seed2, m2,
          cnt, st)
        # the dense temporal conv: data gradient = the same kernel on the transposed, tap-flipped weights
        L("eav_dconv_fwd", P(ws.du3), w3, P(ws.dp2), None, B, F2, C2, T2, 16, 1, st)
        L("eav_dconv_wgrad", P(ws.du3), P(ws.p2), P(ws.part_cw), B, C2, F2, T2, 16, st)
        n3 = F2 * C2 * 16
        L("eav_reduce_partials", P(ws.part_cw), B, n3, n3, 1.0, P(g["separableConv.weight"]), st)
        b2 = P(ws.bn2)
        L("eav_bn_elu_pool_bwd_reduce", P(ws.dp2), P(ws.z), b2, P(ws.part_pb), B, C2, S, 4, drop, seed1, m1, cnt, st)
        L("eav_bn_bwd_finalize", P(ws.part_pb), B, C2, float(B * S), tr, P(g["depthwiseBN.weight"]),
          P(g["depthwiseBN.bias"]), b2 + 16 * C2, b2 + 20 * C2, st)
        L("eav_bn_elu_pool_bwd_apply", P(ws.dp2), P(ws.z), b2, b2 + 16 * C2, P(ws.dz), B, C2, S, 4, drop, seed1, m1,
          cnt, st)
        # depthwiseConv <- ELU <- firstBN (post-renorm depthwise weight, Q2), then the firstConv weight gradient
        b1 = P(ws.bn1)
        L("eav_spatial_bwd", P(ws.y1), P(ws.dz), b1, w2, P(ws.g1), P(ws.part_sst), P(ws.part_sw), B, C, S, F1, D, 1, st)
        L("eav_reduce_partials", P(ws.part_sw), ws.np_s, C2 * C, C2 * C, 1.0, P(g["depthwiseConv.weight"]), st)
        L("eav_bn_bwd_finalize", P(ws.part_sst), ws.np_s, F1, float(B * C * S), tr, P(g["firstBN.weight"]),
          P(g["firstBN.bias"]), b1 + 16 * F1, b1 + 20 * F1, st)
        L("eav_tconv_wgrad", P(x), P(ws.y1), P(ws.g1), b1, P(ws.part_tw), B, C, S, F1, K, st)
        L("eav_reduce_partials", P(ws.part_tw), ws.np_tw, F1 * K, F1 * K, 1.0, P(g["firstConv.weight"]), st)
        named = dict(self.named_parameters())
        return [g[k].view(named[k].shape) if named[k].requires_grad else None for k in _PARAM_ORDER]

    def _launch_backward(self, dprobs, token):
        if self._saved is None or self._saved[0] != token:
            raise _lib.EavError("EEGNet_tor.backward: the activations of this forward were overwritten by a later "
                                "forward (one outstanding forward per backward)")
        if self._generic:
            return self._launch_backward_generic(dprobs)
        L, P, st = self._call, _lib.ptr, _lib.stream_ptr()
        _, x, training, drop, seed1, seed2, masks, cnt, ws, probs = self._saved
        B, C, S, K, nb = x.shape[0], self.Chans, self.Samples, self.kernLength, self.nb_classes
        T2, NF = ws.T2, ws.NF
        flat, gflat, offs = self._flat
        g = {k: gflat[offs[k][0]:offs[k][0] + offs[k][1]] for k in _PARAM_ORDER}
        w2, wd = P(self.depthwiseConv.weight), P(self.dense.weight)
        m1 = P(masks[0]) if masks is not None else None
        m2 = P(masks[1]) if masks is not None else None
        tr = int(training)

        L("eav_dense_softmax_bwd", P(dprobs), P(probs), P(ws.p3), wd, P(g["dense.weight"]), P(g["dense.bias"]),
          P(ws.dp3), B, NF, nb, st)
        # block 2: Dropout <- AvgPool8 <- ELU <- separableBN
        b3 = P(ws.bn3)
        if training:
            L("eav_bn_elu_pool_bwd_reduce", P(ws.dp3), P(ws.u3), b3, P(ws.part_pb), B, 64, T2, 8, drop, seed2, m2, cnt, st)
            L("eav_bn_bwd_finalize", P(ws.part_pb), B, 64, float(B * T2), tr, P(g["separableBN.weight"]),
              P(g["separableBN.bias"]), b3 + 4 * 256, b3 + 4 * 320, st)
            L("eav_bn_elu_pool_bwd_apply", P(ws.dp3), P(ws.u3), b3, b3 + 4 * 256, P(ws.du3), B, 64, T2, 8, drop, seed2, m2,
              cnt, st)
        else:
            # eval-mode step (Q4: every epoch after the first): BatchNorm backward is a plain scale, so the gradient and the
            # sums for the BatchNorm weight / bias leave from ONE pass over u3 / dp3
            L("eav_bn_elu_pool_bwd_eval", P(ws.dp3), P(ws.u3), b3, P(ws.du3), P(ws.part_pb), B, 64, T2, 8, drop, seed2, m2,
              cnt, st)
            L("eav_bn_bwd_finalize", P(ws.part_pb), B, 64, float(B * T2), tr, P(g["separableBN.weight"]),
              P(g["separableBN.bias"]), b3 + 4 * 256, b3 + 4 * 320, st)
        # separableConv: data gradient (flipped/transposed taps, pad 8) and weight gradient
        if self._use_conv_fft(B):
            # (bwd = 2: the filter spectra of the data gradient were prepared by this step's forward call)
            L("eav_conv64_fft_fwd", P(ws.du3), P(self.separableConv.weight), P(ws.dp2), None, P(ws.c64_ws), B, T2, 2, st)
        else:
            L("eav_conv64_fwd", P(ws.du3), P(ws.wTb), P(ws.dp2), None, B, T2, 8, st)
        if self._use_conv_fft(B):
            L("eav_conv64_fft_wgrad", P(ws.du3), P(g["separableConv.weight"]), P(ws.c64_ws), B, T2, st)
        else:
            L("eav_conv64_wgrad", P(ws.du3), P(ws.p2), P(ws.part_cw), B, T2, 7, st)
            L("eav_reduce_partials", P(ws.part_cw), ws.np_cw, 65536, 65536, 1.0, P(g["separableConv.weight"]), st)
        # block 1 tail: Dropout <- AvgPool4 <- ELU <- depthwiseBN
        b2 = P(ws.bn2)
        # depthwiseConv <- ELU <- firstBN (uses the post-renorm depthwise weight, Q2)
        b1 = P(ws.bn1)
        # dz = backward of BN2 -> ELU -> pool -> dropout is formed inside dw_bwd: no dz tensor in HBM
        if training:
            L("eav_bn_elu_pool_bwd_reduce", P(ws.dp2), P(ws.z), b2, P(ws.part_pb), B, 64, S, 4, drop, seed1, m1, cnt, st)
            L("eav_bn_bwd_finalize", P(ws.part_pb), B, 64, float(B * S), tr, P(g["depthwiseBN.weight"]),
              P(g["depthwiseBN.bias"]), b2 + 4 * 256, b2 + 4 * 320, st)
            L("eav_eegnet_dw_bwd_fused", P(ws.y1), P(ws.z), P(ws.dp2), b2, b1, w2, P(ws.g1), P(ws.part_dst),
              P(ws.part_dw2), B, C, S, drop, seed1, m1, cnt, st)
        else:
            # eval-mode step: no sums are needed before dz = scale2 g - they leave from the depthwise pass itself (no reduce
            # launch, no extra read of z and dp2)
            L("eav_eegnet_dw_bwd_fused_eval", P(ws.y1), P(ws.z), P(ws.dp2), b2, b1, w2, P(ws.g1), P(ws.part_dst),
              P(ws.part_dw2), P(ws.part_dw), B, C, S, drop, seed1, m1, cnt, st)
        # the finishing work behind the depthwise pass in ONE launch: depthwiseConv.weight (fixed-order sum of the partial
        # rows), firstBN's backward sums -> its gradients and the m1 / m2 the FIR weight gradient folds in, and - eval-mode
        # step - depthwiseBN's, which left from the same pass
        bn2job = (None, 0, 0, 0.0, 0, None, None, None, None) if training else \
            (P(ws.part_dw), B * ws.nchunk, 64, float(B * S), tr, P(g["depthwiseBN.weight"]), P(g["depthwiseBN.bias"]),
             b2 + 4 * 256, b2 + 4 * 320)
        L("eav_reduce_and_bn_bwd_finalize", P(ws.part_dw2), B * ws.nchunk, 64 * C, 64 * C, P(g["depthwiseConv.weight"]),
          P(ws.part_dst), B * ws.nchunk, 8, float(B * C * S), tr, P(g["firstBN.weight"]), P(g["firstBN.bias"]), b1 + 4 * 32,
          b1 + 4 * 40, *bn2job, st)
        # firstConv weight gradient (BN backward folded into the operand staging)
        if self._use_fft():
            if ws.fft_ws is None:
                ws.fft_ws = torch.empty(_lib.plain("eav_eegnet_fir_wgrad_fft_ws_floats", B, C, S), dtype=torch.float32,
                                        device=ws.y1.device)
            xi = isinstance(x, IndexedBatch)
            L("eav_eegnet_fir_wgrad_fft", P(x.data) if xi else P(x), P(x.idx) if xi else None,
              P(ws.y1) if training else None, P(ws.g1), b1, P(ws.fft_ws), P(g["firstConv.weight"]), B, C, S, K, st)
        else:
            # eval-mode training (every epoch after the first, Q4): BatchNorm backward is a plain scale, y1 is not needed
            if isinstance(x, IndexedBatch):
                L("eav_eegnet_fir_wgrad_indexed", P(x.data), P(x.idx), P(ws.y1) if training else None, P(ws.g1), b1,
                  P(ws.part_fw), B, C, S, K, st)
            else:
                L("eav_eegnet_fir_wgrad", P(x), P(ws.y1) if training else None, P(ws.g1), b1, P(ws.part_fw), B, C, S, K,
                  st)
            L("eav_reduce_partials", P(ws.part_fw), ws.np_fw, 8 * K, 8 * K, 1.0, P(g["firstConv.weight"]), st)
        named = dict(self.named_parameters())
        return [g[k].view(named[k].shape) if named[k].requires_grad else None for k in _PARAM_ORDER]


def gather_batch(xs, ys, idx_dev):
    """(xs[idx], ys[idx]) assembled in HBM by the library's gather kernels (eav_gather_rows / eav_gather_i64)."""
    n = idx_dev.numel()
    if not xs.is_cuda:   # host tensors (CPU-side unit tests of the loader only)
        return xs.index_select(0, idx_dev), ys.index_select(0, idx_dev)
    data = torch.empty((n,) + tuple(xs.shape[1:]), dtype=torch.float32, device=xs.device)
    targets = torch.empty(n, dtype=torch.long, device=xs.device)
    st = _lib.stream_ptr()
    _lib.call("eav_gather_rows", xs.data_ptr(), idx_dev.data_ptr(), data.data_ptr(), n, xs[0].numel(), st)
    _lib.call("eav_gather_i64", ys.data_ptr(), idx_dev.data_ptr(), targets.data_ptr(), n, st)
    return data, targets


class GraphStep:
    """One EEGNet training step (batch gather, forward, CE, backward, [grad sync], fused Adam) captured in a
    hipGraph and replayed: at the reference's own shape ([32,1,30,500]) the step is ~35 tiny kernels and is
    bound by launch overhead, not by the GPU.  Everything that varies between steps lives in device memory
    (batch indices, dropout counter, Adam step count), so a replay needs no host-side argument updates."""

    def __init__(self, model, optimizer, criterion, xs, ys, batch, grad_sync=None, post_step=None):
        if not getattr(optimizer, "capturable", False):
            raise _lib.EavError("GraphStep needs FusedAdam(capturable=True)")
        self.model, self.batch, self.grad_sync = model, batch, grad_sync
        dev = xs.device
        self.idx = torch.zeros(batch, dtype=torch.long, device=dev)

        def compute():       # batch gather + forward + loss + backward
            if hasattr(model, "forward_indexed") and xs.is_cuda and xs.dim() == 4 and xs.is_contiguous():
                # EEGNet_tor reads the batch in place through the index vector: only the labels are gathered - by the
                # model's own step-counter launch where it has one (eav_step_begin: the label gather, the optimiser's step
                # count and the dropout / BatchNorm counters in ONE graph node instead of three)
                targets = torch.empty(batch, dtype=torch.long, device=dev)
                merged = False
                optimizer.step_counted = False       # (a step that failed between its two halves must not leave the flag set)
                if hasattr(model, "_step_begin") and not getattr(model, "_generic", False):
                    cnt = optimizer.device_step_counter()
                    model._step_begin = (cnt.data_ptr(), ys.data_ptr(), self.idx.data_ptr(), targets.data_ptr(), batch)
                    scores = model.forward_indexed(xs, self.idx)
                    merged = model._step_begin is None
                    model._step_begin = None
                    if merged:
                        optimizer.step_counted = True
                else:
                    scores = model.forward_indexed(xs, self.idx)
                if not merged:
                    _lib.call("eav_gather_i64", ys.data_ptr(), self.idx.data_ptr(), targets.data_ptr(), batch, _lib.stream_ptr())
            else:
                data, targets = gather_batch(xs, ys, self.idx)
                scores = model(data)
            loss = criterion(scores, targets)
            optimizer.zero_grad(set_to_none=True)
            loss.backward(gradient=unit_gradient(loss.device))      # no ones_like fill, no scaling launch
            return scores, loss

        def update():        # fused Adam (+ e.g. the max-norm projection of Transformer_EEG.py:195-199)
            optimizer.step()
            if post_step is not None:
                post_step()

        self.warm_steps = 0
        self.graph = None          # compute (and, without a grad_sync, update) graph
        self.graph_update = None   # data parallel: the update is its own graph, the all-reduce runs between the two
        self._compute, self._update = compute, update

    def _eager(self):
        scores, loss = self._compute()
        if self.grad_sync is not None:
            self.grad_sync()
        self._update()
        return scores, loss

    def run(self, idx):
        """idx: sequence of `batch` dataset indices.  The first two calls run eagerly (they are real training
        steps), the third is captured, later ones are replays.  Under data parallelism (grad_sync) the collective is
        not captured: replay(compute) -> all-reduce on the live stream -> replay(update)."""
        self.idx.copy_(torch.as_tensor(idx, dtype=torch.long))     # pageable source: staged, no host race
        if self.graph is None:
            if self.warm_steps < 2:
                self.warm_steps += 1
                scores, loss = self._eager()
                return scores.detach(), loss.detach()     # keep no reference to the autograd graph
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                scores, loss = self._compute()
                self.scores, self.loss = scores.detach(), loss.detach()
                if self.grad_sync is None:
                    self._update()
            del scores, loss
            if getattr(self.model, "_ws", None) is not None:
                self.model._ws.pinned = True       # the graph holds this workspace's raw pointers (cached_workspace)
            if self.grad_sync is not None:
                # the gradients the update graph reads live in the model's flat buffer (static address); capture the
                # update on its own (its launches are recorded, not executed)
                self.graph_update = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_update, pool=self.graph.pool()):
                    self._update()
            # capture does not execute: fall through to a replay so that this call is a real step too
        self.graph.replay()
        if self.grad_sync is not None:
            self.grad_sync()
            self.graph_update.replay()
        return self.scores, self.loss


# ----------------------------------------------------------------------------- data plumbing
class DeviceLoader:
    """DataLoader-shaped iterator over a device-resident TensorDataset.

    The reference builds ``DataLoader(TensorDataset(x, y), batch_size, shuffle)`` on
    the host and copies every batch to the device inside the loop
    (EEGNet_tor.py:91-94,100-101).  Here the whole split lives in HBM once and a
    batch is assembled by one gather; the *index order* is produced by the same
    torch samplers (RandomSampler / SequentialSampler + BatchSampler), consuming
    the torch RNG exactly as ``iter(DataLoader)`` does, so a seeded run visits
    the same batches as the reference.
    """

    def __init__(self, x, y, batch_size, shuffle, device):
        from torch.utils.data import TensorDataset
        self.x = torch.as_tensor(x, dtype=torch.float32).to(device).contiguous()
        self.y = torch.as_tensor(y, dtype=torch.long).to(device).contiguous()
        self.dataset = TensorDataset(self.x, self.y)
        self.batch_size, self.shuffle, self.device = batch_size, shuffle, device
        self.order_override = None  # tests: list of index arrays, one per epoch

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def index_batches(self):
        """The index lists iter(DataLoader) would visit (same samplers, same torch RNG consumption)."""
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        n = len(self.dataset)
        # iter(DataLoader) draws its base seed first (torch/utils/data/dataloader.py, _BaseDataLoaderIter)
        torch.empty((), dtype=torch.int64).random_()
        if self.order_override:
            order = [int(i) for i in self.order_override.pop(0)]
            return [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        sampler = RandomSampler(range(n)) if self.shuffle else SequentialSampler(range(n))
        return list(BatchSampler(sampler, self.batch_size, drop_last=False))

    def gather(self, idx):
        if idx[-1] - idx[0] == len(idx) - 1 and all(b - a == 1 for a, b in zip(idx, idx[1:])):
            return self.x[idx[0]:idx[-1] + 1], self.y[idx[0]:idx[-1] + 1]
        return gather_batch(self.x, self.y, torch.as_tensor(idx, dtype=torch.long, device=self.device))

    def gather_labels(self, idx):
        """The labels of a batch only (a step that already holds the batch's features needs no copy of x)."""
        if idx[-1] - idx[0] == len(idx) - 1 and all(b - a == 1 for a, b in zip(idx, idx[1:])):
            return self.y[idx[0]:idx[-1] + 1]
        i = torch.as_tensor(idx, dtype=torch.long, device=self.device)
        out = torch.empty(len(idx), dtype=torch.long, device=self.device)
        _lib.call("eav_gather_i64", self.y.data_ptr(), i.data_ptr(), out.data_ptr(), len(idx), _lib.stream_ptr())
        return out

    def __iter__(self):
        for idx in self.index_batches():
            yield self.gather(idx)


class Trainer_uni:
    def __init__(self, model, data, lr=1e-4, batch_size=32, num_epochs=10, device=None):
        self.lr = lr
        self.batch_size = batch_size
        self.num_epochs = num_epochs
        self.device = device if device else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.device = torch.device(self.device)
        if self.device.type != "cuda":
            raise _lib.EavError("eav_amd.Trainer_uni needs an MI355X (torch device 'cuda' on ROCm); no CPU fallback")
        self.tr_x, self.tr_y, self.te_x, self.te_y = data
        self.train_dataloader = self._prepare_dataloader(self.tr_x, self.tr_y, shuffle=True)
        self.test_dataloader = self._prepare_dataloader(self.te_x, self.te_y, shuffle=False)

        self.model = model
        self.criterion = CrossEntropyLoss()                       # EEGNet_tor.py:81
        self.optimizer = FusedAdam(self.model.parameters(), lr=self.lr, capturable=True)   # :82 (Adam, wd 0)
        # :86-88 wraps in nn.DataParallel when several GPUs are visible; here multi-GPU is one
        # process per GPU with an RCCL gradient all-reduce (eav_amd.dist), enabled by the launcher.
        self.model.to(self.device)
        self.grad_sync = None  # set by eav_amd.dist.attach(trainer) under torchrun
        self.use_graph = True  # replay full-size batches from a hipGraph (partial batches run eagerly)
        self._graphs = {}

    def _prepare_dataloader(self, x, y, shuffle=False):
        return DeviceLoader(x, y, self.batch_size, shuffle, self.device)

    def train(self):
        self.model.train()  # once, before the epoch loop - as the reference (:97)
        dl = self.train_dataloader
        for epoch in range(self.num_epochs):
            for batch_idx, idx in enumerate(dl.index_batches()):
                if self.use_graph and len(idx) == self.batch_size:
                    # one captured graph per (batch size, BN mode): epochs >= 2 train in eval mode (Q4)
                    key = (len(idx), bool(self.model.training))
                    if key not in self._graphs:
                        self._graphs[key] = GraphStep(self.model, self.optimizer, self.criterion, dl.x, dl.y,
                                                      len(idx), self.grad_sync)
                    scores, loss = self._graphs[key].run(idx)
                else:
                    data, targets = dl.gather(idx)
                    scores = self.model(data)
                    loss = self.criterion(scores, targets)
                    self.optimizer.zero_grad()
                    loss.backward()
                    if self.grad_sync is not None:
                        self.grad_sync()
                    self.optimizer.step()
                if batch_idx % 100 == 0:
                    print(f"Epoch [{epoch+1}/{self.num_epochs}], Step [{batch_idx}/{len(self.train_dataloader)}], "
                          f"Loss: {loss.item():.4f}")
            self.criterion.check()        # out-of-range labels recorded by the captured steps of this epoch
            if self.test_dataloader:
                self.validate()

    def validate(self):
        self.model.eval()
        # EEGNet_tor.py:118-135 reads loss.item() and the hit count back after every batch; here both stay on the device
        # (one slot per batch, one hit counter) and are read once - the sums are formed in the reference's order
        nb = len(self.test_dataloader)
        losses = torch.zeros(max(nb, 1), dtype=torch.float32, device=self.device)
        correct = torch.zeros((), dtype=torch.int32, device=self.device)
        with torch.no_grad():
            for k, (data, targets) in enumerate(self.test_dataloader):
                scores = self.model(data)
                self.criterion.accumulate(scores, targets, losses[k], correct)
        total_loss = 0
        for v in losses[:nb].cpu().tolist():
            total_loss += v
        total_correct = int(correct.item())
        self.criterion.check()
        avg_loss = total_loss / len(self.test_dataloader)
        accuracy = total_correct / len(self.test_dataloader.dataset)
        print(f"Validation - Loss: {avg_loss:.4f}, Accuracy: {accuracy:.4f}")
