"""Fused Adam / AdamW over flat parameter storage + the cross-entropy criterion.

Host-side mirrors of what the reference trainers construct:
  * ``optim.Adam(model.parameters(), lr)``            CNN_torch/EEGNet_tor.py:82
  * ``optim.AdamW(model.parameters(), lr)``           Transformer_Audio.py:30, Transformer_Vision.py:36
    (weight_decay is torch's default 0.01 - the trainer's ctor argument is ignored)
  * ``nn.CrossEntropyLoss()``                          EEGNet_tor.py:81, Transformer_Audio.py:31

The arithmetic runs in libeav_hip.so (``eav_adam_step`` / ``eav_ce_fwd_bwd``).
Parameters that live in one flat buffer (``flatten_parameters``) are updated
with one launch per run of tensors that share a step count - the frozen and
unfrozen fine-tuning phases give the head and the backbone different counts
(SURVEY Q11), and tensors whose ``grad`` is None are skipped like torch does.
"""
from __future__ import annotations

import torch

from . import _lib


def flatten_parameters(module: torch.nn.Module, order=None, pad_after=None):
    """Re-home every parameter of ``module`` as a view of one flat fp32 device
    buffer (and allocate a same-shaped flat gradient buffer).  Parameter objects
    are preserved, so optimisers created earlier stay valid.  Returns
    (flat_param, flat_grad, {name: (offset, numel)}).

    pad_after: {name: n} leaves n zero floats after that tensor in every flat buffer (e.g. to lay a [40,40] weight out
    as the first rows of a zero-padded [64,40] GEMM operand); the padding stays zero under FusedAdam."""
    params = [(n, p) for n, p in module.named_parameters()]
    if order is not None:  # explicit flat layout (e.g. q/k/v weights adjacent for one fused GEMM)
        named = dict(params)
        assert sorted(named) == sorted(order), "order must name every parameter exactly once"
        params = [(n, named[n]) for n in order]
    if not params:
        raise ValueError("module has no parameters")
    dev = params[0][1].device
    offsets, total = {}, 0
    for n, p in params:
        if p.dtype != torch.float32:
            raise TypeError(f"{n}: only fp32 parameters are supported")
        offsets[n] = (total, p.numel())
        total += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        if pad_after and n in pad_after:
            assert pad_after[n] % 4 == 0
            total += pad_after[n]
            p._eav_pad = pad_after[n]
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    gflat = torch.zeros(total, dtype=torch.float32, device=dev)
    for n, p in params:
        off, num = offsets[n]
        view = flat[off:off + num].view(p.shape)
        view.copy_(p.data)
        p.data = view
        p._eav_flat = (flat, gflat, off)
    return flat, gflat, offsets


class FusedAdam(torch.optim.Optimizer):
    """Adam (``decoupled=False``) / AdamW (``decoupled=True``) with torch's
    hyper-parameter names; one ``eav_adam_step`` launch per contiguous run."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=False,
                 capturable=False):
        """capturable=True keeps ONE step count in device memory (incremented by a kernel), so that step()
        can be captured in a hipGraph and replayed; it requires every parameter to receive a gradient on
        every step (true for EEGNet; the two-phase AST/ViT fine-tune uses the default host-side counts)."""
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, decoupled=decoupled)
        super().__init__(params, defaults)
        self._flat_state = {}
        self.capturable = capturable
        self._dev_step = None
        # capturable: set by a caller that increments the device step count itself at the start of the step, in a launch
        # it issues anyway (GraphStep: eav_step_begin) - step() then skips its own eav_counter_inc for that one call
        self.step_counted = False

    def _launch(self, p_ptr, g_ptr, m_ptr, v_ptr, n, group, step):
        b1, b2 = group["betas"]
        _lib.call("eav_adam_step", p_ptr, g_ptr, m_ptr, v_ptr, n, float(group["lr"]), float(b1), float(b2),
                  float(group["eps"]), float(group["weight_decay"]), int(step), int(bool(group["decoupled"])),
                  None if self._dev_step is None else self._dev_step.data_ptr(), _lib.stream_ptr())

    def device_step_counter(self):
        """The device-resident step count of a capturable optimiser (created on first use)."""
        if not self.capturable:
            return None
        if self._dev_step is None:
            dev = next(p for g in self.param_groups for p in g["params"]).device
            self._dev_step = torch.zeros((), dtype=torch.int64, device=dev)
        return self._dev_step

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        if self.capturable:
            if self._dev_step is None:
                dev = next(p for g in self.param_groups for p in g["params"]).device
                self._dev_step = torch.zeros((), dtype=torch.int64, device=dev)
            if self.step_counted:
                self.step_counted = False
            else:
                _lib.call("eav_counter_inc", self._dev_step.data_ptr(), _lib.stream_ptr())
        for group in self.param_groups:
            runs = []  # (p_ptr, g_ptr, m_ptr, v_ptr, numel, step) candidates for merging
            touched = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    if self.capturable:
                        raise _lib.EavError("FusedAdam(capturable=True): every parameter must have a gradient")
                    continue
                if not p.is_cuda:
                    raise _lib.EavError("FusedAdam needs parameters on a ROCm device (no CPU fallback)")
                if not g.is_contiguous():
                    g = g.contiguous()
                fl = getattr(p, "_eav_flat", None)
                if fl is not None and g.data_ptr() != fl[1].data_ptr() + 4 * fl[2]:
                    # the models return views of their flat gradient buffer, which each backward OVERWRITES; autograd
                    # adopts the view only when p.grad was None - anything else means torch accumulated into a copy
                    # (zero_grad(set_to_none=False), or two backward() calls before one step)
                    raise _lib.EavError("FusedAdam: p.grad is not the model's flat gradient buffer - use "
                                        "optimizer.zero_grad() (set_to_none=True) and one backward() per step; gradient "
                                        "accumulation across backward() calls is not supported")
                st = self.state[p]
                if getattr(p, "_eav_flat", None) is not None:
                    touched[p._eav_flat[0].data_ptr()] = p._eav_flat[0]
                if not st:
                    st["step"] = 0
                    flat = getattr(p, "_eav_flat", None)
                    if flat is not None and flat[0].data_ptr() + 4 * flat[2] == p.data_ptr():
                        shared = self._flat_state.setdefault(flat[0].data_ptr(), {})
                        if not shared:
                            shared["m"] = torch.zeros_like(flat[0])
                            shared["v"] = torch.zeros_like(flat[0])
                        st["exp_avg"] = shared["m"][flat[2]:flat[2] + p.numel()].view(p.shape)
                        st["exp_avg_sq"] = shared["v"][flat[2]:flat[2] + p.numel()].view(p.shape)
                    else:
                        st["exp_avg"] = torch.zeros_like(p)
                        st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                runs.append([p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(),
                             p.numel(), st["step"], g, 4 * getattr(p, "_eav_pad", 0)])
            # merge runs that are adjacent (up to 16-B alignment padding, plus declared zero padding - whose
            # gradient and moments are zero, so the update leaves it zero) in all four buffers
            runs.sort(key=lambda r: r[0])
            merged = []
            for r in runs:
                if merged:
                    q = merged[-1]
                    gap = r[0] - (q[0] + 4 * q[4])
                    if (0 <= gap < 16 + q[7] and r[5] == q[5] and r[1] - q[1] == r[0] - q[0]
                            and r[2] - q[2] == r[0] - q[0] and r[3] - q[3] == r[0] - q[0]):
                        q[4] = (r[0] - q[0]) // 4 + r[4]
                        q[7] = r[7]
                        continue
                merged.append(r)
            for r in merged:
                self._launch(r[0], r[1], r[2], r[3], r[4], group, r[5])
                # tell whoever caches derived copies of these parameters (transformer.Encoder's fp16 operand
                # planes) which byte ranges of the flat buffer just changed
                # (only flat buffers whose owner registered interest - `_eav_track_dirty` - are tracked, and the list
                # collapses to its hull beyond a few entries: an eager loop that nobody drains must not grow it)
                for fl in touched.values():
                    lo = fl.data_ptr()
                    if getattr(fl, "_eav_track_dirty", False) and lo <= r[0] < lo + 4 * fl.numel():
                        d = getattr(fl, "_eav_dirty", [])
                        d.append((r[0], r[0] + 4 * r[4]))
                        if len(d) > 64:
                            d = [(min(a for a, _ in d), max(b for _, b in d))]
                        fl._eav_dirty = d
        return loss


class _CEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, targets, flag):
        B, NC = scores.shape
        loss = torch.empty((), dtype=torch.float32, device=scores.device)
        need_grad = ctx.needs_input_grad[0]
        dsc = torch.empty_like(scores) if need_grad else None      # no gradient buffer under no_grad (validate())
        _lib.call("eav_ce_fwd_bwd", scores.data_ptr(), targets.data_ptr(), loss.data_ptr(), _lib.ptr(dsc), None,
                  flag.data_ptr(), B, NC, _lib.stream_ptr())
        ctx.dsc = dsc
        return loss

    @staticmethod
    def backward(ctx, gout):
        # d loss / d scores was produced by the forward launch; the incoming gradient (1.0 for loss.backward()) is
        # applied in place by the library - no torch kernel inside a captured step.  `unit_gradient(device)` is a constant
        # 1.0 that callers may pass as the seed (loss.backward(gradient=...)): recognised here, nothing is launched
        if gout.data_ptr() != _UNIT.get(gout.device, (None, 0))[1]:
            # any other upstream gradient scales a COPY: ctx.dsc stays the unit-gradient result, so a second backward
            # (retain_graph=True) or a re-used graph never scales twice
            out = ctx.dsc.clone()
            _lib.call("eav_scale_by_scalar", out.data_ptr(), gout.contiguous().data_ptr(), out.numel(),
                      _lib.stream_ptr())
            return out, None, None
        return ctx.dsc, None, None


_UNIT = {}


def unit_gradient(device):
    """A device-resident constant 1.0 to seed ``loss.backward(gradient=unit_gradient(dev))`` with: autograd then needs no
    fill kernel for the implicit ones_like(loss), and the CE backward recognises it and skips its scaling launch."""
    device = torch.device(device)
    if device not in _UNIT:
        t = torch.ones((), dtype=torch.float32, device=device)
        _UNIT[device] = (t, t.data_ptr())
    return _UNIT[device][0]


class CrossEntropyLoss:
    """``nn.CrossEntropyLoss()`` (mean reduction) as one fused HIP kernel that
    produces the loss and its input gradient together.

    Like torch, targets equal to -100 (the default ignore_index) are excluded from the mean and get a zero gradient, and
    any other class index outside [0, classes) is rejected: the kernel never indexes with such a label (the row
    contributes nothing to the loss and receives a zero gradient) and records it in a device flag; ``check()`` reads the flag (one host synchronisation) and
    raises.  The trainers call it once per epoch - a per-step check would serialise host and GPU; code that drives the
    criterion directly calls ``check()`` whenever it synchronises anyway."""

    def __init__(self):
        self._flag = None

    def check(self):
        if self._flag is not None:
            bad = int(self._flag.item())
            if bad != 0:
                self._flag.zero_()
                raise _lib.EavError(f"CrossEntropyLoss: target {bad - 1 if bad > 0 else bad} is outside [0, classes) "
                                    "(the reference's labels 1,3,5,7,9 must be mapped to 0..4 first)")

    def accumulate(self, scores, targets, loss_out, ncorrect):
        """Evaluation form (no gradient, nothing read back): writes the mean loss of this batch into the 0-dim device
        tensor `loss_out` and adds the number of argmax hits to the device int32 `ncorrect` - a validation loop reads
        both once per epoch instead of synchronising twice per batch (EEGNet_tor.py:126-130 calls .item() per batch)."""
        if not scores.is_cuda or scores.dim() != 2 or scores.dtype != torch.float32 or not scores.is_contiguous():
            raise _lib.EavError("CrossEntropyLoss.accumulate: scores must be a contiguous fp32 [batch, classes] device tensor")
        # every operand goes to the kernel as a raw pointer: a host tensor would be dereferenced on the GPU
        if not isinstance(targets, torch.Tensor) or targets.device != scores.device or targets.dim() != 1 \
                or targets.numel() != scores.shape[0]:
            raise _lib.EavError(f"CrossEntropyLoss.accumulate: targets must be {scores.shape[0]} labels on {scores.device} "
                                "(no CPU fallback: move the loader's tensors to the device)")
        if not isinstance(loss_out, torch.Tensor) or loss_out.device != scores.device or loss_out.dtype != torch.float32 \
                or loss_out.numel() < 1:
            raise _lib.EavError(f"CrossEntropyLoss.accumulate: loss_out must be an fp32 tensor on {scores.device}")
        if not isinstance(ncorrect, torch.Tensor) or ncorrect.device != scores.device or ncorrect.dtype != torch.int32 \
                or ncorrect.numel() < 1:
            raise _lib.EavError(f"CrossEntropyLoss.accumulate: ncorrect must be an int32 tensor on {scores.device}")
        if targets.dtype != torch.int64:
            targets = targets.long()
        if self._flag is None or self._flag.device != scores.device:
            self._flag = torch.zeros((), dtype=torch.int32, device=scores.device)
        _lib.call("eav_ce_fwd_bwd", scores.data_ptr(), targets.contiguous().data_ptr(), loss_out.data_ptr(), None,
                  ncorrect.data_ptr(), self._flag.data_ptr(), scores.shape[0], scores.shape[1], _lib.stream_ptr())

    def __call__(self, scores, targets):
        if not isinstance(scores, torch.Tensor) or not scores.is_cuda or not targets.is_cuda:
            raise _lib.EavError("CrossEntropyLoss needs device tensors (no CPU fallback)")
        if scores.dim() != 2 or scores.dtype != torch.float32 or not scores.is_contiguous():
            raise _lib.EavError(f"CrossEntropyLoss: scores must be a contiguous fp32 [batch, classes] tensor, got "
                                f"{tuple(scores.shape)} {scores.dtype}")
        if targets.dim() != 1 or targets.numel() != scores.shape[0]:
            raise _lib.EavError(f"CrossEntropyLoss: {targets.numel()} targets for {scores.shape[0]} rows of scores")
        if targets.dtype != torch.int64:
            targets = targets.long()
        if self._flag is None or self._flag.device != scores.device:
            self._flag = torch.zeros((), dtype=torch.int32, device=scores.device)
        return _CEFn.apply(scores, targets.contiguous(), self._flag)
