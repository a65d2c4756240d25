"""Optimiser-state and multi-step pins (VERDICT r1 'harden the loose tests').

(1) After ONE step from identical state the Adam moments are well conditioned (exp_avg = (1-b1) g, exp_avg_sq =
    (1-b2) g^2): they are compared with the CPU oracle for all four trainable models - EEGNet_tor, the canonical EEGNet,
    ShallowConvNet+transformer and the ViT/AST encoder - so the optimiser state itself is pinned, not only the
    (Adam-noise-limited) parameters.
(2) The two alternative EEG encoders are replayed step by step against the CPU oracle on the SAME batches from the SAME
    state, with nothing copied from any golden into the model: the probability gap must stay under a stated bound that
    grows with the step index (these post-norm / scale-free models amplify rounding differences; the per-step growth
    measured on the CPU between two fp32 runs of the reference loop is ~10x for the ShallowConvNet)."""
import numpy as np
import pytest
import torch

from eav_amd import synth
from tests.golden_util import cnn_eeg_weights, eegnet_weights, shallow_tf_weights, tf_weights

pytestmark = pytest.mark.gpu


def moments_close(opt, named, ref_m, ref_v, names, rel_m=2e-3, rel_v=4e-3, relu_flips=False):
    """relu_flips: a ReLU pre-activation within rounding of zero may take the other branch on the GPU (see grad_close in
    test_shallow_tf_gpu.py), which moves the one weight row / bias entry that unit feeds: every element within 5x the
    bound, 99 % within the bound itself (measured on the shallow transformer: the exact-fp32 and the split attention
    each show one or two such rows at 4e-3 .. 7e-3, in different layers)."""
    for k in names:
        st = opt.state[named[k]]
        m, v = st["exp_avg"].detach().cpu().double().numpy(), st["exp_avg_sq"].detach().cpu().double().numpy()
        rm, rv = ref_m[k].double().numpy(), ref_v[k].double().numpy()
        em, ev = np.abs(m - rm), np.abs(v - rv)
        bm, bv = rel_m * np.abs(rm).max() + 1e-12, rel_v * np.abs(rv).max() + 1e-20
        if relu_flips:
            assert (em <= bm).mean() >= 0.99 and (ev <= bv).mean() >= 0.99, f"{k}: fraction within the bound"
            bm, bv = 5 * bm, 5 * bv
        assert em.max() <= bm, f"exp_avg {k}: {em.max():.3e}"
        assert ev.max() <= bv, f"exp_avg_sq {k}: {ev.max():.3e}"


def load(m, sd):
    full = m.state_dict()
    full.update({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()})
    m.load_state_dict(full)
    return m.cuda()


def test_adam_moments_eegnet_tor():
    from eav_amd.eegnet import EEGNet_tor
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from oracle import eegnet_oracle as orc
    S = 500
    sd = eegnet_weights(41, S)
    model = load(EEGNet_tor(5, Chans=30, Samples=S, dropoutRate=0.0), sd).train()
    opt, crit = FusedAdam(model.parameters(), lr=1e-3), CrossEntropyLoss()
    x, y = synth.eeg_batch(410, 8, 30, S)
    crit(model(torch.from_numpy(x).cuda()), torch.from_numpy(y).cuda()).backward()
    opt.step()
    torch.cuda.synchronize()
    st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in orc.PARAM_NAMES},
                     {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=1e-3, drop_p=0.0)
    st.step(torch.from_numpy(x), torch.from_numpy(y), True, None)
    moments_close(opt, dict(model.named_parameters()), st.m, st.v, orc.PARAM_NAMES)


def test_adam_moments_encoder():
    from eav_amd import transformer as T
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from oracle import vit_oracle as vo
    cfg = T.make_config("vit", hidden=64, layers=2, heads=4, ff=128)
    ocfg = vo.cfg_vit(hidden=64, layers=2, heads=4, ff=128)
    W = tf_weights(23, vo.param_shapes(ocfg), std=0.08)
    x, y = synth.frame_batch(230, 3, 224)
    for precision in ("split", "fp32"):
        model = T.Encoder(cfg, W).cuda().train()
        model.precision = precision
        opt = FusedAdam(model.parameters(), lr=1e-3, weight_decay=0.01, decoupled=True)
        CrossEntropyLoss()(model(torch.from_numpy(x).cuda()).logits, torch.from_numpy(y).cuda()).backward()
        opt.step()
        torch.cuda.synchronize()
        st = vo.Stepper({k: torch.from_numpy(v.copy()) for k, v in W.items()}, ocfg, lr=1e-3)
        st.step(torch.from_numpy(x), torch.from_numpy(y), False)
        names = [k for k in W if not k.endswith("k_proj.bias")]       # analytically zero gradient: rounding noise only
        moments_close(opt, dict(model.named_parameters()), st.m, st.v, names, rel_m=3e-3, rel_v=6e-3)


def _replay(model, stepper, opt, crit, x, y, B, nsteps, bounds, post_step=None, squeeze=False):
    """Same batches through the HIP model and the CPU oracle from the same state; returns the per-step gaps."""
    n = x.shape[0]
    gaps = []
    for s in range(nsteps):
        idx = [(5 * s + 3 * j) % n for j in range(B)]
        xb, yb = torch.from_numpy(x[idx]), torch.from_numpy(y[idx])
        out = model((xb[:, 0] if squeeze else xb).cuda())
        loss = crit(out, yb.cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        if post_step is not None:
            post_step()
        ref = stepper.step(xb[:, 0] if squeeze else xb, yb, True, None)[0]
        gaps.append(float((out.detach().cpu() - ref).abs().max()))
        assert gaps[-1] <= bounds[s], f"step {s}: |model - oracle| = {gaps[-1]:.3e} > {bounds[s]:.1e} (so far {gaps})"
    return gaps


def test_shallow_transformer_replay_against_oracle():
    """Six optimiser steps of the 12-layer ShallowConvNet+transformer (the reference trainer's first epochs at its batch
    shape) against the oracle.  Bound per step: 5e-5 at step 0, x8 per step, capped at 1e-2.  Measured on MI355X:
    1.5e-7, 1.5e-4, 9.5e-4, 3.1e-3, 4.9e-3, 8.2e-3 - the jump after the first update is Adam turning the rounding-level
    gradient of the scale-free bias into +-lr steps, the rest is the ~3x per step amplification of 12 post-norm layers
    (two CPU fp32 runs of the reference's own loop end 6e-3 apart)."""
    from eav_amd import _lib
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from eav_amd.transformer_eeg import ShallowConvNet
    from oracle import shallow_tf_oracle as orc
    nb, nl, B = 5, 12, 16
    sd = shallow_tf_weights(57, nb, nl)
    model = load(ShallowConvNet(nb_classes=nb, dropout=0.0, num_layers=nl), sd).train()
    opt, crit = FusedAdam(model.parameters(), lr=1e-3), CrossEntropyLoss()
    st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in orc.param_names(nl)},
                     {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=1e-3, drop_p=0.0, num_layers=nl)
    x, y = synth.eeg_batch(570, 48, 30, 500, n_classes=nb)

    def max_norm():
        w = model.fc.weight
        _lib.call("eav_renorm_rows", _lib.ptr(w), w.shape[0], w.shape[1], 0.5, _lib.stream_ptr())
    bounds = [min(5e-5 * 8 ** s, 1e-2) for s in range(6)]
    gaps = _replay(model, st, opt, crit, x, y, B, 6, bounds, post_step=max_norm)
    print("ShallowConvNet |probs - oracle| per step:", ["%.1e" % g for g in gaps])
    assert gaps[0] < 2e-6          # from identical state the first forward agrees to fp32 rounding


def test_shallow_transformer_adam_moments_after_one_step():
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from eav_amd.transformer_eeg import ShallowConvNet
    from oracle import shallow_tf_oracle as orc
    nb, nl = 5, 12
    sd = shallow_tf_weights(58, nb, nl)
    model = load(ShallowConvNet(nb_classes=nb, dropout=0.0, num_layers=nl), sd).train()
    opt = FusedAdam(model.parameters(), lr=1e-3)
    x, y = synth.eeg_batch(580, 16, 30, 500, n_classes=nb)
    CrossEntropyLoss()(model(torch.from_numpy(x).cuda()), torch.from_numpy(y).cuda()).backward()
    opt.step()
    torch.cuda.synchronize()
    st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in orc.param_names(nl)},
                     {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=1e-3, drop_p=0.0, num_layers=nl)
    st.step(torch.from_numpy(x), torch.from_numpy(y), True, None)
    # gradients that are analytically zero (bias in front of a train-mode BatchNorm) are rounding noise on both sides
    names = [k for k in orc.param_names(nl) if float(st.m[k].abs().max()) > 1e-7]
    assert len(names) >= len(orc.param_names(nl)) - 2
    moments_close(opt, dict(model.named_parameters()), st.m, st.v, names, rel_m=5e-3, rel_v=1e-2, relu_flips=True)


def test_canonical_eegnet_replay_and_moments():
    """Canonical EEGNet (CNN_EEG.py): one-step Adam moments, then eight steps against the oracle from the same state
    (bound 2e-6 on the logits at every step; measured 1.2e-7 .. 1.8e-7)."""
    from eav_amd.cnn_eeg import EEGNet
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from oracle import cnn_eeg_oracle as orc
    nb, chans, S, B = 4, 64, 128, 16
    sd = cnn_eeg_weights(73, nb, chans, S)
    x, y = synth.eeg_batch(730, 48, chans, S, n_classes=nb)
    model = load(EEGNet(nb, Chans=chans, Samples=S, dropoutRate=0.0), sd).train()
    opt, crit = FusedAdam(model.parameters(), lr=1e-3), CrossEntropyLoss()
    st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in orc.PARAM_NAMES},
                     {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=1e-3, drop_p=0.0)
    bounds = [2e-6] * 8
    gaps = _replay(model, st, opt, crit, x, y, B, 1, bounds, squeeze=True)
    names = [k for k in orc.PARAM_NAMES if float(st.m[k].abs().max()) > 1e-7]
    assert len(names) >= len(orc.PARAM_NAMES) - 2
    moments_close(opt, dict(model.named_parameters()), st.m, st.v, names, rel_m=5e-3, rel_v=1e-2)
    n = x.shape[0]
    for s in range(1, 8):
        idx = [(5 * s + 3 * j) % n for j in range(B)]
        xb, yb = torch.from_numpy(x[idx][:, 0]), torch.from_numpy(y[idx])
        out = model(xb.cuda())
        loss = crit(out, yb.cuda())
        opt.zero_grad()
        loss.backward()
        opt.step()
        ref = st.step(xb, yb, True, None)[0]
        gaps.append(float((out.detach().cpu() - ref).abs().max()))
        assert gaps[-1] <= bounds[s], f"step {s}: {gaps}"
    print("canonical EEGNet |logits - oracle| per step:", ["%.1e" % g for g in gaps])
