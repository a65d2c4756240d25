"""ShallowConvNet + transformer / TrainerUni (eav_amd/transformer_eeg.py) on the MI355X
against (a) golden vectors captured from the imported reference Transformer_torch/Transformer_EEG.py and (b) the CPU
oracle on the same seeded inputs.  Output probabilities within 1e-3 is north_star's bound; held to 5e-5 here."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from eav_amd import synth
from tests.golden_util import shallow_tf_weights
from tests.test_eegnet_model_gpu import close
from tests.test_oracle_shallow_tf import load_masks, scale_free

pytestmark = pytest.mark.gpu

BN = ["bn.running_mean", "bn.running_var"]


def build(nb, nl, sd, drop=0.0):
    from eav_amd.transformer_eeg import ShallowConvNet
    m = ShallowConvNet(nb_classes=nb, dropout=drop, num_layers=nl)
    full = m.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(np.ascontiguousarray(v))
    m.load_state_dict(full)
    return m.cuda()


def grad_close(got, ref, rel, what):
    """Every element within 5*rel of the tensor's max, 99% within rel (one flipped unit touches a whole 40-entry row): a ReLU pre-activation within rounding of 0 may
    take the other branch on the GPU (a couple of the 2.5 M activations per layer at B=32), which moves the few
    gradient entries it feeds by ~1e-5.  5e-8: the noise floor of the analytically-zero gradient."""
    got = got.detach().cpu().double().numpy()
    ref = np.asarray(ref, np.float64)
    err, scale = np.abs(got - ref), np.abs(ref).max()
    assert err.max() <= max(5 * rel * scale, 5e-8), f"{what}: max err {err.max():.3e}, ref max {scale:.3e}"
    loose = int((err > max(rel * scale, 5e-8) + 1e-3 * np.abs(ref)).sum())
    assert loose <= max(0.01 * err.size, 2), f"{what}: {loose} of {err.size} elements beyond the tight bound"


@pytest.mark.parametrize("attn", ["split", "fp32"])
@pytest.mark.parametrize("case", ["l12_train", "l12_eval", "l2_dropout"])
def test_steps_match_reference_golden(golden_dir, case, attn, monkeypatch):
    """attention core on the split-operand fp16 matrix-core kernels (the default) and on the exact-fp32 ones: same
    bounds against the reference's own numbers"""
    from eav_amd import _lib
    monkeypatch.setenv("EAV_SHALLOW_ATTENTION", attn)
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    from oracle.shallow_tf_oracle import param_names
    g = np.load(os.path.join(golden_dir, f"shallow_tf_{case}.npz"))
    nb, nl, B, lr = int(g["nb"]), int(g["num_layers"]), int(g["B"]), float(g["lr"])
    training = bool(int(g["train_mode"]))
    names = param_names(nl)
    free = scale_free(nl) if float(g["drop_p"]) == 0 else None
    model = build(nb, nl, shallow_tf_weights(int(g["wseed"]), nb, nl), float(g["drop_p"])).train(training)
    crit, opt = CrossEntropyLoss(), FusedAdam(model.parameters(), lr=lr)
    for s in range(int(g["steps"])):
        x, y = synth.eeg_batch(int(g["xseed"]) + s, B, 30, 500, n_classes=nb)
        masks = load_masks(g, s)
        if masks is not None:
            model.set_dropout_masks([m.to(torch.uint8).cuda().contiguous() for m in masks])
        probs = model(torch.from_numpy(x).cuda())
        loss = crit(probs, torch.from_numpy(y).cuda())
        opt.zero_grad()
        loss.backward()
        loose = s > 0
        close(probs, g[f"probs{s}"], 1e-4, 5e-5 if not loose else 5e-4, f"probs{s}")
        close(loss, g[f"loss{s}"], 1e-5, 1e-5 if not loose else 1e-4, f"loss{s}")
        named = dict(model.named_parameters())
        for k in names:
            grad_close(named[k].grad, g[f"grad{s}.{k}"], 2e-3 if not loose else 5e-2, f"grad{s}.{k}")
        opt.step()
        w = model.fc.weight
        _lib.call("eav_renorm_rows", _lib.ptr(w), w.shape[0], w.shape[1], 0.5, _lib.stream_ptr())
        torch.cuda.synchronize()
        full = model.state_dict()
        for k in names:
            err = np.abs(full[k].cpu().double().numpy() - g[f"post{s}.{k}"].astype(np.float64))
            bound = 2.1 * lr * (s + 1) if (training and k == free) else 0.5 * lr * (s + 1) + 1e-6
            assert err.max() <= bound, f"post{s}.{k}: {err.max():.3e}"
            if not loose and k != free:
                assert (err <= 2e-5 + 1e-4 * np.abs(g[f"post{s}.{k}"])).mean() > 0.95, f"post{s}.{k}: tight fraction"
        for k in BN:
            close(full[k], g[f"post{s}.{k}"], 1e-4, 1e-5 if not loose else 1e-3, f"post{s}.{k}")


@pytest.mark.parametrize("attn", ["split", "fp32"])
@pytest.mark.parametrize("B,S,nl", [(32, 500, 12), (5, 497, 3)])
def test_batch_against_oracle(B, S, nl, attn, monkeypatch):
    """The reference's batch size (and a ragged one with a shorter recording) against the CPU oracle."""
    from eav_amd.optim import CrossEntropyLoss
    monkeypatch.setenv("EAV_SHALLOW_ATTENTION", attn)
    from oracle import shallow_tf_oracle as orc
    nb = 5
    sd = shallow_tf_weights(81, nb, nl)
    names = orc.param_names(nl)
    x, y = synth.eeg_batch(811, B, 30, S, n_classes=nb)
    for training in (True, False):
        model = build(nb, nl, sd).train(training)
        assert model.attention_precision == attn
        probs = model(torch.from_numpy(x).cuda())
        loss = CrossEntropyLoss()(probs, torch.from_numpy(y).cuda())
        loss.backward()
        torch.cuda.synchronize()
        st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in names},
                         {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=1e-3, drop_p=0.0,
                         num_layers=nl)
        pref, lref, grads = st.step(torch.from_numpy(x), torch.from_numpy(y), training, None)
        close(probs, pref.numpy(), 1e-4, 5e-5, "probs")
        close(loss, lref.numpy(), 1e-5, 1e-5, "loss")
        named = dict(model.named_parameters())
        for k in names:
            grad_close(named[k].grad, grads[k].numpy(), 3e-3, f"grad.{k} (train={training})")
        for k in BN:
            close(model.state_dict()[k], st.Bf[k].numpy(), 1e-4, 1e-5, k)
        # the zero pad of the 64-wide attention tile must stay exactly zero
        assert float(model._ws.qkv[0].view(-1, 3, 64)[:, :, 40:].abs().max()) == 0.0
        assert float(model._ws.dqkv.view(-1, 3, 64)[:, :, 40:].abs().max()) == 0.0


def test_generated_dropout_statistics_and_determinism():
    from eav_amd.transformer_eeg import ShallowConvNet
    torch.manual_seed(0)
    m = ShallowConvNet(nb_classes=5, num_layers=2).cuda().train()
    x = torch.from_numpy(synth.normal(6, (4, 1, 30, 500))).cuda()
    m(x)
    kept = (m._ws.f1[0] != 0).float().mean().item()          # ReLU keeps ~half, dropout half of those
    assert 0.2 < kept < 0.3, kept
    feat = m._ws.feat.clone()
    assert 0.45 < (feat != 0).float().mean().item() < 0.55
    m(x)
    assert not torch.equal(feat != 0, m._ws.feat != 0)
    m.eval()
    a = m(x).clone()
    b = m(x).clone()
    assert torch.equal(a, b)


def test_trainer_matches_reference(golden_dir, tmp_path, monkeypatch):
    """TrainerUni.train() (Transformer_EEG.py:183-219) against the reference's own two-epoch run with its recorded
    shuffle orders; eagerly and through the hipGraph replay."""
    from eav_amd.transformer_eeg import TrainerUni
    g = np.load(os.path.join(golden_dir, "shallow_tf_trainer.npz"))
    nb, ntr, nte, bs = (int(g[k]) for k in ("nb", "ntr", "nte", "batch_size"))
    x, y = synth.eeg_batch(int(g["xseed"]), ntr + nte, 30, 500, n_classes=nb)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    monkeypatch.chdir(tmp_path)
    ref = torch.from_numpy(g["final_probs"])
    top2 = ref.sort(1).values
    decided = ((top2[:, -1] - top2[:, -2]) > 0.1).numpy()
    free = scale_free(12)
    for use_graph in (False, True):
        model = build(nb, 12, shallow_tf_weights(int(g["wseed"]), nb))
        buf = io.StringIO()
        with redirect_stdout(buf):
            tr = TrainerUni(model, data=[xt[:ntr], yt[:ntr], xt[ntr:], yt[ntr:]], lr=float(g["lr"]), batch_size=bs,
                            epochs=int(g["epochs"]), subject=7)
            tr.use_graph = use_graph
            tr.train_loader.order_override = [g["order0"], g["order1"]]
            tr.train()
        # Nothing is copied from the golden into the model.  Six Adam steps at lr 1e-3 through 12 post-norm layers amplify
        # rounding differences ~10x per step (two CPU fp32 runs of the reference's own loop differ by 6e-3 at the end), so
        # against the reference's recorded run this test checks the mechanics (batch order, eval switch, max-norm, printed
        # lines, result file) and a loose agreement; the tight step-by-step comparison against the oracle on identical
        # batches, with a stated per-step bound, is tests/test_optimizer_state_gpu.py.
        sd = model.state_dict()
        assert float((sd[free].cpu() - torch.from_numpy(g["final." + free])).abs().max()) <= 2.1 * float(g["lr"]) * 6
        close(sd["bn.running_var"], g["final.bn.running_var"], 5e-2, 1e-2, "running_var")
        model.eval()
        with torch.no_grad():
            probs = model(xt[ntr:].cuda()).cpu()
        # bn.running_mean follows the scale-free bias (a +-lr random walk): it shifts both the running mean and the
        # activations it normalises, so the eval output is insensitive to it up to the 6-step amplification above
        close(probs, ref.numpy(), 0, 6e-2, "final probs")
        assert np.array_equal(probs.argmax(1).numpy()[decided], ref.argmax(1).numpy()[decided])
        assert float(model.fc.weight.norm(dim=1).max()) <= 0.5 + 1e-6
        lines = buf.getvalue().strip().splitlines()
        assert len(lines) == 2 and all(l.startswith("Validation Accuracy: ") for l in lines)
    out = open("eeg_results_new_shallow_.txt").read().strip().splitlines()
    assert len(out) == 2 and all(l.startswith("Subject 7 | Accuracy: ") for l in out)


def test_host_tensor_is_rejected():
    from eav_amd import _lib
    from eav_amd.transformer_eeg import ShallowConvNet
    with pytest.raises(_lib.EavError):
        ShallowConvNet(nb_classes=5, num_layers=1)(torch.zeros(2, 1, 30, 500))
