"""Canonical EEGNet / EEGNetTrainer (eav_amd/cnn_eeg.py) on the MI355X against (a) golden
vectors captured from the imported reference CNN_torch/CNN_EEG.py and (b) the CPU oracle on the same seeded inputs.
Logits within 1e-3 is the north_star tolerance; held to 5e-5 here, gradients to 1e-3 of the tensor's max."""
import io
import os
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from eav_amd import synth
from tests.golden_util import cnn_eeg_weights
from tests.test_eegnet_model_gpu import close

pytestmark = pytest.mark.gpu

PN = ["block1.0.weight", "block1.1.weight", "block1.1.bias", "block1.2.weight", "block1.3.weight", "block1.3.bias",
      "block2.0.weight", "block2.1.weight", "block2.2.weight", "block2.2.bias", "classifier.weight", "classifier.bias"]
BN = ["block1.1.running_mean", "block1.1.running_var", "block1.3.running_mean", "block1.3.running_var",
      "block2.2.running_mean", "block2.2.running_var"]
# analytically zero gradients in train mode (see tests/test_oracle_cnn_eeg.py): compared against a noise floor
SCALE_FREE = ("block1.1.weight", "block1.1.bias")


def build(dims, sd, drop=0.0):
    from eav_amd.cnn_eeg import EEGNet
    m = EEGNet(nb_classes=dims["nb"], Chans=dims["chans"], Samples=dims["S"], dropoutRate=drop,
               kernLength=dims["klen"], F1=dims["F1"], D=dims["D"], F2=dims["F2"])
    full = m.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(np.ascontiguousarray(v))
    full = {k: (torch.zeros_like(v) if k.endswith("num_batches_tracked") else v) for k, v in full.items()}
    m.load_state_dict(full)
    return m.cuda()


def dims_of(g):
    return {k: int(g[k]) for k in ("nb", "chans", "S", "klen", "F1", "D", "F2")}


def grad_close(got, ref, k, training, rel, what):
    floor = 2e-6 if (training and k in SCALE_FREE) else 1e-8     # 1e-8: a 1-tap filter is scale-free as well
    close(got, ref, 1e-3, max(rel * np.abs(ref).max(), floor), what)


@pytest.mark.parametrize("case", ["default_train", "default_eval", "eav_dropout", "wide_ragged"])
def test_steps_match_reference_golden(golden_dir, case):
    from eav_amd.optim import CrossEntropyLoss, FusedAdam
    g = np.load(os.path.join(golden_dir, f"cnn_eeg_{case}.npz"))
    d, B, lr = dims_of(g), int(g["B"]), float(g["lr"])
    training = bool(int(g["train_mode"]))
    sd = cnn_eeg_weights(int(g["wseed"]), d["nb"], d["chans"], d["S"], d["klen"], d["F1"], d["D"], d["F2"])
    model = build(d, sd, float(g["drop_p"])).train(training)
    crit, opt = CrossEntropyLoss(), FusedAdam(model.parameters(), lr=lr)
    for s in range(int(g["steps"])):
        x, y = synth.eeg_batch(int(g["xseed"]) + s, B, d["chans"], d["S"], n_classes=d["nb"])
        if float(g["drop_p"]) > 0:
            model.set_dropout_masks((torch.from_numpy(g[f"mask{2 * s}"]).cuda().contiguous(),
                                     torch.from_numpy(g[f"mask{2 * s + 1}"]).cuda().contiguous()))
        logits = model(torch.from_numpy(x[:, 0]).cuda())          # [B,Chans,Samples], as the reference is fed
        loss = crit(logits, torch.from_numpy(y).cuda())
        opt.zero_grad()
        loss.backward()
        loose = s > 0
        close(logits, g[f"logits{s}"], 1e-4, 5e-5 if not loose else 5e-4, f"logits{s}")
        close(loss, g[f"loss{s}"], 1e-5, 1e-5 if not loose else 2e-4, f"loss{s}")
        named = dict(model.named_parameters())
        for k in PN:
            grad_close(named[k].grad, g[f"grad{s}.{k}"], k, training, 1e-3 if not loose else 2e-2, f"grad{s}.{k}")
        opt.step()
        torch.cuda.synchronize()
        full = model.state_dict()
        for k in PN:
            err = np.abs(full[k].cpu().double().numpy() - g[f"post{s}.{k}"].astype(np.float64))
            if training and k in SCALE_FREE:
                assert err.max() <= 2.1 * lr * (s + 1), f"post{s}.{k}: {err.max():.3e}"
                continue
            assert err.max() <= 0.5 * lr * (s + 1) + 1e-6, f"post{s}.{k}: {err.max():.3e}"
            if not loose:
                assert (err <= 2e-5 + 1e-4 * np.abs(g[f"post{s}.{k}"])).mean() > 0.97, f"post{s}.{k}: tight fraction"
        for k in BN:
            close(full[k], g[f"post{s}.{k}"], 1e-4, 1e-5 if not loose else 5e-4, f"post{s}.{k}")
        assert int(full["block1.1.num_batches_tracked"]) == (s + 1 if training else 0)


def oracle_step(d, sd, x, y, training=True, lr=1e-3):
    from oracle import cnn_eeg_oracle as orc
    P = {k: torch.from_numpy(sd[k].copy()) for k in orc.PARAM_NAMES}
    Bf = {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}
    st = orc.Stepper(P, Bf, lr=lr, drop_p=0.0)
    return st, st.step(torch.from_numpy(x), torch.from_numpy(y), training, None)


@pytest.mark.parametrize("dims,B", [
    (dict(nb=5, chans=30, S=10000, klen=64, F1=8, D=2, F2=16), 16),       # the EAV recording shape
    (dict(nb=5, chans=30, S=500, klen=300, F1=8, D=8, F2=64), 32),        # EEGNet_tor's hyper-parameters
    (dict(nb=2, chans=128, S=1030, klen=512, F1=16, D=4, F2=64), 5),      # every limit of the kernels, ragged tiles
    (dict(nb=3, chans=1, S=33, klen=1, F1=1, D=1, F2=1), 2),              # degenerate sizes
])
def test_shapes_against_oracle(dims, B):
    from eav_amd.optim import CrossEntropyLoss
    sd = cnn_eeg_weights(51, dims["nb"], dims["chans"], dims["S"], dims["klen"], dims["F1"], dims["D"], dims["F2"])
    x, y = synth.eeg_batch(511, B, dims["chans"], dims["S"], n_classes=dims["nb"])
    for training in (True, False):
        model = build(dims, sd).train(training)
        logits = model(torch.from_numpy(x).cuda())
        loss = CrossEntropyLoss()(logits, torch.from_numpy(y).cuda())
        loss.backward()
        torch.cuda.synchronize()
        st, (lref, loss_ref, grads) = oracle_step(dims, sd, x, y, training)
        close(logits, lref.numpy(), 1e-4, 5e-5, "logits")
        close(loss, loss_ref.numpy(), 1e-5, 1e-5, "loss")
        named = dict(model.named_parameters())
        for k in PN:
            grad_close(named[k].grad, grads[k].numpy(), k, training, 2e-3, f"grad.{k} (train={training})")
        full = model.state_dict()
        for k in BN:
            close(full[k], st.Bf[k].numpy(), 1e-4, 1e-5, k)


def test_step_is_deterministic():
    from eav_amd.optim import CrossEntropyLoss
    dims = dict(nb=5, chans=30, S=2000, klen=64, F1=8, D=2, F2=16)
    sd = cnn_eeg_weights(52, **{k: dims[k] for k in ("nb",)}, chans=30, samples=2000)
    x, y = synth.eeg_batch(512, 8, 30, 2000)
    outs = []
    for _ in range(2):
        model = build(dims, sd).train()
        logits = model(torch.from_numpy(x).cuda())
        CrossEntropyLoss()(logits, torch.from_numpy(y).cuda()).backward()
        outs.append((logits.clone(), {k: v.grad.clone() for k, v in model.named_parameters()}))
    assert torch.equal(outs[0][0], outs[1][0])
    for k in PN:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_generated_dropout_keeps_half_and_matches_backward():
    """Counter-based dropout: ~50% of the pooled activations survive, eval mode ignores it, and the backward uses the
    same mask as the forward (checked through the oracle given the mask read back from the activations)."""
    from eav_amd.cnn_eeg import EEGNet
    torch.manual_seed(0)
    m = EEGNet(nb_classes=4, Chans=30, Samples=512, dropoutRate=0.5).cuda().train()
    x = torch.from_numpy(synth.normal(5, (16, 30, 512))).cuda()
    m(x)
    a2 = m._ws.a2
    kept = (a2 != 0).float().mean().item()
    assert 0.47 < kept < 0.53, kept
    first = a2.clone()
    m(x)
    assert not torch.equal(first != 0, m._ws.a2 != 0)      # a fresh mask on every training forward
    m.eval()
    m(x)
    assert (m._ws.a2 != 0).float().mean().item() > 0.99


def test_trainer_matches_reference(golden_dir):
    """EEGNetTrainer.train() + predict() (CNN_EEG.py:138-162) against the reference's own run, replaying its recorded
    shuffle orders; once eagerly and once through the hipGraph replay path."""
    from torch.utils.data import TensorDataset
    from eav_amd.cnn_eeg import EEGNetTrainer
    g = np.load(os.path.join(golden_dir, "cnn_eeg_trainer.npz"))
    nb, chans, S, ntr, nte, bs = (int(g[k]) for k in ("nb", "chans", "S", "ntr", "nte", "batch_size"))
    x, y = synth.eeg_batch(int(g["xseed"]), ntr + nte, chans, S, n_classes=nb)
    x = x[:, 0]
    dims = dict(nb=nb, chans=chans, S=S, klen=64, F1=8, D=2, F2=16)
    results = []
    for use_graph in (False, True):
        model = build(dims, cnn_eeg_weights(int(g["wseed"]), nb, chans, S))
        tr = TensorDataset(torch.from_numpy(x[:ntr]), torch.from_numpy(y[:ntr]))
        te = TensorDataset(torch.from_numpy(x[ntr:]), torch.from_numpy(y[ntr:]))
        buf = io.StringIO()
        with redirect_stdout(buf):
            trainer = EEGNetTrainer(model, tr, te, batch_size=bs, epochs=int(g["epochs"]), lr=float(g["lr"]))
            trainer.use_graph = use_graph
            trainer.train_loader.order_override = [g["order0"], g["order1"]]
            trainer.train()
            preds = trainer.predict()
        model.eval()
        with torch.no_grad():
            logits = model(torch.from_numpy(x[ntr:]).cuda()).cpu()
        results.append(logits)
        ref = torch.from_numpy(g["final_logits"])
        # eval outputs depend on the scale-free parameters' +-lr random walk (tests/test_oracle_cnn_eeg.py)
        close(logits, ref.numpy(), 0, 5e-3, "final logits")
        top2 = ref.sort(1).values
        decided = ((top2[:, -1] - top2[:, -2]) > 1e-2).numpy()
        assert np.array_equal(np.array(preds)[decided], g["preds"][decided])
        ours, theirs = buf.getvalue().strip().splitlines(), str(g["stdout"]).strip().splitlines()
        assert ours[0] == "Using device: cuda" and ours[1] == theirs[1] and len(ours) == len(theirs)
        for a, b in zip(ours[2:], theirs[2:]):      # 'Epoch e/E | Train Loss: x | Val Loss: y | Val Acc: z%'
            fa = [float(t.split(":")[1].strip(" %")) for t in a.split("|")[1:]]
            fb = [float(t.split(":")[1].strip(" %")) for t in b.split("|")[1:]]
            assert a.split("|")[0] == b.split("|")[0]
            assert abs(fa[0] - fb[0]) < 2e-3 and abs(fa[1] - fb[1]) < 5e-3, (a, b)
    close(results[1], results[0].numpy(), 0, 5e-3, "graph replay vs eager")


def test_host_tensor_is_rejected():
    from eav_amd import _lib
    from eav_amd.cnn_eeg import EEGNet
    with pytest.raises(_lib.EavError):
        EEGNet(nb_classes=4)(torch.zeros(2, 64, 128))
