"""Pin the EEGNet oracle (oracle/eegnet_oracle.py) to golden vectors captured
from the imported, shimmed reference CNN_torch/EEGNet_tor.py.  CPU<->CPU, so the
tolerance is tight (1e-5 relative on grads, 1e-6 on probabilities)."""
import os

import numpy as np
import pytest
import torch

from eav_amd import synth
from oracle import eegnet_oracle as orc
from tests.golden_util import eegnet_weights


def _close(a, b, rtol, atol, what):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e} (ref max {np.abs(b).max():.3e})"


def _close_params(a, b, lr, what, frac=0.995):
    """Post-Adam parameters.  Adam's update lr*m/(sqrt(v)+eps) is ill-conditioned
    where |g| ~ eps=1e-8 (a 1e-10 change of g moves the update by a few % of lr),
    so rounding-level gradient differences legitimately move a few elements by a
    fraction of lr.  Require: almost all elements tight, every element within lr/4."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b)
    tight = err <= 2e-6 + 1e-5 * np.abs(b)
    assert tight.mean() >= frac, f"{what}: only {tight.mean():.4f} tight"
    assert err.max() <= 0.25 * lr, f"{what}: max err {err.max():.3e} vs lr {lr}"


def _arch(g):
    """Network shape of a golden: the reference driver's unless the fixture records another (`arch.*`, generic cases)."""
    a = dict(nb=5, chans=30, klen=300, F1=8, D=8, F2=64)
    a.update({k[5:]: int(g[k]) for k in g.files if k.startswith("arch.")})
    return a


@pytest.mark.parametrize("case", ["s500_train", "s500_eval", "s500_maxnorm", "s500_dropout", "s500_dropout2d",
                                  "generic_train", "generic_eval", "generic_odd", "generic_f8"])
def test_oracle_matches_reference_steps(golden_dir, case):
    g = np.load(os.path.join(golden_dir, f"eegnet_{case}.npz"))
    B, S = int(g["B"]), int(g["S"])
    a = _arch(g)
    sd = eegnet_weights(int(g["wseed"]), S, scale=float(g["wscale"]), **a)
    P = {k: torch.from_numpy(sd[k].copy()) for k in orc.PARAM_NAMES}
    Bf = {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}
    st = orc.Stepper(P, Bf, lr=float(g["lr"]), drop_p=float(g["drop_p"]))
    training = bool(int(g["train_mode"]))
    for s in range(int(g["steps"])):
        x, y = synth.eeg_batch(int(g["xseed"]) + s, B, a["chans"], S, a["nb"])
        masks = None
        if float(g["drop_p"]) > 0:
            masks = (torch.from_numpy(g[f"mask{2 * s}"].astype(np.float32)),
                     torch.from_numpy(g[f"mask{2 * s + 1}"].astype(np.float32)))
        probs, loss, grads = st.step(torch.from_numpy(x), torch.from_numpy(y), training, masks)
        _close(probs, g[f"probs{s}"], 1e-5, 1e-6 if s == 0 else 1e-5, f"probs{s}")
        _close(loss, g[f"loss{s}"], 1e-6, 1e-6 if s == 0 else 1e-5, f"loss{s}")
        for k in orc.PARAM_NAMES:
            ref = g[f"grad{s}.{k}"]
            # step >= 1 inherits the (ill-conditioned) Adam differences of step 0
            ga = (1e-5 if s == 0 else 5e-4) * max(np.abs(ref).max(), 1e-12)
            _close(grads[k], ref, 1e-4, ga, f"grad{s}.{k}")
            _close_params(st.P[k].detach(), g[f"post{s}.{k}"], float(g["lr"]), f"post{s}.{k}")
        for k in orc.BUFFER_NAMES:
            _close(st.Bf[k], g[f"post{s}.{k}"], 1e-5, 1e-6, f"post{s}.{k}")


def test_oracle_matches_reference_s10000(golden_dir):
    g = np.load(os.path.join(golden_dir, "eegnet_s10000_train.npz"))
    B, S = int(g["B"]), int(g["S"])
    sd = eegnet_weights(int(g["wseed"]), S)
    P = {k: torch.from_numpy(sd[k].copy()) for k in orc.PARAM_NAMES}
    Bf = {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}
    st = orc.Stepper(P, Bf, lr=float(g["lr"]), drop_p=0.0)
    x, y = synth.eeg_batch(int(g["xseed"]), B, 30, S)
    probs, loss, grads = st.step(torch.from_numpy(x), torch.from_numpy(y), True, None)
    _close(probs, g["probs0"], 1e-5, 1e-6, "probs0")
    _close(loss, g["loss0"], 1e-6, 1e-6, "loss0")
    for k in orc.PARAM_NAMES:
        if f"grad0.{k}" in g:
            ref = g[f"grad0.{k}"]
            _close(grads[k], ref, 1e-4, 1e-5 * np.abs(ref).max(), f"grad0.{k}")
        else:
            ref = g[f"grad0.{k}.sample"]
            _close(grads[k].reshape(-1)[::97], ref, 1e-4, 1e-5 * np.abs(ref).max(), f"grad0.{k}.sample")


def test_renorm_rows_matches_torch():
    w = torch.from_numpy(synth.uniform(5, (64, 30), -1, 1))
    a = w.clone()
    b = w.clone()
    orc.renorm_rows_(a, 1.0)
    b.renorm_(p=2, dim=0, maxnorm=1.0)
    assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
