"""Pin the ShallowConvNet+transformer oracle (oracle/shallow_tf_oracle.py) to golden vectors captured from the imported
reference Transformer_torch/Transformer_EEG.py (tests/golden/make_goldens_alt_eeg.py).  CPU<->CPU: tight."""
import os

import numpy as np
import pytest
import torch

from eav_amd import synth
from oracle import shallow_tf_oracle as orc
from tests.golden_util import shallow_tf_weights
from tests.test_oracle_eegnet import _close, _close_params


def scale_free(num_layers):
    """The last layer's norm2.bias: without dropout it adds a per-feature constant that the train-mode BatchNorm removes
    again, so its gradient is analytically zero (noise ~4e-9)."""
    return f"transformer.{num_layers - 1}.norm2.bias"


def load_masks(g, s):
    n = int(g["nmask_per_step"])
    if n == 0:
        return None
    out = []
    for i in range(s * n, (s + 1) * n):
        shape = tuple(int(v) for v in g[f"mask{i}.shape"])
        out.append(torch.from_numpy(np.unpackbits(g[f"mask{i}"])[:int(np.prod(shape))].reshape(shape).astype(np.float32)))
    return out


def stepper_for(g):
    nb, nl = int(g["nb"]), int(g["num_layers"])
    sd = shallow_tf_weights(int(g["wseed"]), nb, nl)
    names = orc.param_names(nl)
    P = {k: torch.from_numpy(sd[k].copy()) for k in names}
    Bf = {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}
    return orc.Stepper(P, Bf, lr=float(g["lr"]), drop_p=float(g["drop_p"]), num_layers=nl), names


@pytest.mark.parametrize("case", ["l12_train", "l12_eval", "l2_dropout"])
def test_oracle_matches_reference_steps(golden_dir, case):
    g = np.load(os.path.join(golden_dir, f"shallow_tf_{case}.npz"))
    st, names = stepper_for(g)
    training = bool(int(g["train_mode"]))
    SCALE_FREE = scale_free(int(g["num_layers"])) if float(g["drop_p"]) == 0 else None
    for s in range(int(g["steps"])):
        x, y = synth.eeg_batch(int(g["xseed"]) + s, int(g["B"]), 30, 500, n_classes=int(g["nb"]))
        probs, loss, grads = st.step(torch.from_numpy(x), torch.from_numpy(y), training, load_masks(g, s))
        _close(probs, g[f"probs{s}"], 1e-5, 1e-6 if s == 0 else 2e-5, f"probs{s}")
        _close(loss, g[f"loss{s}"], 1e-6, 1e-6 if s == 0 else 1e-5, f"loss{s}")
        for k in names:
            ref = g[f"grad{s}.{k}"]
            # the last layer's norm2.bias is removed again by the train-mode BatchNorm: analytically zero, ~4e-9 of noise
            ga = max((2e-5 if s == 0 else 2e-3) * np.abs(ref).max(), 2e-8)
            _close(grads[k], ref, 1e-4, ga, f"grad{s}.{k}")
            if training and k == SCALE_FREE:      # Adam turns the rounding-noise gradient into a +-lr move
                err = np.abs(st.P[k].detach().numpy() - g[f"post{s}.{k}"]).max()
                assert err <= 2.1 * float(g["lr"]) * (s + 1), (k, err)
            else:
                _close_params(st.P[k].detach(), g[f"post{s}.{k}"], float(g["lr"]), f"post{s}.{k}", frac=0.97)
        for k in orc.BUFFER_NAMES:
            _close(st.Bf[k], g[f"post{s}.{k}"], 1e-5, 1e-6 if s == 0 else 1e-4, f"post{s}.{k}")


def test_oracle_replays_reference_trainer(golden_dir):
    """TrainerUni.train() for two epochs (Transformer_EEG.py:183-206) replayed with the recorded shuffle orders."""
    g = np.load(os.path.join(golden_dir, "shallow_tf_trainer.npz"))
    nb, ntr, nte, bs = (int(g[k]) for k in ("nb", "ntr", "nte", "batch_size"))
    sd = shallow_tf_weights(int(g["wseed"]), nb)
    names = orc.param_names()
    st = orc.Stepper({k: torch.from_numpy(sd[k].copy()) for k in names},
                     {k: torch.from_numpy(sd[k].copy()) for k in orc.BUFFER_NAMES}, lr=float(g["lr"]), drop_p=0.0)
    x, y = synth.eeg_batch(int(g["xseed"]), ntr + nte, 30, 500, n_classes=nb)
    for e in range(int(g["epochs"])):
        order = g[f"order{e}"]
        for i in range(0, ntr, bs):
            idx = order[i:i + bs]
            st.step(torch.from_numpy(x[idx]), torch.from_numpy(y[idx]), True, None)
    # Eval-mode outputs see the scale-free bias (running statistics do not cancel it), which random-walks by +-lr per
    # step on rounding noise, and bn.running_mean follows it.  Those two are bounded by the walk, then taken from the
    # reference so that everything else can be compared tightly.
    free = scale_free(12)
    assert np.abs(st.P[free].detach().numpy() - g["final." + free]).max() <= 2.1 * float(g["lr"]) * 6
    _close(st.Bf["bn.running_mean"], g["final.bn.running_mean"], 0, 5e-2, "running_mean")
    _close(st.Bf["bn.running_var"], g["final.bn.running_var"], 1e-2, 1e-3, "running_var")
    with torch.no_grad():
        st.P[free].copy_(torch.from_numpy(g["final." + free]))
        st.Bf["bn.running_mean"].copy_(torch.from_numpy(g["final.bn.running_mean"]))
        probs = orc.forward(st.P, st.Bf, torch.from_numpy(x[ntr:]), False)
    # what is left is Adam's amplification of rounding differences where |g| ~ eps (the double softmax makes the
    # gradients tiny): two CPU fp32 runs of the same six steps already differ by ~6e-3 here
    _close(probs, g["final_probs"], 0, 3e-2, "final probs")
    ref = torch.from_numpy(g["final_probs"])
    top2 = ref.sort(1).values
    decided = ((top2[:, -1] - top2[:, -2]) > 6e-2).numpy()
    assert np.array_equal(probs.argmax(1).numpy()[decided], ref.argmax(1).numpy()[decided])
    ref_acc = float((ref.argmax(1).numpy() == y[ntr:]).mean())
    assert f"Subject 7 | Accuracy: {ref_acc:.4f}" in str(g["result_file"])
