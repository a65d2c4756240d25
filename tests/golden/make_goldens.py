"""Generate the golden fixtures by running the IMPORTED reference here.

Runs only in the development container (needs /root/reference, read-only).
It imports the reference's own modules - shimmed exactly as SURVEY.md section 8c
lists (S1-S3 for EEGNet) - feeds them inputs/weights from the repo's own
deterministic generator (eav_amd/synth.py) and stores inputs' seeds plus the
reference's outputs as small .npz files next to this script.  No reference
source text is stored: fixtures are data only.

    python tests/golden/make_goldens.py [eegnet|datasplit|ast|vit|all]
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from eav_amd import synth  # noqa: E402


# --------------------------------------------------------------------------- shims
def import_reference_eegnet():
    """S1 (dangling Fusion import, Q5), S2 (missing DataLoader names, Q6)."""
    for name in ("Fusion", "Fusion.VIT_audio", "Fusion.VIT_audio.Transformer_audio"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["Fusion.VIT_audio.Transformer_audio"].Trainer_uni = object
    import importlib
    mod = importlib.import_module("CNN_torch.EEGNet_tor")
    from torch.utils.data import DataLoader, TensorDataset
    mod.TensorDataset = TensorDataset
    mod.DataLoader = DataLoader
    return mod


def fix_hooks(model, norm_rate=1.0):
    """S3: the shipped hooks return the weight and so replace the layer output
    (Q1); re-register them with the intended meaning (renorm in place, output
    untouched)."""
    for m in (model.depthwiseConv, model.dense):
        m._forward_hooks.clear()
        m.register_forward_hook(
            lambda mod, i, o, nr=norm_rate: (mod.weight.data.renorm_(p=2, dim=0, maxnorm=nr), None)[1])
    return model


# --------------------------------------------------------------------------- weights
from tests.golden_util import eegnet_weights  # noqa: E402


def load_eegnet_state(model, sd):
    full = model.state_dict()
    for k, v in sd.items():
        full[k] = torch.from_numpy(np.ascontiguousarray(v))
    model.load_state_dict(full)


PNAMES = ["firstConv.weight", "firstBN.weight", "firstBN.bias", "depthwiseConv.weight",
          "depthwiseBN.weight", "depthwiseBN.bias", "separableConv.weight", "separableBN.weight",
          "separableBN.bias", "dense.weight", "dense.bias"]
BNAMES = ["firstBN.running_mean", "firstBN.running_var", "depthwiseBN.running_mean",
          "depthwiseBN.running_var", "separableBN.running_mean", "separableBN.running_var"]


def eegnet_case(mod, name, B, S, wseed, xseed, train_mode, wscale=1.0, masks=False, lr=1e-3, steps=2,
                dropout_type="Dropout", arch=None):
    """Run `steps` reference training steps (Trainer_uni.train body, :104-110)
    and record everything the parity tests compare."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    drop = 0.5 if masks else 0.0
    # arch: a non-default network shape (the reference constructor takes any F1 / D / F2 / kernLength / Chans, :16-17)
    a = dict(nb=5, chans=30, klen=300, F1=8, D=8, F2=64)
    a.update(arch or {})
    model = mod.EEGNet_tor(nb_classes=a["nb"], Chans=a["chans"], Samples=S, kernLength=a["klen"], F1=a["F1"], D=a["D"],
                           F2=a["F2"], dropoutRate=drop, dropoutType=dropout_type)
    fix_hooks(model)
    sd = eegnet_weights(wseed, S, scale=wscale, **a)
    load_eegnet_state(model, sd)
    model.train(train_mode)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    out = {"B": B, "S": S, "wseed": wseed, "xseed": xseed, "train_mode": int(train_mode),
           "wscale": wscale, "lr": lr, "steps": steps, "drop_p": drop}
    if arch:
        out.update({"arch." + k: v for k, v in a.items()})
    captured = []
    if masks:
        # capture the Bernoulli keep-masks the reference draws: wrap F.dropout
        import torch.nn.functional as F
        orig = F.dropout

        def cap(inp, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return inp
            keep = (torch.rand_like(inp) >= p).to(inp.dtype)
            captured.append(keep.numpy().astype(np.uint8))
            return inp * keep / (1.0 - p)
        F.dropout = cap
        orig2d = F.dropout2d

        def cap2d(inp, p=0.5, training=True, inplace=False):      # nn.Dropout2d: one draw per (sample, channel) map
            if not training or p == 0.0:
                return inp
            keep = (torch.rand(inp.shape[0], inp.shape[1], *([1] * (inp.dim() - 2))) >= p).to(inp.dtype).expand_as(inp)
            captured.append(keep.numpy().astype(np.uint8))
            return inp * keep / (1.0 - p)
        F.dropout2d = cap2d
    try:
        for s in range(steps):
            x, y = synth.eeg_batch(xseed + s, B, a["chans"], S, a["nb"])
            xt, yt = torch.from_numpy(x), torch.from_numpy(y)
            scores = model(xt)
            loss = crit(scores, yt)
            opt.zero_grad()
            loss.backward()
            out[f"probs{s}"] = scores.detach().numpy().copy()
            out[f"loss{s}"] = np.float32(loss.item())
            named = dict(model.named_parameters())
            for k in PNAMES:
                out[f"grad{s}.{k}"] = named[k].grad.detach().numpy().copy()
            opt.step()
            full = model.state_dict()
            for k in PNAMES + BNAMES:
                out[f"post{s}.{k}"] = full[k].detach().numpy().copy()
    finally:
        if masks:
            import torch.nn.functional as F
            F.dropout = orig
            F.dropout2d = orig2d
    for i, m in enumerate(captured):
        out[f"mask{i}"] = m
    # keep the big S=10000 fixture small: grads of dense.weight / firstConv kept, rest summarised
    if S > 2000:
        for key in list(out.keys()):
            v = out[key]
            if isinstance(v, np.ndarray) and v.size > 20000 and not key.startswith("mask"):
                out[key + ".sample"] = v.reshape(-1)[::97].copy()
                out[key + ".sum"] = np.float64(v.astype(np.float64).sum())
                out[key + ".abssum"] = np.float64(np.abs(v.astype(np.float64)).sum())
                del out[key]
    np.savez_compressed(os.path.join(HERE, f"eegnet_{name}.npz"), **out)
    print("wrote", name, {k: (v.shape if isinstance(v, np.ndarray) else v) for k, v in list(out.items())[:12]})


def make_eegnet():
    mod = import_reference_eegnet()
    eegnet_case(mod, "s500_train", B=4, S=500, wseed=11, xseed=101, train_mode=True)
    eegnet_case(mod, "s500_eval", B=4, S=500, wseed=11, xseed=101, train_mode=False)
    eegnet_case(mod, "s500_maxnorm", B=4, S=500, wseed=12, xseed=102, train_mode=True, wscale=3.0)
    eegnet_case(mod, "s500_dropout", B=4, S=500, wseed=13, xseed=103, train_mode=True, masks=True)
    eegnet_case(mod, "s10000_train", B=2, S=10000, wseed=14, xseed=104, train_mode=True, steps=1)
    eegnet_case(mod, "s500_dropout2d", B=6, S=500, wseed=15, xseed=105, train_mode=True, masks=True,
                dropout_type="SpatialDropout2D")
    make_eegnet_generic(mod)
    make_eegnet_loop(mod)


def make_eegnet_generic(mod=None):
    """Widths other than the reference driver's (EEGNet_tor.py:16-17 accepts any): the canonical EEGNet shape (train
    mode with max-norm active, eval mode), an odd shape with captured dropout masks, and an F1 = 8 shape whose
    firstConv takes the MFMA kernels while the rest is generic."""
    mod = mod or import_reference_eegnet()
    canon = dict(nb=4, chans=64, klen=64, F1=4, D=2, F2=16)
    eegnet_case(mod, "generic_train", B=4, S=256, wseed=31, xseed=131, train_mode=True, arch=canon, wscale=2.0)
    eegnet_case(mod, "generic_eval", B=4, S=256, wseed=31, xseed=131, train_mode=False, arch=canon)
    eegnet_case(mod, "generic_odd", B=3, S=352, wseed=32, xseed=132, train_mode=True, masks=True,
                arch=dict(nb=5, chans=19, klen=37, F1=6, D=3, F2=24))
    eegnet_case(mod, "generic_f8", B=3, S=320, wseed=33, xseed=133, train_mode=True,
                arch=dict(nb=5, chans=30, klen=128, F1=8, D=2, F2=32))


def make_eegnet_loop(mod):
    """Trainer_uni.train() for 2 epochs (EEGNet_tor.py:96-135) on 40 train / 20
    test items of [1,30,500]; the DataLoader shuffle order is recorded as index
    lists so the HIP trainer can replay it; Q4 (eval-mode training from epoch 2)
    happens inside the reference itself."""
    import io
    import contextlib
    S, ntr, nte = 500, 40, 20
    x, y = synth.eeg_batch(777, ntr + nte, 30, S)
    tr_x, tr_y, te_x, te_y = x[:ntr], y[:ntr], x[ntr:], y[ntr:]
    torch.manual_seed(0)
    model = mod.EEGNet_tor(nb_classes=5, Chans=30, Samples=S, kernLength=300, F1=8, D=8, F2=64, dropoutRate=0.0)
    fix_hooks(model)
    load_eegnet_state(model, eegnet_weights(21, S))
    trainer = mod.Trainer_uni(model=model, data=[tr_x, tr_y, te_x, te_y], lr=1e-3, batch_size=16,
                              num_epochs=2, device=torch.device("cpu"))
    # record the shuffle order the reference loop really uses: wrap RandomSampler.__iter__
    from torch.utils.data import sampler as _sampler
    orders = []
    orig_iter = _sampler.RandomSampler.__iter__

    def rec_iter(self):
        idx = list(orig_iter(self))
        orders.append(np.array(idx, dtype=np.int64))
        return iter(idx)
    _sampler.RandomSampler.__iter__ = rec_iter
    torch.manual_seed(1234)
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            trainer.train()
    finally:
        _sampler.RandomSampler.__iter__ = orig_iter
    assert len(orders) == 2
    out = {"S": S, "ntr": ntr, "nte": nte, "xseed": 777, "wseed": 21, "lr": 1e-3, "batch_size": 16,
           "epochs": 2, "order0": orders[0], "order1": orders[1], "stdout": np.array(buf.getvalue())}
    full = model.state_dict()
    for k in PNAMES + BNAMES:
        out[f"final.{k}"] = full[k].numpy().copy()
    model.eval()
    with torch.no_grad():
        out["final_probs"] = model(torch.from_numpy(te_x)).numpy().copy()
    np.savez_compressed(os.path.join(HERE, "eegnet_loop.npz"), **out)
    print("wrote loop;", buf.getvalue().strip().splitlines()[-1])


def make_datasplit():
    import importlib
    ref = importlib.import_module("EAV_datasplit")
    out = {}
    x = np.arange(400)
    for i in range(6):
        y = np.repeat(np.arange(5), 80)
        if i > 0:
            perm = np.argsort(synth.splitmix64(900 + i, 400))
            y = y[perm]
        out[f"y{i}"] = y.astype(np.int64)
        for h in (40, 56):
            tr, try_, te, tey = ref.EAVDataSplit(x, y).get_split(h_idx=h)
            out[f"tr{i}_{h}"] = tr
            out[f"te{i}_{h}"] = te
            out[f"try{i}_{h}"] = try_
            out[f"tey{i}_{h}"] = tey
    # shape / squeeze behaviour: 3-D features with a singleton axis, and 5-D
    y = out["y1"]
    x3 = np.arange(400 * 1 * 3, dtype=np.float32).reshape(400, 1, 3)
    tr, _, te, _ = ref.EAVDataSplit(x3, y).get_split(40)
    out["x3_tr"], out["x3_te"] = tr, te
    np.savez_compressed(os.path.join(HERE, "datasplit.npz"), **out)
    print("wrote datasplit")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.chdir("/tmp")
    if what in ("datasplit", "all"):
        make_datasplit()
    if what in ("eegnet", "all"):
        make_eegnet()
    if what == "eegnet_generic":
        make_eegnet_generic()
    if what in ("ast", "vit", "all"):
        from make_goldens_tf import make_ast, make_vit  # noqa
        if what in ("ast", "all"):
            make_ast()
        if what in ("vit", "all"):
            make_vit()
