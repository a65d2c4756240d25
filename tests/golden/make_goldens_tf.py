"""Goldens for the AST / ViT path, produced by the Hugging Face classes the reference calls
(Transformer_Audio.py:22, Transformer_Vision.py:29) and by the reference trainers themselves
(shims S4/S5 of SURVEY.md section 8c).  Development container only; data-only fixtures."""
from __future__ import annotations

import io
import os
import sys
import tempfile
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
for p in (ROOT, REF):
    if p not in sys.path:
        sys.path.insert(0, p)

from eav_amd import synth  # noqa: E402
from oracle import vit_oracle as vo  # noqa: E402
from tests.golden_util import tf_weights  # noqa: E402


def hf_model(cfg):
    from transformers import ASTConfig, ASTForAudioClassification, ViTConfig, ViTForImageClassification
    if cfg["kind"] == "ast":
        c = ASTConfig(hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                      intermediate_size=cfg["ff"], patch_size=cfg["patch"], frequency_stride=cfg["fstride"],
                      time_stride=cfg["tstride"], max_length=cfg["frames"], num_mel_bins=cfg["mel"],
                      layer_norm_eps=cfg["eps"], num_labels=cfg["num_labels"])
        return ASTForAudioClassification(c)
    c = ViTConfig(hidden_size=cfg["hidden"], num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"],
                  intermediate_size=cfg["ff"], image_size=cfg["image"], patch_size=cfg["patch"],
                  num_channels=cfg["channels"], layer_norm_eps=cfg["eps"], num_labels=cfg["num_labels"])
    return ViTForImageClassification(c)


def load(model, W):
    sd = model.state_dict()
    assert set(sd) == set(W), (set(sd) ^ set(W))
    model.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in W.items()})


def batch(cfg, seed, B):
    if cfg["kind"] == "ast":
        return synth.mel_batch(seed, B, cfg["frames"], cfg["mel"])
    return synth.frame_batch(seed, B, cfg["image"])


def reduced_case(kind):
    """2 layers, hidden 64, 4 heads, ff 128, full token count (1214 / 197): logits, loss, all
    gradients, parameters after an AdamW step; then a frozen-backbone step (Q11)."""
    cfg = vo.cfg_ast(hidden=64, layers=2, heads=4, ff=128) if kind == "ast" else vo.cfg_vit(hidden=64, layers=2, heads=4, ff=128)
    W = tf_weights(5 if kind == "ast" else 6, vo.param_shapes(cfg), std=0.08)
    torch.manual_seed(0)
    model = hf_model(cfg)
    load(model, W)
    model.train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)      # wd defaults to 0.01 (Q10)
    B = 2
    out = {"kind": kind, "B": B, "wseed": 5 if kind == "ast" else 6, "xseed": 50, "lr": 1e-3, "std": 0.08}
    hk = set(vo.head_keys(cfg))
    for s, freeze in enumerate((False, True)):
        x, y = batch(cfg, 50 + s, B)
        for k, p in model.named_parameters():
            p.requires_grad = (not freeze) or (k in hk)
        opt.zero_grad()
        o = model(torch.from_numpy(x), labels=torch.from_numpy(y)) if kind == "vit" else model(torch.from_numpy(x))
        loss = o.loss if kind == "vit" else torch.nn.CrossEntropyLoss()(o.logits, torch.from_numpy(y))
        loss.backward()
        out[f"logits{s}"] = o.logits.detach().numpy().copy()
        out[f"loss{s}"] = np.float32(loss.item())
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[f"grad{s}.{k}"] = p.grad.numpy().copy()
        opt.step()
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[f"post{s}.{k}"] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, f"{kind}_reduced.npz"), **out)
    print("wrote", kind, "reduced", out["logits0"])


def full_case(kind):
    """Full-size 12-layer model, generator-seeded weights: only the seeds and the [2,5] logits are stored."""
    cfg = vo.cfg_ast() if kind == "ast" else vo.cfg_vit()
    W = tf_weights(7 if kind == "ast" else 8, vo.param_shapes(cfg), std=0.02)
    model = hf_model(cfg)
    load(model, W)
    model.eval()
    x, y = batch(cfg, 70, 2)
    with torch.no_grad():
        logits = model(torch.from_numpy(x)).logits.numpy()
    np.savez_compressed(os.path.join(HERE, f"{kind}_full.npz"), kind=kind, wseed=7 if kind == "ast" else 8,
                        xseed=70, B=2, logits=logits, nparams=sum(v.size for v in W.values()))
    print("wrote", kind, "full", logits)


def trainer_case(kind):
    """The UNMODIFIED reference trainer: train(1, 5e-4, freeze=True) then train(1, 5e-6, freeze=False) on
    6 train / 4 test synthetic items; outputs_test (Q15) is the parity surface.  Reduced model config
    (the trainers take whatever from_pretrained(model_path) yields)."""
    if kind == "ast":
        cfg = vo.cfg_ast(hidden=64, layers=2, heads=4, ff=128)
    else:
        cfg = vo.cfg_vit(hidden=64, layers=2, heads=4, ff=128)
    W = tf_weights(9 if kind == "ast" else 10, vo.param_shapes(cfg), std=0.08)
    model = hf_model(cfg)
    load(model, W)
    tmp = tempfile.mkdtemp()
    model.save_pretrained(tmp)
    out = {"kind": kind, "wseed": 9 if kind == "ast" else 10}
    os.chdir(tmp)
    buf = io.StringIO()
    if kind == "ast":
        import Transformer_torch.Transformer_Audio as TA
        # synthetic waveforms [N,80000] -> the trainer's own ASTFeatureExtractor (numpy fallback)
        wav = synth.normal(90, (10, 80000), 0.0, 0.1)
        y = synth.labels(91, 10)
        data = [wav[:6], y[:6], wav[6:], y[6:]]
        torch.manual_seed(0)
        with redirect_stdout(buf):
            tr = TA.AudioModelTrainer(data, tmp, sub="s", num_classes=5, batch_size=4)
            # the freshly created head is torch-default-initialised from the global RNG: record it
            out["head.weight"] = tr.model.classifier.dense.weight.detach().numpy().copy()
            out["head.bias"] = tr.model.classifier.dense.bias.detach().numpy().copy()
            out["tr_x"] = tr.tr_x.numpy().copy()
            out["te_x"] = tr.te_x.numpy().copy()
            orders = _record_orders(lambda: (tr.train(epochs=1, lr=5e-4, freeze=True), tr.train(epochs=1, lr=5e-6, freeze=False)))
    else:
        import Transformer_torch.Transformer_Vision as TV
        from transformers.models.vit.image_processing_pil_vit import ViTImageProcessorPil

        class _Shim:  # S5: AutoImageProcessor needs torchvision
            @staticmethod
            def from_pretrained(p):
                return ViTImageProcessorPil.from_pretrained(p)
        ViTImageProcessorPil().save_pretrained(tmp)
        TV.AutoImageProcessor = _Shim
        frames = (synth.uniform(92, (10, 2, 56, 56, 3)) * 255).astype(np.uint8)
        y = synth.labels(93, 10)
        data = [frames[:6], y[:6], frames[6:], y[6:]]
        torch.manual_seed(0)
        with redirect_stdout(buf):
            tr = TV.ImageClassifierTrainer(data, tmp, sub="s", num_labels=5, batch_size=4)
            tr.model.config.num_labels = 5
            out["head.weight"] = tr.model.classifier.weight.detach().numpy().copy()
            out["head.bias"] = tr.model.classifier.bias.detach().numpy().copy()
            out["tr_x"] = tr.train_dataloader.dataset.tensors[0].numpy().copy()
            out["te_x"] = tr.test_dataloader.dataset.tensors[0].numpy().copy()
            orders = _record_orders(lambda: (tr.train(epochs=1, lr=5e-4, freeze=True), tr.train(epochs=1, lr=5e-6, freeze=False)))
    out["tr_y"], out["te_y"] = y[:6], y[6:]
    for i, o in enumerate(orders):
        out[f"order{i}"] = o
    out["outputs_test"] = tr.outputs_test.copy()
    out["stdout"] = np.array(buf.getvalue())
    np.savez_compressed(os.path.join(HERE, f"{kind}_trainer.npz"), **out)
    print("wrote", kind, "trainer", tr.outputs_test, buf.getvalue().strip().splitlines()[-1])


def _record_orders(fn):
    from torch.utils.data import sampler as _sampler
    orders = []
    orig = _sampler.RandomSampler.__iter__

    def rec(self):
        idx = list(orig(self))
        orders.append(np.array(idx, dtype=np.int64))
        return iter(idx)
    _sampler.RandomSampler.__iter__ = rec
    try:
        fn()
    finally:
        _sampler.RandomSampler.__iter__ = orig
    return orders


def make_ast():
    reduced_case("ast")
    full_case("ast")
    trainer_case("ast")


def make_vit():
    reduced_case("vit")
    full_case("vit")
    trainer_case("vit")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    cwd = os.getcwd()
    if what in ("ast", "all"):
        make_ast()
    os.chdir(cwd)
    if what in ("vit", "all"):
        make_vit()
