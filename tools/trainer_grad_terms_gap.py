#!/usr/bin/env python3
"""The two trainer goldens (tests/golden/{ast,vit}_trainer.npz: frozen epoch + unfrozen epoch of the UNMODIFIED reference
trainers) re-run with the opt-in fp16-operand gradients (Encoder.grad_terms = 1): how far does `outputs_test` move from the
reference's, next to the default three-term arithmetic?  Decides whether that mode could ever be a default."""
import io
import os
import sys
import tempfile
from contextlib import redirect_stdout
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from eav_amd import synth  # noqa: E402
from tests.test_transformer_trainers_gpu import _save_model_dir  # noqa: E402


def run(kind, terms):
    os.environ["EAV_GRAD_TERMS"] = str(terms)
    os.environ["EAV_ENCODER_PRECISION"] = "split"
    g = np.load(os.path.join(ROOT, "tests", "golden", f"{kind}_trainer.npz"))
    with tempfile.TemporaryDirectory() as td:
        path = _save_model_dir(Path(td), kind, int(g["wseed"]))
        cwd = os.getcwd()
        os.chdir(td)
        try:
            with redirect_stdout(io.StringIO()):
                if kind == "ast":
                    from eav_amd.audio import AudioModelTrainer
                    wav, y = synth.normal(90, (10, 80000), 0.0, 0.1), synth.labels(91, 10)
                    tr = AudioModelTrainer([wav[:6], y[:6], wav[6:], y[6:]], path, sub="s", num_classes=5, batch_size=4)
                else:
                    from eav_amd.vision import ImageClassifierTrainer
                    fr, y = (synth.uniform(92, (10, 2, 56, 56, 3)) * 255).astype(np.uint8), synth.labels(93, 10)
                    tr = ImageClassifierTrainer([fr[:6], y[:6], fr[6:], y[6:]], path, sub="s", num_labels=5, batch_size=4)
                assert tr.model.grad_terms == terms
                tr.model.reset_head(g["head.weight"], g["head.bias"])
                tr.optimizer = type(tr.optimizer)(tr.model.parameters(), lr=tr.initial_lr, weight_decay=0.01, decoupled=True)
                tr.train_dataloader.order_override = [g["order0"], g["order1"]]
                tr.train(epochs=1, lr=5e-4, freeze=True)
                tr.train(epochs=1, lr=5e-6, freeze=False)
        finally:
            os.chdir(cwd)
    return float(np.abs(tr.outputs_test - g["outputs_test"]).max()), float(np.abs(g["outputs_test"]).max())


if __name__ == "__main__":
    for kind in ("ast", "vit"):
        e3, m = run(kind, 3)
        e1, _ = run(kind, 1)
        print(f"{kind} trainer golden: max |outputs_test - reference| = {e3:.2e} (three-term, default) / {e1:.2e} "
              f"(grad_terms = 1); max |logit| {m:.2f}")
