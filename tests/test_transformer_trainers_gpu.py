"""AudioModelTrainer / ImageClassifierTrainer against the UNMODIFIED reference trainers
(tests/golden/{ast,vit}_trainer.npz): frozen epoch then unfrozen epoch on 6 train / 4 test items;
the parity surface is outputs_test within 1e-3, plus the printed accuracy lines."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from eav_amd import synth
from tests.golden_util import tf_weights

pytestmark = pytest.mark.gpu


def _save_model_dir(tmp_path, kind, seed):
    """HF-format directory (config.json + model.safetensors [+ preprocessor_config.json]) with the
    same generator-seeded weights the golden run used."""
    from safetensors.numpy import save_file
    from oracle import vit_oracle as vo
    ocfg = vo.cfg_ast(hidden=64, layers=2, heads=4, ff=128) if kind == "ast" else vo.cfg_vit(hidden=64, layers=2, heads=4, ff=128)
    W = tf_weights(seed, vo.param_shapes(ocfg), std=0.08)
    save_file({k: np.ascontiguousarray(v) for k, v in W.items()}, str(tmp_path / "model.safetensors"))
    common = {"hidden_size": 64, "num_hidden_layers": 2, "num_attention_heads": 4, "intermediate_size": 128,
              "patch_size": 16, "layer_norm_eps": 1e-12, "hidden_act": "gelu",
              "id2label": {str(i): f"LABEL_{i}" for i in range(5)}}
    if kind == "ast":
        cfg = dict(common, model_type="audio-spectrogram-transformer", num_mel_bins=128, max_length=1024,
                   frequency_stride=10, time_stride=10)
    else:
        cfg = dict(common, model_type="vit", image_size=224, num_channels=3)
        json.dump({"do_normalize": True, "do_rescale": True, "do_resize": True, "image_mean": [0.5, 0.5, 0.5],
                   "image_std": [0.5, 0.5, 0.5], "image_processor_type": "ViTImageProcessor", "resample": 2,
                   "rescale_factor": 1 / 255, "size": {"height": 224, "width": 224}},
                  open(tmp_path / "preprocessor_config.json", "w"))
    json.dump(cfg, open(tmp_path / "config.json", "w"))
    return str(tmp_path)


def _compare_lines(got, ref):
    g = [l for l in got.strip().splitlines() if l.startswith("Epoch")]
    r = [l for l in ref.strip().splitlines() if l.startswith("Epoch")]
    assert g == r, (g, r)


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_audio_trainer_matches_reference(golden_dir, tmp_path, monkeypatch, precision):
    from eav_amd.audio import AudioModelTrainer
    monkeypatch.setenv("EAV_ENCODER_PRECISION", precision)
    g = np.load(os.path.join(golden_dir, "ast_trainer.npz"))
    path = _save_model_dir(tmp_path, "ast", int(g["wseed"]))
    monkeypatch.chdir(tmp_path)
    wav = synth.normal(90, (10, 80000), 0.0, 0.1)
    y = synth.labels(91, 10)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr = AudioModelTrainer([wav[:6], y[:6], wav[6:], y[6:]], path, sub="s", num_classes=5, batch_size=4)
        # the reference's own ASTFeatureExtractor call is kept: same input_values
        assert np.allclose(tr.tr_x.numpy(), g["tr_x"], atol=1e-5) and np.allclose(tr.te_x.numpy(), g["te_x"], atol=1e-5)
        tr.model.reset_head(g["head.weight"], g["head.bias"])       # the head the reference drew from its RNG
        tr.optimizer = type(tr.optimizer)(tr.model.parameters(), lr=tr.initial_lr, weight_decay=0.01, decoupled=True)
        tr.train_dataloader.order_override = [g["order0"], g["order1"]]
        tr.train(epochs=1, lr=5e-4, freeze=True)
        assert not hasattr(tr, "outputs_test")                      # Q15: only after the unfrozen phase
        tr.train(epochs=1, lr=5e-6, freeze=False)
    err = np.abs(tr.outputs_test - g["outputs_test"]).max()
    assert tr.outputs_test.shape == g["outputs_test"].shape and tr.outputs_test.dtype == np.float32
    assert err < 1e-3, err
    _compare_lines(buf.getvalue(), str(g["stdout"]))
    assert open("training_performance_audio.txt").read().count("Epoch") == 2      # Q17


@pytest.mark.parametrize("precision", ["split", "fp32"])
def test_vision_trainer_matches_reference(golden_dir, tmp_path, monkeypatch, precision):
    from eav_amd.vision import ImageClassifierTrainer, trial_vote
    monkeypatch.setenv("EAV_ENCODER_PRECISION", precision)
    g = np.load(os.path.join(golden_dir, "vit_trainer.npz"))
    path = _save_model_dir(tmp_path, "vit", int(g["wseed"]))
    monkeypatch.chdir(tmp_path)
    frames = (synth.uniform(92, (10, 2, 56, 56, 3)) * 255).astype(np.uint8)
    y = synth.labels(93, 10)
    buf = io.StringIO()
    with redirect_stdout(buf):
        tr = ImageClassifierTrainer([frames[:6], y[:6], frames[6:], y[6:]], path, sub="s", num_labels=5, batch_size=4)
        assert np.allclose(tr.train_dataloader.x.cpu().numpy(), g["tr_x"], atol=1e-6)
        tr.model.reset_head(g["head.weight"], g["head.bias"])
        tr.optimizer = type(tr.optimizer)(tr.model.parameters(), lr=tr.initial_lr, weight_decay=0.01, decoupled=True)
        tr.train_dataloader.order_override = [g["order0"], g["order1"]]
        tr.train(epochs=1, lr=5e-4, freeze=True)
        tr.train(epochs=1, lr=5e-6, freeze=False)
    err = np.abs(tr.outputs_test - g["outputs_test"]).max()
    assert tr.outputs_test.shape == g["outputs_test"].shape
    assert err < 1e-3, err
    _compare_lines(buf.getvalue(), str(g["stdout"]))
    pred, acc, f1 = trial_vote(tr.outputs_test, g["te_y"], frames_per_trial=2)
    ref_pred = np.argmax(g["outputs_test"].reshape(-1, 2, 5).mean(1), 1)
    assert np.array_equal(pred, ref_pred) and 0.0 <= acc <= 1.0 and 0.0 <= f1 <= 1.0


@pytest.mark.parametrize("precision", ["fp32", "split"])
@pytest.mark.parametrize("kind", ["ast", "vit"])
def test_frozen_phase_feature_cache_changes_nothing(golden_dir, tmp_path, monkeypatch, kind, precision):
    """train(freeze=True) keeps the classifier's input of every sample after the first epoch and runs the later frozen
    epochs on the head alone (finetune.FineTuneBase; Transformer_Audio.py:44-56,113 / Transformer_Vision.py:61-77,151 re-run
    the backbone every epoch).  Three frozen epochs + one unfrozen epoch with and without the cache, ragged last batches
    included: in the exact-fp32 arithmetic outputs_test, the head's weights and its AdamW state are BIT-equal; in the split
    arithmetic (whose patch-plane scale depends on which samples share a batch) they agree to rounding."""
    from eav_amd.audio import AudioModelTrainer
    from eav_amd.vision import ImageClassifierTrainer
    monkeypatch.setenv("EAV_ENCODER_PRECISION", precision)
    path = _save_model_dir(tmp_path, kind, 5)
    monkeypatch.chdir(tmp_path)
    if kind == "ast":
        x = synth.normal(95, (11, 80000), 0.0, 0.1)
        y = synth.labels(96, 11)
        data = [x[:7], y[:7], x[7:], y[7:]]
    else:
        x = (synth.uniform(97, (11, 2, 56, 56, 3)) * 255).astype(np.uint8)
        y = synth.labels(98, 11)
        data = [x[:7], y[:7], x[7:], y[7:]]

    def run(cache):
        import torch
        torch.manual_seed(3)                                    # same head init, same shuffles
        with redirect_stdout(io.StringIO()):
            tr = (AudioModelTrainer(data, path, sub="s", num_classes=5, batch_size=4) if kind == "ast" else
                  ImageClassifierTrainer(data, path, sub="s", num_labels=5, batch_size=4))
            tr.cache_frozen_features = cache
            tr.train(epochs=3, lr=5e-4, freeze=True)
            tr.train(epochs=1, lr=5e-6, freeze=False)
        head = {k: v.detach().clone() for k, v in tr.model.classifier.state_dict().items()}
        state = [(tr.optimizer.state[p]["step"], tr.optimizer.state[p]["exp_avg"].clone(),
                  tr.optimizer.state[p]["exp_avg_sq"].clone()) for p in tr.model.classifier.parameters()]
        return tr.outputs_test, head, state

    out1, head1, st1 = run(True)
    out0, head0, st0 = run(False)
    if precision == "fp32":
        assert np.array_equal(out1, out0)
        assert all(bool((head1[k] == head0[k]).all()) for k in head0)
        assert all(a[0] == b[0] and bool((a[1] == b[1]).all()) and bool((a[2] == b[2]).all()) for a, b in zip(st1, st0))
    else:
        assert np.abs(out1 - out0).max() < 2e-5
        assert all(float((head1[k] - head0[k]).abs().max()) < 2e-5 for k in head0)
        assert all(a[0] == b[0] for a, b in zip(st1, st0))


@pytest.mark.parametrize("kind", ["ast", "vit"])
def test_trainer_goldens_with_fp16_operand_gradients(kind, monkeypatch):
    """The opt-in `grad_terms = 1` (backward GEMMs on the hi.hi term alone) on the two reference-trainer goldens: after the
    frozen + unfrozen epoch `outputs_test` stays within 1e-4 of the unmodified reference trainer's (measured 2.5e-6 AST /
    7.2e-6 ViT; three-term default 5e-7 / 1e-6) - far inside north_star's 1e-3.  It remains opt-in for what these two-step
    goldens cannot show: over 40 AdamW steps the held-out logits drift by 1.6-2.1e-3 (tools/encoder_trajectory.py,
    DESIGN.md Appendix B)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "gap_tool", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "trainer_grad_terms_gap.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    env = {k: os.environ.get(k) for k in ("EAV_GRAD_TERMS", "EAV_ENCODER_PRECISION")}
    try:
        e1, _ = tool.run(kind, 1)
    finally:
        for k, v in env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print(f"{kind}: grad_terms = 1: max |outputs_test - reference| = {e1:.2e}")
    assert e1 < 1e-4
