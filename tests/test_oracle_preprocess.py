"""Pin oracle/preprocess_oracle.py: Pillow-exact resize vs PIL itself and vs the frames the reference
trainer's own HF processor produced (tests/golden/vit_trainer.npz)."""
import os

import numpy as np

from eav_amd import synth
from oracle import preprocess_oracle as po


def test_resize_matches_pillow_bit_exact():
    from PIL import Image
    for seed, (h, w, oh, ow) in enumerate([(56, 56, 224, 224), (56, 56, 100, 75), (64, 48, 32, 24), (7, 9, 20, 31)]):
        img = (synth.uniform(500 + seed, (h, w, 3)) * 256).astype(np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), resample=Image.BILINEAR))
        got = po.resize_bilinear_u8(img, oh, ow)
        assert np.array_equal(got, ref), (h, w, oh, ow, np.abs(got.astype(int) - ref.astype(int)).max())


def test_vit_preprocess_matches_reference_trainer_frames(golden_dir):
    g = np.load(os.path.join(golden_dir, "vit_trainer.npz"))
    frames = (synth.uniform(92, (10, 2, 56, 56, 3)) * 255).astype(np.uint8)
    got = po.vit_preprocess(frames[:6].reshape(-1, 56, 56, 3))
    assert got.shape == g["tr_x"].shape
    assert np.array_equal(got, g["tr_x"])                    # bit-exact, float32
    got_te = po.vit_preprocess(frames[6:].reshape(-1, 56, 56, 3))
    assert np.array_equal(got_te, g["te_x"])


def test_trial_vote():
    o = synth.normal(7, (50, 5))
    assert np.array_equal(po.trial_vote(o, 25), o.reshape(2, 25, 5).mean(1).argmax(1))


def test_ast_fbank_matches_reference_trainer_features(golden_dir):
    g = np.load(os.path.join(golden_dir, "ast_trainer.npz"))
    wav = synth.normal(90, (10, 80000), 0.0, 0.1)
    got = po.ast_fbank(wav[:6])
    assert got.shape == g["tr_x"].shape == (6, 1024, 128) and got.dtype == np.float32
    assert np.abs(got - g["tr_x"]).max() < 2e-6
    assert np.abs(po.ast_fbank(wav[6:]) - g["te_x"]).max() < 2e-6
    assert abs(float(got[0, 600, 0]) - 0.4670) < 1e-4          # padded frames carry the pad value


def test_resample_and_sosfilt_restatements_match_scipy():
    from scipy import signal
    x = synth.normal(61, (3, 1003)).astype(np.float64)
    for down in (5, 4, 2):
        assert np.abs(po.resample_poly_down(x, down) - signal.resample_poly(x, 1, down, axis=1)).max() < 1e-12
    sos = signal.butter(5, [5, 30], btype='bandpass', fs=100, output='sos')
    y = po.sosfilt_loop(sos, x[0])
    assert np.abs(y - signal.sosfilt(sos, x[0])).max() < 1e-12


def test_eeg_pipeline_restatement_matches_reference_golden(golden_dir):
    """oracle resample -> scipy sosfilt (the reference's own engine; the pure-Python oracle loop is pinned to it
    above) -> oracle segmentation == the reference DataLoadEEG run (tests/golden/eeg_preprocess.npz)."""
    from scipy import signal
    from tests.golden.make_goldens_eeg import synthetic_recording
    g = np.load(os.path.join(golden_dir, "eeg_preprocess.npz"))
    x, lab = synthetic_recording(int(g["seed"]))
    ch, t, tri = x.shape
    tm = np.reshape(x, [ch, t * tri], order='F')
    down = np.reshape(po.resample_poly_down(tm, 5), [ch, t // 5, tri], order='F')
    assert np.abs(down[::3, ::41, ::17] - g["down_sample"]).max() < 1e-10
    sos = signal.butter(5, [5, 30], btype='bandpass', fs=100, output='sos')
    filt = signal.sosfilt(sos, np.reshape(down, [ch, -1], order='F'), axis=1).reshape((ch, t // 5, tri), order='F')
    out, labels = po.eeg_segment(filt, lab)
    assert out.shape == tuple(g["shape"]) and np.array_equal(labels, g["labels"])
    assert np.abs(out[::7, ::3, ::11] - g["out_sample"]).max() < 1e-9
    assert abs(np.abs(out).sum() - float(g["out_abssum"])) < 1e-6 * float(g["out_abssum"])
