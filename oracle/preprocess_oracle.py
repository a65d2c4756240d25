"""Oracle: the host pre-processing either side of the encoders, restated in numpy.

TEST INFRASTRUCTURE - see oracle/__init__.py.

resize_bilinear_u8 restates Pillow's 8-bit resampler (third-party dependency of the reference via the
HF image processor, Transformer_Vision.py:56; Pillow's published algorithm in src/libImaging/Resample.c:
precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc): triangle filter,
support scaled by max(in/out, 1), coefficients normalised in float64 and rounded to 22-bit fixed point,
horizontal pass rounded to uint8 before the vertical pass.  vit_preprocess adds the HF steps
(image * (1/255) in float64 -> float32, then (x - mean) / std in float32).  Pinned against the processed
frames the reference trainer itself produced (tests/golden/vit_trainer.npz: tr_x / te_x).
"""
from __future__ import annotations

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """-> (bounds int32 [out,2] = (first, count), coeffs int32 [out, ksize]) - Pillow precompute_coeffs +
    normalize_coeffs_8bpc for the bilinear (triangle, support 1.0) filter over the whole input range."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) * ss)) for x in range(xmax)], np.float64)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64),
                     (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)).astype(np.int32)
    return bounds, fixed


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H,W,C] -> uint8 [out_h,out_w,C] exactly as PIL.Image.resize(..., BILINEAR) does."""
    H, W, C = img.shape
    bx, kx = precompute_coeffs(W, out_w)
    by, ky = precompute_coeffs(H, out_h)
    src = img.astype(np.int64)
    tmp = np.zeros((H, out_w, C), np.uint8)
    for xx in range(out_w):
        x0, n = bx[xx]
        acc = (1 << (PRECISION_BITS - 1)) + (src[:, x0:x0 + n, :] * kx[xx, :n].astype(np.int64)[None, :, None]).sum(1)
        tmp[:, xx, :] = _clip8(acc)
    t64 = tmp.astype(np.int64)
    out = np.zeros((out_h, out_w, C), np.uint8)
    for yy in range(out_h):
        y0, n = by[yy]
        acc = (1 << (PRECISION_BITS - 1)) + (t64[y0:y0 + n] * ky[yy, :n].astype(np.int64)[:, None, None]).sum(0)
        out[yy] = _clip8(acc)
    return out


def vit_preprocess(frames, size=224, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale=1 / 255):
    """frames uint8 [n,H,W,3] -> float32 [n,3,size,size] (HF ViTImageProcessor defaults, Transformer_Vision.py:52-59)."""
    out = np.zeros((len(frames), 3, size, size), np.float32)
    m = np.asarray(mean, np.float32)[:, None, None]
    s = np.asarray(std, np.float32)[:, None, None]
    for i, f in enumerate(frames):
        r = resize_bilinear_u8(f, size, size).transpose(2, 0, 1)
        x = (r.astype(np.float64) * rescale).astype(np.float32)
        out[i] = (x - m) / s
    return out


def trial_vote(outputs_test, frames_per_trial=25):
    """reshape(n_trials, frames, classes).mean(1).argmax(1) - Transformer_Vision.py:177-180."""
    a = np.reshape(outputs_test, (-1, frames_per_trial, outputs_test.shape[-1]), 'C')
    return np.argmax(np.mean(a, 1), axis=1)


# ---------------------------------------------------------------------------------------------------
# AST log-mel front-end: HF ASTFeatureExtractor numpy path (transformers 5.15.0,
# models/audio_spectrogram_transformer/feature_extraction_audio_spectrogram_transformer.py:91-158 and
# audio_utils.spectrogram / mel_filter_bank), called by the reference at Transformer_Audio.py:38-42.
# Pinned against tests/golden/ast_trainer.npz (tr_x / te_x produced by the reference trainer itself).
def kaldi_mel_filters(nbins=257, nmel=128, fmin=20.0, fmax=8000.0, sr=16000):
    """[nbins, nmel] float64: triangles in kaldi-mel space (1127 ln(1 + f/700)), no normalisation."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)  # noqa: E731
    mel_freqs = np.linspace(mel(fmin), mel(fmax), nmel + 2)
    fft_freqs = mel(sr / ((nbins - 1) * 2) * np.arange(nbins))
    diff = np.diff(mel_freqs)
    slopes = mel_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    return np.maximum(0.0, np.minimum(down, up))


def ast_fbank(wav, max_length=1024, nmel=128, mean=-4.2677393, std=4.5689974):
    """wav float32 [n,L] @16 kHz -> input_values float32 [n,max_length,nmel]."""
    wav = np.asarray(wav, np.float32)
    window = np.hanning(400).astype(np.float64)
    filt = kaldi_mel_filters(257, nmel)
    out = []
    for w in wav:
        x = w.astype(np.float64)
        nfr = 1 + (x.size - 400) // 160
        idx = np.arange(400)[None, :] + 160 * np.arange(nfr)[:, None]
        fr = x[idx]
        fr = fr - fr.mean(1, keepdims=True)                                  # remove_dc_offset
        pe = np.empty_like(fr)
        pe[:, 1:] = fr[:, 1:] - 0.97 * fr[:, :-1]                            # preemphasis
        pe[:, 0] = fr[:, 0] * (1 - 0.97)
        buf = np.zeros((nfr, 512))
        buf[:, :400] = pe * window
        spec = np.fft.rfft(buf, axis=1).astype(np.complex64)                 # the reference stores complex64
        power = np.abs(spec, dtype=np.float64) ** 2.0
        melspec = np.maximum(1.192092955078125e-07, filt.T @ power.T)        # [nmel, nfr]
        fb = np.log(melspec).astype(np.float32).T                            # [nfr, nmel]
        if nfr < max_length:
            fb = np.concatenate([fb, np.zeros((max_length - nfr, nmel), np.float32)], 0)
        else:
            fb = fb[:max_length]
        out.append((fb - mean) / (std * 2))
    return np.stack(out).astype(np.float32)


# ---------------------------------------------------------------------------------------------------
# EEG pre-processing (Dataload_eeg.py:85-152).  The arithmetic is scipy's (third-party; installed here: see
# tests/golden/make_goldens_eeg.py for the version stamp): resample_poly = zero-phase decimating FIR with a
# Kaiser(5.0)-windowed sinc of 2*10*down+1 taps; sosfilt = cascade of direct-form-II-transposed biquads.
def resample_poly_down(x, down):
    """scipy.signal.resample_poly(x, 1, down, axis=-1) for a float64 array [..., n]."""
    half = 10 * down
    m = np.arange(-half, half + 1, dtype=np.float64)
    h = (1.0 / down) * np.sinc(m / down) * np.kaiser(2 * half + 1, 5.0)
    h /= h.sum()
    n = x.shape[-1]
    n_out = n // down + bool(n % down)
    flat = x.reshape(-1, n)
    out = np.empty((flat.shape[0], n_out))
    for i, row in enumerate(flat):
        full = np.convolve(row, h)                     # full[i] = sum_j h[j] row[i-j]
        out[i] = full[half::down][:n_out]              # y[m] = full[m*down + half]
    return out.reshape(x.shape[:-1] + (n_out,))


def sosfilt_loop(sos, x):
    """scipy.signal.sosfilt(sos, x) for 1-D float64 x - plain Python recurrence (small cases only)."""
    sos = np.asarray(sos, np.float64)
    z = np.zeros((sos.shape[0], 2))
    y = np.empty_like(x, dtype=np.float64)
    for n, v in enumerate(np.asarray(x, np.float64)):
        for s in range(sos.shape[0]):
            b0, b1, b2, _, a1, a2 = sos[s]
            o = b0 * v + z[s, 0]
            z[s, 0] = b1 * v - a1 * o + z[s, 1]
            z[s, 1] = b2 * v - a2 * o
            v = o
        y[n] = v
    return y


def eeg_segment(seg_f, label, selected=(1, 3, 5, 7, 9), win=500):
    """segment_and_select_classes (Dataload_eeg.py:123-152) for seg_f [ch, t, trials], one-hot label [10, trials]."""
    ch, t, tri = seg_f.shape
    nwin = t // win
    tm1 = seg_f.reshape((ch, win, nwin, tri), order='F')
    div = tm1.reshape((ch, win, nwin * tri), order='F')
    lab = np.repeat(label, repeats=nwin, axis=1)
    mask = np.isin(np.argmax(lab, axis=0), list(selected))
    return np.transpose(div[:, :, mask], (2, 0, 1)), np.argmax(lab[:, mask], axis=0)
