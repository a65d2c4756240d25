"""Device-side pre-processing either side of the encoders (SURVEY.md section 8f "next" rows).

frames_to_pixel_values: what ImageClassifierTrainer.preprocess_images does frame by frame on the host
through the HF image processor (Transformer_Vision.py:52-59) - Pillow's 8-bit bilinear resize, 1/255
rescale, (x-mean)/std - for a whole batch of uint8 HWC frames in one kernel launch
(`eav_resize_normalize_u8`).  The coefficient tables follow Pillow's precompute_coeffs /
normalize_coeffs_8bpc (src/libImaging/Resample.c) so that the result is bit-identical to the host path.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_PRECISION_BITS = 32 - 8 - 2


def pillow_bilinear_tables(in_size: int, out_size: int):
    """(bounds int32 [out,2] = (first input index, tap count), coeffs int32 [out,ksize])."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = filterscale                      # bilinear support 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.float64)[None, :]
    w = np.maximum(0.0, 1.0 - np.abs((taps + xmin[:, None] - center[:, None] + 0.5) / filterscale))
    w = np.where(taps < xmax[:, None], w, 0.0)
    ww = w.sum(1, keepdims=True)
    w = np.where(ww != 0.0, w / np.where(ww == 0.0, 1.0, ww), w)
    fixed = np.where(w < 0, (-0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64),
                     (0.5 + w * (1 << _PRECISION_BITS)).astype(np.int64)).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, fixed


def frames_to_pixel_values(frames, size=(224, 224), mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), rescale=1.0 / 255.0,
                           device="cuda"):
    """frames: uint8 array/tensor [n,H,W,C] (C <= 3) -> float32 device tensor [n,C,OH,OW]."""
    fr = torch.as_tensor(np.ascontiguousarray(frames) if isinstance(frames, np.ndarray) else frames)
    if fr.dtype != torch.uint8 or fr.dim() != 4 or fr.shape[-1] > 3:
        raise ValueError("frames must be uint8 [n,H,W,C] with C <= 3")
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.EavError("frames_to_pixel_values runs on the MI355X only (no CPU fallback)")
    fr = fr.to(dev).contiguous()
    n, H, W, C = fr.shape
    OH, OW = size
    bx, kx = pillow_bilinear_tables(W, OW)
    by, ky = pillow_bilinear_tables(H, OH)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    kxd, bxd, kyd, byd = t(kx), t(bx), t(ky), t(by)
    out = torch.empty(n, C, OH, OW, dtype=torch.float32, device=dev)
    m = np.asarray(list(mean) + [0.0, 0.0], np.float32)[:3].copy()
    s = np.asarray(list(std) + [1.0, 1.0], np.float32)[:3].copy()
    _lib.call("eav_resize_normalize_u8", fr.data_ptr(), kxd.data_ptr(), bxd.data_ptr(), kyd.data_ptr(), byd.data_ptr(),
              out.data_ptr(), n, H, W, C, OH, OW, kx.shape[1], ky.shape[1], float(rescale), m.ctypes.data,
              s.ctypes.data, _lib.stream_ptr())
    torch.cuda.current_stream().synchronize()      # tables / host constants must outlive the launch
    return out


# ------------------------------------------------------------------------------------------------
def kaldi_mel_filters(nbins=257, nmel=128, fmin=20.0, fmax=8000.0, sr=16000):
    """[nbins, nmel] float64 kaldi-mel triangular filters (HF mel_filter_bank(..., mel_scale='kaldi',
    triangularize_in_mel_space=True, norm=None), as ASTFeatureExtractor builds them)."""
    mel = lambda f: 1127.0 * np.log(1.0 + f / 700.0)  # noqa: E731
    mel_freqs = np.linspace(mel(fmin), mel(fmax), nmel + 2)
    fft_freqs = mel(sr / ((nbins - 1) * 2) * np.arange(nbins))
    diff = np.diff(mel_freqs)
    slopes = mel_freqs[None, :] - fft_freqs[:, None]
    return np.maximum(0.0, np.minimum(-slopes[:, :-2] / diff[:-1], slopes[:, 2:] / diff[1:]))


_FBANK_TABLES = {}


def waveforms_to_input_values(wav, max_length=1024, nmel=128, mean=-4.2677393, std=4.5689974, device="cuda"):
    """AudioModelTrainer._feature_extract on the GPU: wav [n,L] float (16 kHz mono) ->
    input_values float32 [n,max_length,nmel] (device tensor), the HF ASTFeatureExtractor recipe."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.EavError("waveforms_to_input_values runs on the MI355X only (no CPU fallback)")
    w = torch.as_tensor(np.asarray(wav, dtype=np.float32) if not isinstance(wav, torch.Tensor) else wav).float()
    if w.dim() == 1:
        w = w[None]
    w = w.to(dev).contiguous()
    n, L = w.shape
    key = (str(dev), nmel)
    if key not in _FBANK_TABLES:
        k = np.arange(256)
        tw = np.stack([np.cos(2 * np.pi * k / 512), -np.sin(2 * np.pi * k / 512)], 1)
        melT = np.ascontiguousarray(kaldi_mel_filters(257, nmel).T)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)  # noqa: E731
        _FBANK_TABLES[key] = (t(np.hanning(400)), t(tw), t(melT))
    win, tw, melT = _FBANK_TABLES[key]
    out = torch.empty(n, max_length, nmel, dtype=torch.float32, device=dev)
    _lib.call("eav_ast_fbank", w.data_ptr(), win.data_ptr(), tw.data_ptr(), melT.data_ptr(), out.data_ptr(), n, L,
              max_length, nmel, 0.97, 1.192092955078125e-07, float(np.float32(mean)), float(np.float32(std * 2)),
              _lib.stream_ptr())
    return out
