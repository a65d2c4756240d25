"""Deterministic synthetic-data generator (repo-owned, numpy only).

The reference ships no data and never seeds anything, so
every parity fixture in this repo is regenerated from a 64-bit seed with the
counter-based splitmix64 generator below.  It is a pure function of
(seed, index): the GPU box regenerates bit-identical inputs and weights from a
seed without carrying files and without depending on the stability of
``torch.manual_seed`` streams across torch builds.

Shapes follow BASELINE.json: EEG ``[B,30,10000]``, mel ``[B,1024,128]``
(reference layout, time-major - Transformer_Audio.py:40-42) and frames
``[B,3,224,224]``.
"""
from __future__ import annotations

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, n: int, offset: int = 0) -> np.ndarray:
    """n 64-bit words: word i = mix(seed + (offset+i+1)*GOLDEN)."""
    with np.errstate(over="ignore"):
        idx = np.arange(offset + 1, offset + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    """float32 uniform in [lo, hi): 24 random bits per value, exact in fp32."""
    n = int(np.prod(shape))
    u = (splitmix64(seed, n) >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normal(seed: int, shape, mean: float = 0.0, std: float = 1.0) -> np.ndarray:
    """float32 approximately N(mean, std^2): centred sum of four 16-bit uniforms
    from one splitmix64 word (Irwin-Hall, variance 4/12) - no transcendental,
    hence bit-identical on every host."""
    n = int(np.prod(shape))
    z = splitmix64(seed, n)
    m = np.uint64(0xFFFF)
    s = ((z & m).astype(np.float64) + ((z >> np.uint64(16)) & m).astype(np.float64)
         + ((z >> np.uint64(32)) & m).astype(np.float64) + (z >> np.uint64(48)).astype(np.float64))
    s = (s * (1.0 / 65536.0) - 2.0) * np.sqrt(3.0)
    return (mean + std * s).astype(np.float32).reshape(shape)


def labels(seed: int, n: int, num_classes: int = 5) -> np.ndarray:
    """int64 labels uniform in [0, num_classes)."""
    return (splitmix64(seed, n) % np.uint64(num_classes)).astype(np.int64)


def eeg_batch(seed: int, batch: int, chans: int = 30, samples: int = 10000, n_classes: int = 5):
    """x [B,1,chans,samples] fp32 N(0,1), y [B] int64 (BASELINE configs 1-2)."""
    x = normal(seed, (batch, 1, chans, samples))
    y = labels(seed ^ 0x5EED, batch, n_classes)
    return x, y


def eeg_subject(subject: int, trials: int = 200, chans: int = 30, samples: int = 10000):
    """One synthetic subject: x [trials,chans,samples], y[i] = i mod 5
    (SURVEY.md section 8d, config 1)."""
    x = normal(1000 + subject, (trials, chans, samples))
    y = (np.arange(trials) % 5).astype(np.int64)
    return x, y


def mel_batch(seed: int, batch: int, frames: int = 1024, mels: int = 128):
    """AST input_values [B,1024,128] fp32, N(0,0.5^2); in every second item the
    rows >= 498 hold the feature extractor's pad value 0.4670."""
    x = normal(seed, (batch, frames, mels), 0.0, 0.5)
    x[1::2, 498:, :] = np.float32(0.4670)
    y = labels(seed ^ 0x5EED, batch)
    return x, y


def frame_batch(seed: int, batch: int, size: int = 224):
    """ViT pixel_values [B,3,size,size] fp32 uniform[-1,1) (post-processor range)."""
    x = uniform(seed, (batch, 3, size, size), -1.0, 1.0)
    y = labels(seed ^ 0x5EED, batch)
    return x, y
