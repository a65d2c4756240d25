"""Deterministic weight builders shared by the golden generator and the tests
(pure functions of a seed, built on eav_amd.synth - no reference code)."""
import numpy as np

from eav_amd import synth


def eegnet_weights(seed, samples, chans=30, klen=300, F1=8, D=8, F2=64, nb=5, scale=1.0):
    nflat = F2 * (samples // 4 // 8)
    u = synth.uniform

    def ku(s, shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return u(s, shape, -b, b) * np.float32(scale)

    return {
        "firstConv.weight": ku(seed + 1, (F1, 1, 1, klen), klen),
        "firstBN.weight": u(seed + 2, (F1,), 0.8, 1.2),
        "firstBN.bias": u(seed + 3, (F1,), -0.1, 0.1),
        "depthwiseConv.weight": ku(seed + 4, (F1 * D, 1, chans, 1), chans),
        "depthwiseBN.weight": u(seed + 5, (F1 * D,), 0.8, 1.2),
        "depthwiseBN.bias": u(seed + 6, (F1 * D,), -0.1, 0.1),
        "separableConv.weight": ku(seed + 7, (F2, F1 * D, 1, 16), F1 * D * 16),
        "separableBN.weight": u(seed + 8, (F2,), 0.8, 1.2),
        "separableBN.bias": u(seed + 9, (F2,), -0.1, 0.1),
        "dense.weight": ku(seed + 10, (nb, nflat), nflat),
        "dense.bias": ku(seed + 11, (nb,), nflat),
        "firstBN.running_mean": u(seed + 12, (F1,), -0.05, 0.05),
        "firstBN.running_var": u(seed + 13, (F1,), 0.5, 1.5),
        "depthwiseBN.running_mean": u(seed + 14, (F1 * D,), -0.05, 0.05),
        "depthwiseBN.running_var": u(seed + 15, (F1 * D,), 0.5, 1.5),
        "separableBN.running_mean": u(seed + 16, (F2,), -0.05, 0.05),
        "separableBN.running_var": u(seed + 17, (F2,), 0.5, 1.5),
    }


def tf_weights(seed, shapes, std=0.02):
    """Deterministic transformer weights keyed by HF names: matrices ~ N(0, std^2) (HF
    initializer_range 0.02), biases / tokens / position embeddings small non-zero, LayerNorm
    gains around 1 - so that no term of the forward or backward is trivially zero."""
    out = {}
    for i, (k, shp) in enumerate(shapes.items()):
        s = seed * 1000 + i
        if k.endswith("layernorm.weight") or "layernorm_before.weight" in k or "layernorm_after.weight" in k:
            out[k] = synth.uniform(s, shp, 0.8, 1.2)
        elif k.endswith(".bias"):
            out[k] = synth.normal(s, shp, 0.0, 0.02)
        elif "token" in k or "position_embeddings" in k:
            out[k] = synth.normal(s, shp, 0.0, 0.02)
        else:
            out[k] = synth.normal(s, shp, 0.0, std)
    return out


def cnn_eeg_weights(seed, nb, chans=64, samples=128, klen=64, F1=8, D=2, F2=16, k2=16):
    """Deterministic state for the canonical EEGNet (keys of CNN_torch/CNN_EEG.py's state_dict)."""
    C2, nflat = F1 * D, F2 * (samples // 4 // 8)
    u = synth.uniform

    def ku(s, shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return u(s, shape, -b, b)

    out = {
        "block1.0.weight": ku(seed + 1, (F1, 1, 1, klen), klen),
        "block1.2.weight": ku(seed + 2, (C2, 1, chans, 1), chans),
        "block2.0.weight": ku(seed + 3, (C2, 1, 1, k2), k2),
        "block2.1.weight": ku(seed + 4, (F2, C2, 1, 1), C2),
        "classifier.weight": ku(seed + 5, (nb, nflat), nflat),
        "classifier.bias": ku(seed + 6, (nb,), nflat),
    }
    for i, (name, n) in enumerate((("block1.1", F1), ("block1.3", C2), ("block2.2", F2))):
        out[name + ".weight"] = u(seed + 10 + 4 * i, (n,), 0.8, 1.2)
        out[name + ".bias"] = u(seed + 11 + 4 * i, (n,), -0.1, 0.1)
        out[name + ".running_mean"] = u(seed + 12 + 4 * i, (n,), -0.05, 0.05)
        out[name + ".running_var"] = u(seed + 13 + 4 * i, (n,), 0.5, 1.5)
    return out


def shallow_tf_weights(seed, nb, num_layers=12):
    """Deterministic state for ShallowConvNet + transformer (keys of Transformer_torch/Transformer_EEG.py's
    state_dict): matrices U(+-1/sqrt(fan_in)), LayerNorm/BatchNorm gains around 1, small non-zero biases."""
    u = synth.uniform
    out, s = {}, [seed * 1000]

    def nxt():
        s[0] += 1
        return s[0]

    def ku(shape, fan_in):
        b = 1.0 / np.sqrt(fan_in)
        return u(nxt(), shape, -b, b)

    out["conv.weight"] = ku((40, 1, 1, 13), 13)
    out["bn.weight"] = u(nxt(), (40,), 0.8, 1.2)
    out["bn.bias"] = u(nxt(), (40,), -0.1, 0.1)
    out["bn.running_mean"] = u(nxt(), (40,), -0.05, 0.05)
    out["bn.running_var"] = u(nxt(), (40,), 0.5, 1.5)
    for i in range(40):
        out[f"embedding.value_proj.{i}.weight"] = ku((1, 30), 30)
    for l in range(num_layers):
        p = f"transformer.{l}."
        for n in "qkv":
            out[p + f"attn.W_{n}.weight"] = ku((40, 40), 40)
        out[p + "ffn.net.0.weight"] = ku((160, 40), 40)
        out[p + "ffn.net.0.bias"] = ku((160,), 40)
        out[p + "ffn.net.3.weight"] = ku((40, 160), 160)
        out[p + "ffn.net.3.bias"] = ku((40,), 160)
        for n in ("norm1", "norm2"):
            out[p + n + ".weight"] = u(nxt(), (40,), 0.8, 1.2)
            out[p + n + ".bias"] = u(nxt(), (40,), -0.1, 0.1)
    out["fc.weight"] = ku((nb, 2600), 2600)
    return out
