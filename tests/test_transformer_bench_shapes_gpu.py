"""The 12-layer encoders at the shapes bench.py times - AST mel [8,1024,128] (M = 9 712 token rows) and ViT-B/16 frames
[128,3,224,224] (M = 25 216): one UNFROZEN training step against oracle/vit_oracle.Stepper (CPU autograd restatement of the
HF forward, pinned to the HF classes by tests/golden/*_full.npz) - logits, loss and every gradient tensor, in both
arithmetic paths.  These are the shapes where the persistent-tile remap, the XCD mapping, the split-K plans, the
token-contracting weight-gradient kernel with ragged token counts (9 712 = 303.5 x 32) and the 512-resident-workgroup
paths are all active at once; the B = 2 tests of test_transformer_model_gpu.py do not reach them.

Second part: the same models with the activation outliers real AST / ViT checkpoints have - LayerNorm gains of 10^3 on a
few channels and one "sink" token whose residual stream is 100x the others' - again against the oracle.
Reference call sites: Transformer_Audio.py:70-79, Transformer_Vision.py:89-100."""
import os

import numpy as np
import pytest
import torch

from eav_amd import synth
from tests.golden_util import tf_weights

pytestmark = pytest.mark.gpu


def _cfgs(kind):
    from eav_amd import transformer as T
    from oracle import vit_oracle as vo
    return T.make_config(kind), (vo.cfg_ast() if kind == "ast" else vo.cfg_vit())


def _oracle_step(ocfg, W, x, y):
    from oracle import vit_oracle as vo
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    st = vo.Stepper({k: torch.from_numpy(v.copy()) for k, v in W.items()}, ocfg, lr=1e-3)
    logits, loss, grads = st.step(torch.from_numpy(x), torch.from_numpy(y), False)
    return logits.numpy(), float(loss), {k: g.numpy() for k, g in grads.items()}


def _gpu_step(cfg, W, x, y, precision):
    from eav_amd import transformer as T
    from eav_amd.optim import CrossEntropyLoss
    model = T.Encoder(cfg, W).cuda().train()
    model.precision = precision
    out = model(torch.from_numpy(x).cuda())
    loss = CrossEntropyLoss()(out.logits, torch.from_numpy(y).cuda())
    loss.backward()
    torch.cuda.synchronize()
    res = (out.logits.detach().cpu().numpy(), float(loss.detach()), {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()})
    del model, out, loss
    torch.cuda.empty_cache()
    return res


def _compare(got, ref, grad_rtol, logit_tol, tag):
    lg, ls, gg = got
    lr_, lsr, gr = ref
    assert np.abs(lg - lr_).max() <= logit_tol, f"{tag}: logits differ by {np.abs(lg - lr_).max():.3e}"
    assert abs(ls - lsr) <= logit_tol, f"{tag}: loss {ls} vs {lsr}"
    assert sorted(gg) == sorted(gr)
    worst = ("", 0.0)
    for k in gr:
        err = np.abs(gg[k].astype(np.float64) - gr[k]).max()
        scale = max(np.abs(gr[k]).max(), 1e-30)
        if err / scale > worst[1] and not k.endswith("k_proj.bias"):
            worst = (k, err / scale)
        # (k_proj.bias has an analytically zero gradient - softmax is shift invariant: pure rounding noise either side)
        if k.endswith("k_proj.bias"):
            assert err <= 1e-3 * max(np.abs(gr[k.replace("k_proj", "q_proj")]).max(), 1e-30) + 1e-9, (tag, k, err)
        else:
            assert err <= grad_rtol * scale + 1e-9, f"{tag}: grad {k}: max err {err:.3e} vs max |ref| {scale:.3e}"
    return worst


@pytest.mark.parametrize("kind,B", [("ast", 8), ("vit", 128)])
def test_unfrozen_step_at_the_bench_batch_matches_the_oracle(kind, B):
    cfg, ocfg = _cfgs(kind)
    from oracle import vit_oracle as vo
    W = tf_weights(17, vo.param_shapes(ocfg), std=0.02)
    x, y = (synth.mel_batch(71, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(71, B, cfg.H))
    ref = _oracle_step(ocfg, W, x, y)
    for precision in ("split", "fp32"):
        worst = _compare(_gpu_step(cfg, W, x, y, precision), ref, 2e-3, 1e-4, f"{kind} B={B} {precision}")
        print(f"{kind} B={B} {precision}: worst gradient tensor {worst[0]} at {worst[1]:.2e} of its maximum")


def _outlier_weights(kind, ocfg):
    """Synthetic weights with the outlier structure of trained checkpoints: in three layers the LayerNorm gains of four
    channels are 10^3 (and the columns of the projections that read those channels 10^-2 of the rest, as in trained
    models, so that the network stays a network); token 0's position embedding is 100x - an attention-sink token whose
    residual stream dwarfs the others'."""
    from oracle import vit_oracle as vo
    W = tf_weights(23, vo.param_shapes(ocfg), std=0.02)
    p = ocfg["prefix"] if isinstance(ocfg, dict) else ocfg.prefix
    ch = [5, 200, 413, 700]
    for layer in (0, 6, 11):
        L = f"{p}.layers.{layer}"
        for ln, readers in (("layernorm_before", ("attention.q_proj", "attention.k_proj", "attention.v_proj")),
                            ("layernorm_after", ("mlp.fc1",))):
            W[f"{L}.{ln}.weight"] = W[f"{L}.{ln}.weight"].copy()
            W[f"{L}.{ln}.weight"][ch] = 1000.0
            for r in readers:
                W[f"{L}.{r}.weight"] = W[f"{L}.{r}.weight"].copy()
                W[f"{L}.{r}.weight"][:, ch] *= 1e-2
    pe = f"{p}.embeddings.position_embeddings"
    W[pe] = W[pe].copy()
    W[pe][0, 0] = 100.0 * np.sign(W[pe][0, 0] + 1e-12) * (np.abs(W[pe][0, 0]) + 0.02)
    return W


@pytest.mark.parametrize("kind", ["ast", "vit"])
def test_outlier_channels_and_a_sink_token_keep_parity(kind):
    """Logits within north_star's 1e-3 (held to 2e-4) and all gradients to the bounds of the plain full-size test, in both
    precisions.  The split path sees operand tensors whose largest entries are 10^3 x the typical ones (a-priori scales of
    the LayerNorm / GELU planes grow with max|gamma|; measured scales of the gradient tensors are set by the sink token)."""
    cfg, ocfg = _cfgs(kind)
    W = _outlier_weights(kind, ocfg)
    B = 2
    x, y = (synth.mel_batch(72, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(72, B, cfg.H))
    ref = _oracle_step(ocfg, W, x, y)
    assert np.isfinite(ref[0]).all() and all(np.isfinite(g).all() for g in ref[2].values())
    for precision in ("split", "fp32"):
        worst = _compare(_gpu_step(cfg, W, x, y, precision), ref, 2e-3, 2e-4, f"{kind} outliers {precision}")
        print(f"{kind} outliers {precision}: worst gradient tensor {worst[0]} at {worst[1]:.2e} of its maximum")


def _oracle_step_microbatched(ocfg, W, x, y, mb, stats=None):
    """The oracle's step on a batch too large to hold its autograd graph in host memory at once: the mean-CE gradient of
    the whole batch accumulated over sub-batches of `mb` (exact for these models: LayerNorm only, every dropout 0.0 - no op
    couples the samples of a batch)."""
    from oracle import vit_oracle as vo
    import torch.nn.functional as F
    torch.set_num_threads(min(os.cpu_count() or 1, 64))
    P = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in W.items()}
    B = x.shape[0]
    logits, loss = [], 0.0
    for a in range(0, B, mb):
        lg = vo.forward(P, torch.from_numpy(x[a:a + mb]), ocfg, stats=stats)
        ls = F.cross_entropy(lg, torch.from_numpy(y[a:a + mb]), reduction="sum") / B
        ls.backward()
        logits.append(lg.detach().numpy())
        loss += float(ls.detach())
    return np.concatenate(logits), loss, {k: p.grad.numpy() for k, p in P.items()}


def test_ast_unfrozen_step_at_the_throughput_batch_32_matches_the_oracle():
    """SURVEY.md:616 names B = 8 (reference) AND B = 32 (throughput) for the AST configuration; bench.py times both.
    M = 38 848 token rows: 304 tile-rows of the persistent GEMM grid, 1 214 K-tiles in the weight-gradient products."""
    cfg, ocfg = _cfgs("ast")
    from oracle import vit_oracle as vo
    W = tf_weights(17, vo.param_shapes(ocfg), std=0.02)
    x, y = synth.mel_batch(73, 32, cfg.W, cfg.H)
    ref = _oracle_step_microbatched(ocfg, W, x, y, 8)
    for precision in ("split", "fp32"):
        worst = _compare(_gpu_step(cfg, W, x, y, precision), ref, 2e-3, 1e-4, f"ast B=32 {precision}")
        print(f"ast B=32 {precision}: worst gradient tensor {worst[0]} at {worst[1]:.2e} of its maximum")


def _trained_like_weights(ocfg, qk_gain, head_gain):
    """N(0, 0.02) weights give near-uniform attention and |logits| << 1, where "within 1e-3" is a loose relative bound.
    Trained checkpoints do not look like that: scale the q / k projections (weights and biases) so that the softmax rows
    are peaked, and the classifier so that |logits| reaches 5-10 - the regime where 1e-3 absolute is 1e-4 relative."""
    from oracle import vit_oracle as vo
    W = tf_weights(29, vo.param_shapes(ocfg), std=0.02)
    for k in list(W):
        if ".attention.q_proj." in k or ".attention.k_proj." in k:
            W[k] = W[k] * np.float32(qk_gain)
        if k in ("classifier.weight", "classifier.dense.weight"):
            W[k] = W[k] * np.float32(head_gain)
    return W


@pytest.mark.parametrize("kind", ["ast", "vit"])
def test_trained_like_statistics_keep_parity(kind):
    """Peaked attention (a share of the softmax rows above 0.9 in every layer) and |logits| of 5-10 at full size, both
    precisions: logits within north_star's 1e-3, every gradient tensor to the bounds of the plain full-size tests
    (Transformer_Audio.py:72, Transformer_Vision.py:92 - HF attention with trained weights)."""
    cfg, ocfg = _cfgs(kind)
    W = _trained_like_weights(ocfg, 4.5, 9.0)
    B = 2
    x, y = (synth.mel_batch(74, B, cfg.W, cfg.H) if kind == "ast" else synth.frame_batch(74, B, cfg.H))
    stats = {}
    ref = _oracle_step_microbatched(ocfg, W, x, y, B, stats=stats)
    print(f"{kind}: share of softmax rows with a probability above 0.9 per layer: "
          f"{[round(v, 3) for v in stats['peaked_rows']]}; max |logit| {np.abs(ref[0]).max():.2f}")
    assert min(stats["peaked_rows"]) > 0.01 and np.abs(ref[0]).max() > 4.0, "the weights are not trained-like"
    assert np.isfinite(ref[0]).all() and all(np.isfinite(g).all() for g in ref[2].values())
    for precision in ("split", "fp32"):
        worst = _compare(_gpu_step(cfg, W, x, y, precision), ref, 2e-3, 1e-3, f"{kind} trained-like {precision}")
        print(f"{kind} trained-like {precision}: worst gradient tensor {worst[0]} at {worst[1]:.2e} of its maximum")
