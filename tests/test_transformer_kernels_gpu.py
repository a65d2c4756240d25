"""Kernel-level parity of the AST/ViT building blocks (through the C ABI) against fp64 torch on the
host.  The GEMM is an exact-fp32 MFMA fma chain, so tolerances are fp32 summation-order level."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from eav_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from eav_amd import _lib
    _lib.load()
    return _lib


_KEEP = []


def dev(a):
    """Host array -> device tensor kept alive for the module (temporaries would be recycled by the
    caching allocator before the asynchronous kernel ran)."""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    if len(_KEEP) > 64:
        torch.cuda.synchronize()
        del _KEEP[:32]
    return t


def close(got, ref, rtol, atol, what=""):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    err = np.abs(got - ref)
    assert (err <= atol + rtol * np.abs(ref)).all(), f"{what}: max err {err.max():.3e}, ref max {np.abs(ref).max():.3e}"


def gemm(L, A, B, C, M, N, K, lda, ldb, ldc, tA=0, tB=0, batch=1, heads=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha=1.0,
         bias=None, gelu=0, pre=None, resid=None, ldr=0, acc=0):
    p = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())  # noqa: E731
    L.call("eav_gemm_f32", p(A), p(B), p(C), M, N, K, lda, ldb, ldc, tA, tB, batch, heads, sA[0], sA[1], sB[0], sB[1],
           sC[0], sC[1], alpha, p(bias), gelu, p(pre), p(resid), ldr, acc, None)


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (300, 200, 100), (77, 64, 48), (1214, 1214, 64), (257, 40, 1214),
                                   (9, 5, 12), (130, 3072, 768)])
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_all_layouts(L, M, N, K, tA, tB):
    pad = lambda v: (v + 3) // 4 * 4  # noqa: E731
    a = synth.normal(1, (K, M) if tA else (M, K))
    b = synth.normal(2, (K, N) if tB else (N, K))
    lda, ldb = pad(a.shape[1]), pad(b.shape[1])
    ap, bp = np.zeros((a.shape[0], lda), np.float32), np.zeros((b.shape[0], ldb), np.float32)
    ap[:, :a.shape[1]], bp[:, :b.shape[1]] = a, b
    A, B = dev(ap), dev(bp)
    C = torch.full((M, N + 3), 7.0, device="cuda")
    gemm(L, A, B, C, M, N, K, lda, ldb, N + 3, tA, tB, alpha=0.5)
    ad = torch.from_numpy(a).double()
    bd = torch.from_numpy(b).double()
    ref = 0.5 * (ad.t() if tA else ad) @ (bd if tB else bd.t())
    close(C[:, :N], ref, 1e-5, 2e-5 * float(ref.abs().max()), "C")
    assert torch.all(C[:, N:] == 7.0)                    # never writes outside [M,N]


@pytest.mark.parametrize("M,N,K", [(768, 768, 9712), (3072, 768, 2500), (64, 256, 2424), (200, 40, 700), (8, 8, 64)])
def test_gemm_splitk_weight_gradient(L, M, N, K):
    a, b = synth.normal(31, (K, M)), synth.normal(32, (K, N))
    A, B = dev(a), dev(b)
    ns = L.plain("eav_gemm_f32_splitk_plan", M, N, K)
    ws = torch.empty(max(ns, 1) * M * N, device="cuda")
    C = torch.empty(M, N, device="cuda")
    L.call("eav_gemm_f32_splitk", A.data_ptr(), B.data_ptr(), C.data_ptr(), ws.data_ptr(), M, N, K, M, N, 1, 1, None)
    ref = torch.from_numpy(a).double().t() @ torch.from_numpy(b).double()
    close(C, ref, 1e-5, 2e-5 * float(ref.abs().max()), f"split-K x{ns}")
    C2 = torch.empty_like(C)
    L.call("eav_gemm_f32_splitk", A.data_ptr(), B.data_ptr(), C2.data_ptr(), ws.data_ptr(), M, N, K, M, N, 1, 1, None)
    assert torch.equal(C, C2)                               # fixed-order reduction: bit-reproducible


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).double()


@pytest.mark.parametrize("M,N,K", [(300, 200, 100), (1214, 64, 1214), (130, 3072, 768), (9, 5, 12)])
@pytest.mark.parametrize("tA,tB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.skipif(not __import__("eav_amd._lib", fromlist=["x"]).have_extras(),
                    reason="comparison-only kernel: build with `make -C eav_amd/csrc BENCH_EXTRAS=1`")
def test_gemm_bf16_operands(L, M, N, K, tA, tB):
    """bf16-MFMA variant == the fp64 product of the bf16-rounded (RNE) operands, up to fp32 accumulation."""
    pad = lambda v: (v + 3) // 4 * 4  # noqa: E731
    a = synth.normal(41, (K, M) if tA else (M, K))
    b = synth.normal(42, (K, N) if tB else (N, K))
    lda, ldb = pad(a.shape[1]), pad(b.shape[1])
    ap, bp = np.zeros((a.shape[0], lda), np.float32), np.zeros((b.shape[0], ldb), np.float32)
    ap[:, :a.shape[1]], bp[:, :b.shape[1]] = a, b
    A, B = dev(ap), dev(bp)
    C = torch.full((M, N + 3), 7.0, device="cuda")
    p = lambda t: t.data_ptr()  # noqa: E731
    L.call("eav_gemm_bf16", p(A), p(B), p(C), M, N, K, lda, ldb, N + 3, tA, tB, 1, 1, 0, 0, 0, 0, 0, 0, 0.5, None, 0,
           None, None, 0, 0, None)
    ad, bd = _bf16_round(a), _bf16_round(b)
    ref = 0.5 * (ad.t() if tA else ad) @ (bd if tB else bd.t())
    close(C[:, :N], ref, 1e-5, 2e-5 * float(ref.abs().max()), "C (bf16 operands)")
    assert torch.all(C[:, N:] == 7.0)
    if tA and tB and (N & 3) == 0:
        ns = L.plain("eav_gemm_f32_splitk_plan", M, N, K)
        ws, C2 = torch.empty(max(ns, 1) * M * N, device="cuda"), torch.empty(M, N, device="cuda")
        L.call("eav_gemm_bf16_splitk", p(A), p(B), p(C2), p(ws), M, N, K, lda, ldb, 1, 1, None)
        close(C2, 2.0 * ref, 1e-5, 4e-5 * float(ref.abs().max()), "split-K (bf16 operands)")


def test_gemm_epilogues(L):
    M, N, K = 200, 136, 72
    a, b = synth.normal(3, (M, K)), synth.normal(4, (N, K), 0, 0.2)
    bias, resid = synth.normal(5, (N,)), synth.normal(6, (M, N))
    A, B, bi, rs = dev(a), dev(b), dev(bias), dev(resid)
    C, pre = torch.zeros(M, N, device="cuda"), torch.zeros(M, N, device="cuda")
    gemm(L, A, B, C, M, N, K, K, K, N, bias=bi, gelu=1, pre=pre, resid=rs, ldr=N)
    z = torch.from_numpy(a).double() @ torch.from_numpy(b).double().t() + torch.from_numpy(bias).double()
    close(pre, z, 1e-5, 1e-5, "pre")
    ref = z * 0.5 * (1 + torch.erf(z / np.sqrt(2))) + torch.from_numpy(resid).double()
    close(C, ref, 1e-5, 2e-5, "gelu+resid")
    C2 = C.clone()
    gemm(L, A, B, C2, M, N, K, K, K, N, acc=1)
    close(C2, ref + (z - torch.from_numpy(bias).double()), 1e-5, 4e-5, "accumulate")


def test_gemm_batched_head_strides(L):
    """The attention products exactly as the encoder issues them: Q.K^T, P.V, dV, dK over (image, head)."""
    Bn, H, N, hd = 2, 3, 197, 16
    D = H * hd
    ldn = 200
    qkv = synth.normal(7, (Bn * N, 3 * D))
    Q = dev(qkv)
    S = torch.zeros(Bn * H, N, ldn, device="cuda")
    sQ, sP, sO = (N * 3 * D, hd), (H * N * ldn, N * ldn), (N * D, hd)
    gemm(L, Q.data_ptr(), Q.data_ptr() + 4 * D, S, N, N, hd, 3 * D, 3 * D, ldn, batch=Bn * H, heads=H, sA=sQ, sB=sQ,
         sC=sP, alpha=0.25)
    t = torch.from_numpy(qkv).double().view(Bn, N, 3, H, hd)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))       # [B,H,N,hd]
    sref = 0.25 * q @ k.transpose(2, 3)
    close(S.view(Bn, H, N, ldn)[..., :N], sref, 1e-5, 1e-5, "scores")
    assert torch.all(S.view(Bn, H, N, ldn)[..., N:] == 0)
    L.call("eav_softmax_fwd", S.data_ptr(), Bn * H * N, N, ldn, None)
    pref = torch.softmax(sref, -1)
    close(S.view(Bn, H, N, ldn)[..., :N], pref, 1e-5, 1e-7, "softmax")
    O = torch.zeros(Bn * N, D, device="cuda")
    gemm(L, S, Q.data_ptr() + 8 * D, O, N, hd, N, ldn, 3 * D, D, tB=1, batch=Bn * H, heads=H, sA=sP, sB=sQ, sC=sO)
    oref = (pref @ v).permute(0, 2, 1, 3).reshape(Bn * N, D)
    close(O, oref, 1e-5, 1e-6, "P.V")
    do = synth.normal(8, (Bn * N, D))
    dO = dev(do)
    dqkv = torch.zeros(Bn * N, 3 * D, device="cuda")
    gemm(L, S, dO, dqkv.data_ptr() + 8 * D, N, hd, N, ldn, D, 3 * D, tA=1, tB=1, batch=Bn * H, heads=H, sA=sP, sB=sO, sC=sQ)
    dot = torch.from_numpy(do).double().view(Bn, N, H, hd).permute(0, 2, 1, 3)
    dv = (pref.transpose(2, 3) @ dot).permute(0, 2, 1, 3).reshape(Bn * N, D)
    close(dqkv[:, 2 * D:], dv, 1e-5, 1e-6, "dV")
    dP = torch.zeros(Bn * H, N, ldn, device="cuda")
    gemm(L, dO, Q.data_ptr() + 8 * D, dP, N, N, hd, D, 3 * D, ldn, batch=Bn * H, heads=H, sA=sO, sB=sQ, sC=sP)
    dpref = dot @ v.transpose(2, 3)
    close(dP.view(Bn, H, N, ldn)[..., :N], dpref, 1e-5, 1e-5, "dP")
    L.call("eav_softmax_bwd", S.data_ptr(), dP.data_ptr(), Bn * H * N, N, ldn, None)
    dsref = pref * (dpref - (dpref * pref).sum(-1, keepdim=True))
    close(dP.view(Bn, H, N, ldn)[..., :N], dsref, 1e-4, 1e-7, "dS")
    gemm(L, dP, Q, dqkv.data_ptr() + 4 * D, N, hd, N, ldn, 3 * D, 3 * D, tA=1, tB=1, batch=Bn * H, heads=H, sA=sP,
         sB=sQ, sC=sQ, alpha=0.25)
    dk = 0.25 * (dsref.transpose(2, 3) @ q).permute(0, 2, 1, 3).reshape(Bn * N, D)
    close(dqkv[:, D:2 * D], dk, 1e-4, 1e-7, "dK")


@pytest.mark.parametrize("M,D", [(9, 64), (1000, 768), (33, 1024)])
def test_layernorm_fwd_bwd(L, M, D):
    x = synth.normal(11, (M, D), 0.3, 2.0)
    x[1] = 0.467                                           # constant row (AST padding value): var = 0, eps = 1e-12
    g, b = synth.uniform(12, (D,), 0.5, 1.5), synth.normal(13, (D,), 0, 0.1)
    X, G, Bt = dev(x), dev(g), dev(b)
    y, st = torch.empty(M, D, device="cuda"), torch.empty(2, M, device="cuda")
    L.call("eav_layernorm_fwd", X.data_ptr(), G.data_ptr(), Bt.data_ptr(), y.data_ptr(), st.data_ptr(),
           st.data_ptr() + 4 * M, M, D, 1e-12, None)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    gt = torch.from_numpy(g).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    ref = F.layer_norm(xt, (D,), gt, bt, 1e-12)
    keep = np.ones(M, bool)
    keep[1] = False                                        # the degenerate row is checked for finiteness only
    close(y[torch.from_numpy(keep).cuda()], ref[torch.from_numpy(keep)], 1e-4, 2e-5, "ln fwd")
    assert torch.isfinite(y).all()
    dy = synth.normal(14, (M, D))
    dy[1] = 0
    ref.backward(torch.from_numpy(dy).double())
    dx = torch.full((M, D), 1.0, device="cuda")
    npart = L.plain("eav_layernorm_bwd_nparts", M)
    part = torch.zeros(npart, 2 * D, device="cuda")
    L.call("eav_layernorm_bwd", dev(dy).data_ptr(), X.data_ptr(), G.data_ptr(), st.data_ptr(), st.data_ptr() + 4 * M,
           dx.data_ptr(), 1, part.data_ptr(), M, D, None)
    torch.cuda.synchronize()
    close((dx - 1.0)[torch.from_numpy(keep).cuda()], xt.grad[torch.from_numpy(keep)], 1e-3, 1e-4, "ln dx (accumulated)")
    ps = part.sum(0).cpu().double()
    # the constant row contributes xhat = 0 * 1e6-ish garbage-free terms only through dy = 0
    close(ps[:D], gt.grad, 1e-3, 1e-3, "dgamma")
    close(ps[D:], bt.grad, 1e-4, 1e-4, "dbeta")


def test_gelu_bwd_colsum_im2col_embed(L):
    n = 4096
    pre, d = synth.normal(21, (n,), 0, 2.0), synth.normal(22, (n,))
    D_ = dev(d)
    L.call("eav_gelu_bwd", D_.data_ptr(), dev(pre).data_ptr(), n, None)
    pt = torch.from_numpy(pre).double().requires_grad_(True)
    (pt * 0.5 * (1 + torch.erf(pt / np.sqrt(2)))).backward(torch.from_numpy(d).double())
    close(D_, pt.grad, 1e-5, 1e-6, "gelu bwd")
    M, N = 1000, 200
    dy = synth.normal(23, (M, N))
    npart = L.plain("eav_colsum_nparts", M)
    part = torch.zeros(npart, N, device="cuda")
    L.call("eav_colsum", dev(dy).data_ptr(), part.data_ptr(), M, N, N, None)
    close(part.sum(0), dy.astype(np.float64).sum(0), 1e-5, 1e-4, "colsum")
    # im2col: AST view (transposed input, stride 10) and ViT view
    x = synth.normal(24, (2, 40, 36))                     # [B, frames=40, mel=36]
    col = torch.zeros(2 * 3 * 3, 256, device="cuda")      # ny = (36-16)/10+1 = 3, nx = (40-16)/10+1 = 3
    L.call("eav_im2col", dev(x).data_ptr(), col.data_ptr(), 2, 1, 36, 40, 16, 10, 10, 1, None)
    img = torch.from_numpy(x).unsqueeze(1).transpose(2, 3)
    ref = F.unfold(img, 16, stride=10).transpose(1, 2).reshape(-1, 256)
    assert torch.equal(col.cpu(), ref)
    xv = synth.normal(25, (2, 3, 32, 32))
    colv = torch.zeros(2 * 4, 768, device="cuda")
    L.call("eav_im2col", dev(xv).data_ptr(), colv.data_ptr(), 2, 3, 32, 32, 16, 16, 16, 0, None)
    refv = F.unfold(torch.from_numpy(xv), 16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(colv.cpu(), refv)
    # embedding finish / backward / token rows / pair mean
    B, ntok, D, nx = 3, 7, 8, 2
    h = synth.normal(26, (B, ntok, D))
    cls, dist, pos = synth.normal(27, (D,)), synth.normal(28, (D,)), synth.normal(29, (ntok, D))
    H_ = dev(h)
    L.call("eav_embed_finish", H_.data_ptr(), dev(cls).data_ptr(), dev(dist).data_ptr(), dev(pos).data_ptr(), B, ntok,
           D, nx, None)
    ref = h.copy()
    ref[:, 0], ref[:, 1] = cls, dist
    ref += pos
    assert np.allclose(H_.cpu().numpy(), ref, atol=1e-7)
    dpos, demb = torch.zeros(ntok, D, device="cuda"), torch.zeros(B * (ntok - nx), D, device="cuda")
    L.call("eav_embed_bwd", H_.data_ptr(), dpos.data_ptr(), demb.data_ptr(), B, ntok, D, nx, None)
    assert np.allclose(dpos.cpu().numpy(), ref.sum(0), atol=1e-6)
    assert np.allclose(demb.cpu().numpy(), ref[:, nx:].reshape(-1, D))
    rows = torch.zeros(B * nx, D, device="cuda")
    L.call("eav_token_rows", H_.data_ptr(), rows.data_ptr(), B, ntok, D, nx, 0, None)
    assert np.allclose(rows.cpu().numpy(), ref[:, :nx].reshape(-1, D))
    pooled = torch.zeros(B, D, device="cuda")
    L.call("eav_pair_mean", rows.data_ptr(), pooled.data_ptr(), B, D, 0, None)
    assert np.allclose(pooled.cpu().numpy(), ref[:, :2].mean(1), atol=1e-7)


def test_frame_preprocessing_bit_exact(golden_dir):
    """GPU frames -> pixel_values == the oracle (Pillow-exact) == what the reference trainer's HF processor made."""
    import os
    from eav_amd.preprocess import frames_to_pixel_values, pillow_bilinear_tables
    from oracle import preprocess_oracle as po
    for a, b in ((56, 224), (48, 100), (224, 56), (7, 31)):
        rb, rk = po.precompute_coeffs(a, b)
        gb, gk = pillow_bilinear_tables(a, b)
        assert np.array_equal(rb, gb) and np.array_equal(rk, gk), (a, b)
    frames = (synth.uniform(92, (10, 2, 56, 56, 3)) * 255).astype(np.uint8).reshape(-1, 56, 56, 3)
    got = frames_to_pixel_values(frames).cpu().numpy()
    assert np.array_equal(got, po.vit_preprocess(frames))
    g = np.load(os.path.join(golden_dir, "vit_trainer.npz"))
    assert np.array_equal(got[:12], g["tr_x"]) and np.array_equal(got[12:], g["te_x"])
    odd = (synth.uniform(93, (3, 40, 64, 3)) * 256).astype(np.uint8)          # non-square, down- and up-scaling
    got = frames_to_pixel_values(odd, size=(96, 48), mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)).cpu().numpy()
    ref = po.vit_preprocess
    exp = np.stack([((po.resize_bilinear_u8(f, 96, 48).transpose(2, 0, 1).astype(np.float64) / 255).astype(np.float32)
                     - np.float32([0.485, 0.456, 0.406])[:, None, None]) / np.float32([0.229, 0.224, 0.225])[:, None, None]
                    for f in odd])
    assert np.allclose(got, exp, atol=1e-6)


def test_ast_logmel_frontend(golden_dir):
    """GPU log-mel == oracle (numpy restatement of the HF recipe) == the reference trainer's features."""
    import os
    from eav_amd.preprocess import waveforms_to_input_values
    from oracle import preprocess_oracle as po
    g = np.load(os.path.join(golden_dir, "ast_trainer.npz"))
    wav = synth.normal(90, (10, 80000), 0.0, 0.1)
    got = waveforms_to_input_values(wav).cpu().numpy()
    assert got.shape == (10, 1024, 128) and got.dtype == np.float32
    assert np.abs(got[:6] - g["tr_x"]).max() < 5e-6 and np.abs(got[6:] - g["te_x"]).max() < 5e-6
    short = synth.normal(94, (2, 3000), 0.0, 0.3)                # 17 frames, 1007 padded
    long_ = synth.normal(95, (1, 170000), 0.0, 0.3)              # 1060 frames -> truncated to 1024
    for w in (short, long_):
        assert np.abs(waveforms_to_input_values(w).cpu().numpy() - po.ast_fbank(w)).max() < 5e-6
    silent = np.zeros((1, 16000), np.float32)                    # log of the floor, no NaN/inf
    assert np.isfinite(waveforms_to_input_values(silent).cpu().numpy()).all()


def test_eeg_decimate_and_sosfilt():
    """GPU resample_poly / sosfilt (float64) vs scipy on ragged sizes; the chunk-parallel IIR is exact for any chunk
    length (incl. chunks longer than the record and a near-unit-circle 0.3 Hz band)."""
    from scipy import signal
    from eav_amd import eeg_data as ed
    x = synth.normal(71, (5, 20011)).astype(np.float64) + 2.0
    xd = torch.from_numpy(x).cuda()
    for down in (5, 4):
        got = ed.decimate(xd, down).cpu().numpy()
        assert np.abs(got - signal.resample_poly(x, 1, down, axis=1)).max() < 1e-11
    for band in ([5, 30], [0.3, 45]):
        sos = signal.butter(5, band, btype='bandpass', fs=100, output='sos')
        ref = signal.sosfilt(sos, x, axis=1)
        for chunk in (257, 2048, 50000):
            got = ed.sosfilt(sos, xd, chunk=chunk).cpu().numpy()
            assert np.abs(got - ref).max() < 1e-9 * max(1.0, np.abs(ref).max()), (band, chunk)


def test_eeg_dataload_pipeline_matches_reference(golden_dir):
    """eav_amd.DataLoadEEG.downsampling/bandpass_filter/segment_and_select_classes == the reference class on the
    same synthetic recording [30,10000,200] (sample + checksum fixture)."""
    import os
    from eav_amd.eeg_data import DataLoadEEG
    from tests.golden.make_goldens_eeg import synthetic_recording
    g = np.load(os.path.join(golden_dir, "eeg_preprocess.npz"))
    x, lab = synthetic_recording(int(g["seed"]))
    d = DataLoadEEG(subject=1, band=[5, 30], fs_orig=500, fs_target=100)
    d.seg, d.label = x, lab
    d.downsampling()
    assert np.abs(d.seg.cpu().numpy()[::3, ::41, ::17] - g["down_sample"]).max() < 1e-10
    d.bandpass_filter()
    assert np.abs(d.seg_f.cpu().numpy()[::3, ::41, ::17] - g["filt_sample"]).max() < 1e-9
    d.segment_and_select_classes()
    assert d.seg_f_div.shape == tuple(g["shape"]) and d.seg_f_div.dtype == np.float64
    assert np.array_equal(d.label_div, g["labels"])
    assert np.abs(d.seg_f_div[::7, ::3, ::11] - g["out_sample"]).max() < 1e-9
    assert abs(np.abs(d.seg_f_div).sum() - float(g["out_abssum"])) < 1e-9 * float(g["out_abssum"])
    d2 = DataLoadEEG(band=[5, 30], remap_labels=True)
    d2.seg_f, d2.label = d.seg_f, lab
    d2.segment_and_select_classes()
    assert set(np.unique(d2.label_div)) <= {0, 1, 2, 3, 4} and np.array_equal(d2.label_div, (g["labels"] - 1) // 2)


def _attn_ref(qkv, B, H, N, hd):
    t = torch.from_numpy(qkv).double().view(B, N, 3, H, hd)
    q, k, v = (t[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(2, 3)) * hd ** -0.5
    return q, k, v, s


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 1214), (2, 2, 70), (1, 1, 33), (1, 1, 128)])
def test_fused_attention_forward(L, B, H, N):
    hd, D = 64, H * 64
    qkv = synth.normal(101, (B * N, 3 * D), 0.0, 1.5)
    Q = dev(qkv)
    ao = torch.full((B * N, D), 9.0, device="cuda")
    lse = torch.zeros(B * H, N, device="cuda")
    L.call("eav_attn_fwd", Q.data_ptr(), ao.data_ptr(), lse.data_ptr(), B, H, N, hd, hd ** -0.5, None)
    q, k, v, s = _attn_ref(qkv, B, H, N, hd)
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    close(ao, ref, 1e-4, 2e-6, "attention output")
    close(lse.view(B, H, N), torch.logsumexp(s, -1), 1e-5, 1e-5, "lse")


@pytest.mark.parametrize("B,H,N", [(2, 3, 197), (1, 2, 1214), (2, 2, 70), (1, 1, 33)])
def test_fused_attention_backward(L, B, H, N):
    hd, D = 64, H * 64
    qkv = synth.normal(111, (B * N, 3 * D), 0.0, 1.2)
    do = synth.normal(112, (B * N, D))
    Q, dO = dev(qkv), dev(do)
    ao, lse = torch.empty(B * N, D, device="cuda"), torch.empty(B * H, N, device="cuda")
    L.call("eav_attn_fwd", Q.data_ptr(), ao.data_ptr(), lse.data_ptr(), B, H, N, hd, hd ** -0.5, None)
    delta = torch.empty(B * H, N, device="cuda")
    dqkv = torch.full((B * N, 3 * D), 5.0, device="cuda")
    L.call("eav_attn_bwd", Q.data_ptr(), ao.data_ptr(), dO.data_ptr(), lse.data_ptr(), delta.data_ptr(), dqkv.data_ptr(),
           B, H, N, hd, hd ** -0.5, None)
    t = torch.from_numpy(qkv).double().requires_grad_(True)
    tv = t.view(B, N, 3, H, hd)
    q, k, v = (tv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    o = (torch.softmax((q @ k.transpose(2, 3)) * hd ** -0.5, -1) @ v).permute(0, 2, 1, 3).reshape(B * N, D)
    o.backward(torch.from_numpy(do).double())
    close(dqkv, t.grad, 2e-4, 2e-5 * float(t.grad.abs().max()), "dqkv")
