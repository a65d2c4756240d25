"""Split-operand kernels (csrc/gemm_sp.hip, csrc/attention_sp.hip) through the C ABI: every result is compared with a
float64 reference BESIDE the exact-fp32 MFMA kernel of the same product - the split path must not be worse than the
fp32 kernel (same style as test_fir_fwd_split_is_fp32_grade) - plus the operand format itself, the epilogues and the
ragged / padded edges."""
import numpy as np
import pytest
import torch

from eav_amd import _lib

pytestmark = pytest.mark.gpu
P = _lib.ptr
SLOT = 4128


def kpad(k):
    return (k + 31) // 32 * 32


def planes(x, want=True, wantT=False):
    R, C = x.shape
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, x.stride(0), P(slot), None)
    d = torch.zeros(R, 2 * kpad(C), dtype=torch.float16, device="cuda") if want else None
    dT = torch.zeros(C, 2 * kpad(R), dtype=torch.float16, device="cuda") if wantT else None
    _lib.call("eav_sp_convert", P(x), R, C, x.stride(0), P(slot), P(d), P(dT), None)
    return slot, d, dT


def gemm_sp(A, B, **kw):
    M, K = A.shape
    N = B.shape[0]
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = kw.pop("C", None)
    if C is None:
        C = torch.empty(M, N, device="cuda")
    _lib.call("eav_gemm_sp", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, kw.get("alpha", 1.0),
              P(kw.get("bias")), kw.get("gelu", 0), P(kw.get("pre")), P(kw.get("resid")), N if "resid" in kw else 0,
              kw.get("acc", 0), P(kw.get("amax")), None)
    return C


def gemm_f32(A, B):
    M, K = A.shape
    N = B.shape[0]
    C = torch.empty(M, N, device="cuda")
    _lib.call("eav_gemm_f32", P(A), P(B), P(C), M, N, K, K, K, N, 0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, None, 0, None, None,
              0, 0, None)
    return C


def test_planes_encode_the_tensor():
    """hi + lo 2^-11 reproduces sigma*x to fp32 accuracy, sigma is the power of two that puts max|x| in [2^14, 2^15),
    the K padding is zero, and the transposed planes hold the same numbers."""
    torch.manual_seed(0)
    x = torch.randn(77, 100, device="cuda") * 3.7
    slot, d, dT = planes(x, True, True)
    sigma = float(slot[2048])
    assert sigma == 2.0 ** (14 - np.floor(np.log2(float(x.abs().max()))))
    assert float(slot[2049]) == 1.0 / sigma
    v = d.view(77, kpad(100) // 8, 2, 8).double()
    rec = (v[:, :, 0, :] + v[:, :, 1, :] / 2048.0).reshape(77, kpad(100))
    assert (rec[:, 100:] == 0).all()
    err = (rec[:, :100] - x.double() * sigma).abs().max().item() / (float(x.abs().max()) * sigma)
    assert err < 2.0 ** -22, err
    vT = dT.view(100, kpad(77) // 8, 2, 8).double()
    recT = (vT[:, :, 0, :] + vT[:, :, 1, :] / 2048.0).reshape(100, kpad(77))
    assert (recT[:, 77:] == 0).all()
    assert torch.equal(recT[:, :77], rec[:, :100].t())


@pytest.fixture
def tile_hook():
    """eav_gemm_sp_set_tile for one test (process-global tuning hook, include/eav_hip_tuning.h): restored afterwards."""
    def set_tile(v):
        _lib.call("eav_gemm_sp_set_tile", int(v))
    yield set_tile
    _lib.call("eav_gemm_sp_set_tile", 0)


@pytest.mark.parametrize("shape", [(512, 384, 768), (1000, 200, 100), (9712, 768, 3072), (300, 130, 40), (64, 5, 36),
                                   (9712, 2304, 768)])     # the last one: 1368 tiles on 512 persistent workgroups
@pytest.mark.parametrize("amp", [(1.0, 0.02), (2e-5, 3e4)])
@pytest.mark.parametrize("tile", [0, 64, 128])      # heuristic / the 64 x 128 form wherever it applies / never
def test_gemm_sp_is_fp32_grade(shape, amp, tile, tile_hook):
    M, N, K = shape
    tile_hook(tile)
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda") * amp[0]
    B = torch.randn(N, K, device="cuda") * amp[1]
    ref = A.double() @ B.double().t()
    den = A.double().abs() @ B.double().abs().t()
    e_sp = ((gemm_sp(A, B).double() - ref).abs() / den).max().item()
    e_32 = ((gemm_f32(A, B).double() - ref).abs() / den).max().item()
    assert e_sp <= 1.05 * e_32 + 1e-9, (e_sp, e_32)
    assert e_sp < 5e-7


def test_gemm_sp_rows_far_below_the_tensor_maximum():
    """Per-tensor scale: a row 2^-20 below the maximum still comes out to ~1e-6 of its own magnitude (lo is lifted by 2^11,
    so both pieces stay normal fp16 numbers down to 2^-29 of the maximum)."""
    torch.manual_seed(3)
    A = torch.randn(256, 512, device="cuda")
    A[7] *= 2.0 ** -20
    B = torch.randn(128, 512, device="cuda")
    ref = A.double() @ B.double().t()
    den = A.double().abs() @ B.double().abs().t()
    err = ((gemm_sp(A, B).double() - ref).abs() / den)
    assert err[7].max().item() < 2e-6 and err.max().item() < 2e-6


def test_gemm_sp_epilogues_and_batch():
    torch.manual_seed(5)
    M, N, K = 300, 200, 96
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.1
    bias = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    pre = torch.empty(M, N, device="cuda")
    amax = torch.zeros(SLOT, device="cuda")
    lin = 0.5 * (A.double() @ B.double().t()) + bias.double()
    want = torch.nn.functional.gelu(lin) + resid.double()
    got = gemm_sp(A, B, alpha=0.5, bias=bias, gelu=1, pre=pre, resid=resid, amax=amax)
    assert (got.double() - want).abs().max().item() < 2e-5
    assert (pre.double() - lin).abs().max().item() < 2e-5
    assert float(amax[:2048:32].view(torch.int32).max().view(torch.float32)) == float(got.abs().max())
    acc = gemm_sp(A, B, C=got.clone(), acc=1)
    assert (acc.double() - (want + A.double() @ B.double().t())).abs().max().item() < 3e-5
    # gelu = 2: the product times gelu'(pre) with `pre` read (fc2 data gradient + GELU backward in one pass); ragged and
    # full tiles, against autograd in float64
    for (m2, n2) in ((300, 200), (256, 256)):
        A2 = torch.randn(m2, K, device="cuda")
        B2 = torch.randn(n2, K, device="cuda") * 0.1
        x = (torch.randn(m2, n2, device="cuda") * 2.0)
        xd = x.double().requires_grad_(True)
        torch.nn.functional.gelu(xd).backward(A2.double() @ B2.double().t())
        amax2 = torch.zeros(SLOT, device="cuda")
        got2 = gemm_sp(A2, B2, gelu=2, pre=x.clone(), amax=amax2)
        assert (got2.double() - xd.grad).abs().max().item() < 2e-5 * max(1.0, float(xd.grad.abs().max()))
        assert float(amax2[:2048:32].view(torch.int32).max().view(torch.float32)) == float(got2.abs().max())
    # batched over A and C (the patch-embedding call: per-image row blocks, shared weight)
    nb, m = 3, 100
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.zeros(nb, m + 2, N, device="cuda")
    _lib.call("eav_gemm_sp", P(pa), P(pb), P(C) + 4 * 2 * N, P(sa), P(sb), m, N, K, N, nb, m * 4 * kpad(K), (m + 2) * N,
              1.0, None, 0, None, None, 0, 0, None, None)
    ref = (A.double() @ B.double().t()).view(nb, m, N)
    assert (C[:, 2:].double() - ref).abs().max().item() < 2e-5 and (C[:, :2] == 0).all()


def row_planes(x):
    """Row planes with the rows padded to a multiple of 32 with zeros (what the token-contracting GEMM requires)."""
    R, C = x.shape
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, x.stride(0), P(slot), None)
    d = torch.zeros((R + 31) // 32 * 32, 2 * kpad(C), dtype=torch.float16, device="cuda")
    _lib.call("eav_sp_convert", P(x), R, C, x.stride(0), P(slot), P(d), None, None)
    return slot, d


@pytest.mark.parametrize("shape", [(9712, 768, 256), (1214, 200, 136), (50, 64, 64), (4096, 3072, 768), (197, 40, 2304),
                                   (33, 128, 128)])
def test_weight_gradient_contracts_over_the_rows_of_row_planes(shape):
    """dW[N,K] = dY^T X straight from the ROW planes of dY [tokens,N] and X [tokens,K]: the kernel contracts over the rows
    (tokens) with transposing LDS reads (ds_read_b64_tr_b16) - no transposed planes exist.  Asymmetric operands (a
    transposed or mis-swizzled fragment cannot pass), ragged token counts (zero pad rows), ragged and sub-tile feature
    counts, split-K accumulate mode, bit-reproducibility."""
    tokens, N, K = shape
    torch.manual_seed(7)
    dY = torch.randn(tokens, N, device="cuda") * 1e-3 * (1 + torch.arange(N, device="cuda") % 7)
    X = torch.randn(tokens, K, device="cuda") * (1 + 0.25 * (torch.arange(tokens, device="cuda") % 5))[:, None]
    sa, pa = row_planes(dY)
    sb, pb = row_planes(X)
    C = torch.empty(N, K, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", N, K, tokens)
    ws = torch.empty(max(ns, 1) * N * K, device="cuda")
    _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), N, K, tokens, 0, None)
    ref = dY.double().t() @ X.double()
    den = dY.double().abs().t() @ X.double().abs()
    err = ((C.double() - ref).abs() / den).max().item()
    assert err < 2e-7, err
    C2 = C.clone()
    _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C2), P(ws), P(sa), P(sb), N, K, tokens, 1, None)
    assert torch.allclose(C2, 2 * C, rtol=1e-6, atol=0)
    C3 = torch.empty_like(C)   # bit-reproducible (fixed-order split-K reduction)
    _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C3), P(ws), P(sa), P(sb), N, K, tokens, 0, None)
    assert torch.equal(C3, C)


def test_convert_colsum_and_producer_amax():
    torch.manual_seed(9)
    R, C = 1000, 192
    x = torch.randn(R, C, device="cuda")
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, C, P(slot), None)
    npart = _lib.plain("eav_sp_convert_colsum_nparts", R)
    part = torch.zeros(npart, C, device="cuda")
    d = torch.empty(R, 2 * kpad(C), dtype=torch.float16, device="cuda")
    _lib.call("eav_sp_convert_colsum", P(x), R, C, C, P(slot), P(d), None, P(part), None)
    assert (part.double().sum(0) - x.double().sum(0)).abs().max().item() < 1e-4
    _, d2, _ = planes(x)
    assert torch.equal(d, d2)
    # LayerNorm forward / backward and GELU backward leave max|output| in the slot
    g, b = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    y, mean, rstd = torch.empty_like(x), torch.empty(R, device="cuda"), torch.empty(R, device="cuda")
    s1 = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_layernorm_fwd_amax", P(x), P(g), P(b), P(y), P(mean), P(rstd), R, C, 1e-12, P(s1), None)
    assert float(s1[:2048:32].view(torch.int32).max().view(torch.float32)) == float(y.abs().max())
    dy, dx = torch.randn_like(x), torch.zeros_like(x)
    s2 = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_layernorm_bwd_amax", P(dy), P(x), P(g), P(mean), P(rstd), P(dx), 0, None, R, C, P(s2), None)
    assert float(s2[:2048:32].view(torch.int32).max().view(torch.float32)) == float(dx.abs().max())
    s3 = torch.zeros(SLOT, device="cuda")
    da = dy.clone()
    _lib.call("eav_gelu_bwd_amax", P(da), P(x), R * C, P(s3), None)
    assert float(s3[:2048:32].view(torch.int32).max().view(torch.float32)) == float(da.abs().max())


def _attn_ref(qkv, dO, B, H, N):
    D = H * 64
    x = qkv.double().view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    s = (x[0] @ x[1].transpose(-1, -2)) * 0.125
    o = torch.softmax(s, -1) @ x[2]
    out = o.permute(0, 2, 1, 3).reshape(B * N, D)
    out.backward(dO.double())
    return out.detach(), x.grad.permute(1, 3, 0, 2, 4).reshape(B * N, 3 * D), torch.logsumexp(s, -1).reshape(B * H, N).detach()


def _attn_prep(x, B, N, ncols, secw, tmask):
    Npad = _lib.plain("eav_attn_sp_npad", N)
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_absmax", P(x), B * N, ncols, ncols, P(slot), None)
    rowp = torch.empty(B * N, 2 * ncols, dtype=torch.float16, device="cuda")
    tp = torch.empty(B, ncols // 64, 64, 2 * Npad, dtype=torch.float16, device="cuda")
    _lib.call("eav_attn_sp_prep", P(x), P(slot), P(rowp), P(tp), B, N, ncols, secw, tmask, None)
    return slot, rowp, tp


# (3, 4, .) / (3, 3, .): 12 / 9 image-heads = one full group of 8 on the XCD-aware workgroup map plus a remainder;
# N = 128 / 129: either side of the 64-row / 128-row workgroup switch
@pytest.mark.parametrize("cfg", [(1, 2, 64, 1.0, False), (2, 3, 197, 1.0, False), (1, 2, 1214, 1.0, False),
                                 (2, 2, 300, 3.0, True), (1, 1, 33, 1.0, False), (3, 4, 129, 1.0, False),
                                 (3, 3, 128, 1.0, False), (4, 4, 70, 1.0, False),
                                 # the software-pipelined forward (N >= 512): 16 / 18 / 17 key tiles = every tail of its
                                 # three-way unrolled tile loop, the last one ragged
                                 (1, 2, 512, 1.0, False), (2, 1, 576, 1.0, False), (1, 3, 530, 2.0, True)])
def test_attention_sp_is_fp32_grade(cfg):
    B, H, N, qs, spike = cfg
    D = H * 64
    torch.manual_seed(B * 1000 + N)
    qkv = torch.randn(B * N, 3 * D, device="cuda") * qs
    if spike:      # one key aligned with one query: the running maximum jumps inside a key tile (rescale branch)
        qkv[5, D:D + 64] = qkv[3, :64] * 6.0
    dO = torch.randn(B * N, D, device="cuda") * 1e-3
    ro, rg, rl = _attn_ref(qkv, dO, B, H, N)
    # split
    s_qkv, rowp, tp = _attn_prep(qkv, B, N, 3 * D, D, 7)
    ao, lse = torch.empty(B * N, D, device="cuda"), torch.empty(B * H, N, device="cuda")
    amax = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), P(amax), B, H, N, 64, 0.125, None)
    assert float(amax[:2048:32].view(torch.int32).max().view(torch.float32)) == float(ao.abs().max())
    s_do, dorow, dotp = _attn_prep(dO, B, N, D, D, 1)
    s_ds, delta = torch.zeros(SLOT, device="cuda"), torch.empty(B * H, N, device="cuda")
    dqkv = torch.empty(B * N, 3 * D, device="cuda")
    _lib.call("eav_attn_bwd_sp", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds), P(ao), P(dO), P(lse),
              P(delta), P(dqkv), None, B, H, N, 64, 0.125, None)
    # exact-fp32 kernels
    ao32, lse32 = torch.empty_like(ao), torch.empty_like(lse)
    _lib.call("eav_attn_fwd", P(qkv), P(ao32), P(lse32), B, H, N, 64, 0.125, None)
    dq32 = torch.empty_like(dqkv)
    _lib.call("eav_attn_bwd", P(qkv), P(ao32), P(dO), P(lse32), P(delta), P(dq32), B, H, N, 64, 0.125, None)

    # the output as the o-proj operand planes (eav_attn_fwd_sp_planes): scale = qkv's sigma, copied into the output's slot;
    # planes == the fp32 output of the same launch to split precision; the fp32 copy is optional; pad rows stay untouched
    s_ao = torch.zeros(SLOT, device="cuda")
    aop = torch.full(((B * N + 31) // 32 * 32, 2 * kpad(D)), 7.0, dtype=torch.float16, device="cuda")
    ao2, lse2 = torch.empty_like(ao), torch.empty_like(lse)
    _lib.call("eav_attn_fwd_sp_planes", P(rowp), P(tp), P(s_qkv), P(ao2), P(lse2), None, P(aop), P(s_ao), B, H, N, 64,
              0.125, None)
    assert torch.equal(ao2, ao) and torch.equal(lse2, lse)
    sig = float(s_qkv[2048])
    assert float(s_ao[2048]) == sig and float(s_ao[2049]) == float(s_qkv[2049])
    assert float(ao.abs().max()) * sig < 2.0 ** 15
    got = decode_planes(aop, B * N, D, sig)
    assert (got - ao.double()).abs().max().item() <= 2.0 ** -21 * (2.0 ** 15 / sig)
    assert (aop[B * N:] == 7).all()
    aop3 = torch.zeros_like(aop)
    _lib.call("eav_attn_fwd_sp_planes", P(rowp), P(tp), P(s_qkv), None, P(lse2), None, P(aop3), P(s_ao), B, H, N, 64,
              0.125, None)
    assert torch.equal(aop3[:B * N], aop[:B * N])

    # the backward writing dqkv as the operand planes of the q/k/v projection's gradient products (eav_attn_bwd_sp_planes):
    # scale from the rigorous bound (eav_attn_dqkv_bound) >= max|dqkv|; planes == the fp32 dqkv of the plain call to split
    # precision in ABSOLUTE terms (2^-21 of the scale's range: the bound is loose, small elements sit far below it); the
    # column sums of every 32-row tile add up to the bias gradient; the fp32 copy is optional; pad rows untouched
    s_g = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_attn_dqkv_bound", P(s_g), P(s_do), P(s_qkv), N, 0.125, None)
    gsig = float(s_g[2048])
    assert float(dqkv.abs().max()) * gsig < 2.0 ** 15 and float(s_g[2049]) == 1.0 / gsig
    nrb = (N + 31) // 32
    gpl = torch.full(((B * N + 31) // 32 * 32, 2 * kpad(3 * D)), 7.0, dtype=torch.float16, device="cuda")
    cs = torch.zeros(B * nrb, 3 * D, device="cuda")
    s_ds2, delta2 = torch.zeros(SLOT, device="cuda"), torch.empty(B * H, N, device="cuda")
    dq2 = torch.empty_like(dqkv)
    _lib.call("eav_attn_bwd_sp_planes", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds2), P(ao), P(dO), P(lse),
              P(delta2), P(dq2), None, P(gpl), P(s_g), P(cs), None, None, B, H, N, 64, 0.125, None)
    assert torch.equal(dq2, dqkv)
    got = decode_planes(gpl, B * N, 3 * D, gsig)
    assert (got - dqkv.double()).abs().max().item() <= 2.0 ** -21 * (2.0 ** 15 / gsig) * 2.0 ** -10 + \
        2.0 ** -21 * float(dqkv.abs().max())
    assert (gpl[B * N:] == 7).all()
    want_cs = dqkv.double().sum(0)
    assert (cs.double().sum(0) - want_cs).abs().max().item() <= 1e-5 * float(dqkv.abs().max()) * (B * N) ** 0.5
    gpl3 = torch.zeros_like(gpl)
    _lib.call("eav_attn_bwd_sp_planes", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds2), P(ao), P(dO), P(lse),
              P(delta2), None, None, P(gpl3), P(s_g), None, None, None, B, H, N, 64, 0.125, None)
    assert torch.equal(gpl3[:B * N], gpl[:B * N])
    # delta = dO . O from the planes of dO and of the attention output (no fp32 O, dO read): the row sums to 2^-20 of
    # |dO| |O| sqrt(64), dqkv to the same bounds as the fp32-delta run
    delta3, dq3 = torch.empty_like(delta2), torch.empty_like(dqkv)
    _lib.call("eav_attn_bwd_sp_planes", P(rowp), P(tp), P(dorow), P(dotp), P(s_qkv), P(s_do), P(s_ds2), None, None, P(lse),
              P(delta3), P(dq3), None, None, None, None, P(aop), P(s_ao), B, H, N, 64, 0.125, None)
    assert (delta3 - delta2).abs().max().item() <= 2.0 ** -20 * 8 * float(dO.abs().max()) * float(ao.abs().max()) + 1e-30
    for i in range(3):
        sl = slice(i * D, (i + 1) * D)
        assert ((dq3[:, sl] - dqkv[:, sl]).abs().max() / dqkv[:, sl].abs().max()).item() <= 2e-6, i

    def rel(a, r):
        return ((a.double() - r).abs().max() / r.abs().max()).item()
    assert rel(ao, ro) <= 1.5 * rel(ao32, ro) + 2e-7
    # (+ 3 ulp of the largest lse: with a spike it reaches 64, one fp32 ulp there is 7.6e-6)
    assert (lse.double() - rl).abs().max().item() <= 1.5 * (lse32.double() - rl).abs().max().item() + 1e-6 + \
        3 * 2.0 ** -24 * rl.abs().max().item()
    for i in range(3):
        sl = slice(i * D, (i + 1) * D)
        assert rel(dqkv[:, sl], rg[:, sl]) <= 1.5 * rel(dq32[:, sl], rg[:, sl]) + 3e-7, i


@pytest.mark.parametrize("shape", [(256, 256, 64), (300, 200, 96), (128, 130, 40)])
@pytest.mark.parametrize("opts", [dict(bias=True), dict(bias=True, gelu=1, pre=True), dict(bias=True, resid=True),
                                  dict(gelu=1, resid=True, acc=True, pre=True), dict(acc=True), dict(gelu=2),
                                  dict(bias=True, gelu=2, resid=True)])
@pytest.mark.parametrize("tile", [64, 128])         # 64 x 128 tiles (these launches are far below one tile per CU) / 128 x 128
def test_gemm_sp_epilogue_option_matrix(shape, opts, tile, tile_hook):
    """Every epilogue option combination on tiles inside the matrix (16-byte path) and ragged ones (element path),
    against float64: C = [accumulate C0 +] [resid +] act(alpha A.B^T + bias), act = GELU or . x gelu'(pre)."""
    M, N, K = shape
    tile_hook(tile)
    torch.manual_seed(M + N + len(opts))
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.2
    bias = torch.randn(N, device="cuda") if opts.get("bias") else None
    resid = torch.randn(M, N, device="cuda") if opts.get("resid") else None
    C0 = torch.randn(M, N, device="cuda")
    gelu = opts.get("gelu", 0)
    xin = torch.randn(M, N, device="cuda") * 1.5
    pre = xin.clone() if gelu == 2 else (torch.empty(M, N, device="cuda") if opts.get("pre") else None)
    lin = 0.75 * (A.double() @ B.double().t()) + (bias.double() if bias is not None else 0.0)
    if gelu == 1:
        want = torch.nn.functional.gelu(lin)
    elif gelu == 2:
        xd = xin.double().requires_grad_(True)
        torch.nn.functional.gelu(xd).backward(torch.ones_like(xd))
        want = lin * xd.grad
    else:
        want = lin
    if resid is not None:
        want = want + resid.double()
    if opts.get("acc"):
        want = want + C0.double()
    kw = dict(alpha=0.75, gelu=gelu, C=C0.clone(), acc=1 if opts.get("acc") else 0)
    if bias is not None:
        kw["bias"] = bias
    if resid is not None:
        kw["resid"] = resid
    if pre is not None:
        kw["pre"] = pre
    amax = torch.zeros(SLOT, device="cuda")
    got = gemm_sp(A, B, amax=amax, **kw)
    tol = 3e-5 * max(1.0, float(want.abs().max()))
    assert (got.double() - want).abs().max().item() < tol
    if gelu == 1 and pre is not None:
        assert (pre.double() - lin).abs().max().item() < tol
    if gelu == 2:
        assert torch.equal(pre, xin)             # read-only in the backward mode
    assert float(amax[:2048:32].view(torch.int32).max().view(torch.float32)) == float(got.abs().max())


def test_pre_activation_only_epilogue_and_gelu_conversion():
    """gelu = 3 + eav_sp_convert_gelu (fc1 of the encoder: no fp32 activation tensor) give the same planes, bit for bit,
    as gelu = 1 followed by eav_sp_convert of the stored activation."""
    torch.manual_seed(9)
    M, N, K = 300, 256, 96
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.3
    bias = torch.randn(N, device="cuda")
    pre1 = torch.empty(M, N, device="cuda")
    s1 = torch.zeros(SLOT, device="cuda")
    act = gemm_sp(A, B, bias=bias, gelu=1, pre=pre1, amax=s1)
    p1 = torch.zeros(M, 2 * kpad(N), dtype=torch.float16, device="cuda")
    p1T = torch.zeros(N, 2 * kpad(M), dtype=torch.float16, device="cuda")
    _lib.call("eav_sp_convert", P(act), M, N, N, P(s1), P(p1), P(p1T), None)
    s3 = torch.zeros(SLOT, device="cuda")
    pre3 = gemm_sp(A, B, bias=bias, gelu=3, amax=s3)
    p3, p3T = torch.zeros_like(p1), torch.zeros_like(p1T)
    _lib.call("eav_sp_convert_gelu", P(pre3), M, N, N, P(s3), P(p3), P(p3T), None)
    assert torch.equal(pre3, pre1)
    assert float(s3[:2048:32].view(torch.int32).max()) == float(s1[:2048:32].view(torch.int32).max())
    assert torch.equal(p3.view(torch.int16), p1.view(torch.int16)) and torch.equal(p3T.view(torch.int16), p1T.view(torch.int16))


def decode_planes(pl, R, C, sigma):
    """planes [Rp, 2*Cp] f16 ([R][Cp/8][2][8]) -> fp64 [R, C]: (hi + lo 2^-11) / sigma"""
    v = pl[:R].view(R, -1, 2, 8).double()
    return ((v[:, :, 0, :] + v[:, :, 1, :] / 2048.0).reshape(R, -1)[:, :C]) / sigma


@pytest.mark.parametrize("M,D", [(37, 64), (1214, 768), (300, 1024), (70, 8)])
def test_layernorm_backward_writes_the_operand_planes_itself(M, D):
    """eav_layernorm_bwd_planes: dx (accumulated) bit-equal to eav_layernorm_bwd_amax, dgamma / dbeta partials identical, the
    third section of the partials sums to the column sums of the stored value, the planes hold the stored value to split
    precision under the scale of eav_layernorm_bwd_bound (>= the measured maximum, which lands in the slot's shards)."""
    torch.manual_seed(3 * M + D)
    x = torch.randn(M, D, device="cuda") * 2 + 0.3
    x[M // 3] = 0.467                                   # a constant row: rstd = 1e6
    dy = torch.randn(M, D, device="cuda") * 1e-3
    g = torch.rand(D, device="cuda") * 3 + 0.2
    b = torch.randn(D, device="cuda")
    old = torch.randn(M, D, device="cuda") * 1e-2
    y, mean, rstd = torch.empty_like(x), torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    _lib.call("eav_layernorm_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-12, None)
    npart = _lib.plain("eav_layernorm_bwd_nparts", M)
    dx1, part1, s1 = old.clone(), torch.zeros(npart, 2 * D, device="cuda"), torch.zeros(SLOT, device="cuda")
    _lib.call("eav_layernorm_bwd_amax", P(dy), P(x), P(g), P(mean), P(rstd), P(dx1), 1, P(part1), M, D, P(s1), None)
    s_old, s_dy, s2 = (torch.zeros(SLOT, device="cuda") for _ in range(3))
    _lib.call("eav_sp_absmax", P(old), M, D, D, P(s_old), None)
    _lib.call("eav_sp_absmax", P(dy), M, D, D, P(s_dy), None)
    _lib.call("eav_layernorm_bwd_bound", P(s2), P(s_old), P(s_dy), P(g), P(rstd), M, D, None, None)
    sig = float(s2[2048])
    # the same bound from the forward's slot: eav_layernorm_fwd_amax leaves max(rstd) in word 1 of the shard lines
    s_f, s3 = torch.zeros(SLOT, device="cuda"), torch.zeros(SLOT, device="cuda")
    _lib.call("eav_layernorm_fwd_amax", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-12, P(s_f), None)
    assert float(s_f[1:2048:32].max()) == float(rstd.max())
    _lib.call("eav_layernorm_bwd_bound", P(s3), P(s_old), P(s_dy), P(g), None, M, D, P(s_f), None)
    assert float(s3[2048]) == sig
    dx2, part2 = old.clone(), torch.zeros(npart, 3 * D, device="cuda")
    pl = torch.full(((M + 31) // 32 * 32, 2 * kpad(D)), 7.0, dtype=torch.float16, device="cuda")
    _lib.call("eav_layernorm_bwd_planes", P(dy), P(x), P(g), P(mean), P(rstd), P(dx2), 1, P(part2), M, D, P(s2), P(pl), None,
              None, None, None)
    # the bound formed inside the launch (slots given) = the one of the stand-alone call, same planes
    s4, dx4, part4, pl4 = torch.zeros(SLOT, device="cuda"), old.clone(), torch.zeros(npart, 3 * D, device="cuda"), torch.zeros_like(pl)
    _lib.call("eav_layernorm_bwd_planes", P(dy), P(x), P(g), P(mean), P(rstd), P(dx4), 1, P(part4), M, D, P(s4), P(pl4), P(s_old),
              P(s_dy), P(s_f), None)
    assert float(s4[2048]) == sig and float(s4[2049]) == 1.0 / sig and torch.equal(dx4, dx2) and torch.equal(part4, part2)
    # a slot_rstd the matching forward never wrote (word 1 still zero) is not trusted: both forms walk rstd[M] instead and
    # arrive at the same bound - a zero max(rstd) would shrink it to max|dx_old| and overflow the fp16 pieces (advisor r4)
    s_z, s5, s6 = (torch.zeros(SLOT, device="cuda") for _ in range(3))
    _lib.call("eav_layernorm_bwd_bound", P(s5), P(s_old), P(s_dy), P(g), P(rstd), M, D, P(s_z), None)
    assert float(s5[2048]) == sig
    dx6, part6, pl6 = old.clone(), torch.zeros(npart, 3 * D, device="cuda"), torch.zeros_like(pl)
    _lib.call("eav_layernorm_bwd_planes", P(dy), P(x), P(g), P(mean), P(rstd), P(dx6), 1, P(part6), M, D, P(s6), P(pl6), P(s_old),
              P(s_dy), P(s_z), None)
    assert float(s6[2048]) == sig and torch.equal(dx6, dx2) and torch.equal(pl6[:M, :2 * D], pl[:M, :2 * D])
    assert torch.equal(dx2, dx1)
    assert torch.equal(pl4[:M, :2 * D], pl[:M, :2 * D])
    assert torch.equal(part2[:, :2 * D], part1)
    mx = float(dx1.abs().max())
    assert mx * sig < 2.0 ** 15
    assert float(s2[:2048:32].view(torch.int32).max().view(torch.float32)) == mx
    want = dx1.double().sum(0)
    assert (part2[:, 2 * D:].double().sum(0) - want).abs().max().item() <= 1e-5 * mx * M ** 0.5 + 1e-12
    got = decode_planes(pl, M, D, sig)
    assert (got - dx1.double()).abs().max().item() <= 2.0 ** -21 * mx + 2.0 ** -35 / sig
    assert (pl[M:] == 7).all() and (pl[:M, 2 * D:] == 7).all()


@pytest.mark.parametrize("M,D", [(37, 64), (1214, 768), (300, 1024), (5, 8)])
def test_layernorm_writes_the_operand_planes_itself(M, D):
    """eav_layernorm_fwd_planes: the LayerNorm output as row planes scaled by a slot's sigma (no fp32 copy, no
    conversion pass) == the fp32 kernel's output to split precision (2^-22 of the scale), statistics identical, pad
    columns / rows untouched."""
    torch.manual_seed(M + D)
    x = torch.randn(M, D, device="cuda") * 3 + 0.5
    x[M // 2] = 0.467                                   # a constant row (AST padding value): finite, zero output + beta
    g = torch.rand(D, device="cuda") + 0.5
    b = torch.randn(D, device="cuda") * 0.1
    y = torch.empty_like(x)
    st = torch.empty(2, M, device="cuda")
    _lib.call("eav_layernorm_fwd", P(x), P(g), P(b), P(y), P(st), P(st) + 4 * M, M, D, 1e-12, None)
    slot = torch.zeros(SLOT, device="cuda")
    sigma = 2.0 ** (14 - np.floor(np.log2(float(np.sqrt(D) * g.abs().max() + b.abs().max()))))
    slot[2048], slot[2049] = sigma, 1.0 / sigma
    pl = torch.full(((M + 31) // 32 * 32, 2 * kpad(D)), 7.0, dtype=torch.float16, device="cuda")
    st2 = torch.empty(2, M, device="cuda")
    y2 = torch.empty_like(x)
    _lib.call("eav_layernorm_fwd_planes", P(x), P(g), P(b), P(y2), P(pl), P(slot), P(st2), P(st2) + 4 * M, M, D, 1e-12, None)
    assert torch.equal(y2, y) and torch.equal(st2, st)
    got = decode_planes(pl, M, D, sigma)
    assert (got - y.double()).abs().max().item() <= 2.0 ** -21 * (2.0 ** 15 / sigma)
    assert float(y.abs().max()) * sigma < 2.0 ** 15                      # the bound holds
    assert (pl[M:] == 7).all() and (pl[:, 2 * D:] == 7).all()             # pad rows / columns are not written
    _lib.call("eav_layernorm_fwd_planes", P(x), P(g), P(b), None, P(pl), P(slot), P(st2), P(st2) + 4 * M, M, D, 1e-12, None)
    assert torch.equal(decode_planes(pl, M, D, sigma), got)               # y is optional


@pytest.mark.parametrize("shape", [(256, 384, 96), (1000, 200, 100), (9712, 3072, 768), (130, 128, 64)])
@pytest.mark.parametrize("gelu", [0, 1])
def test_gemm_epilogue_writes_the_next_operand_planes(shape, gelu):
    """eav_gemm_sp_planes: bias [+ erf-GELU] in the epilogue, the result leaves as the row planes of the next product
    (aligned tiles and ragged last tiles, N % 8 == 0) - equal to the fp32 output of the same launch to split precision;
    the pre-activation is stored as before; C may be NULL."""
    M, N, K = shape
    torch.manual_seed(M + N + K + gelu)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda") * 0.3
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    pre = torch.empty(M, N, device="cuda")
    ref = A.double() @ B.double().t() + bias.double()
    bound = float((A.norm(dim=1).max() * B.norm(dim=1).max() + bias.abs().max()))
    sigma = 2.0 ** (14 - np.floor(np.log2(bound)))
    slot = torch.zeros(SLOT, device="cuda")
    slot[2048], slot[2049] = sigma, 1.0 / sigma
    pl = torch.full(((M + 31) // 32 * 32, 2 * kpad(N)), 7.0, dtype=torch.float16, device="cuda")
    _lib.call("eav_gemm_sp_planes", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), gelu,
              P(pre) if gelu else None, None, 0, 0, None, P(pl), P(slot), None)
    want = torch.nn.functional.gelu(ref) if gelu else ref
    assert (C.double() - want).abs().max().item() < 2e-5
    if gelu:
        assert (pre.double() - ref).abs().max().item() < 2e-5
    got = decode_planes(pl, M, N, sigma)
    assert (got - C.double()).abs().max().item() <= 2.0 ** -21 * (2.0 ** 15 / sigma)
    assert (pl[M:] == 7).all() and (pl[:, 2 * N:] == 7).all()
    pl2 = torch.zeros_like(pl)
    _lib.call("eav_gemm_sp_planes", P(pa), P(pb), None, P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), gelu,
              P(pre) if gelu else None, None, 0, 0, None, P(pl2), P(slot), None)
    assert torch.equal(pl2[:M, :2 * N], pl[:M, :2 * N])                   # no fp32 copy needed


def test_forward_scales_are_bounds_and_row_norms_are_exact():
    torch.manual_seed(5)
    R, C = 3072, 768
    w = torch.randn(R, C, device="cuda") * 0.02
    w[77] *= 9
    out = torch.zeros(1, device="cuda")
    _lib.call("eav_rownorm_max", P(w), R, C, C, P(out), None)
    assert abs(float(out) - float(w.double().norm(dim=1).max())) < 1e-5 * float(out)
    for r, c, ld in ((5, 1300, 1300), (9, 1027, 1031), (130, 40, 64)):     # long / unaligned (scalar path) / short rows
        wv = torch.randn(r, ld, device="cuda")
        o2 = torch.zeros(1, device="cuda")
        _lib.call("eav_rownorm_max", P(wv), r, c, ld, P(o2), None)
        assert abs(float(o2) - float(wv[:, :c].double().norm(dim=1).max())) < 1e-5 * float(o2)
    # one layer laid out as [g1 | b1 | g2 | b2 | bfc1]
    D, FF = C, R
    g1, b1, g2, b2 = (torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.2,
                      torch.rand(D, device="cuda") * 3, torch.randn(D, device="cuda") * 0.2)
    bf = torch.randn(FF, device="cuda") * 0.5
    flat = torch.cat([g1, b1, g2, b2, bf]).contiguous()
    slots = torch.zeros(5, SLOT, device="cuda")
    _lib.call("eav_tf_forward_scales", P(flat), 0, 1, 0, D, 2 * D, 3 * D, 4 * D, D, FF, P(out), P(slots), 0, 0, 3, 4, None)
    sd = np.sqrt(D)
    b_y1 = sd * float(g1.abs().max()) + float(b1.abs().max())
    b_y2 = sd * float(g2.abs().max()) + float(b2.abs().max())
    b_act = (sd * float(g2.abs().max()) + float(b2.double().norm())) * float(out) + float(bf.abs().max())
    for k, bnd in ((0, b_y1), (3, b_y2), (4, b_act)):
        sg = float(slots[k, 2048])
        assert sg == 2.0 ** (14 - np.floor(np.log2(bnd * 1.0001))) and float(slots[k, 2049]) == 1.0 / sg, (k, sg, bnd)
    assert float(slots[1, 2048]) == 0 and float(slots[2, 2048]) == 0      # the other slots are not touched
    # the bounds hold on data: LayerNorm rows and the MLP activation they feed
    x = torch.randn(500, D, device="cuda") * torch.exp(torch.randn(500, 1, device="cuda"))
    xh = (x - x.mean(1, keepdim=True)) / x.var(1, unbiased=False, keepdim=True).sqrt()
    y2 = xh * g2 + b2
    assert float(y2.abs().max()) <= b_y2
    act = torch.nn.functional.gelu(y2 @ w.t() + bf)
    assert float(act.abs().max()) <= b_act


@pytest.mark.parametrize("shape", [(2048, 768, 768), (9712, 768, 3072), (1000, 200, 100)])
def test_rows_of_any_magnitude_keep_fp32_grade_relative_precision(shape):
    """Per-row-block operand scales: with log-normal ROW scales of sigma = 4 (ten decimal orders between the rows - what a
    gradient tensor looks like when a few tokens carry almost all of the loss) every output element is as accurate,
    relative to sum |a||b| of ITS OWN row, as the exact-fp32 MFMA kernel's (bound: 2x).  A 32-row block (one MFMA tile of rows)
    whose maximum is >= 2^8 below the tensor's gets its own power of two (EAV_SLOT_BMAX / EAV_SLOT_BEXP); one scale per tensor
    loses the lo piece of rows 2^29 below the maximum (6.7e-5 on this case before).  With the rows in RANDOM order (the
    `wide=True` case of tools/gemm_sp_bench.py) a block's scale is set by its largest row: every row is then precise relative
    to its BLOCK's maximum, i.e. element-wise fp32-grade for the rows within 2^29 of it - all but the few ordinary rows that
    share a block of 32 with one of the giant rows (asserted: >= 97 % of the rows meet the 2x bound; per-ROW scales would be
    needed for the rest, section 7.1 of DESIGN.md)."""
    M, N, K = shape
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.02
    rows = torch.exp(torch.randn(M, 1, device="cuda") * 4)
    rows, _ = rows.sort(dim=0)                      # slowly varying along the rows (blocks differ, rows of a block agree)
    A = A * rows
    ref = A.double() @ B.double().t()
    den = A.double().abs() @ B.double().abs().t()
    e_sp = ((gemm_sp(A, B).double() - ref).abs() / den).max().item()
    e_32 = ((gemm_f32(A, B).double() - ref).abs() / den).max().item()
    assert e_sp <= 2 * e_32, (e_sp, e_32)
    slot, pl, _ = planes(A)
    bexp = slot[3104:3104 + (M + 31) // 32].view(torch.int32)
    assert int(bexp.max()) >= 8 and int(bexp.min()) == 0           # small blocks were boosted, the largest was not
    # rows in random order: the same element-wise bound (a block's scale is set by its largest row; the other 31 rows of a
    # block are at most ~2^24 below it), and in any case precise relative to their BLOCK's maximum
    for seed in range(3):
        g = torch.Generator(device="cuda").manual_seed(seed)
        A2 = A[torch.randperm(M, device="cuda", generator=g)]
        ref2 = A2.double() @ B.double().t()
        den2 = A2.double().abs() @ B.double().abs().t()
        C2 = gemm_sp(A2, B).double()
        row_sp = ((C2 - ref2).abs() / den2).max(dim=1).values
        e_322 = ((gemm_f32(A2, B).double() - ref2).abs() / den2).max().item()
        frac = float((row_sp <= 2 * e_322).double().mean())
        assert frac >= 0.97, (seed, frac)
        blockmax = torch.stack([A2[i:i + 32].abs().max() for i in range(0, M, 32)]).repeat_interleave(32)[:M].double()
        e2 = ((C2 - ref2).abs() / (blockmax[:, None] * B.double().abs().sum(1)[None, :])).max().item()
        assert e2 < 1e-6


def test_weight_gradient_undoes_the_row_block_boosts():
    """The token-contracting product reads the same planes: a boosted token block's fragments are scaled back before the
    MFMAs.  dY with token blocks 2^12 .. 2^30 below the largest, X plain: dW equals the float64 product to 2e-7 of
    sum |a||b| - with and without the boosts - and the bias-gradient partials of the boosted conversion are the plain
    column sums."""
    tokens, N, K = 4096, 256, 384
    torch.manual_seed(11)
    dY = torch.randn(tokens, N, device="cuda")
    X = torch.randn(tokens, K, device="cuda")
    scale = torch.ones(tokens, 1, device="cuda")
    scale[128:256] = 2.0 ** -12
    scale[1024:1536] = 2.0 ** -20
    scale[2048:2176] = 2.0 ** -30
    dY = dY * scale
    sa, pa = row_planes(dY)
    sb, pb = row_planes(X)
    bexp = sa[3104:3104 + tokens // 32].view(torch.int32).cpu().tolist()           # one entry per 32 tokens
    # (a block's boost = exponent of the tensor maximum - exponent of ITS maximum: 12 or 13 for a block scaled by 2^-12)
    assert all(bexp[i] in (12, 13) for i in range(4, 8)) and all(bexp[i] in (20, 21) for i in range(32, 48))
    assert all(bexp[i] in (30, 31) for i in range(64, 68)) and bexp[0] == 0 and bexp[8] == 0 and bexp[68] == 0
    C = torch.empty(N, K, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", N, K, tokens)
    ws = torch.empty(max(ns, 1) * N * K, device="cuda")
    _lib.call("eav_gemm_sp_splitk", P(pa), P(pb), P(C), P(ws), P(sa), P(sb), N, K, tokens, 0, None)
    ref = dY.double().t() @ X.double()
    den = dY.double().abs().t() @ X.double().abs()
    assert ((C.double() - ref).abs() / den).max().item() < 2e-7
    # only the small blocks contribute (X zero elsewhere): their un-boosted fragments still carry them exactly enough
    X2 = X.clone()
    X2[:128] = 0
    X2[256:1024] = 0
    X2[1536:] = 0
    sb2, pb2 = row_planes(X2)
    _lib.call("eav_gemm_sp_splitk", P(pa), P(pb2), P(C), P(ws), P(sa), P(sb2), N, K, tokens, 0, None)
    ref2 = dY.double().t() @ X2.double()
    den2 = dY.double().abs().t() @ X2.double().abs()
    assert ((C.double() - ref2).abs() / den2).max().item() < 1e-5
    # bias gradient of the boosted conversion
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_absmax", P(dY), tokens, N, N, P(slot), None)
    npart = _lib.plain("eav_sp_convert_colsum_nparts", tokens)
    part = torch.empty(npart, N, device="cuda")
    pl = torch.zeros(tokens, 2 * kpad(N), dtype=torch.float16, device="cuda")
    _lib.call("eav_sp_convert_colsum", P(dY), tokens, N, N, P(slot), P(pl), None, P(part), None)
    got = part.double().sum(0)
    want = dY.double().sum(0)
    assert (got - want).abs().max().item() <= 1e-5 * dY.double().abs().sum(0).max().item()


@pytest.mark.parametrize("shape", [(512, 384, 768), (1000, 200, 100), (9712, 768, 3072), (300, 130, 40), (2500, 2304, 768)])
def test_one_term_products_are_the_fp16_product_of_the_planes(shape):
    """eav_gemm_sp_x1 / eav_gemm_sp_splitk_x1 (Encoder.grad_terms = 1): the hi.hi term alone - EXACTLY the product of the
    hi planes (fp16 values, fp32 accumulation) up to summation order, i.e. within a few 1e-7 of the float64 product of the
    decoded hi pieces, and within ~2^-11 per operand (tolerance 1e-3 of sum|a||b|) of the true product; same epilogues."""
    M, N, K = shape
    torch.manual_seed(M + N)
    A = torch.randn(M, K, device="cuda") * (1 + torch.arange(K, device="cuda") % 5)
    B = torch.randn(N, K, device="cuda") * 0.02
    bias = torch.randn(N, device="cuda")
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    C = torch.empty(M, N, device="cuda")
    _lib.call("eav_gemm_sp_x1", P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, P(bias), 0, None, None, 0, 0,
              None, None)
    Kp = kpad(K)

    def hi(pl, R):      # decode the hi pieces: planes [R][Kp/8][2][8]
        return pl.view(R, Kp // 8, 2, 8)[:, :, 0, :].reshape(R, Kp)[:, :K].double()
    ref_hi = hi(pa, M) @ hi(pb, N).t() * (sa[2049].double() * sb[2049].double()) + bias.double()
    den = A.double().abs() @ B.double().abs().t() + bias.double().abs()
    assert ((C.double() - ref_hi).abs() / den).max().item() < 3e-7
    ref = A.double() @ B.double().t() + bias.double()
    assert ((C.double() - ref).abs() / den).max().item() < 1e-3
    assert ((C.double() - ref).norm() / ref.norm()).item() < 6e-4
    # token-contracting form on the transposed problem: dW[N', K'] = sum_t X[t, n] Y[t, k]
    X, Y = A[:, :min(K, 256)].contiguous(), A[:, -min(K, 136):].contiguous() * 1e-3
    sx, px = row_planes(X)
    sy, py = row_planes(Y)
    n1, n2 = X.shape[1], Y.shape[1]
    W = torch.empty(n1, n2, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", n1, n2, M)
    ws = torch.empty(max(ns, 1) * n1 * n2, device="cuda")
    _lib.call("eav_gemm_sp_splitk_x1", P(px), P(py), P(W), P(ws), P(sx), P(sy), n1, n2, M, 0, None)
    refw = X.double().t() @ Y.double()
    denw = X.double().abs().t() @ Y.double().abs()
    assert ((W.double() - refw).abs() / denw).max().item() < 1e-3
    assert ((W.double() - refw).norm() / refw.norm()).item() < 6e-4
    W2 = torch.empty_like(W)
    _lib.call("eav_gemm_sp_splitk_x1", P(px), P(py), P(W2), P(ws), P(sx), P(sy), n1, n2, M, 0, None)
    assert torch.equal(W, W2)


@pytest.mark.parametrize("T,n1,n2", [(512, 256, 136), (1000, 200, 100), (9712, 768, 384), (25216, 64, 768), (300, 132, 40)])
def test_two_term_weight_gradient_rounds_only_the_activation_operand(T, n1, n2):
    """eav_gemm_sp_splitk_x2 (Encoder.wgrad_terms = 2): hi_A.hi_B + lo_A.hi_B - operand A (the gradient tensor of a weight
    gradient) keeps both pieces, operand B (the activation) is rounded to fp16 under its scale.  EXACTLY (up to summation
    order) the float64 product of A with the decoded hi pieces of B: within 3e-7 of sum|a||b| - the three-term kernel's
    accuracy class; against the true product the error is B's fp16 rounding (2^-12 relative per element, random signs), far
    below the one-term product's, which rounds both operands.  A carries a loose scale (1e-3 of its slot's bound, like the
    producers' a-priori gradient planes) without loss; bit-reproducible."""
    torch.manual_seed(T + n1)
    G = torch.randn(T, n1, device="cuda") * 1e-3 * (1 + torch.arange(n1, device="cuda") % 7)      # "gradient"
    X = torch.randn(T, n2, device="cuda") * (1 + torch.arange(n2, device="cuda") % 3)              # "activation"
    sg, pg = row_planes(G)
    sx, px = row_planes(X)
    W = torch.empty(n1, n2, device="cuda")
    ns = _lib.plain("eav_gemm_sp_splitk_plan", n1, n2, T)
    ws = torch.empty(max(ns, 1) * n1 * n2, device="cuda")
    _lib.call("eav_gemm_sp_splitk_x2", P(pg), P(px), P(W), P(ws), P(sg), P(sx), n1, n2, T, 0, None)
    n2p = kpad(n2)
    Tp = px.numel() // (2 * n2p)
    xhi = px.view(Tp, n2p // 8, 2, 8)[:T, :, 0, :].reshape(T, n2p)[:, :n2].double() * sx[2049].double()
    ref_hi = G.double().t() @ xhi
    den = G.double().abs().t() @ X.double().abs()
    assert ((W.double() - ref_hi).abs() / den).max().item() < 3e-7
    ref = G.double().t() @ X.double()
    e2 = ((W.double() - ref).norm() / ref.norm()).item()
    W1 = torch.empty_like(W)
    _lib.call("eav_gemm_sp_splitk_x1", P(pg), P(px), P(W1), P(ws), P(sg), P(sx), n1, n2, T, 0, None)
    e1 = ((W1.double() - ref).norm() / ref.norm()).item()
    W3 = torch.empty_like(W)
    _lib.call("eav_gemm_sp_splitk", P(pg), P(px), P(W3), P(ws), P(sg), P(sx), n1, n2, T, 0, None)
    e3 = ((W3.double() - ref).norm() / ref.norm()).item()
    print(f"wgrad T={T} {n1}x{n2}: relative error three terms {e3:.2e}, two terms {e2:.2e}, one term {e1:.2e}")
    assert e2 < 3e-4 and e2 < 0.85 * e1 and e3 < 1e-6
    assert ((W.double() - ref).abs() / den).max().item() < 3e-4
    W2 = torch.empty_like(W)
    _lib.call("eav_gemm_sp_splitk_x2", P(pg), P(px), P(W2), P(ws), P(sg), P(sx), n1, n2, T, 0, None)
    assert torch.equal(W, W2)


@pytest.mark.parametrize("shape", [(512, 384, 768), (1000, 200, 100), (9712, 768, 3072), (25216, 768, 64), (700, 2304, 96)])
def test_three_stage_256x128_form_gives_the_same_bits(shape):
    """The 256 x 128 / 8-wave / three-LDS-stage form of the kernel (tuning hook eav_gemm_sp_set_tile(2); epilogue patches
    aliased onto stage 2) against the default 128 x 128 form: the same MFMA sequence per output element, so bit-equal -
    with every epilogue option, ragged edges, K of one to many K-tiles and several tiles per persistent workgroup."""
    M, N, K = shape
    torch.manual_seed(M)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") * 0.05
    bias = torch.randn(N, device="cuda")
    resid = torch.randn(M, N, device="cuda")
    outs = []
    for tile in (1, 2):
        _lib.call("eav_gemm_sp_set_tile", tile)
        try:
            pre = torch.zeros(M, N, device="cuda")
            amax = torch.zeros(SLOT, device="cuda")
            C = gemm_sp(A, B, bias=bias, gelu=1, pre=pre, resid=resid, amax=amax)
            C1 = gemm_sp(A, B, C=C.clone(), acc=1, alpha=0.5)
            _lib.call("eav_gemm_sp_x1", *x1_args(A, B, C2 := torch.empty(M, N, device="cuda")))
        finally:
            _lib.call("eav_gemm_sp_set_tile", 0)
        outs.append((C, pre, C1, C2, amax[:2048].max()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def x1_args(A, B, C):
    M, K = A.shape
    N = B.shape[0]
    sa, pa, _ = planes(A)
    sb, pb, _ = planes(B)
    x1_args.keep = (sa, pa, sb, pb)
    return (P(pa), P(pb), P(C), P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 0, None, None, 0, 0, None, None)


def test_table_refresh_equals_the_per_matrix_passes():
    """eav_sp_refresh_planes (every GEMM weight of an encoder in two launches) against eav_sp_absmax + eav_sp_convert per
    matrix: the same planes, transposed planes, scales and row-block entries, bit for bit - matrices of different shapes,
    with and without transposed planes, rows far below the matrix maximum (boost exponents) included."""
    torch.manual_seed(3)
    shapes = [(768, 256), (2304, 768), (768, 768), (3072, 768), (768, 3072), (40, 64), (200, 136)]
    mats = [torch.randn(r, c, device="cuda") * (0.02 * (i + 1)) for i, (r, c) in enumerate(shapes)]
    mats[3][256:512] *= 1e-4                      # a row block far below the maximum
    for with_T in (True, False):
        ref, rows, keep = [], [], []
        slots = torch.zeros(len(mats), SLOT, device="cuda")
        for i, w in enumerate(mats):
            r, c = w.shape
            s, d, dT = planes(w, True, with_T)
            ref.append((s, d, dT))
            d2 = torch.zeros_like(d)
            dT2 = torch.zeros_like(dT) if with_T else None
            keep.append((d2, dT2))
            rows.append([w.data_ptr(), d2.data_ptr(), dT2.data_ptr() if with_T else 0, slots[i].data_ptr(), r | (c << 32)])
        jobs = torch.tensor(rows, dtype=torch.int64).cuda()
        _lib.call("eav_sp_refresh_planes", P(jobs), len(mats), max(r for r, _ in shapes), max(c for _, c in shapes), None)
        for i, (s, d, dT) in enumerate(ref):
            assert torch.equal(keep[i][0], d), (with_T, i)
            if with_T:
                assert torch.equal(keep[i][1], dT), (with_T, i)
            assert torch.equal(slots[i][2048:2050], s[2048:2050])                      # sigma, 1 / sigma
            assert torch.equal(slots[i][2080:].view(torch.int32), s[2080:].view(torch.int32))   # block maxima, boosts
            assert float(slots[i][:2048].max()) == float(s[:2048].max())
        if not with_T:
            assert int(slots[3][3104:].view(torch.int32).max()) >= 8     # (the small row block of matrix 3 really is boosted)


@pytest.mark.parametrize("shape", [(512, 384, 96), (9712, 3072, 768), (1000, 200, 100), (25216, 264, 64), (70, 72, 40)])
@pytest.mark.parametrize("tile", [0, 2])
def test_gelu_backward_epilogue_leaves_planes_and_bias_gradient_partials(shape, tile):
    """fc2's data gradient as the encoder backward launches it (eav_gemm_sp_ex: gelu = 2, C = NULL, planes_out, colsum_part):
    the product times gelu'(pre) leaves as the row planes of the NEXT products - scaled by the bound eav_sp_bound_scale
    derives from max|dh| and the column norms of W2 (eav_colnorm_max), which must really bound it - together with the
    64-row column-sum partials of fc1's bias gradient.  Full and ragged tiles, both tile forms; bit-reproducible."""
    M, N, K = shape          # tokens, FF, D
    torch.manual_seed(M + N)
    dh = torch.randn(M, K, device="cuda") * 1e-3 * (1 + torch.arange(M, device="cuda") % 3)[:, None]
    W2 = torch.randn(K, N, device="cuda") * 0.03            # fc2.weight [D, FF]
    pre = torch.randn(M, N, device="cuda") * 1.5
    sa, pa, _ = planes(dh)
    sb, pbT = planes(W2, want=False, wantT=True)[0::2]      # planes of W2^T [FF, D]: B operand, contraction over D
    cn = torch.zeros(1, device="cuda")
    _lib.call("eav_colnorm_max", P(W2), K, N, N, P(cn), None)
    assert abs(float(cn) - float(W2.double().norm(dim=0).max())) < 1e-5 * float(cn)
    slot = torch.zeros(SLOT, device="cuda")
    _lib.call("eav_sp_bound_scale", P(slot), P(sa), P(cn), 1.13 * float(np.sqrt(K)), None)
    sigma = float(slot[2048])
    x = pre.double()
    gp = 0.5 * (1 + torch.erf(x / np.sqrt(2))) + x * torch.exp(-x * x / 2) / np.sqrt(2 * np.pi)
    ref = (dh.double() @ W2.double()) * gp
    assert sigma == 2.0 ** np.round(np.log2(sigma)) and float(ref.abs().max()) * sigma < 2.0 ** 15      # a bound, a power of two
    nparts = (M + 63) // 64
    outs = []
    _lib.call("eav_gemm_sp_set_tile", tile)
    try:
        for _ in range(2):
            pl = torch.zeros((M + 31) // 32 * 32, 2 * kpad(N), dtype=torch.float16, device="cuda")
            part = torch.full((nparts, N), 7.0, device="cuda")
            _lib.call("eav_gemm_sp_ex", P(pa), P(pbT), None, P(sa), P(sb), M, N, K, N, 1, 0, 0, 1.0, None, 2, P(pre), None,
                      0, 0, None, P(pl), P(slot), P(part), 2, None)
            outs.append((pl, part))
    finally:
        _lib.call("eav_gemm_sp_set_tile", 0)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    pl, part = outs[0]
    got = decode_planes(pl, M, N, sigma)
    tol = 3e-7 * float((dh.double().abs() @ W2.double().abs()).max()) * 1.2 + 2.0 ** -21 * (2.0 ** 15 / sigma)
    assert (got - ref).abs().max().item() < tol
    assert (pl[M:] == 0).all()
    # partial p = column sums over rows [64 p, 64 p + 64) of the value the planes hold
    refp = torch.stack([ref[64 * p:64 * p + 64].sum(0) for p in range(nparts)])
    assert (part.double() - refp).abs().max().item() < 64 * tol
    assert ((part.double().sum(0) - ref.sum(0)).abs() / ref.abs().sum(0).clamp_min(1e-30)).max().item() < 1e-5


@pytest.mark.parametrize("cfg", [(2, 3, 197), (1, 2, 1214), (3, 1, 33), (2, 2, 64)])
def test_qkv_projection_writes_the_attention_row_planes(cfg):
    """The fused q/k/v projection as the encoder forward launches it: eav_gemm_sp_ex with EAV_GEMM_PLANES_NOLIFT writes the
    attention kernels' row planes (lo = fp16(sigma x - hi), no lift) scaled by the bound of eav_tf_forward_scales_qkv - which
    must bound the real output; eav_attn_sp_prep on the values the planes hold gives the same operands."""
    B, H, N = cfg
    D, M = 64 * H, B * N
    torch.manual_seed(B + H + N)
    g1, b1 = torch.rand(D, device="cuda") + 0.5, torch.randn(D, device="cuda") * 0.2
    x = torch.randn(M, D, device="cuda") * 2
    y1 = torch.nn.functional.layer_norm(x, (D,), g1, b1, 1e-12)
    Wqkv = torch.randn(3 * D, D, device="cuda") * 0.05
    bqkv = torch.randn(3 * D, device="cuda") * 0.1
    # one layer laid out as [g1 | b1 | g2 | b2 | bfc1 (FF = D) | bqkv]
    flat = torch.cat([g1, b1, g1, b1, torch.zeros(D, device="cuda"), bqkv]).contiguous()
    wn1, wnq = torch.ones(1, device="cuda"), torch.zeros(1, device="cuda")
    _lib.call("eav_rownorm_max", P(Wqkv), 3 * D, D, D, P(wnq), None)
    slots = torch.zeros(5, SLOT, device="cuda")
    _lib.call("eav_tf_forward_scales_qkv", P(flat), 0, 1, 0, D, 2 * D, 3 * D, 4 * D, 5 * D, D, D, P(wn1), P(wnq), P(slots), 0,
              0, 3, 4, 1, None)
    ref = y1.double() @ Wqkv.double().t() + bqkv.double()
    sig = float(slots[1, 2048])
    assert sig == 2.0 ** np.round(np.log2(sig)) and float(ref.abs().max()) * sig < 2.0 ** 15
    assert float(ref.abs().max()) * sig > 2.0 ** 6          # (and not absurdly loose)
    sy, py, _ = planes(y1)
    sw, pw, _ = planes(Wqkv)
    rowp = torch.full((M, 6 * D), 7.0, dtype=torch.float16, device="cuda")
    _lib.call("eav_gemm_sp_ex", P(py), P(pw), None, P(sy), P(sw), M, 3 * D, D, 3 * D, 1, 0, 0, 1.0, P(bqkv), 0, None, None, 0,
              0, None, P(rowp), P(slots[1]), None, 4, None)
    v = rowp.view(M, -1, 2, 8).double()
    got = (v[:, :, 0, :] + v[:, :, 1, :]).reshape(M, -1) / sig            # lo NOT lifted
    assert (got - ref).abs().max().item() < 4e-7 * float((y1.double().abs() @ Wqkv.double().abs().t()).max()) + 2.0 ** -24 / sig * 4
    # and eav_attn_sp_prep on the values the planes hold gives the same operands (up to fp16 rounding ties in hi)
    held = ((v[:, :, 0, :] + v[:, :, 1, :]).reshape(M, -1) / sig).float().contiguous()
    s2 = torch.zeros(SLOT, device="cuda")
    s2[0] = 2.0 ** 14.5 / sig
    rowp2 = torch.empty_like(rowp)
    _lib.call("eav_attn_sp_prep", P(held), P(s2), P(rowp2), None, B, N, 3 * D, D, 0, None)
    assert float(s2[2048]) == sig
    v2 = rowp2.view(M, -1, 2, 8).double()
    assert torch.equal(v2[:, :, 0, :] + v2[:, :, 1, :], v[:, :, 0, :] + v[:, :, 1, :])
