"""Per-kernel parity of the EEGNet HIP kernels (through the C ABI) against the
fp32 CPU restatement of the same op.  Tolerances are stated per test; the
reference arithmetic is fp32, so differences are summation-order rounding."""
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
    """Host array -> device tensor that stays alive until the test module is torn down
    (a temporary's memory would be recycled by the caching allocator before the kernel ran)."""
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
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f"{what}: max err {err.max():.3e}, ref max {np.abs(ref).max():.3e}"


def bn_buf(nch, mean, invstd, scale, shift, m1=None, m2=None):
    z = np.zeros(nch, np.float32)
    return dev(np.concatenate([mean, invstd, scale, shift, z if m1 is None else m1, z if m2 is None else m2]).astype(np.float32))


@pytest.mark.parametrize("B,C,S,K", [(2, 30, 500, 300), (1, 30, 10000, 300), (2, 5, 333, 300), (1, 3, 2100, 64),
                                     (3, 32, 130, 7)])
def test_fir_fwd(L, B, C, S, K):
    x = synth.normal(1, (B, C, S))
    w = synth.uniform(2, (8, K), -0.1, 0.1)
    xd, wd = dev(x), dev(w)
    y = torch.empty(B, 8, C, S, device="cuda")
    npart = L.plain("eav_eegnet_fir_fwd_nparts", B, C, S)
    part = torch.zeros(npart, 16, device="cuda")
    L.call("eav_eegnet_fir_fwd", xd.data_ptr(), wd.data_ptr(), y.data_ptr(), part.data_ptr(), B, C, S, K, None)
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).unsqueeze(1)
    left = (K - 1) // 2
    ref = F.conv2d(F.pad(xt, (left, K - 1 - left)), torch.from_numpy(w).view(8, 1, 1, K))
    close(y, ref, 1e-4, 2e-5, "y1")
    st = part.sum(0).cpu().double().numpy()
    close(st[:8], ref.double().sum((0, 2, 3)), 1e-5, 1e-3, "sum")
    close(st[8:], (ref.double() ** 2).sum((0, 2, 3)), 1e-5, 1e-3, "sumsq")


@pytest.mark.parametrize("B,C,S,K", [(2, 30, 500, 300), (1, 30, 10000, 300), (2, 4, 333, 300), (1, 3, 1100, 64)])
def test_fir_wgrad(L, B, C, S, K):
    x = synth.normal(3, (B, C, S))
    y1 = synth.normal(4, (B, 8, C, S))
    g1 = synth.normal(5, (B, 8, C, S))
    mean, invstd = synth.uniform(6, (8,), -0.2, 0.2), synth.uniform(7, (8,), 0.5, 2.0)
    scale, m1, m2 = synth.uniform(8, (8,), 0.5, 1.5), synth.uniform(9, (8,), -0.1, 0.1), synth.uniform(10, (8,), -0.1, 0.1)
    bn = bn_buf(8, mean, invstd, scale, np.zeros(8, np.float32), m1, m2)
    npart = L.plain("eav_eegnet_fir_wgrad_nparts", B, C, S)
    part = torch.zeros(npart, 8 * K, device="cuda")
    L.call("eav_eegnet_fir_wgrad", dev(x).data_ptr(), dev(y1).data_ptr(), dev(g1).data_ptr(), bn.data_ptr(),
           part.data_ptr(), B, C, S, K, None)
    out = torch.empty(8, K, device="cuda")
    L.call("eav_reduce_partials", part.data_ptr(), npart, 8 * K, 8 * K, 1.0, out.data_ptr(), None)
    torch.cuda.synchronize()
    bc = lambda v: torch.from_numpy(v).double()[None, :, None, None]  # noqa: E731
    dy = bc(scale) * (torch.from_numpy(g1).double() - bc(m1) - (torch.from_numpy(y1).double() - bc(mean)) * bc(invstd) * bc(m2))
    left = (K - 1) // 2
    xp = F.pad(torch.from_numpy(x).double(), (left, K - 1 - left))      # [B,C,S+K-1]
    win = xp.unfold(2, S, 1)                                            # [B,C,K,S]
    ref = torch.einsum("bfcs,bcks->fk", dy, win)
    close(out, ref, 2e-4, 2e-4 * float(ref.abs().max()), "dW1")


# (S <= 512 / <= 256: the two- / four-channel-group forms of dw_fwd for short rows)
@pytest.mark.parametrize("B,C,S", [(2, 30, 500), (1, 30, 2050), (2, 7, 333), (3, 30, 200), (2, 5, 256), (2, 30, 512),
                                   (2, 3, 130), (1, 30, 513)])
def test_dw_fwd_bwd(L, B, C, S):
    y1 = synth.normal(11, (B, 8, C, S))
    w2 = synth.uniform(12, (64, C), -0.3, 0.3)
    mean, invstd = synth.uniform(13, (8,), -0.2, 0.2), synth.uniform(14, (8,), 0.5, 2.0)
    gamma, beta = synth.uniform(15, (8,), 0.5, 1.5), synth.uniform(16, (8,), -0.2, 0.2)
    scale = gamma * invstd
    shift = beta - mean * scale
    bn = bn_buf(8, mean, invstd, scale, shift)
    nchunk = (S + 1023) // 1024
    z = torch.empty(B, 64, S, device="cuda")
    part = torch.zeros(B * nchunk, 128, device="cuda")
    y1d, w2d = dev(y1), dev(w2)
    L.call("eav_eegnet_dw_fwd", y1d.data_ptr(), bn.data_ptr(), w2d.data_ptr(), z.data_ptr(), part.data_ptr(), B, C, S, None)
    torch.cuda.synchronize()
    yt = torch.from_numpy(y1).double().requires_grad_(True)
    wt = torch.from_numpy(w2).double().requires_grad_(True)
    a1 = F.elu(yt * torch.from_numpy(scale).double()[None, :, None, None] + torch.from_numpy(shift).double()[None, :, None, None])
    zr = torch.einsum("bfcs,fdc->bfds", a1, wt.view(8, 8, C)).reshape(B, 64, S)
    close(z, zr, 1e-4, 1e-5, "z")
    st = part.sum(0).cpu().double().numpy()
    close(st[:64], zr.detach().sum((0, 2)), 1e-4, 1e-3, "sum z")
    close(st[64:], (zr.detach() ** 2).sum((0, 2)), 1e-4, 1e-3, "sum z2")
    # backward
    dz = synth.normal(17, (B, 64, S))
    zr.backward(torch.from_numpy(dz).double())
    g_ref = yt.grad / torch.from_numpy(scale).double()[None, :, None, None]   # dL/d(BN out) = dL/dy1 / scale
    g1 = torch.empty(B, 8, C, S, device="cuda")
    pst = torch.zeros(B * nchunk, 16, device="cuda")
    pw = torch.zeros(B * nchunk, 64 * C, device="cuda")
    L.call("eav_eegnet_dw_bwd", y1d.data_ptr(), dev(dz).data_ptr(), bn.data_ptr(), w2d.data_ptr(), g1.data_ptr(),
           pst.data_ptr(), pw.data_ptr(), B, C, S, None)
    torch.cuda.synchronize()
    close(g1, g_ref, 1e-4, 1e-5, "g1")
    close(pw.sum(0).view(64, C), wt.grad, 1e-4, 1e-4 * float(wt.grad.abs().max()), "dW2")
    yhat = (torch.from_numpy(y1).double() - torch.from_numpy(mean).double()[None, :, None, None]) * torch.from_numpy(invstd).double()[None, :, None, None]
    s = pst.sum(0).cpu().double().numpy()
    close(s[:8], g_ref.sum((0, 2, 3)), 1e-4, 1e-3, "sum g")
    close(s[8:], (g_ref * yhat).sum((0, 2, 3)), 1e-4, 1e-3, "sum g*yhat")


@pytest.mark.parametrize("B,T,P,drop", [(2, 500, 4, 0.0), (2, 125, 8, 0.0), (1, 2500, 8, 0.5), (2, 1000, 4, 0.5), (1, 333, 4, 0.0)])
def test_pool_fwd_bwd(L, B, T, P, drop):
    CH = 64
    u = synth.normal(21, (B, CH, T))
    mean, invstd = synth.uniform(22, (CH,), -0.2, 0.2), synth.uniform(23, (CH,), 0.5, 2.0)
    gamma, beta = synth.uniform(24, (CH,), 0.5, 1.5), synth.uniform(25, (CH,), -0.2, 0.2)
    scale, shift = gamma * invstd, beta - mean * gamma * invstd
    m1, m2 = synth.uniform(26, (CH,), -0.01, 0.01), synth.uniform(27, (CH,), -0.01, 0.01)
    bn = bn_buf(CH, mean, invstd, scale, shift, m1, m2)
    To = T // P
    mask = (synth.uniform(28, (B, CH, To)) >= 0.5).astype(np.uint8) if drop > 0 else None
    maskd = dev(mask) if mask is not None else None
    mp = maskd.data_ptr() if maskd is not None else None
    out = torch.empty(B, CH, To, device="cuda")
    ud = dev(u)
    L.call("eav_bn_elu_pool_fwd", ud.data_ptr(), bn.data_ptr(), out.data_ptr(), B, CH, T, P, drop, 0, mp, None, None)
    torch.cuda.synchronize()
    ut = torch.from_numpy(u).double().requires_grad_(True)
    bc = lambda v: torch.from_numpy(v).double()[None, :, None]  # noqa: E731
    act = F.elu(ut * bc(scale) + bc(shift))
    ref = F.avg_pool1d(act, P)
    if mask is not None:
        ref = ref * torch.from_numpy(mask).double() / (1 - drop)
    close(out, ref, 1e-5, 1e-6, "pool out")
    dp = synth.normal(29, (B, CH, To))
    ref.backward(torch.from_numpy(dp).double())
    g_ref = ut.grad / bc(scale)                      # gradient w.r.t. the BN output
    part = torch.zeros(B, 2 * CH, device="cuda")
    dpd = dev(dp)
    L.call("eav_bn_elu_pool_bwd_reduce", dpd.data_ptr(), ud.data_ptr(), bn.data_ptr(), part.data_ptr(), B, CH, T, P,
           drop, 0, mp, None, None)
    du = torch.empty(B, CH, T, device="cuda")
    L.call("eav_bn_elu_pool_bwd_apply", dpd.data_ptr(), ud.data_ptr(), bn.data_ptr(), bn.data_ptr() + 4 * 4 * CH,
           du.data_ptr(), B, CH, T, P, drop, 0, mp, None, None)
    torch.cuda.synchronize()
    uhat = (torch.from_numpy(u).double() - bc(mean)) * bc(invstd)
    s = part.sum(0).cpu().double().numpy()
    close(s[:CH], g_ref.sum((0, 2)), 1e-4, 1e-4, "sum g")
    close(s[CH:], (g_ref * uhat).sum((0, 2)), 1e-4, 1e-4, "sum g uhat")
    du_ref = bc(scale) * (g_ref - bc(m1) - uhat * bc(m2))
    close(du, du_ref, 1e-4, 1e-6, "du")
    # eval-mode step (BatchNorm on its running statistics, m1 = m2 = 0): gradient and sums from ONE pass
    du2 = torch.full((B, CH, T), float("nan"), device="cuda")
    part2 = torch.full((B, 2 * CH), float("nan"), device="cuda")
    L.call("eav_bn_elu_pool_bwd_eval", dpd.data_ptr(), ud.data_ptr(), bn.data_ptr(), du2.data_ptr(), part2.data_ptr(), B, CH,
           T, P, drop, 0, mp, None, None)
    torch.cuda.synchronize()
    close(du2, bc(scale) * g_ref, 1e-4, 1e-6, "du (eval)")
    s2 = part2.sum(0).cpu().double().numpy()
    close(s2[:CH], g_ref.sum((0, 2)), 1e-4, 1e-4, "sum g (eval)")
    close(s2[CH:], (g_ref * uhat).sum((0, 2)), 1e-4, 1e-4, "sum g uhat (eval)")
    close(part2, part, 1e-4, 1e-5 * float(part.abs().max()), "per-sample sums, one pass against two")


def test_elu_is_expm1_to_an_ulp_and_propagates_nan(L):
    """elu_f (csrc/eav_common.h: branch-free 13-instruction expm1 on the clamped argument) through eav_bn_elu_pool_fwd with an
    identity BatchNorm and four equal samples per pool window (their mean is exact): within 1.5 ulp of float64 expm1 for
    v <= 0 (libm's class; nn.ELU = expm1, EEGNet_tor.py:53,56,61), v itself for v > 0, NaN in -> NaN out (torch.nn.functional
    .elu propagates it: a diverged run must stay visible to loss / NaN checks), -inf -> -1, +inf -> +inf."""
    CH, n = 64, 4096
    x = np.concatenate([-np.logspace(-38, np.log10(40.0), CH * n - 4096 - 8), np.linspace(-20, 6, 4096),
                        [0.0, -0.0, -17.5, -17.4999, -88.0, 1e-30, 3.0, -1e-45]]).astype(np.float32)
    x[5] = np.nan
    x[6] = -np.inf
    x[7] = np.inf
    u = np.repeat(x.reshape(1, CH, n), 4, axis=2)
    one, zero = np.ones(CH, np.float32), np.zeros(CH, np.float32)
    bn = bn_buf(CH, zero, one, one, zero)
    out = torch.empty(1, CH, n, device="cuda")
    L.call("eav_bn_elu_pool_fwd", dev(u).data_ptr(), bn.data_ptr(), out.data_ptr(), 1, CH, 4 * n, 4, 0.0, 0, None, None, None)
    torch.cuda.synchronize()
    got = out.cpu().numpy().reshape(-1)
    assert np.isnan(got[5]) and got[6] == -1.0 and got[7] == np.inf
    ok = np.isfinite(x)
    ref = np.where(x[ok] > 0, x[ok].astype(np.float64), np.expm1(x[ok].astype(np.float64)))
    ulp = np.spacing(np.abs(ref.astype(np.float32))).astype(np.float64)
    err = np.abs(got[ok].astype(np.float64) - ref) / ulp
    assert err.max() <= 1.5, f"ELU: {err.max():.2f} ulp at v = {x[ok][err.argmax()]!r}"
    assert (got[ok][x[ok] > 0] == x[ok][x[ok] > 0]).all()


def test_dropout_generator_statistics(L):
    B, CH, T, P = 4, 64, 4000, 4
    u = np.ones((B, CH, T), np.float32)
    one, zero = np.ones(CH, np.float32), np.zeros(CH, np.float32)
    bn = bn_buf(CH, zero, one, one, zero)
    out = torch.empty(B, CH, T // P, device="cuda")
    L.call("eav_bn_elu_pool_fwd", dev(u).data_ptr(), bn.data_ptr(), out.data_ptr(), B, CH, T, P, 0.5, 1234, None, None, None)
    o = out.cpu().numpy()
    assert set(np.unique(o)) == {0.0, 2.0}
    assert abs((o == 0).mean() - 0.5) < 0.01
    out2 = torch.empty_like(out)
    L.call("eav_bn_elu_pool_fwd", dev(u).data_ptr(), bn.data_ptr(), out2.data_ptr(), B, CH, T, P, 0.5, 1234, None, None, None)
    assert torch.equal(out, out2)            # same seed -> same mask (needed by the backward)
    L.call("eav_bn_elu_pool_fwd", dev(u).data_ptr(), bn.data_ptr(), out2.data_ptr(), B, CH, T, P, 0.5, 1235, None, None, None)
    assert not torch.equal(out, out2)


@pytest.mark.parametrize("B,T", [(2, 125), (1, 2500), (3, 300)])
def test_conv64(L, B, T):
    x = synth.normal(31, (B, 64, T))
    w = synth.uniform(32, (64, 64, 16), -0.05, 0.05)
    wTf, wTb = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
    wd, xd = dev(w), dev(x)
    L.call("eav_conv64_prep_weights", wd.data_ptr(), wTf.data_ptr(), wTb.data_ptr(), None)
    nt = L.plain("eav_conv64_fwd_nparts", B, T)
    out = torch.empty(B, 64, T, device="cuda")
    part = torch.zeros(nt, 128, device="cuda")
    L.call("eav_conv64_fwd", xd.data_ptr(), wTf.data_ptr(), out.data_ptr(), part.data_ptr(), B, T, 7, None)
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    ref = F.conv1d(F.pad(xt, (7, 8)), wt)
    close(out, ref, 1e-4, 1e-5, "conv out")
    st = part.sum(0).cpu().double().numpy()
    close(st[:64], ref.detach().sum((0, 2)), 1e-4, 1e-3, "sum")
    close(st[64:], (ref.detach() ** 2).sum((0, 2)), 1e-4, 1e-3, "sumsq")
    du = synth.normal(33, (B, 64, T))
    ref.backward(torch.from_numpy(du).double())
    dud = dev(du)
    dx = torch.empty(B, 64, T, device="cuda")
    L.call("eav_conv64_fwd", dud.data_ptr(), wTb.data_ptr(), dx.data_ptr(), None, B, T, 8, None)
    npart = L.plain("eav_conv64_wgrad_nparts", B, T)
    pw = torch.zeros(npart, 65536, device="cuda")
    L.call("eav_conv64_wgrad", dud.data_ptr(), xd.data_ptr(), pw.data_ptr(), B, T, 7, None)
    dw = torch.empty(64, 64, 16, device="cuda")
    L.call("eav_reduce_partials", pw.data_ptr(), npart, 65536, 65536, 1.0, dw.data_ptr(), None)
    torch.cuda.synchronize()
    close(dx, xt.grad, 1e-4, 1e-5, "dgrad")
    close(dw, wt.grad, 1e-4, 1e-4 * float(wt.grad.abs().max()), "wgrad")


def test_bn_finalize_and_bwd_finalize(L):
    nch, nparts, count = 64, 7, 12345.0
    part = synth.uniform(41, (nparts, 2 * nch), 0.0, 1.0)
    part[:, :nch] *= 100.0
    part[:, nch:] = part[:, nch:] * 100.0 + 50000.0
    gamma, beta = synth.uniform(42, (nch,), 0.5, 1.5), synth.uniform(43, (nch,), -0.2, 0.2)
    rm, rv = synth.uniform(44, (nch,), -0.1, 0.1), synth.uniform(45, (nch,), 0.5, 1.5)
    for training in (1, 0):
        rmd, rvd = dev(rm), dev(rv)
        buf = torch.zeros(4 * nch, device="cuda")
        b0 = buf.data_ptr()
        L.call("eav_bn_finalize", dev(part).data_ptr(), nparts, nch, count, dev(gamma).data_ptr(), dev(beta).data_ptr(),
               rmd.data_ptr(), rvd.data_ptr(), training, 0.1, 1e-5, b0, b0 + 4 * nch, b0 + 8 * nch, b0 + 12 * nch, None)
        torch.cuda.synchronize()
        s, q = part[:, :nch].astype(np.float64).sum(0), part[:, nch:].astype(np.float64).sum(0)
        if training:
            mean = s / count
            var = q / count - mean ** 2
            close(rmd, 0.9 * rm + 0.1 * mean, 1e-6, 1e-7, "running mean")
            close(rvd, 0.9 * rv + 0.1 * var * count / (count - 1), 1e-6, 1e-7, "running var")
        else:
            mean, var = rm.astype(np.float64), rv.astype(np.float64)
            assert torch.equal(rmd.cpu(), torch.from_numpy(rm))
        invstd = 1 / np.sqrt(var + 1e-5)
        b = buf.cpu().numpy()
        close(b[:nch], mean, 1e-6, 1e-7, "mean")
        close(b[nch:2 * nch], invstd, 1e-6, 1e-7, "invstd")
        close(b[2 * nch:3 * nch], gamma * invstd, 1e-6, 1e-7, "scale")
        close(b[3 * nch:], beta - mean * gamma * invstd, 1e-5, 1e-6, "shift")
    out = torch.zeros(4 * nch, device="cuda")
    o = out.data_ptr()
    L.call("eav_bn_bwd_finalize", dev(part).data_ptr(), nparts, nch, count, 1, o, o + 4 * nch, o + 8 * nch, o + 12 * nch, None)
    r = out.cpu().numpy()
    close(r[:nch], q, 1e-6, 1e-3, "dgamma")
    close(r[nch:2 * nch], s, 1e-6, 1e-3, "dbeta")
    close(r[2 * nch:3 * nch], s / count, 1e-6, 1e-7, "m1")
    close(r[3 * nch:], q / count, 1e-6, 1e-6, "m2")


def test_renorm_rows(L):
    w = synth.uniform(51, (64, 30), -0.5, 0.5)
    w[::2] *= 0.1
    wd = dev(w)
    L.call("eav_renorm_rows", wd.data_ptr(), 64, 30, 1.0, None)
    ref = torch.from_numpy(w.copy())
    ref.renorm_(p=2, dim=0, maxnorm=1.0)
    close(wd, ref, 1e-6, 1e-7, "renorm")
    assert (np.linalg.norm(w, axis=1) > 1).any() and (np.linalg.norm(w, axis=1) < 1).any()


@pytest.mark.parametrize("B,NF,NC", [(4, 960, 5), (64, 19968, 5), (3, 77, 7)])
def test_dense_softmax_ce(L, B, NF, NC):
    x = synth.normal(61, (B, NF))
    w = synth.uniform(62, (NC, NF), -0.02, 0.02)
    b = synth.uniform(63, (NC,), -0.1, 0.1)
    y = synth.labels(64, B, NC)
    xd, wd, bd = dev(x), dev(w), dev(b)
    probs, logits = torch.empty(B, NC, device="cuda"), torch.empty(B, NC, device="cuda")
    L.call("eav_dense_softmax_fwd", xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), logits.data_ptr(), probs.data_ptr(), B, NF, NC, None)
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    bt = torch.from_numpy(b).double().requires_grad_(True)
    lg = F.linear(xt, wt, bt)
    pr = torch.softmax(lg, 1)
    close(logits, lg, 1e-5, 1e-5, "logits")
    close(probs, pr, 1e-5, 1e-6, "probs")
    loss = torch.empty((), device="cuda")
    dpr = torch.empty(B, NC, device="cuda")
    nc = torch.zeros((), dtype=torch.int32, device="cuda")
    L.call("eav_ce_fwd_bwd", probs.data_ptr(), dev(y).data_ptr(), loss.data_ptr(), dpr.data_ptr(), nc.data_ptr(), None, B, NC, None)
    pr2 = pr.detach().clone().requires_grad_(True)
    lref = F.cross_entropy(pr2, torch.from_numpy(y))          # CE on probabilities: the double softmax (Q3)
    lref.backward()
    close(loss, lref, 1e-5, 1e-6, "loss")
    close(dpr, pr2.grad, 1e-4, 1e-7, "dprobs")
    assert int(nc.item()) == int((pr.argmax(1) == torch.from_numpy(y)).sum())
    pr.backward(pr2.grad)
    dw, db, dx = torch.empty(NC, NF, device="cuda"), torch.empty(NC, device="cuda"), torch.empty(B, NF, device="cuda")
    L.call("eav_dense_softmax_bwd", dpr.data_ptr(), probs.data_ptr(), xd.data_ptr(), wd.data_ptr(), dw.data_ptr(),
           db.data_ptr(), dx.data_ptr(), B, NF, NC, None)
    torch.cuda.synchronize()
    close(dw, wt.grad, 1e-3, 1e-4 * float(wt.grad.abs().max()), "dW")
    close(db, bt.grad, 1e-3, 1e-4 * float(bt.grad.abs().max()), "db")
    close(dx, xt.grad, 1e-3, 1e-4 * float(xt.grad.abs().max()), "dx")


@pytest.mark.parametrize("decoupled,wd", [(0, 0.0), (1, 0.01), (0, 0.01)])
def test_adam_step_matches_torch(L, decoupled, wd):
    n = 10007
    p0, g = synth.normal(71, (n,)), synth.normal(72, (n,), 0, 1e-3)
    g[::50] = 1e-9 * g[::50]
    ref_p = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = (torch.optim.AdamW if decoupled else torch.optim.Adam)([ref_p], lr=1e-3, weight_decay=wd)
    pd, m, v = dev(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 4):
        gs = g * np.float32(step)
        ref_p.grad = torch.from_numpy(gs.copy())
        opt.step()
        L.call("eav_adam_step", pd.data_ptr(), dev(gs).data_ptr(), m.data_ptr(), v.data_ptr(), n, 1e-3, 0.9, 0.999,
               1e-8, wd, step, decoupled, None, None)
        torch.cuda.synchronize()
        close(pd, ref_p.detach(), 1e-6, 2e-7, f"p step {step}")


def test_gather_rows(L):
    from eav_amd.eegnet import gather_batch
    x = synth.normal(81, (20, 1, 3, 37))          # 111 floats per row: scalar path
    x4 = synth.normal(82, (20, 1, 30, 500))       # vector path
    y = synth.labels(83, 20)
    idx = torch.tensor([5, 0, 19, 5, 7], device="cuda")
    for arr in (x, x4):
        d, t = gather_batch(dev(arr), dev(y), idx)
        assert torch.equal(d.cpu(), torch.from_numpy(arr)[idx.cpu()]) and torch.equal(t.cpu(), torch.from_numpy(y)[idx.cpu()])


def test_cross_entropy_rejects_labels_outside_the_class_range():
    """torch's CrossEntropyLoss asserts on a class index outside [0, classes); here such a label never indexes anything
    and check() raises - e.g. the reference's raw EEG labels 1,3,5,7,9 fed to a 5-class head."""
    from eav_amd._lib import EavError
    from eav_amd.optim import CrossEntropyLoss
    crit = CrossEntropyLoss()
    scores = torch.rand(6, 5, device="cuda", requires_grad=True)
    good = torch.tensor([0, 1, 2, 3, 4, 0], device="cuda")
    crit(scores, good).backward()
    crit.check()
    g_ok = scores.grad.clone()
    assert torch.isfinite(g_ok).all()
    for bad_value in (9, -1):
        bad = good.clone()
        bad[3] = bad_value
        scores.grad = None
        loss = crit(scores, bad)
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(scores.grad).all()      # nothing read out of bounds
        with pytest.raises(EavError, match="outside"):
            crit.check()
        crit.check()                                                           # the flag is cleared by the raise
    with pytest.raises(EavError, match="targets for"):
        crit(scores, good[:4])
    with torch.no_grad():                                                      # validate(): no gradient buffer needed
        assert torch.isfinite(crit(scores.detach(), good))


def test_cross_entropy_matches_torch_semantics_for_ignored_rows_and_repeated_backward():
    """nn.CrossEntropyLoss (EEGNet_tor.py:81, Transformer_Audio.py:31): targets equal to ignore_index = -100 are left out
    of the mean and get a zero gradient; a bad label contributes a ZERO gradient row (never a wrong update); a second
    backward through the same graph (retain_graph) or a non-unit upstream gradient scales a copy, not the stored
    gradient."""
    from eav_amd._lib import EavError
    from eav_amd.optim import CrossEntropyLoss
    torch.manual_seed(3)
    scores = torch.randn(7, 5, device="cuda", requires_grad=True)
    y = torch.tensor([0, -100, 2, 3, -100, 1, 4], device="cuda")
    ref_s = scores.detach().cpu().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss()(ref_s, y.cpu())
    ref.backward()
    crit = CrossEntropyLoss()
    loss = crit(scores, y)
    loss.backward(retain_graph=True)
    crit.check()                                                    # -100 is not an error
    assert abs(float(loss) - float(ref)) < 1e-6
    assert torch.allclose(scores.grad.cpu(), ref_s.grad, atol=1e-7)
    assert (scores.grad[1] == 0).all() and (scores.grad[4] == 0).all()
    # second backward with an upstream gradient of 3: exactly 3x the unit result, and a third one with 1 again is not
    # contaminated by the scaling
    g1 = scores.grad.clone()
    scores.grad = None
    loss.backward(gradient=torch.tensor(3.0, device="cuda"), retain_graph=True)
    assert torch.allclose(scores.grad, 3 * g1, rtol=1e-6, atol=0)
    scores.grad = None
    loss.backward()
    assert torch.equal(scores.grad, g1)
    # a bad label: zero gradient row, the other rows as if it were ignored, flag raised
    bad = y.clone()
    bad[0] = 7
    scores.grad = None
    crit(scores, bad).backward()
    assert (scores.grad[0] == 0).all()
    y2 = y.clone()
    y2[0] = -100
    ref_s.grad = None
    torch.nn.CrossEntropyLoss()(ref_s, y2.cpu()).backward()
    assert torch.allclose(scores.grad.cpu(), ref_s.grad, atol=1e-7)
    with pytest.raises(EavError, match="outside"):
        crit.check()


def test_scores_returned_by_eegnet_must_not_be_edited_in_place_before_backward():
    from eav_amd._lib import EavError
    from eav_amd.eegnet import EEGNet_tor
    from eav_amd.optim import CrossEntropyLoss
    torch.manual_seed(0)
    m = EEGNet_tor(5, Chans=30, Samples=500, dropoutRate=0.0).cuda().train()
    x = torch.randn(4, 1, 30, 500, device="cuda")
    y = torch.tensor([0, 1, 2, 3], device="cuda")
    s = m(x)
    loss = CrossEntropyLoss()(s, y)
    with torch.no_grad():
        s.clamp_(min=1e-3)                      # aliases the probabilities the backward reads
    with pytest.raises(EavError, match="modified in place"):
        loss.backward()


def test_workspace_cache_is_bounded_for_eager_batch_sizes():
    from eav_amd.eegnet import EEGNet_tor
    torch.manual_seed(0)
    m = EEGNet_tor(5, Chans=30, Samples=500).cuda().eval()
    with torch.no_grad():
        for B in (1, 2, 3, 5, 7, 2, 9):
            m(torch.randn(B, 1, 30, 500, device="cuda"))
    assert len(m._wss) <= 2, list(m._wss)


# ---------------------------------------------------------------------------------------------- firstConv by FFT
# csrc/eegnet_fir_fft.hip: the same exact-fp32 linear maps as test_fir_fwd / test_fir_wgrad (nn.Conv2d(1, 8, (1, K),
# padding='same'), EEGNet_tor.py:24,51 and its weight gradient) evaluated by overlap-save FFT blocks - SAME tolerances.
# Shapes: the bench shape, one block exactly (704), one sample more, odd electrode counts (last pair half empty),
# recordings shorter than a block, the longest supported kernel (321 taps), a short kernel.
@pytest.mark.parametrize("B,C,S,K", [(1, 30, 10000, 300), (2, 30, 500, 300), (2, 5, 704, 300), (3, 3, 705, 300),
                                     (2, 1, 1409, 321), (1, 7, 2100, 64), (2, 32, 130, 7)])
@pytest.mark.parametrize("indexed", [False, True])
def test_fir_fwd_fft(L, B, C, S, K, indexed):
    N = B + 3 if indexed else B
    xs = synth.normal(1, (N, C, S))
    idx = np.array([(3 * i + 1) % N for i in range(B)], np.int64)
    x = xs[idx] if indexed else xs
    w = synth.uniform(2, (8, K), -0.1, 0.1)
    xd, wd = dev(xs), dev(w)
    y = torch.full((B, 8, C, S), float("nan"), device="cuda")
    npart = L.plain("eav_eegnet_fir_fwd_fft_nparts", B, C, S)
    part = torch.zeros(npart, 16, device="cuda")
    L.call("eav_eegnet_fir_fwd_fft", xd.data_ptr(), dev(idx).data_ptr() if indexed else None, wd.data_ptr(), y.data_ptr(),
           part.data_ptr(), B, C, S, K, None)
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).unsqueeze(1)
    left = (K - 1) // 2
    ref = F.conv2d(F.pad(xt, (left, K - 1 - left)), torch.from_numpy(w).view(8, 1, 1, K))
    close(y, ref, 1e-4, 2e-5, "y1")
    st = part.sum(0).cpu().double().numpy()
    close(st[:8], ref.double().sum((0, 2, 3)), 1e-5, 1e-3, "sum")
    close(st[8:], (ref.double() ** 2).sum((0, 2, 3)), 1e-5, 1e-3, "sumsq")
    # against float64 the FFT form must not be worse than the direct fp32 MFMA kernel
    if K <= 300:
        y2 = torch.empty(B, 8, C, S, device="cuda")
        part2 = torch.zeros(L.plain("eav_eegnet_fir_fwd_nparts", B, C, S), 16, device="cuda")
        L.call("eav_eegnet_fir_fwd", dev(x).data_ptr(), wd.data_ptr(), y2.data_ptr(), part2.data_ptr(), B, C, S, K, None)
        ref64 = F.conv2d(F.pad(xt.double(), (left, K - 1 - left)), torch.from_numpy(w).double().view(8, 1, 1, K))
        e_fft = float((y.double().cpu() - ref64).abs().max())
        e_dir = float((y2.double().cpu() - ref64).abs().max())
        print(f"fir_fwd_fft B={B} C={C} S={S} K={K}: max error vs float64 {e_fft:.2e} (direct fp32 MFMA kernel {e_dir:.2e})")
        assert e_fft <= 4.0 * e_dir + 1e-6


@pytest.mark.parametrize("B,C,S,K", [(1, 30, 10000, 300), (2, 30, 500, 300), (2, 4, 704, 300), (2, 3, 705, 300),
                                     (3, 1, 1409, 321), (1, 3, 1100, 64)])
@pytest.mark.parametrize("eval_mode", [False, True])
def test_fir_wgrad_fft(L, B, C, S, K, eval_mode):
    x = synth.normal(3, (B, C, S))
    y1 = synth.normal(4, (B, 8, C, S))
    g1 = synth.normal(5, (B, 8, C, S))
    mean, invstd = synth.uniform(6, (8,), -0.2, 0.2), synth.uniform(7, (8,), 0.5, 2.0)
    scale = synth.uniform(8, (8,), 0.5, 1.5)
    m1, m2 = synth.uniform(9, (8,), -0.1, 0.1), synth.uniform(10, (8,), -0.1, 0.1)
    if eval_mode:
        m1, m2 = np.zeros(8, np.float32), np.zeros(8, np.float32)
    bn = bn_buf(8, mean, invstd, scale, np.zeros(8, np.float32), m1, m2)
    ws = torch.empty(L.plain("eav_eegnet_fir_wgrad_fft_ws_floats", B, C, S), device="cuda")
    out = torch.full((8, K), float("nan"), device="cuda")
    L.call("eav_eegnet_fir_wgrad_fft", dev(x).data_ptr(), None, None if eval_mode else dev(y1).data_ptr(),
           dev(g1).data_ptr(), bn.data_ptr(), ws.data_ptr(), out.data_ptr(), B, C, S, K, None)
    out2 = torch.full((8, K), float("nan"), device="cuda")
    L.call("eav_eegnet_fir_wgrad_fft", dev(x).data_ptr(), None, None if eval_mode else dev(y1).data_ptr(),
           dev(g1).data_ptr(), bn.data_ptr(), ws.data_ptr(), out2.data_ptr(), B, C, S, K, None)
    torch.cuda.synchronize()
    assert torch.equal(out, out2), "the FFT weight gradient is not bit-reproducible"
    bc = lambda v: torch.from_numpy(v).double()[None, :, None, None]  # noqa: E731
    dy = bc(scale) * (torch.from_numpy(g1).double() - bc(m1) - (torch.from_numpy(y1).double() - bc(mean)) * bc(invstd) * bc(m2))
    left = (K - 1) // 2
    xp = F.pad(torch.from_numpy(x).double(), (left, K - 1 - left))
    win = xp.unfold(2, S, 1)
    ref = torch.einsum("bfcs,bcks->fk", dy, win)
    close(out, ref, 2e-4, 2e-4 * float(ref.abs().max()), "dW1")
    err = float((out.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    print(f"fir_wgrad_fft B={B} C={C} S={S} K={K} eval={eval_mode}: max error / max |dW| = {err:.2e}")


# ---------------------------------------------------------------------------------------------- separableConv by FFT
# csrc/eegnet_conv64_fft.hip: the same maps as test_conv64 (forward + BatchNorm sums, data gradient) in the frequency domain -
# SAME tolerances.  T: the bench shape (2500 = 52 blocks of 49), the reference's own (125), one block (49), one block + 1, an
# odd number of blocks (3 x 49: the last pair is half empty), a batch whose column count is not a multiple of 32.
@pytest.mark.parametrize("B,T", [(1, 2500), (2, 125), (3, 49), (2, 50), (2, 147), (5, 300)])
def test_conv64_fft(L, B, T):
    x = synth.normal(31, (B, 64, T))
    w = synth.uniform(32, (64, 64, 16), -0.05, 0.05)
    wd, xd = dev(w), dev(x)
    ws = torch.zeros(L.plain("eav_conv64_fft_ws_floats", B, T), device="cuda")
    nt = L.plain("eav_conv64_fft_nparts", B, T)
    out = torch.full((B, 64, T), float("nan"), device="cuda")
    part = torch.zeros(nt, 128, device="cuda")
    L.call("eav_conv64_fft_fwd", xd.data_ptr(), wd.data_ptr(), out.data_ptr(), part.data_ptr(), ws.data_ptr(), B, T, 0, None)
    torch.cuda.synchronize()
    xt = torch.from_numpy(x).double().requires_grad_(True)
    wt = torch.from_numpy(w).double().requires_grad_(True)
    ref = F.conv1d(F.pad(xt, (7, 8)), wt)
    close(out, ref, 1e-4, 1e-5, "conv out")
    st = part.sum(0).cpu().double().numpy()
    close(st[:64], ref.detach().sum((0, 2)), 1e-4, 1e-3, "sum")
    close(st[64:], (ref.detach() ** 2).sum((0, 2)), 1e-4, 1e-3, "sumsq")
    du = synth.normal(33, (B, 64, T))
    ref.backward(torch.from_numpy(du).double())
    dx = torch.full((B, 64, T), float("nan"), device="cuda")
    # (bwd = 2: the data gradient on the filter spectra the forward call left in ws - the model's sequence; it must leave the
    # forward's input spectra alone: the weight gradient below reads them)
    L.call("eav_conv64_fft_fwd", dev(du).data_ptr(), wd.data_ptr(), dx.data_ptr(), None, ws.data_ptr(), B, T, 2, None)
    dx1 = torch.full((B, 64, T), float("nan"), device="cuda")
    ws1 = torch.zeros_like(ws)
    L.call("eav_conv64_fft_fwd", dev(du).data_ptr(), wd.data_ptr(), dx1.data_ptr(), None, ws1.data_ptr(), B, T, 1, None)
    torch.cuda.synchronize()
    close(dx, xt.grad, 1e-4, 1e-5, "dgrad")
    assert torch.equal(dx, dx1), "bwd = 1 (own filter spectra) and bwd = 2 (the forward's) differ"
    dw = torch.full((64, 64, 16), float("nan"), device="cuda")
    L.call("eav_conv64_fft_wgrad", dev(du).data_ptr(), dw.data_ptr(), ws.data_ptr(), B, T, None)
    dw2 = torch.full((64, 64, 16), float("nan"), device="cuda")
    L.call("eav_conv64_fft_wgrad", dev(du).data_ptr(), dw2.data_ptr(), ws.data_ptr(), B, T, None)
    torch.cuda.synchronize()
    close(dw, wt.grad, 1e-4, 1e-4 * float(wt.grad.abs().max()), "wgrad")
    assert torch.equal(dw, dw2), "the frequency-domain weight gradient is not bit-reproducible"
    print(f"conv64_fft wgrad B={B} T={T}: max error / max |dW| = "
          f"{float((dw.double().cpu() - wt.grad).abs().max() / wt.grad.abs().max()):.2e}")
    # against float64, beside the direct fp32 MFMA kernel
    wTf, wTb = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
    L.call("eav_conv64_prep_weights", wd.data_ptr(), wTf.data_ptr(), wTb.data_ptr(), None)
    out2 = torch.empty(B, 64, T, device="cuda")
    L.call("eav_conv64_fwd", xd.data_ptr(), wTf.data_ptr(), out2.data_ptr(), None, B, T, 7, None)
    e_fft = float((out.double().cpu() - ref.detach()).abs().max())
    e_dir = float((out2.double().cpu() - ref.detach()).abs().max())
    print(f"conv64_fft B={B} T={T}: max error vs float64 {e_fft:.2e} (direct fp32 MFMA kernel {e_dir:.2e})")
    assert e_fft <= 4.0 * e_dir + 1e-6


# ----------------------------------------------------------------------------------------------------------------------
# Launch mergers of the launch-bound shapes: same results as the separate launches they replace.
def test_renorm_rows2_is_the_two_hooks_in_one_launch(L):
    a, b = synth.uniform(52, (64, 30), -0.5, 0.5), synth.uniform(53, (5, 960), -0.1, 0.1)
    a[::2] *= 0.1
    b[1] *= 0.01
    a1, b1, a2, b2 = dev(a), dev(b), dev(a), dev(b)
    L.call("eav_renorm_rows", a1.data_ptr(), 64, 30, 0.25, None)
    L.call("eav_renorm_rows", b1.data_ptr(), 5, 960, 0.25, None)
    L.call("eav_renorm_rows2", a2.data_ptr(), 64, 30, b2.data_ptr(), 5, 960, 0.25, None)
    torch.cuda.synchronize()
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    assert not torch.equal(a1.cpu(), torch.from_numpy(a)) and not torch.equal(b1.cpu(), torch.from_numpy(b))


def test_step_prologue_is_prep_weights_plus_the_step_counters(L):
    w = dev(synth.normal(54, (64, 64, 16)))
    ref_f, ref_b = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
    got_f, got_b = torch.empty(1024, 64, device="cuda"), torch.empty(1024, 64, device="cuda")
    cnt = torch.tensor([5, 0, 41, 7], dtype=torch.int64, device="cuda")
    c = cnt.data_ptr()
    L.call("eav_conv64_prep_weights", w.data_ptr(), ref_f.data_ptr(), ref_b.data_ptr(), None)
    for _ in range(3):
        L.call("eav_eegnet_step_prologue", w.data_ptr(), got_f.data_ptr(), got_b.data_ptr(), c, None, c + 16, c + 24, None)
    torch.cuda.synchronize()
    assert torch.equal(got_f, ref_f) and torch.equal(got_b, ref_b)
    assert cnt.tolist() == [8, 0, 44, 10]
    assert torch.equal(ref_f.view(64, 16, 64), w.permute(1, 2, 0))        # wT_fwd[(i*16+k)][o] = W[o][i][k]


def test_step_begin_is_the_counters_plus_the_label_gather(L):
    """eav_step_begin (GraphStep's first node): five distinct counters + 1, NULL ones skipped, labels gathered - what
    eav_counter_inc4 + eav_counter_inc + eav_gather_i64 did in three launches."""
    cnt = torch.tensor([5, 0, 41, 7, 100, 9], dtype=torch.int64, device="cuda")
    c = cnt.data_ptr()
    labels = torch.arange(200, dtype=torch.int64, device="cuda") % 5
    idx = torch.randperm(200, generator=torch.Generator().manual_seed(3))[:64].cuda()
    out = torch.full((64,), -1, dtype=torch.int64, device="cuda")
    for _ in range(3):
        L.call("eav_step_begin", c, None, c + 16, c + 24, c + 32, labels.data_ptr(), idx.data_ptr(), out.data_ptr(), 64, None)
    L.call("eav_step_begin", None, None, None, None, c + 40, None, None, None, 0, None)       # counters only
    torch.cuda.synchronize()
    assert cnt.tolist() == [8, 0, 44, 10, 103, 10]
    assert torch.equal(out, labels[idx])
    ref = torch.empty_like(out)
    L.call("eav_gather_i64", labels.data_ptr(), idx.data_ptr(), ref.data_ptr(), 64, None)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    out2 = torch.full((300,), -1, dtype=torch.int64, device="cuda")                              # more than one block of labels
    idx2 = torch.randint(0, 200, (300,), generator=torch.Generator().manual_seed(4)).cuda()
    L.call("eav_step_begin", None, None, None, None, None, labels.data_ptr(), idx2.data_ptr(), out2.data_ptr(), 300, None)
    torch.cuda.synchronize()
    assert torch.equal(out2, labels[idx2])


@pytest.mark.parametrize("nparts,n,two", [(640, 1920, False), (640, 1920, True), (3, 448, True), (20, 64 * 30, False),
                                           (640, 64 * 128, True), (7, 450, False)])
def test_reduce_and_bn_bwd_finalize_is_the_separate_launches(L, nparts, n, two):
    """eav_reduce_and_bn_bwd_finalize (one graph node behind eav_eegnet_dw_bwd_fused) against eav_reduce_partials + one / two
    eav_bn_bwd_finalize: the same bits - also for the shapes it hands to the separate kernels itself (n >= 8192, n % 4 != 0)."""
    part = dev(synth.normal(71, (nparts, n)))
    pa, pb = dev(synth.normal(72, (nparts, 16))), dev(synth.normal(73, (nparts * 2, 128)))
    ref_out = torch.empty(n, device="cuda")
    refs = [[torch.empty(ch, device="cuda") for _ in range(4)] for ch in (8, 64)]
    L.call("eav_reduce_partials", part.data_ptr(), nparts, n, n, 1.0, ref_out.data_ptr(), None)
    L.call("eav_bn_bwd_finalize", pa.data_ptr(), nparts, 8, 1234.0, 1, *[t.data_ptr() for t in refs[0]], None)
    L.call("eav_bn_bwd_finalize", pb.data_ptr(), 2 * nparts, 64, 77.0, 0, *[t.data_ptr() for t in refs[1]], None)
    out = torch.full((n,), float("nan"), device="cuda")
    got = [[torch.full((ch,), float("nan"), device="cuda") for _ in range(4)] for ch in (8, 64)]
    jb = (pb.data_ptr(), 2 * nparts, 64, 77.0, 0, *[t.data_ptr() for t in got[1]]) if two else \
        (None, 0, 0, 0.0, 0, None, None, None, None)
    L.call("eav_reduce_and_bn_bwd_finalize", part.data_ptr(), nparts, n, n, out.data_ptr(), pa.data_ptr(), nparts, 8, 1234.0, 1,
           *[t.data_ptr() for t in got[0]], *jb, None)
    torch.cuda.synchronize()
    assert torch.equal(out, ref_out)
    for a, b in zip(got[0], refs[0]):
        assert torch.equal(a, b)
    if two:
        for a, b in zip(got[1], refs[1]):
            assert torch.equal(a, b)
        assert float(got[1][2].abs().max()) == 0.0          # training = 0: m1 = m2 = 0
    close(out, part.double().sum(0), 1e-5, 1e-5, "reduced partials")


@pytest.mark.parametrize("B,C,S,drop", [(2, 30, 500, 0.5), (3, 30, 200, 0.25), (2, 7, 132, 0.0), (1, 30, 2052, 0.5),
                                        (2, 3, 512, -0.5), (2, 30, 256, 0.5), (1, 5, 516, 0.0)])
def test_dw_bwd_fused_matches_apply_then_dw_bwd(L, B, C, S, drop):
    """eav_eegnet_dw_bwd_fused (dz formed in the prologue; two / four channel groups for S <= 512 / 256) against
    eav_bn_elu_pool_bwd_apply -> eav_eegnet_dw_bwd (checked against torch autograd above)."""
    y1, z = synth.normal(21, (B, 8, C, S)), synth.normal(22, (B, 64, S))
    dp2 = synth.normal(23, (B, 64, S // 4))
    w2 = synth.uniform(24, (64, C), -0.3, 0.3)

    def bn(seed, nch):
        mean, invstd = synth.uniform(seed, (nch,), -0.2, 0.2), synth.uniform(seed + 1, (nch,), 0.5, 2.0)
        gamma, beta = synth.uniform(seed + 2, (nch,), 0.5, 1.5), synth.uniform(seed + 3, (nch,), -0.2, 0.2)
        m1, m2 = synth.uniform(seed + 4, (nch,), -0.01, 0.01), synth.uniform(seed + 5, (nch,), -0.01, 0.01)
        return bn_buf(nch, mean, invstd, gamma * invstd, beta - mean * gamma * invstd, m1, m2)

    b1, b2 = bn(30, 8), bn(40, 64)
    nchunk = (S + 1023) // 1024
    y1d, zd, dpd, w2d = dev(y1), dev(z), dev(dp2), dev(w2)
    dz = torch.empty(B, 64, S, device="cuda")
    L.call("eav_bn_elu_pool_bwd_apply", dpd.data_ptr(), zd.data_ptr(), b2.data_ptr(), b2.data_ptr() + 4 * 256,
           dz.data_ptr(), B, 64, S, 4, drop, 77, None, None, None)
    outs = []
    for fused in (False, True):
        g1 = torch.full((B, 8, C, S), float("nan"), device="cuda")
        pst = torch.full((B * nchunk, 16), float("nan"), device="cuda")
        pw = torch.full((B * nchunk, 64 * C), float("nan"), device="cuda")
        if fused:
            L.call("eav_eegnet_dw_bwd_fused", y1d.data_ptr(), zd.data_ptr(), dpd.data_ptr(), b2.data_ptr(), b1.data_ptr(),
                   w2d.data_ptr(), g1.data_ptr(), pst.data_ptr(), pw.data_ptr(), B, C, S, drop, 77, None, None, None)
        else:
            L.call("eav_eegnet_dw_bwd", y1d.data_ptr(), dz.data_ptr(), b1.data_ptr(), w2d.data_ptr(), g1.data_ptr(),
                   pst.data_ptr(), pw.data_ptr(), B, C, S, None)
        torch.cuda.synchronize()
        outs.append((g1, pst.double().sum(0), pw.double().sum(0)))
    (ga, sa, wa), (gb, sb, wb) = outs
    assert torch.isfinite(gb).all()
    close(gb, ga, 1e-5, 1e-6 * float(ga.abs().max()), "g1")
    close(sb, sa, 1e-4, 1e-5 * float(sa.abs().max()), "sum g, sum g*yhat")
    close(wb, wa, 1e-4, 1e-5 * float(wa.abs().max()), "dW2")
    # eval-mode step: depthwiseBN on its running statistics (m1 = m2 = 0 whatever the slots hold) and its two sums from the
    # same pass - against the fused kernel on a buffer with zeroed m1 / m2 and against eav_bn_elu_pool_bwd_reduce
    b2z = b2.clone()
    b2z[4 * 64:] = 0.0
    g1z = torch.empty(B, 8, C, S, device="cuda")
    pstz, pwz = torch.empty(B * nchunk, 16, device="cuda"), torch.empty(B * nchunk, 64 * C, device="cuda")
    L.call("eav_eegnet_dw_bwd_fused", y1d.data_ptr(), zd.data_ptr(), dpd.data_ptr(), b2z.data_ptr(), b1.data_ptr(),
           w2d.data_ptr(), g1z.data_ptr(), pstz.data_ptr(), pwz.data_ptr(), B, C, S, drop, 77, None, None, None)
    g1e = torch.full((B, 8, C, S), float("nan"), device="cuda")
    pste, pwe = torch.full((B * nchunk, 16), float("nan"), device="cuda"), torch.full((B * nchunk, 64 * C), float("nan"), device="cuda")
    p2e = torch.full((B * nchunk, 128), float("nan"), device="cuda")
    L.call("eav_eegnet_dw_bwd_fused_eval", y1d.data_ptr(), zd.data_ptr(), dpd.data_ptr(), b2.data_ptr(), b1.data_ptr(),
           w2d.data_ptr(), g1e.data_ptr(), pste.data_ptr(), pwe.data_ptr(), p2e.data_ptr(), B, C, S, drop, 77, None, None, None)
    pr = torch.empty(B, 128, device="cuda")
    L.call("eav_bn_elu_pool_bwd_reduce", dpd.data_ptr(), zd.data_ptr(), b2.data_ptr(), pr.data_ptr(), B, 64, S, 4, drop, 77,
           None, None, None)
    torch.cuda.synchronize()
    assert torch.equal(g1e, g1z) and torch.equal(pste, pstz) and torch.equal(pwe, pwz)
    got, want = p2e.double().sum(0), pr.double().sum(0)
    close(got, want, 1e-4, 1e-5 * float(want.abs().max()), "depthwiseBN sums from the depthwise pass")
