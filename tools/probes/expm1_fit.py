"""Coefficients and error of the branch-free expm1 used by elu_f (csrc/eav_common.h) for arguments in [-104, 0].

    n = rint(x log2 e); r = x - n ln2_hi - n ln2_lo; p = r + r^2 q(r); s = 2^n; expm1(x) = fma(p, s, s - 1)

q = weighted least-squares fit (Chebyshev nodes, weight |r|: the relative error of p) of (expm1(r) - r) / r^2 on
[-ln2/2, ln2/2], rounded to fp32.  The evaluation is emulated in numpy fp32 (fma = float64 product-sum rounded once) and
compared with float64 expm1 on a dense grid; prints the max error in ulps of the fp32 result.  Run on CPU."""
import numpy as np

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def fit(deg):
    k = np.arange(4001)
    r = 0.5 * np.log(2.0) * 1.0001 * np.cos(np.pi * (k + 0.5) / 4001)
    q = np.where(np.abs(r) > 1e-6, (np.expm1(r) - r) / (r * r), 0.5 + r / 6)
    w = np.abs(r)
    V = np.vander(r, deg + 1, increasing=True)
    c = np.linalg.lstsq(V * w[:, None], q * w, rcond=None)[0]
    # a few Remez-style reweighting rounds (Lawson) toward the minimax solution
    lw = np.ones_like(r)
    for _ in range(60):
        c = np.linalg.lstsq(V * (w * lw)[:, None], q * w * lw, rcond=None)[0]
        err = np.abs((V @ c - q) * w)
        lw = lw * (0.5 + err / err.max())
        lw /= lw.mean()
    return c.astype(f32)


def expm1_f32(x, c):
    """The instruction sequence of elu_f: 13 VALU operations, no branch, no conversion."""
    x = np.maximum(x.astype(f32), f32(-17.5))                 # (v_med3_f32(v, -17.5, 0) in the kernel: expm1 = -1 below)
    L2E, LN2H, LN2L = f32(1.4426950408889634), f32(0.693145751953125), f32(1.42860682030941723212e-6)
    MAGIC = f32(12582912.0)                                   # 1.5 * 2^23: the sum's low mantissa bits = rint(x log2 e)
    t = fma(x, np.full_like(x, L2E), np.full_like(x, MAGIC))
    n = (t - MAGIC).astype(f32)
    r = fma(-n, np.full_like(x, LN2H), x)
    r = fma(-n, np.full_like(x, LN2L), r)
    q = np.full_like(x, c[-1])
    for ck in c[-2::-1]:
        q = fma(q, r, np.full_like(x, ck))
    r2 = (r * r).astype(f32)
    p = fma(r2, q, r)
    sbits = ((t.view(np.uint32) << np.uint32(23)) + np.uint32(0x3f800000)).astype(np.uint32)   # v_lshl_add_u32
    s = sbits.view(f32)
    assert np.array_equal(s, np.ldexp(f32(1), n.astype(np.int32)).astype(f32))
    return fma(p, s, (s - f32(1)).astype(f32))


def ulps(got, ref):
    ref32 = ref.astype(f32)
    u = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref) / u


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    xs = np.concatenate([-np.logspace(-30, np.log10(104.0), 2_000_000), -rng.uniform(0, 20, 4_000_000),
                         -rng.uniform(0, 1, 2_000_000), np.array([0.0, -104.0, -88.0, -17.0])]).astype(f32)
    ref = np.expm1(xs.astype(np.float64))
    for deg in (4, 5):
        c = fit(deg)
        u = ulps(expm1_f32(xs, c), ref)
        print(f"q degree {deg}: coefficients", ", ".join(f"{float(v):.9e}f" for v in c))
        print(f"   max error {u.max():.3f} ulp at x = {xs[u.argmax()]!r}; mean {u.mean():.3f}; share > 1 ulp {np.mean(u > 1):.2e}")
