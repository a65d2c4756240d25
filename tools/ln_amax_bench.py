import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib
_lib.load(); P = _lib.ptr
for M in (9712, 25216):
    D = 768
    x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    y = torch.empty_like(x); mean = torch.empty(M, device="cuda"); rstd = torch.empty(M, device="cuda")
    slot = torch.zeros(4128, device="cuda")
    def t(f, n=20):
        for _ in range(3): f()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize()
        return a.elapsed_time(e) / n * 1e3
    t0 = t(lambda: _lib.call("eav_layernorm_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-12, None))
    t1 = t(lambda: _lib.call("eav_layernorm_fwd_amax", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-12, P(slot), None))
    print(f"M={M}: layernorm_fwd {t0:.1f} us, with amax {t1:.1f} us")
    dy = torch.randn(M, D, device="cuda"); dx = torch.zeros(M, D, device="cuda")
    npart = _lib.plain("eav_layernorm_bwd_nparts", M)
    part = torch.empty(npart, 2 * D, device="cuda")
    _lib.call("eav_layernorm_fwd", P(x), P(g), P(b), P(y), P(mean), P(rstd), M, D, 1e-12, None)
    t2 = t(lambda: _lib.call("eav_layernorm_bwd_amax", P(dy), P(x), P(g), P(mean), P(rstd), P(dx), 1, P(part), M, D, P(slot), None))
    print(f"M={M}: layernorm_bwd (accumulate, amax) {t2:.1f} us = {4 * M * D * 4 / t2 / 1e6:.2f} TB/s")
    t3 = t(lambda: _lib.call("eav_layernorm_bwd", P(dy), P(x), P(g), P(mean), P(rstd), P(dx), 1, P(part), M, D, None))
    t4 = t(lambda: _lib.call("eav_layernorm_bwd", P(dy), P(x), P(g), P(mean), P(rstd), P(dx), 0, P(part), M, D, None))
    print(f"M={M}: layernorm_bwd without the maxima {t3:.1f} us = {4 * M * D * 4 / t3 / 1e6:.2f} TB/s; not accumulating "
          f"{t4:.1f} us = {3 * M * D * 4 / t4 / 1e6:.2f} TB/s")
