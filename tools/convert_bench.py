#!/usr/bin/env python3
"""eav_sp_convert rate against the resident-block cap (eav_sp_set_convert_blocks).  Run on the GPU box."""
import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd import _lib
from tools.gemm_sp_bench import P, kpad, timeit
_lib.load()
raw = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "eav_amd", "libeav_hip.so"))
for R, C in ((9712, 768), (9712, 3072), (25216, 768), (25216, 3072), (25216, 2304)):
    x = torch.randn(R, C, device="cuda")
    slot = torch.zeros(4128, device="cuda")
    _lib.call("eav_sp_absmax", P(x), R, C, C, P(slot), None)
    d = torch.empty(R, 2 * kpad(C), dtype=torch.float16, device="cuda")
    dT = torch.empty(C, 2 * kpad(R), dtype=torch.float16, device="cuda")
    part = torch.empty(_lib.plain("eav_sp_convert_colsum_nparts", R), C, device="cuda")
    gb = R * C * 4 / 1e9
    line = f"[{R},{C}]"
    for cap in (0, 256, 512, 1024, 2048, 4096):
        raw.eav_sp_set_convert_blocks(cap)
        ms1 = timeit(lambda: _lib.call("eav_sp_convert", P(x), R, C, C, P(slot), P(d), None, None))
        ms2 = timeit(lambda: _lib.call("eav_sp_convert", P(x), R, C, C, P(slot), P(d), P(dT), None))
        ms3 = timeit(lambda: _lib.call("eav_sp_convert_colsum", P(x), R, C, C, P(slot), P(d), P(dT), P(part), None))
        line += f" | cap {cap or 'all'}: {ms1*1e3:.0f}/{ms2*1e3:.0f}/{ms3*1e3:.0f} us ({2*gb/ms1:.1f}/{3*gb/ms2:.1f} TB/s)"
    raw.eav_sp_set_convert_blocks(512)
    print(line)
