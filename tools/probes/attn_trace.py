#!/usr/bin/env python3
"""Workgroup timeline of the pipelined attention forward (library built with -DATTN_TRACE, EAV_LIB_PATH): start / end wall
clock (100 MHz) and CU of every workgroup at the AST shape; prints occupancy over time and per-CU schedules."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from eav_amd import _lib  # noqa: E402
from attn_sp_check import P, prep  # noqa: E402

_lib.load()
B, N, H, D = 8, 1214, 12, 768
qkv = torch.randn(B * N, 3 * D, device="cuda")
s_qkv, rowp, tp = prep(qkv, B, N, 3 * D, D, 7)
ao = torch.empty(B * N, D, device="cuda")
RW = int(os.environ.get("ROWS_PER_WG", "128"))
nwg = (N + RW - 1) // RW * B * H
PP = False          # ping-pong kernel: two trace records per workgroup (waves 0 and 4)
lse = torch.zeros(B * H * N + 8 + 32 * nwg, device="cuda")
for _ in range(3):
    _lib.call("eav_attn_fwd_sp", P(rowp), P(tp), P(s_qkv), P(ao), P(lse), None, B, H, N, 64, 0.125, None)
torch.cuda.synchronize()
off = (B * H * N + 3) & ~3
tr = lse[off:off + 32 * nwg].view(torch.int64).view(2 * nwg, 8).cpu().numpy()
tr4 = tr[1::2] if PP else None
tr = tr[0::2] if PP else tr[:nwg]
t0 = tr[:, 0].min()
st, en = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0          # microseconds
hw, xcc = tr[:, 2] & 0xffffffff, tr[:, 3] & 0xf
ph = [tr[:, 2] >> 32, tr[:, 3] >> 32, tr[:, 4] & 0xffffffff, tr[:, 4] >> 32, tr[:, 5] & 0xffffffff, tr[:, 5] >> 32]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
print(f"{nwg} workgroups, span {en.max():.1f} us; duration min / mean / max {(en - st).min():.1f} / "
      f"{(en - st).mean():.1f} / {(en - st).max():.1f} us; distinct CUs {len(set(cu.tolist()))}")
for t in range(0, int(en.max()) + 1, 5):
    live = ((st <= t) & (en > t)).sum()
    print(f"  t = {t:4d} us: {live:4d} workgroups resident")
per = collections.defaultdict(list)
for i in range(nwg):
    per[int(cu[i])].append((float(st[i]), float(en[i]), i))
for k in list(sorted(per))[:6]:
    print("  CU", hex(k), " ".join(f"[{a:.0f}-{b:.0f} wg{i}]" for a, b, i in sorted(per[k])))
cnt = collections.Counter(len(v) for v in per.values())
print("  workgroups per CU:", dict(sorted(cnt.items())))

import numpy as np  # noqa: E402
names = ["wait + barrier (even)", "M phase", "barrier (odd)", "V phase", "-"] if PP else ["rebase + loop", "top (wait, barrier, DMA issue)", "head + A", "B", "C", "D"]
dur = en - st
order = np.argsort(dur)
for label, sel in (("fastest 10 %", order[:nwg // 10]), ("slowest 10 %", order[-(nwg // 10):]), ("all", order)):
    tot = sum(p[sel].mean() for p in ph)
    print(f"  wave 0 of the {label} workgroups ({dur[sel].mean():.1f} us): cycles per tile " +
          ", ".join(f"{n} {p[sel].mean() / 38:.0f}" for n, p in zip(names, ph)) + f"; sum {tot / 38:.0f}")

if PP:
    nm = ["wait + barrier (even)", "PV MFMAs -> end of M", "barrier (odd)", "split -> end of V", "max + rebase", "K fragment reads", "exponentials",
          "V fragment reads", "score MFMAs"]
    for lab, t_ in (("wave 0 (leading)", tr), ("wave 4 (trailing)", tr4)):
        p_ = [t_[:, 2] >> 32, t_[:, 3] >> 32, t_[:, 4] & 0xffffffff, t_[:, 4] >> 32, t_[:, 5] & 0xffffffff, t_[:, 5] >> 32,
              t_[:, 6] & 0xffffffff, t_[:, 6] >> 32, t_[:, 7] & 0xffffffff]
        print(f"  {lab}: cycles per tile: " + ", ".join(f"{n} {q.mean() / 38:.0f}" for n, q in zip(nm, p_)))
