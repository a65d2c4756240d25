"""Eager step time of the canonical EEGNet (eav_amd/cnn_eeg.py) at three shapes; argv[1] selects one (for rocprofv3)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd.cnn_eeg import EEGNet  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402
SHAPES = [(64, 30, 10000, 64), (32, 30, 500, 64), (32, 64, 128, 64)]
if len(sys.argv) > 1:
    SHAPES = [SHAPES[int(sys.argv[1])]]
for (B, C, S, K) in SHAPES:
    torch.manual_seed(0)
    m = EEGNet(5, Chans=C, Samples=S, kernLength=K).cuda().train()
    opt, crit = FusedAdam(m.parameters(), lr=1e-3), CrossEntropyLoss()
    x = torch.randn(B, C, S, device="cuda"); y = torch.randint(0, 5, (B,), device="cuda")
    def step():
        l = crit(m(x), y); opt.zero_grad(); l.backward(); opt.step()
    for _ in range(3): step()
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / n
    print(f"B={B} C={C} S={S} K={K}: {dt*1e3:.3f} ms/step, {B/dt:.0f} samples/s")
