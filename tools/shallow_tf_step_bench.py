"""Step time of ShallowConvNet + 12-layer transformer (eav_amd/transformer_eeg.py) at the reference's batch shape
[32,1,30,500]: eager and hipGraph replay."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eav_amd.eegnet import GraphStep  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402
from eav_amd.transformer_eeg import ShallowConvNet  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
m = ShallowConvNet(5).cuda().train()
opt, crit = FusedAdam(m.parameters(), lr=1e-3, capturable=True), CrossEntropyLoss()
xs = torch.randn(4 * B, 1, 30, 500, device="cuda")
ys = torch.randint(0, 5, (4 * B,), device="cuda")
x, y = xs[:B], ys[:B]


def step():
    loss = crit(m(x), y)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
print(f"eager  B={B}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} samples/s")
gs = GraphStep(m, opt, crit, xs, ys, B)
idx = list(range(B))
for _ in range(4):
    gs.run(idx)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(50):
    gs.run(idx)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 50
print(f"graph  B={B}: {dt * 1e3:.3f} ms/step, {B / dt:.0f} samples/s")
