#!/usr/bin/env python3
"""Where the encoder step goes at the per-rank batch sizes of a strong-scaling run: wall time per step, the CPU time the
launch loop itself takes (step issued without waiting, then synchronised), and the number of library launches.
    python tools/small_batch_profile.py vit 16 [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eav_amd import _lib, synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam  # noqa: E402


def main():
    kind, B = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = T.Encoder(T.make_config(kind)).to(dev).train()
    x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
    x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    opt = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True)
    crit = CrossEntropyLoss()
    ncalls = [0]
    orig = _lib.call

    def counting(name, *a):
        ncalls[0] += 1
        return orig(name, *a)

    def step():
        opt.zero_grad()
        crit(model(x).logits, y).backward()
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    _lib.call = counting
    T._lib.call = counting
    step()
    torch.cuda.synchronize()
    _lib.call = orig
    T._lib.call = orig
    if os.environ.get("CPROFILE"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(steps):
            step()
        pr.disable()
        torch.cuda.synchronize()
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    print(f"{kind} B={B}: {t_all / steps * 1e3:.3f} ms per step wall, {t_issue / steps * 1e3:.3f} ms of it issuing from the host "
          f"({ncalls[0]} library launches per step)")


if __name__ == "__main__":
    main()
