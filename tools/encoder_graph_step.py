#!/usr/bin/env python3
"""Encoder train step at small per-rank batches: eager two-stream (default) vs eager single-stream vs ONE hipGraph replay of
the single-stream step (forward, CE, backward, AdamW).  Prints ms per step and checks that the replayed losses equal the
eager single-stream ones bit for bit.
    python tools/encoder_graph_step.py vit 16 [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eav_amd import synth, transformer as T  # noqa: E402
from eav_amd.optim import CrossEntropyLoss, FusedAdam, unit_gradient  # noqa: E402


def build(kind, B, dev, overlap, capturable):
    torch.manual_seed(0)
    model = T.Encoder(T.make_config(kind)).to(dev).train()
    model.overlap_wgrad = overlap
    x, y = (synth.mel_batch(5, B) if kind == "ast" else synth.frame_batch(5, B))
    x, y = torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev)
    opt = FusedAdam(model.parameters(), lr=5e-6, weight_decay=0.01, decoupled=True, capturable=capturable)
    crit = CrossEntropyLoss()

    def step():
        opt.zero_grad(set_to_none=True)
        loss = crit(model(x).logits, y)
        loss.backward(gradient=unit_gradient(dev))
        opt.step()
        return loss.detach()
    return model, step


def timeit(step, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    kind, B = sys.argv[1], int(sys.argv[2])
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dev = torch.device("cuda", 0)
    _, s2 = build(kind, B, dev, True, False)
    for _ in range(4):
        s2()
    t2 = timeit(s2, steps)
    del s2
    torch.cuda.empty_cache()
    _, s1 = build(kind, B, dev, False, True)
    ref = [float(s1()) for _ in range(4)]
    t1 = timeit(s1, steps)
    ref += [float(s1()) for _ in range(3)]
    del s1
    torch.cuda.empty_cache()
    mg, sg = build(kind, B, dev, False, True)
    got = [float(sg()) for _ in range(4)]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        loss = sg()
    if getattr(mg, "_ws", None) is not None:
        mg._ws.pinned = True

    def replay():
        g.replay()
        return loss
    tg = timeit(replay, steps)
    # (capture does not execute: `steps` replays = steps 5 .. 4 + steps of the trajectory)
    after = [float(replay()) for _ in range(3)]
    print(f"{kind} B={B}: eager two-stream {t2:.3f} ms, eager single-stream {t1:.3f} ms, graph replay (single stream) {tg:.3f} ms")
    print("  losses eager single-stream:", [f"{v:.6f}" for v in ref])
    print("  losses graph             :", [f"{v:.6f}" for v in got + after])
    ok = ref[:4] == got and ref[4:] == after
    print("  trajectories bit-equal:", ok)


if __name__ == "__main__":
    main()
