#!/usr/bin/env python3
"""What makes the replays after a subject change slow?  30 replays, then one of: nothing | a trivial torch kernel | the device-table
reset (3 torch launches) | a 50 ms host sleep; then 16 replays timed one by one (ms each)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
run = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
for i in range(3):
    run.step(i)
run.prepare_resets([1001])
tiny = torch.zeros(16, device="cuda")
for what in ("nothing", "tiny kernel", "reset", "sleep 50 ms", "nothing", "reset", "tiny kernel"):
    for i in range(30):
        run.step(i)
    if what == "tiny kernel":
        tiny.add_(1.0)
    elif what == "reset":
        run.reset_model(1001)
    elif what.startswith("sleep"):
        torch.cuda.synchronize(); time.sleep(0.05)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(17)]
    ev[0].record()
    for k in range(16):
        run.step(k)
        ev[k + 1].record()
    torch.cuda.synchronize()
    print(f"{what:12s}", " ".join(f"{ev[k].elapsed_time(ev[k + 1]):.3f}" for k in range(16)))
