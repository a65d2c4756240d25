#!/usr/bin/env python3
"""How the graph-replayed EEGNet step time settles after capture: HIP events every 10 replays (first timed block of bench.py
= replays 3 .. 52).  usage: eeg_step_ramp.py [idle_ms before the replays]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
run = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
for i in range(3):
    run.step(i)                      # eager, eager, capture + first replay
torch.cuda.synchronize()
if len(sys.argv) > 1:
    time.sleep(float(sys.argv[1]) * 1e-3)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for k in range(40):
    for i in range(10):
        run.step(i)
    ev[k + 1].record()
torch.cuda.synchronize()
torch.cuda.synchronize()
run2 = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
for i in range(3):
    run2.step(i)
torch.cuda.synchronize()
e2 = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
e2[0].record()
for k in range(20):
    run2.step(k)
    e2[k + 1].record()
torch.cuda.synchronize()
print("first 20 replays after capture, ms each:", " ".join(f"{e2[k].elapsed_time(e2[k + 1]):.3f}" for k in range(20)))
print("ms/step per group of 10 replays:", " ".join(f"{ev[k].elapsed_time(ev[k + 1]) / 10:.3f}" for k in range(40)))
