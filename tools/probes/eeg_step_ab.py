#!/usr/bin/env python3
"""Graph-replayed EEGNet bench step, 5 x 200 steps: run it alternately under two settings of an environment switch on ONE box
(box-to-box spread is ~1 %, larger than most single-kernel changes).   usage: [VAR=..] eeg_step_ab.py [VAR|-] [eval]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
run = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
if len(sys.argv) > 2 and sys.argv[2] == "eval":
    run.model.eval()
for i in range(6): run.step(i)
ts = []
for r in range(5):
    dt, _ = run.timed(200, 0)
    ts.append(dt / 200 * 1e3)
var = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else ""
print(f"{var}={os.environ.get(var, '')}" if var else "", " ".join(f"{t:.4f}" for t in ts), "ms/step")
