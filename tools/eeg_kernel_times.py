#!/usr/bin/env python3
"""Per-kernel times of one EEGNet train step at the bench shape via HIP events (run on the GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
run = bench.EEGRun(torch.device("cuda", 0), 0, 1, 64, 16)
if mode == "eval":
    run.model.eval()
for i in range(4):
    run.step(i)
names = ["eav_eegnet_fir_fwd", "eav_eegnet_fir_wgrad", "eav_eegnet_dw_fwd", "eav_eegnet_dw_bwd", "eav_eegnet_dw_bwd_fused", "eav_conv64_fwd",
         "eav_conv64_wgrad", "eav_bn_elu_pool_fwd_absmax", "eav_bn_elu_pool_fwd", "eav_bn_elu_pool_bwd_reduce",
         "eav_bn_elu_pool_bwd_apply_absmax", "eav_dense_softmax_fwd", "eav_dense_softmax_bwd", "eav_reduce_partials",
         "eav_bn_finalize", "eav_bn_bwd_finalize", "eav_conv64_prep_weights", "eav_renorm_rows"]
run.model.kernel_events = {k: [] for k in names}
for i in range(8):
    run.eager_step(4 + i)          # the graph-replayed step has no per-launch events
torch.cuda.synchronize()
tot = 0.0
for k, v in run.model.kernel_events.items():
    if v:
        ms = sum(a.elapsed_time(b) for a, b in v) / 8
        tot += ms
        print(f"{k:36s} {len(v) // 8} x {ms / (len(v) // 8) * 1e3:8.1f} us = {ms * 1e3:8.1f} us/step")
run.model.kernel_events = None
dt, _ = run.timed(20, 0)
print(f"sum of listed kernels {tot:.3f} ms; step {dt / 20 * 1e3:.3f} ms ({mode})")
