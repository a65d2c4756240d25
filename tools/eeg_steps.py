"""N graph-replayed EEGNet train steps at the bench shape and nothing else - the command tools/collect_profiles.sh puts
under rocprofv3 (kernel trace + the separate FETCH_SIZE / WRITE_SIZE passes), so that per-kernel means AND the traffic of a
whole step (all launches / number of adam_kernel launches) come from one clean run.

    python3 tools/eeg_steps.py [steps=20] [eval]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda", 0)
    run = bench.EEGRun(dev, 0, 1, bench.B_PER_GPU, steps + 3)
    if len(sys.argv) > 2 and sys.argv[2] == "eval":
        run.model.eval()
    for i in range(3):
        run.step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        run.step(3 + i)
    torch.cuda.synchronize()
    print(f"{steps} steps, {(time.perf_counter() - t0) / steps * 1e3:.4f} ms per step")


if __name__ == "__main__":
    main()
