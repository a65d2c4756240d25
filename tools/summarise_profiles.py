#!/usr/bin/env python3
"""gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) -> the small files kept under profiles/:

  <tag>_<run>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary, as emitted
  <tag>_<run>_pmc.json             per kernel, mean per launch: duration, HBM bytes (FETCH_SIZE / WRITE_SIZE in KiB;
                                   FETCH doubled: gfx950 tallies 128-B requests at 64 B - MI355X guide), MFMA pipe busy
                                   (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8 XCDs)) and the
                                   wave-cycle split (parked / issue-stalled).
usage: summarise_profiles.py <tag>"""
import collections
import csv
import json
import os
import re
import shutil
import sys

import subprocess

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:        # the tree the counters were collected from (run right after the gpurun call, before further edits)
    commit = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except OSError:
    commit = None
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.sub(r"\(.*", "", name).strip()


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def durations(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return agg


for run in ("eeg", "ast", "vit"):
    stats = os.path.join(src, f"{run}_trace", f"{run}_kernel_stats.csv")
    if not os.path.exists(stats):
        continue
    shutil.copy(stats, os.path.join(dst, f"{tag}_{run}_kernel_stats.csv"))
    dur = durations(os.path.join(src, f"{run}_trace", f"{run}_kernel_trace.csv"))
    fetch = counters(os.path.join(src, f"{run}_fetch", f"{run}_counter_collection.csv"))
    write = counters(os.path.join(src, f"{run}_write", f"{run}_counter_collection.csv"))
    mfma = counters(os.path.join(src, f"{run}_mfma", f"{run}_counter_collection.csv"))
    total = sum(sum(v) for v in dur.values())
    # whole-step figures (EEGNet run = tools/eeg_steps.py: train steps only): every launch of every kernel, divided by the
    # number of optimiser steps (= adam_kernel launches)
    step = None
    nsteps = sum(len(v) for k, v in dur.items() if k.startswith("adam_kernel"))
    if run == "eeg" and nsteps:
        fb = sum(sum(c.get("FETCH_SIZE", [])) for c in fetch.values())
        wb = sum(sum(c.get("WRITE_SIZE", [])) for c in write.values())
        step = {"steps_profiled": nsteps, "hbm_bytes_per_step": round((2.0 * fb + wb) * 1024 / nsteps),
                "hbm_read_bytes_per_step": round(2.0 * fb * 1024 / nsteps), "hbm_write_bytes_per_step": round(wb * 1024 / nsteps),
                "kernel_ms_per_step": round(total / 1e6 / nsteps, 4), "launches_per_step": round(sum(len(v) for v in dur.values()) / nsteps, 1)}
    out = {}
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) < 0.002 * total:
            continue
        mean = lambda d, c: (sum(d[k][c]) / len(d[k][c])) if d[k].get(c) else None   # noqa: E731
        e = {"calls": len(v), "avg_us": round(sum(v) / len(v) / 1e3, 2), "share_of_kernel_time": round(sum(v) / total, 4)}
        f, w = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE")
        if f is not None and w is not None:
            e["hbm_read_bytes"] = round(2.0 * f * 1024)
            e["hbm_write_bytes"] = round(w * 1024)
            e["hbm_gb_per_s"] = round((2.0 * f + w) * 1024 / (sum(v) / len(v)), 1)      # bytes / ns = GB/s
        busy, gui = mean(mfma, "SQ_VALU_MFMA_BUSY_CYCLES"), mean(mfma, "GRBM_GUI_ACTIVE")
        if busy is not None and gui:
            e["mfma_pipe_busy"] = round(busy / (1024.0 * gui / 8.0), 4)
            wc = mean(mfma, "SQ_WAVE_CYCLES")
            if wc:
                e["wave_cycles_parked"] = round(mean(mfma, "SQ_WAIT_ANY") / wc, 3)
                e["wave_cycles_issue_stalled"] = round(mean(mfma, "SQ_WAIT_INST_ANY") / wc, 3)
        out[k] = e
    note = ("two HIP streams run concurrently in the encoder step (weight gradients beside the main chain): a kernel's "
            "duration there includes the time its workgroups wait for CUs the other stream holds, so mfma_pipe_busy / "
            "hbm_gb_per_s of the two-stream run are per-kernel LOWER bounds that ADD across concurrent kernels; the "
            "*_single_stream fields divide the same counters by the kernel's duration in the NO_OVERLAP=1 run.  "
            "mean per launch over the profiled run; hbm_read = 2 x FETCH_SIZE x 1024 (gfx950 correction), hbm_write = "
            "WRITE_SIZE x 1024; mfma_pipe_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); "
            "counters come from separate --pmc passes of the same command (tools/collect_profiles.sh)")
    serial = os.path.join(src, f"{run}_serial", f"{run}_kernel_stats.csv")
    if os.path.exists(serial):      # single-stream run of the same step: durations without the two-stream stretch
        shutil.copy(serial, os.path.join(dst, f"{tag}_{run}_serial_kernel_stats.csv"))
        sd = durations(os.path.join(src, f"{run}_serial", f"{run}_kernel_trace.csv"))
        for k, e in out.items():
            if k in sd:
                e["avg_us_single_stream"] = round(sum(sd[k]) / len(sd[k]) / 1e3, 2)
                if "mfma_pipe_busy" in e:      # same busy cycles over the un-stretched duration
                    e["mfma_pipe_busy_single_stream"] = round(e["mfma_pipe_busy"] * e["avg_us"] / e["avg_us_single_stream"], 4)
                if "hbm_gb_per_s" in e:
                    e["hbm_gb_per_s_single_stream"] = round(e["hbm_gb_per_s"] * e["avg_us"] / e["avg_us_single_stream"], 1)
    json.dump({"note": note, "commit": commit, "total_kernel_ms": round(total / 1e6, 3), "step": step, "kernels": out},
              open(os.path.join(dst, f"{tag}_{run}_pmc.json"), "w"), indent=1)
    print(run, "total kernel ms", round(total / 1e6, 2))
    for k, e in list(out.items())[:14]:
        print(f"  {k[:44]:44s} {e['calls']:5d} x {e['avg_us']:9.1f} us  share {e['share_of_kernel_time']:.3f}  "
              f"HBM {e.get('hbm_gb_per_s', '-'):>7} GB/s  MFMA busy {e.get('mfma_pipe_busy', '-')}")
# bench.py's roofline.traffic reads <tag>_eegnet_hbm_traffic.json: keep that name / shape for the EEGNet run
p = os.path.join(dst, f"{tag}_eeg_pmc.json")
if os.path.exists(p):
    d = json.load(open(p))
    ker = {k if k.endswith("_kernel") or "<" in k else k: {"read_bytes": e.get("hbm_read_bytes"),
                                                            "write_bytes": e.get("hbm_write_bytes"),
                                                            "total_bytes": (e.get("hbm_read_bytes") or 0) + (e.get("hbm_write_bytes") or 0),
                                                            "avg_us": e.get("avg_us"), "calls": e.get("calls")}
           for k, e in d["kernels"].items() if "hbm_read_bytes" in e}
    for k in list(ker):            # bare template names too ("fir_wgrad_kernel<10, false>" -> "fir_wgrad_kernel")
        b = k.split("<")[0]
        ker.setdefault(b, ker[k])
    json.dump({"note": d["note"], "commit": d.get("commit"), "step": d.get("step"), "kernels": ker},
              open(os.path.join(dst, f"{tag}_eegnet_hbm_traffic.json"), "w"), indent=1)
