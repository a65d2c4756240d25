#!/usr/bin/env python3
"""Summarise rocprofv3 outputs into the small files kept under profiles/.

    python tools/pmc_summary.py <round-tag> <kernel_stats.csv> <fetch counter csv> <write counter csv>

Writes profiles/<tag>_kernel_stats.csv (copy of rocprofv3 --stats summary), and
profiles/<tag>_hbm_traffic.json: per kernel, mean HBM bytes per launch with the gfx950 correction the
MI355X guide prescribes (FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports half the bytes of a wide
coalesced read stream, so it is doubled; WRITE_SIZE is taken as is)."""
import collections
import csv
import json
import os
import shutil
import sys

tag, stats, fetch, write = sys.argv[1:5]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))


def mean_by_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


f = mean_by_kernel(fetch, "FETCH_SIZE")
w = mean_by_kernel(write, "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    rd = 2.0 * f.get(k, 0.0) * 1024.0
    wr = w.get(k, 0.0) * 1024.0
    out[short] = {"read_bytes": round(rd), "write_bytes": round(wr), "total_bytes": round(rd + wr),
                  "raw_FETCH_SIZE_KiB": round(f.get(k, 0.0), 1), "raw_WRITE_SIZE_KiB": round(w.get(k, 0.0), 1)}
# templated kernels with a single instantiation in the run are also listed under their bare name
bases = collections.Counter(k.split("<")[0] for k in out if "<" in k)
for k in list(out):
    b = k.split("<")[0]
    if "<" in k and bases[b] == 1 and b not in out:
        out[b] = dict(out[k], instantiation=k)
json.dump({"note": "mean per launch; read = 2*FETCH_SIZE*1024 (gfx950 correction), write = WRITE_SIZE*1024",
           "kernels": out}, open(os.path.join(root, "profiles", f"{tag}_hbm_traffic.json"), "w"), indent=1)
print("wrote", tag, len(out), "kernels")
