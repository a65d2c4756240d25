#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration, share) of a rocprofv3 `--kernel-trace` result database
(rocpd sqlite, the default output format of ROCm 7.2).  usage: rocpd_stats.py results.db [out.csv] [skip_first_calls]"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 90 else name[:87] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, (end - start) from kernels order by start").fetchall()
    agg = {}
    for name, dur in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        a[0] += 1
        a[1] += dur
        a[2] = min(a[2], dur)
        a[3] = max(a[3], dur)
    total = sum(a[1] for a in agg.values())
    out = sorted(agg.items(), key=lambda kv: -kv[1][1])
    w = csv.writer(open(sys.argv[2], "w")) if len(sys.argv) > 2 else None
    if w:
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    print(f"{'kernel':90s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}")
    for name, (n, t, mn, mx) in out:
        if w:
            w.writerow([name, n, t, f"{t / n:.1f}", f"{100.0 * t / total:.2f}", mn, mx])
        print(f"{short(name):90s} {n:7d} {t / 1e6:10.3f} {t / n / 1e3:9.2f} {100.0 * t / total:6.2f}")
    print(f"total kernel time {total / 1e6:.3f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main()
