#!/usr/bin/env python3
"""Critical-path view of one training step from a rocprofv3 --kernel-trace csv: per stream (queue) busy time, idle gaps
on the main stream, and per-kernel time split into 'alone' vs 'overlapped with the other stream'.

    python tools/timeline.py <kernel_trace.csv> [steps-to-skip-from-the-end=1]
The last complete step (delimited by adam_kernel launches) is analysed."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
rows.sort(key=lambda r: r["s"])
adam = [i for i, r in enumerate(rows) if r["n"].startswith("adam_kernel")]
# a step = after the last adam launch of the previous step ... the last adam launch of this step
groups, cur = [], []
for i in adam:
    if cur and rows[i]["s"] - rows[cur[-1]]["e"] > 2_000_000:
        groups.append(cur)
        cur = []
    cur.append(i)
groups.append(cur)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g0, g1 = groups[-1 - skip - 1], groups[-1 - skip]
step = rows[g0[-1] + 1:g1[-1] + 1]
t0, t1 = step[0]["s"], step[-1]["e"]
print(f"step: {(t1 - t0) / 1e6:.3f} ms, {len(step)} kernels")
qs = collections.Counter(r["Queue_Id"] for r in step)
main = qs.most_common(1)[0][0]
for q, n in qs.most_common():
    ks = [r for r in step if r["Queue_Id"] == q]
    busy = sum(r["e"] - r["s"] for r in ks)
    print(f"  queue {q}: {n} kernels, busy {busy / 1e6:.3f} ms" + ("  (main)" if q == main else ""))
mk = [r for r in step if r["Queue_Id"] == main]
gaps = sum(max(0, b["s"] - a["e"]) for a, b in zip(mk, mk[1:]))
print(f"  main-stream idle gaps: {gaps / 1e6:.3f} ms; biggest:")
big = sorted(((b["s"] - a["e"], a["n"], b["n"]) for a, b in zip(mk, mk[1:])), reverse=True)[:6]
for d, a, b in big:
    print(f"     {d / 1e3:8.1f} us between {a[:40]} -> {b[:40]}")
side = sorted(((r["s"], r["e"]) for r in step if r["Queue_Id"] != main))


def overlap(s, e):
    return sum(max(0, min(e, b) - max(s, a)) for a, b in side)


agg = collections.defaultdict(lambda: [0, 0, 0])
for r in step:
    a = agg[(r["n"][:70], "main" if r["Queue_Id"] == main else "side")]
    a[0] += 1
    a[1] += r["e"] - r["s"]
    if r["Queue_Id"] == main:
        a[2] += overlap(r["s"], r["e"])
print(f"{'kernel':72s} {'strm':4s} {'n':>4s} {'total ms':>9s} {'mean us':>8s} {'ovl ms':>7s}")
for (n, q), (c, t, o) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:72s} {q:4s} {c:4d} {t / 1e6:9.3f} {t / c / 1e3:8.1f} {o / 1e6:7.3f}")
