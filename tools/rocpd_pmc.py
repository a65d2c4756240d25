#!/usr/bin/env python3
"""Per-kernel counter means from a rocprofv3 rocpd database (the default output of `rocprofv3 --pmc ... -d DIR`).

    python tools/rocpd_pmc.py <results.db> [kernel-name-substring]
Prints, per kernel name: launches, mean duration (us) and the mean of every collected counter per launch."""
import collections
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    c = db.cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda n: [x for x in tabs if x.startswith(n)][0]  # noqa: E731
    ev, pm, kd, ks = t("rocpd_pmc_event"), t("rocpd_info_pmc"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
    kcols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    namecol = "kernel_name" if "kernel_name" in kcols else ("display_name" if "display_name" in kcols else "name")
    names = dict(c.execute(f"select id, {namecol} from {ks}"))
    pmc = dict(c.execute(f"select id, name from {pm}"))
    disp = {}
    for did, kid, st, en, evid in c.execute(f"select id, kernel_id, start, end, event_id from {kd}"):
        disp[evid] = (names.get(kid, str(kid)), (en - st) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for evid, (n, d) in disp.items():
        dur[n].append(d)
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for evid, pid, val in c.execute(f"select event_id, pmc_id, value from {ev}"):
        if evid in disp:
            per[evid][pmc[pid]] += val          # a counter is reported per instance (XCD / SE): sum them
    for evid, cs in per.items():
        for k, v in cs.items():
            agg[disp[evid][0]][k].append(v)
    return dur, agg


if __name__ == "__main__":
    dur, agg = load(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    for n in sorted(dur, key=lambda k: -sum(dur[k])):
        if sub not in n:
            continue
        short = n.replace("(anonymous namespace)::", "").replace("void ", "")[:110]
        print(f"{short}\n    launches {len(dur[n])}  mean {sum(dur[n]) / len(dur[n]):9.1f} us")
        for k, v in sorted(agg[n].items()):
            print(f"    {k:28s} {sum(v) / len(v):16.1f}")
