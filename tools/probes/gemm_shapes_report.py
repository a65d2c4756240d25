#!/usr/bin/env python3
"""gemm_shapes_report.py vit|ast [batch] <db> [<db> ...]: per product of tools/probes/gemm_shapes.py, mean duration and the
mean of every counter in the given rocpd databases (gemm_sp launches grouped by position, REPS per product), beside the
product's algorithmic operand / output bytes."""
import collections
import sqlite3
import sys

sys.path.insert(0, __import__("os").path.dirname(__file__))
from gemm_shapes import REPS, shapes  # noqa: E402


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
    disp = []
    for did, kid, st, en, evid in c.execute(f"select id, kernel_id, start, end, event_id from {kd} order by start"):
        if "gemm_sp_kernel" in names.get(kid, ""):
            disp.append((evid, (en - st) / 1e3))
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for evid, pid, val in c.execute(f"select event_id, pmc_id, value from {ev}"):
        per[evid][pmc[pid]] += val
    return disp, per


if __name__ == "__main__":
    kind = sys.argv[1]
    rest = sys.argv[2:]
    batch = int(rest.pop(0)) if rest and rest[0].isdigit() else None
    nt, tr = shapes(kind, batch)
    rows = [(n, 4 * M * K + 4 * N * K, 4 * M * N) for n, M, N, K in nt] + [(n, 4 * T * (M + N), 4 * M * N) for n, M, N, T in tr]
    out = collections.OrderedDict((r[0], {"alg_read_MB": r[1] / 1e6, "alg_write_MB": r[2] / 1e6}) for r in rows)
    for path in rest:
        disp, per = load(path)
        assert len(disp) == REPS * len(rows), (len(disp), len(rows))
        for i, r in enumerate(rows):
            grp = disp[i * REPS + 1:(i + 1) * REPS]           # first launch of a product: cold
            out[r[0]]["us"] = sum(d for _, d in grp) / len(grp)
            cs = collections.defaultdict(list)
            for evid, _ in grp:
                for k, v in per[evid].items():
                    cs[k].append(v)
            for k, v in cs.items():
                out[r[0]][k] = sum(v) / len(v)
    for n, d in out.items():
        extra = ""
        if "FETCH_SIZE" in d:
            extra += f"  hbm_read {2 * d['FETCH_SIZE'] * 1024 / 1e6:8.1f} MB ({2 * d['FETCH_SIZE'] * 1024 / 1e6 / d['alg_read_MB']:.2f}x)"
        if "WRITE_SIZE" in d:
            extra += f"  hbm_write {d['WRITE_SIZE'] * 1024 / 1e6:8.1f} MB"
        if "TCC_HIT_sum" in d:
            extra += f"  L2 hit {d['TCC_HIT_sum'] / (d['TCC_HIT_sum'] + d['TCC_MISS_sum']):.3f}"
        print(f"{n:12s} {d.get('us', 0):8.1f} us  alg read {d['alg_read_MB']:7.1f} MB write {d['alg_write_MB']:7.1f} MB{extra}")
