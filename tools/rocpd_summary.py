#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small text tables committed under profiles/.

    python tools/rocpd_summary.py trace <trace_results.db>         # == rocprofv3 --kernel-trace --stats
    python tools/rocpd_summary.py pmc <pmc_results.db> [...]       # per-kernel counter sums / per-dispatch means
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:60]


def trace(path):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                            "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows)
    print(f"# kernel-trace stats from {path}\n# name calls total_us avg_us min_us max_us pct")
    for n, c, s, a, mn, mx in rows:
        print(f"{short(n):60s} {c:6d} {s / 1e3:12.1f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100 * s / tot:6.2f}")


def pmc(paths):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for kn, cn, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            acc[short(kn)][cn] += v
            cnt[short(kn)][cn] += 1
    names = sorted({c for k in acc for c in acc[k]})
    print("# per-kernel PMC: mean per dispatch (dispatch count in parentheses) from " + ", ".join(paths))
    for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", acc[k].get(names[0], 0))):
        n = max(cnt[k].values())
        print(f"{k} ({n} dispatches)")
        for c in names:
            if c in acc[k]:
                print(f"    {c:32s} {acc[k][c] / cnt[k][c]:18.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    else:
        pmc(sys.argv[2:])
