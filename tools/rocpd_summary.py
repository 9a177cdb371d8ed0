#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs into the small text tables committed under profiles/.

    python tools/rocpd_summary.py trace <trace_results.db>         # == rocprofv3 --kernel-trace --stats
    python tools/rocpd_summary.py pmc <pmc_results.db> [...]       # per-kernel counter sums / per-dispatch means
    python tools/rocpd_summary.py streams <trace_results.db> [n]   # per-stream busy time / inter-kernel gaps over the LAST n launches
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


def streams(path, last=300):
    """Are the rollout streams back-to-back?  Per stream (queue): busy time and the gaps between consecutive kernels over the
    last `last` launches of the run (the timed region), then the listing of the first ~40 of them."""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = list(cur.execute(f"select start, end, name, {qcol or '0'} from kernels order by start"))[-last:]
    t0 = rows[0][0]
    print(f"# window: {len(rows)} launches over {(rows[-1][1] - t0) / 1e3:.0f} us ({path})")
    qs = sorted({r[3] for r in rows})
    for q in qs:
        mine = [r for r in rows if r[3] == q]
        busy = sum(r[1] - r[0] for r in mine)
        gaps = [max(0, b[0] - a[1]) for a, b in zip(mine[:-1], mine[1:])]
        span = mine[-1][1] - mine[0][0]
        print(f"# queue {q}: {len(mine)} launches, busy {busy / 1e3:.0f} us of {span / 1e3:.0f} ({100 * busy / max(span, 1):.0f} %), "
              f"gaps: mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us, max {max(gaps or [0]) / 1e3:.1f} us, > 2 us: {sum(g > 2000 for g in gaps)}")
    print("# start_us end_us dur_us queue kernel")
    for st, en, n, q in rows[:40]:
        print(f"{(st - t0) / 1e3:9.1f} {(en - t0) / 1e3:9.1f} {(en - st) / 1e3:8.1f}  q{qs.index(q)} {short(n)[:40]}")


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2])
    elif sys.argv[1] == "streams":
        streams(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 300)
    else:
        pmc(sys.argv[2:])
