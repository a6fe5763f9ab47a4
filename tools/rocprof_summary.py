"""Summarise rocprofv3 rocpd databases (gpurun_out/<run>/*_results.db) into small text tables
under profiles/.   usage: rocprof_summary.py stats <db> | pmc <db> [kernel-substring]"""
import glob
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name.replace("vfi::", "")


def stats(db):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"{'kernel':60s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
    for n, c, t, a, p in rows:
        if len(n) > 400:
            n = n[:57] + "..."
        print(f"{short(n)[:60]:60s} {c:6d} {t:12.1f} {a:10.2f} {p:7.2f}")


def pmc(db, sub=None):
    con = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(duration) desc")
    print(f"{'kernel':60s} {'counter':28s} {'n':>5s} {'avg_value':>16s} {'avg_dur_us':>10s}")
    for k, c, n, s, a, d in con.execute(q):
        if sub and sub not in k:
            continue
        if len(k) > 400:
            continue
        print(f"{short(k)[:60]:60s} {c:28s} {n:5d} {a:16.1f} {(d or 0) / 1e3:10.2f}")


def phases(db, sub, labels, counter=None):
    """Kernels whose name contains `sub`, in dispatch order, assigned round-robin to `labels` (each "name*count"): the RIFE step
    launches ONE Winograd instantiation for all four trunk widths — 8 x c192, 8 x c128, 8 x c96, 8 x c64 per step — so the
    per-width average (what bench.py's roofline object quotes for resconv_c64) has to be split by position.  With `counter`
    the same split for a PMC pass (counters_collection)."""
    con = sqlite3.connect(db)
    pat = []
    for l in labels.split(","):
        name, cnt = l.split("*")
        pat += [name] * int(cnt)
    if counter:
        cols = [r[1] for r in con.execute("pragma table_info('counters_collection')")]
        order = "dispatch_id" if "dispatch_id" in cols else ("start" if "start" in cols else "rowid")
        rows = list(con.execute(f"select value, duration from counters_collection where kernel_name like ? and counter_name = ? order by {order}",
                                (f"%{sub}%", counter)))
    else:
        rows = list(con.execute("select duration, duration from kernels where name like ? order by start", (f"%{sub}%",)))
    agg = {}
    for i, (v, d) in enumerate(rows):
        a = agg.setdefault(pat[i % len(pat)], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += v
        a[2] += d or 0
    what = counter or "duration_ns"
    print(f"{'phase of ' + sub:40s} {'n':>5s} {'avg ' + what:>22s} {'avg_dur_us':>12s}")
    for k, (n, sv, sd) in agg.items():
        print(f"{k:40s} {n:5d} {sv / n:22.1f} {sd / n / 1e3:12.2f}")


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    import os
    cands = glob.glob(db) if not os.path.isdir(db) else []
    if not cands:  # a directory (or a pattern that missed): look for the rocpd database below it
        root = db if os.path.isdir(db) else os.path.dirname(db.split('*')[0]) or '.'
        cands = sorted(glob.glob(os.path.join(root, '**', '*_results.db'), recursive=True))
    db = cands[0]
    if mode == "phases":
        phases(db, sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    elif mode == "stats":
        stats(db)
    else:
        pmc(db, sys.argv[3] if len(sys.argv) > 3 else None)
