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


if __name__ == "__main__":
    mode, db = sys.argv[1], sys.argv[2]
    import os
    cands = glob.glob(db) if not os.path.isdir(db) else []
    if not cands:  # a directory (or a pattern that missed): look for the rocpd database below it
        root = db if os.path.isdir(db) else os.path.dirname(db.split('*')[0]) or '.'
        cands = sorted(glob.glob(os.path.join(root, '**', '*_results.db'), recursive=True))
    db = cands[0]
    if mode == "stats":
        stats(db)
    else:
        pmc(db, sys.argv[3] if len(sys.argv) > 3 else None)
