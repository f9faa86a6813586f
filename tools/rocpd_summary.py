#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) result: per-kernel launch count / average / total duration, and --
when the run collected PMC counters -- per-kernel counter averages.  Output is plain text suitable for
committing under profiles/ (the .db itself is scratch).   usage: rocpd_summary.py results.db [filter]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else cols[0]
    rows = cur.execute(f"select {name_col}, count(*), avg(end - start), sum(end - start), min(end - start), max(end - start) "
                       f"from kernels group by {name_col} order by 4 desc").fetchall()
    total = sum(r[3] for r in rows) or 1
    print(f"{'kernel':90s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for name, n, avg, tot, mn, mx in rows:
        if flt and flt not in name:
            continue
        print(f"{name[:90]:90s} {n:6d} {avg/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {tot/1e6:10.3f} {100*tot/total:6.2f}")
    try:
        pm = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                         "group by kernel_name, counter_name order by kernel_name").fetchall()
    except Exception as e:    # schema differs between versions
        pm = []
        try:
            cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            print("counters_collection columns:", cols)
        except Exception:
            pass
    if pm:
        print("\nPMC averages per dispatch")
        last = None
        for k, c, v, n in pm:
            if flt and flt not in k:
                continue
            if k != last:
                print(f"  {k[:100]}")
                last = k
            print(f"      {c:28s} {v:18.1f}   (n={n})")


if __name__ == "__main__":
    main()
