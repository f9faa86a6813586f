#!/usr/bin/env python3
"""Tuning aid: per-dispatch timeline of one bench step from a rocprofv3 --kernel-trace (+ --memory-copy-trace)
rocpd database: start offset, duration and the idle gap before every kernel.   usage: timeline.py results.db [step]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    cur = con.cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    # one step starts at every k_fast_stats
    starts = [i for i, r in enumerate(rows) if "k_fast_stats" in r[0]]
    if len(starts) <= step + 1:
        step = max(0, len(starts) - 2)
    a, b = starts[step], starts[step + 1]
    t0 = rows[a][1]
    prev_end = None
    print(f"{'kernel':60s} {'start_us':>10s} {'dur_us':>9s} {'gap_us':>8s}")
    for name, s, e in rows[a:b]:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        print(f"{name[:60]:60s} {(s - t0) / 1e3:10.1f} {(e - s) / 1e3:9.1f} {gap:8.1f}")
        prev_end = e
    print(f"step span: {(rows[b][1] - t0) / 1e3:.1f} us (to the next step's first kernel)")


if __name__ == "__main__":
    main()
