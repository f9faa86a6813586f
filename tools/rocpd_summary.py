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


def traffic_json(fetch_db, write_db, size, out_path):
    """bytes per launch per bench kernel group from two PMC passes (FETCH_SIZE and WRITE_SIZE, both in KiB).
    gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled (MI355X_MICROARCH.md, HBM section)."""
    import json
    groups = {"k_fast_encode1": "fast_encode1", "k_fast_stats": "fast_stats_sizes", "k_fast_pack": "fast_pack", "k_fast_decode": "fast_decode", "k_fast_decode_one": "fast_decode_one", "k_fast_decode_scan": "fast_decode_scan",
              "k_fast_discover": "fast_discover", "k_fast_scan_decide": "fast_scan_decide"}

    def per_kernel(db, counter):
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select kernel_name, avg(value) from counters_collection where counter_name = ? group by kernel_name",
                           (counter,)).fetchall()
        return {k: v for k, v in rows}
    fetch, write = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    out = {}
    for kname, f in fetch.items():
        for key, group in groups.items():
            if key + "<" in kname or key + "(" in kname:
                w = write.get(kname, 0.0)
                out[group] = int(2 * f * 1024 + w * 1024)
    # the sources these passes ran on (bench.py refuses a traffic file taken on other kernels)
    import hashlib
    import os
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lerc_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".cpp", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(d, fn), "rb").read())
    with open(out_path, "w") as fh:
        json.dump({"size": size, "unit": "bytes", "note": "2 x FETCH_SIZE + WRITE_SIZE per launch (KiB counters)",
                   "csrc_digest": h.hexdigest()[:16], "bytes_per_launch": out}, fh, indent=1)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--traffic":
        traffic_json(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5])
    else:
        main()
