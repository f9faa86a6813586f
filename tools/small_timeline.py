#!/usr/bin/env python3
"""Tuning aid: what one small call consists of.  Run under rocprofv3 with kernel and memory-copy tracing, then list the
last call's kernels and copies in time order with the idle gaps between them.
    rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/p -o t -- python tools/small_timeline.py run <case>
    python tools/small_timeline.py show /tmp/p/.../t_results.db
cases: california (decode of the 400 x 400 masked float blob), dec256 / enc256 (256 x 256 float32), masked (encode of the 8192 x 8192 raster with a 10 % mask)"""
import os
import sqlite3
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import capi
    import cases
    P = capi.product()
    rng = np.random.default_rng(1)
    if case == "masked":    # device-resident 8192 x 8192 float32 with a 10 % mask, one encode call
        import ctypes as ct
        import torch
        from lerc_amd import api, synth
        dev = torch.device("cuda:0")
        x = synth.c2_float32().to(dev)
        i = torch.arange(8192).view(-1, 1)
        j = torch.arange(8192).view(1, -1)
        m = (((i // 97) + (j // 131)) % 10 != 0).to(torch.uint8).contiguous().to(dev)
        codec = api.DeviceCodec(torch.cuda.current_stream().cuda_stream)
        out = torch.empty(x.numel() * 4 + (1 << 20), dtype=torch.uint8, device=dev)
        f = lambda: codec.encode(x.data_ptr(), 6, 1, 8192, 8192, 1, 0.01, out.data_ptr(), out.numel(), m.data_ptr(), 1)
    elif case == "california":
        blob = open(os.path.join(ROOT, "tests", "golden", "california_400_400_1_float.lerc2"), "rb").read()
        f = lambda: P.decode(blob)
    else:
        t = cases.terrain(256, 256, rng).astype(np.float32)
        tb = P.encode(t, 0.01)[1]
        f = (lambda: P.decode(tb)) if case == "dec256" else (lambda: P.encode(t, 0.01, buf_size=t.nbytes))
    for _ in range(5):
        f()
    time.sleep(0.02)    # a visible gap in front of the call that is looked at
    t0 = time.perf_counter()
    f()
    print("last call: %.1f us wall" % (1e6 * (time.perf_counter() - t0)))


def show(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = [(s, e, n[:70]) for n, s, e in cur.execute("select name, start, end from kernels")]
    mc = [t for t in tabs if "memory_cop" in t]
    if mc:
        cols = [r[1] for r in cur.execute(f"pragma table_info({mc[0]})")]
        name = "name" if "name" in cols else cols[0]
        size = "size" if "size" in cols else None
        q = f"select {name}, start, end" + (f", {size}" if size else "") + f" from {mc[0]}"
        for r in cur.execute(q):
            ev.append((r[1], r[2], "copy " + str(r[0]) + (f" {r[3]} B" if size else "")))
    ev.sort()
    # the last call: everything behind the longest idle gap
    cut = max(range(1, len(ev)), key=lambda i: ev[i][0] - ev[i - 1][1])
    t0, prev = ev[cut][0], None
    for s, e, n in ev[cut:]:
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {n}")
        prev = e
    print(f"span {(ev[-1][1] - t0) / 1e3:.1f} us, {len(ev) - cut} events")


if __name__ == "__main__":
    run(sys.argv[2]) if sys.argv[1] == "run" else show(sys.argv[2])
