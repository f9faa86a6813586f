"""Tuning aid: latency of the stock host-pointer calls on small rasters (launch + sync bound), next to the reference on
this host if oracle/_ref travelled.   gpurun -- 'python tools/time_small.py'"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import capi  # noqa: E402
import cases  # noqa: E402


def clock(f, reps=20):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    return 1e3 * (time.perf_counter() - t0) / reps


def main():
    P, R = capi.product(), capi.ref()
    rng = np.random.default_rng(1)
    blob = open(os.path.join(ROOT, "tests", "golden", "california_400_400_1_float.lerc2"), "rb").read()
    rows = [("decode california 400x400 f32 masked", lambda L: L.decode(blob))]
    x = cases.terrain(400, 400, rng).astype(np.float32)
    m = (rng.random((400, 400)) > 0.3).astype(np.uint8)
    rows.append(("encode 400x400 f32 masked", lambda L: L.encode(x, 0.01, mask=m, buf_size=x.nbytes)))
    t = cases.terrain(256, 256, rng).astype(np.float32)
    rows.append(("encode 256x256 f32", lambda L: L.encode(t, 0.01, buf_size=t.nbytes)))
    tb = P.encode(t, 0.01)[1]
    rows.append(("decode 256x256 f32", lambda L: L.decode(tb)))
    u = (cases.terrain(1024, 1024, rng)).astype(np.uint16)
    rows.append(("encode 1024x1024 u16", lambda L: L.encode(u, 0, buf_size=u.nbytes)))
    ub = P.encode(u, 0)[1]
    rows.append(("decode 1024x1024 u16", lambda L: L.decode(ub)))
    for name, f in rows:
        a = clock(lambda: f(P))
        b = clock(lambda: f(R)) if R is not None else float("nan")
        print(f"{name:40s} lerc_amd {a:7.3f} ms   reference {b:7.3f} ms")


if __name__ == "__main__":
    main()
