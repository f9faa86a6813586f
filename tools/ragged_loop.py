import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, capi, cases
P = capi.product()
rng = np.random.default_rng(41)
x = cases._cast(cases.terrain(2050, 4099, rng, amp=300, base=1000, sigma=2.0), np.float32)
r, b = P.encode(x, 0.01)
fall_e = fall_d = 0
for i in range(150):
    c0 = P.path_counters()
    r2, b2 = P.encode(x, 0.01)
    d = P.decode(b)
    c1 = P.path_counters()
    assert r2 == 0 and b2 == b and d[0] == 0
    if c1[0] - c0[0] != 2: fall_e += 1; print("encode fell back at", i, c0, c1, P.last_note())
    if c1[2] - c0[2] != 1: fall_d += 1
print("encode fallbacks", fall_e, "decode fallbacks", fall_d, "last note:", P.last_note())
