"""CPU soak of the checker itself: random rasters through the oracle (restatement) and oracle/_ref (the real reference).
prints mismatches.  Needs oracle/_ref (make -C oracle ref; /root/reference present).
    python tools/fuzz_oracle_vs_reference.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, capi, cases
S = capi.oracle(); O = capi.ref()
def same(a, b):
    if a is None or b is None: return a is None and b is None
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
t0 = time.time(); n = bad = 0
while time.time() - t0 < budget:
    dt = cases.ALL_DTYPES[rng.integers(0, 8)]
    r, c = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    if rng.random() < 0.4: r, c = max(8, r - r % 8), max(8, c - c % 8)
    nd = int(rng.choice([1, 1, 1, 2, 3])); nb = int(rng.choice([1, 1, 1, 2]))
    kind = np.dtype(dt).kind
    planes = []
    for _ in range(nb):
        x = cases.terrain(r, c, rng, amp=float(rng.choice([5, 50, 500])), base=float(rng.choice([0, 100, 1000])), sigma=float(rng.choice([0, 0.3, 3])))
        st = rng.integers(0, 5)
        if st == 1: x = np.floor(x / 16) * 16
        if st == 2: x = np.round(x, 1)
        if st == 3: x[::9, ::7] *= 1e6
        x = np.stack([x + k for k in range(nd)], axis=-1) if nd > 1 else x
        planes.append(x)
    x = np.stack(planes) if nb > 1 else planes[0]
    if np.dtype(dt).itemsize == 1: x = x / 8
    x = np.ascontiguousarray(cases._cast(x, dt))
    e = float(rng.choice([0, 0.001, 0.01, 0.5, 1, 3])) if kind == "f" else float(rng.choice([0, 0, 1, 4]))
    kw = dict(n_depth=nd, n_bands=nb) if nb > 1 else dict(n_depth=nd)
    if rng.random() < 0.5:
        m = (rng.random((r, c)) > 0.2).astype(np.uint8)
        if rng.random() < 0.3: m[: r // 2] = 0
        kw["mask"] = m
    tag = f"{np.dtype(dt).name} {nb}x{r}x{c}x{nd} e={e} mask={'mask' in kw}"
    try:
        r1, b1 = O.encode(x, e, **kw); r2, b2 = S.encode(x, e, **kw)
    except TypeError:
        kw.pop("n_bands", None); x = x[0] if nb > 1 else x; nb = 1
        r1, b1 = O.encode(x, e, **kw); r2, b2 = S.encode(x, e, **kw)
    n += 1
    if r1 != r2 or (r1 == 0 and len(b1) != len(b2)) or (r1 == 0 and bytes(b1) != bytes(b2) and not (kind == "f" and e == 0)):
        print("ENC MISMATCH", tag, r1, r2, len(b1), len(b2)); bad += 1; continue
    if r1 == 0:
        d1, d2 = O.decode(b1), S.decode(b1)
        if d1[0] != d2[0] or not same(d1[1], d2[1]) or not same(d1[2], d2[2]):
            print("DEC MISMATCH", tag, d1[0], d2[0]); bad += 1
        k = int(rng.integers(0, len(b1))); y = bytearray(b1); y[k] ^= 1 << int(rng.integers(0, 8))
        g1, g2 = O.decode(bytes(y)), S.decode(bytes(y))
        if (g1[0] == 0) != (g2[0] == 0) or (g1[0] == 0 and not (same(g1[1], g2[1]) and same(g1[2], g2[2]))):
            print("DAMAGED MISMATCH", tag, k, g1[0], g2[0]); bad += 1
print("cases", n, "mismatches", bad)
