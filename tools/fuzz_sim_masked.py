"""Emulator soak of the general path: small rasters of every data type with masks of several kinds (and none, ragged), some damaged
copies; the emulator library against the oracle.   python tools/fuzz_sim_masked.py [seed] [seconds]"""
import sys, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, capi, cases
S, O = capi.sim(), capi.oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
t0 = time.time(); n = bad = 0
while time.time() - t0 < budget:
    dt = cases.ALL_DTYPES[rng.integers(0, 8)]
    r, c = int(rng.integers(9, 260)), int(rng.integers(9, 700))
    if rng.random() < 0.4: r -= r % 8; c -= c % 8
    r, c = max(r, 8), max(c, 8)
    kind = np.dtype(dt).kind
    x = cases.terrain(r, c, rng, amp=float(rng.choice([5, 50, 500])), base=float(rng.choice([0, 100, 1000])), sigma=float(rng.choice([0, 0.3, 3])))
    style = int(rng.integers(0, 4))
    if style == 1: x = np.floor(x / 16) * 16
    if style == 2: x = np.round(x, 1)
    if np.dtype(dt).itemsize == 1: x = x / 8
    nd = int(rng.choice([1, 1, 1, 2, 3]))
    if nd > 1: x = np.stack([x + k for k in range(nd)], axis=-1)
    x = cases._cast(x, dt)
    e = float(rng.choice([0, 0.001, 0.01, 0.5, 1, 3])) if kind == "f" else float(rng.choice([0, 0, 1, 4]))
    kw = {"n_depth": nd} if nd > 1 else {}
    mk = int(rng.integers(0, 8))
    if mk:
        m = np.ones((r, c), np.uint8)
        if mk == 1:
            for _ in range(int(rng.integers(1, 12))):
                i0, j0 = int(rng.integers(0, r)), int(rng.integers(0, c)); m[i0:i0 + int(rng.integers(1, 40)), j0:j0 + int(rng.integers(1, 90))] = 0
        elif mk == 2: m = (rng.random((r, c)) > rng.random() * 0.7).astype(np.uint8)
        elif mk == 3: m[:, : c // 2] = 0; m[r // 2, 3] = 1
        elif mk == 4: m = (((np.arange(r)[:, None] // 7) + (np.arange(c)[None, :] // 19)) % 3 != 0).astype(np.uint8)
        elif mk == 5:    # an ellipse: curved edges (blocks of one or two valid pixels: raw blocks), the rest of each row invalid
            ii, jj = np.mgrid[0:r, 0:c]; m = (((ii - r / 2) / (r * rng.uniform(0.2, 0.6))) ** 2 + ((jj - c / 2) / (c * rng.uniform(0.2, 0.6))) ** 2 < 1).astype(np.uint8)
        elif mk == 6:    # diagonal bands
            ii, jj = np.mgrid[0:r, 0:c]; w = int(rng.integers(20, 400)); m = ((ii * int(rng.integers(1, 4)) + jj) % w < w * rng.uniform(0.2, 0.9)).astype(np.uint8)
        else:            # long runs: columns and rows without a valid pixel
            j0 = int(rng.integers(0, c)); m[:, j0:j0 + int(rng.integers(c // 4, c))] = 0; i0 = int(rng.integers(0, r)); m[i0:i0 + int(rng.integers(1, 60))] = int(rng.integers(0, 2))
        if not m.any(): m[0, 0] = 1
        kw["mask"] = m
    tag = f"{np.dtype(dt).name} {r}x{c}x{nd} e={e} style={style} mask={mk}"
    if "-v" in sys.argv: print(n, tag, flush=True)
    if "-dump" in sys.argv: np.savez("/tmp/fuzz_last.npz", x=x, e=e, m=kw.get("mask", np.zeros(0, np.uint8)))
    r1, b1 = O.encode(x, e, **kw); r2, b2 = S.encode(x, e, **kw)
    n += 1
    if r1 != r2 or (b1 != b2 and not (kind == 'f' and e == 0)):
        bad += 1; print("ENC MISMATCH", tag, r1, r2, len(b1), len(b2)); continue
    if r1: continue
    d1, d2 = O.decode(b1), S.decode(b1)
    ok = d1[0] == d2[0] == 0 and ((d1[2] is None and d2[2] is None) or np.array_equal(d1[2], d2[2]))
    if ok:
        v = (d1[2].reshape(r, c) != 0) if d1[2] is not None else np.ones((r, c), bool)
        v = v[..., None] if nd > 1 else v
        sh = (r, c, nd) if nd > 1 else (r, c)
        a1, a2 = d1[1].reshape(sh), d2[1].reshape(sh)
        ok = np.array_equal(np.where(v, a1, 0).view(np.uint8), np.where(v, a2, 0).view(np.uint8)) if not (kind == 'f' and e == 0) else np.array_equal(np.where(v, a1, 0), np.where(v, a2, 0), equal_nan=True)
    if not ok:
        bad += 1; print("DEC MISMATCH", tag)
    if rng.random() < 0.3 and len(b1) > 200:
        bb = bytearray(b1); bb[int(rng.integers(100, len(bb)))] ^= 1 << int(rng.integers(0, 8))
        if (O.decode(bytes(bb))[0] == 0) != (S.decode(bytes(bb))[0] == 0):
            bad += 1; print("VERDICT MISMATCH on a damaged copy", tag)
f = S.decode_forms()
print("cases", n, "mismatches", bad, "; masked bands cut into blocks by the scan:", f[0])
