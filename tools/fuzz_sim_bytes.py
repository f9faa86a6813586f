"""Soak test on the emulator (no GPU): random 8-bit rasters -- whole 8 x 8 blocks or ragged, 1 .. 4 values per pixel, smooth /
stepped / flat / noisy / regional content, lossless and lossy -- through the emulator build of the product library and the
oracle; prints every mismatch.  These inputs go through the byte kernels of round 2 (lane-per-block tile sizes, grouped
histograms, 16-byte statistics), the one-pass Huffman packer and the chained Huffman decoders.
    python tools/fuzz_sim_bytes.py [seed] [seconds]"""
import sys, os, time, subprocess
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import numpy as np, capi, cases
subprocess.check_call(["make", "-s", "-C", os.path.join(capi.ROOT, "lerc_amd", "csrc"), "sim", "-j8"])
S = capi.sim(); O = capi.oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120
t0 = time.time(); n = 0; modes = {}
while time.time() - t0 < budget:
    dt = np.uint8 if rng.random() < 0.6 else np.int8
    r, c = int(rng.integers(8, 200)), int(rng.integers(8, 260))
    if rng.random() < 0.75: r -= r % 8; c -= c % 8
    r, c = max(r, 8), max(c, 8)
    nd = int(rng.choice([1, 1, 2, 3, 3, 4]))
    planes = []
    for m in range(nd):
        style = int(rng.integers(0, 7))
        base = np.cumsum(rng.integers(-2, 3, (r, c)), axis=1) + np.cumsum(rng.integers(-1, 2, (r, 1)), axis=0) * 2
        if style == 0: x = base
        elif style == 1: x = np.repeat(rng.integers(0, 9, (r, (c + 7) // 8)), 8, axis=1)[:, :c] * 20 + (rng.random((r, c)) < 0.03) * 7
        elif style == 2: x = rng.integers(0, 256, (r, c))
        elif style == 3: x = base // 4 + (rng.random((r, c)) < 0.5)
        elif style == 4: x = np.full((r, c), int(rng.integers(0, 256)))
        elif style == 5 and planes: x = planes[-1] + rng.integers(0, 3, (r, c))
        else: x = base + (rng.random((r, c)) < 0.02) * np.minimum(rng.geometric(0.07, (r, c)), 120) * rng.choice([-1, 1], (r, c))
        if rng.random() < 0.3: x[: r // 2, : c // 3] = int(rng.integers(0, 4)) * (rng.random() < 0.5)
        planes.append(x)
    a = np.stack(planes, axis=-1) & 255
    arr = a.astype(np.uint8).view(dt) if dt is np.int8 else a.astype(np.uint8)
    if nd == 1: arr = arr[:, :, 0]
    e = float(rng.choice([0, 0, 0, 1, 3]))
    kw = dict(n_depth=nd) if nd > 1 else {}
    tag = f"{np.dtype(dt).name} {r}x{c}x{nd} e={e}"
    r1, b1 = O.encode(arr, e, **kw); r2, b2 = S.encode(arr, e, **kw)
    if r1 != r2 or b1 != b2:
        print("ENC MISMATCH", tag, r1, r2, len(b1), len(b2)); continue
    if r1 == 0:
        d1, d2 = O.decode(b1), S.decode(b1)
        if d1[0] != d2[0] or not np.array_equal(np.asarray(d1[1]), np.asarray(d2[1])): print("DEC MISMATCH", tag, d1[0], d2[0])
        if e == 0:
            at = 90 + 4 + 2 * nd
            if len(b1) > at + 1: modes[b1[at + 1] if b1[at] == 0 else 9] = modes.get(b1[at + 1] if b1[at] == 0 else 9, 0) + 1
    n += 1
print("cases", n, "done; lossless image modes (0 tiling, 1 delta Huffman, 2 Huffman, 9 one sweep):", dict(sorted(modes.items())))
