"""Shared case matrix for differential tests: (name, array, kwargs) tuples that force every encoder
branch listed in SURVEY.md App. D-2 (LUT / const / raw blocks, 16x16 retry, TryRaiseMaxZError,
bIsInt promotion, integer lossy, bit-plane cheat code, one-sweep, Huffman / DeltaHuffman, masks,
nDepth > 1 with slice-difference encoding, multi band, NaN)."""
import numpy as np

ALL_DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32, np.float64]


def _cast(x, dt):
    dt = np.dtype(dt)
    if dt.kind in "iu":
        info = np.iinfo(dt)
        return np.clip(np.rint(x), info.min, info.max).astype(dt)
    return x.astype(dt)


def terrain(rows, cols, rng, amp=500.0, base=1000.0, sigma=1.0):
    i = np.arange(rows, dtype=np.float64)[:, None]
    j = np.arange(cols, dtype=np.float64)[None, :]
    return base + amp * np.sin(j / 37.0) * np.cos(i / 23.0) + sigma * rng.standard_normal((rows, cols))


def mixed_regions(rows, cols, rng, dt):
    """flat | stepped | noisy thirds -> const, LUT, bit-stuffed and (lossless float) raw blocks."""
    x = terrain(rows, cols, rng, amp=300.0, base=400.0, sigma=3.0)
    c1, c2 = cols // 3, 2 * cols // 3
    x[:, :c1] = 77.0
    x[:, c1:c2] = np.floor(x[:, c1:c2] / 64.0) * 64.0
    x[: rows // 4, :c1] = 0.0
    return _cast(x, dt)


def basic_cases():
    """Small but branch-complete; used by the CPU differential tests and (subset) GPU parity tests."""
    rng = np.random.default_rng(7)
    cases = []
    shapes = [(8, 8), (64, 64), (129, 257), (1, 257), (257, 1), (5, 3), (100, 100), (16, 24)]
    for dt in ALL_DTYPES:
        kind = np.dtype(dt).kind
        small = np.dtype(dt).itemsize == 1
        for (r, c) in shapes:
            amp, base = (40.0, 60.0) if small else (500.0, 1000.0)
            x = terrain(r, c, rng, amp=amp, base=base, sigma=1.0 if kind == "f" else 2.0)
            errs = [0.01, 0.5, 5.0] if kind == "f" else [0, 1, 5]
            for e in errs:
                cases.append((f"terrain-{np.dtype(dt).name}-{r}x{c}-e{e}", _cast(x, dt), dict(max_z_err=e)))
        cases.append((f"mixed-{np.dtype(dt).name}", mixed_regions(100, 143, rng, dt), dict(max_z_err=0 if kind != "f" else 0.01)))
        cases.append((f"mixed-lossy-{np.dtype(dt).name}", mixed_regions(77, 143, rng, dt), dict(max_z_err=2.0)))
        cases.append((f"const-{np.dtype(dt).name}", np.full((33, 47), 5, dt), dict(max_z_err=0.5)))
        cases.append((f"zeros-{np.dtype(dt).name}", np.zeros((33, 47), dt), dict(max_z_err=0.5)))
    # ---- float specials
    f = np.float32
    x = terrain(256, 256, rng)
    cases.append(("f32-allint", np.rint(x).astype(f), dict(max_z_err=0.01)))
    cases.append(("f32-allint-e2.7", np.rint(x).astype(f), dict(max_z_err=2.7)))
    cases.append(("f32-round1", np.round(1000 + 50 * np.sin(np.arange(256)[None, :] / 30.0) + rng.standard_normal((256, 256)), 1).astype(f), dict(max_z_err=0.01)))
    cases.append(("f32-round2", np.round(1000 + 50 * np.sin(np.arange(256)[None, :] / 30.0) + rng.standard_normal((256, 256)), 2).astype(f), dict(max_z_err=0.001)))
    cases.append(("f64-round1", np.round(terrain(100, 120, rng), 1), dict(max_z_err=0.01)))
    y = np.full((256, 256), 100.0, f)
    y[rng.random((256, 256)) < 0.05] += 0.02
    cases.append(("f32-mb16", y, dict(max_z_err=0.01)))
    cases.append(("f32-smooth-mb16", (100 + 0.001 * np.arange(256)[None, :] * np.ones((256, 1))).astype(f), dict(max_z_err=0.01)))
    z = terrain(64, 80, rng).astype(f)
    z[3, 5] = np.nan
    z[10:20, 30:40] = np.nan
    cases.append(("f32-nan", z, dict(max_z_err=0.01)))
    cases.append(("f32-huge-range", (terrain(64, 64, rng) * 1e30).astype(f), dict(max_z_err=0.01)))
    cases.append(("f32-tiny-err", terrain(64, 64, rng).astype(f), dict(max_z_err=1e-7)))
    cases.append(("f64-wide", terrain(64, 64, rng) * 1e6, dict(max_z_err=1e-4)))
    zr = terrain(64, 96, rng).astype(f)
    zr[5, 7] = 3e30
    zr[40:43, 50:70] *= 1e25
    cases.append(("f32-some-raw", zr, dict(max_z_err=0.01)))
    cases.append(("f32-neg", (-terrain(70, 90, rng)).astype(f), dict(max_z_err=0.1)))
    cases.append(("f32-int16-offsets", (np.rint(terrain(64, 64, rng, amp=100, base=0, sigma=0) / 8) * 8 + 0.25 * rng.integers(0, 3, (64, 64))).astype(f), dict(max_z_err=0.01)))
    # ---- integer specials
    cases.append(("u16-lossy5", _cast(terrain(128, 128, rng), np.uint16), dict(max_z_err=5)))
    cases.append(("u16-777", rng.integers(0, 65535, (128, 128)).astype(np.uint16), dict(max_z_err=777)))
    cases.append(("i32-777", (rng.integers(-1000, 1000, (128, 128)) * 16 + rng.integers(0, 16, (128, 128))).astype(np.int32), dict(max_z_err=777)))
    cases.append(("i32-wide", rng.integers(-2**31, 2**31 - 1, (64, 64)).astype(np.int32), dict(max_z_err=0)))
    cases.append(("u32-wide", rng.integers(0, 2**32 - 1, (64, 64)).astype(np.uint32), dict(max_z_err=0)))
    cases.append(("u32-big-offsets", (rng.integers(0, 1000, (64, 64)) + 3_000_000_000).astype(np.uint32), dict(max_z_err=0)))
    cases.append(("i16-neg-offsets", (rng.integers(0, 100, (64, 64)) - 100).astype(np.int16), dict(max_z_err=0)))
    # ---- 8-bit: Huffman / DeltaHuffman / one sweep
    i = np.arange(96)[:, None]; j = np.arange(128)[None, :]
    smooth = 128 + 100 * np.sin(j / 40.0) * np.cos(i / 31.0)
    rgb = np.stack([_cast(smooth + 4 * rng.standard_normal(smooth.shape) + 5 * k, np.uint8) for k in range(3)], axis=-1)
    cases.append(("u8-rgb-deltahuff", rgb, dict(max_z_err=0, n_depth=3)))
    cases.append(("u8-rgb-lossy", rgb, dict(max_z_err=2, n_depth=3)))
    cases.append(("u8-random-onesweep", rng.integers(0, 256, (64, 64, 3)).astype(np.uint8), dict(max_z_err=0, n_depth=3)))
    cases.append(("u8-fewvals-huff", rng.choice(np.array([3, 50, 51, 200], np.uint8), (80, 90), p=[0.7, 0.1, 0.1, 0.1]), dict(max_z_err=0)))
    cases.append(("i8-smooth", _cast(smooth - 128 + 3 * rng.standard_normal(smooth.shape), np.int8), dict(max_z_err=0)))
    cases.append(("i8-rand30", (rng.integers(0, 30, (257, 713 // 8, 3))).astype(np.int8), dict(max_z_err=0, n_depth=3)))
    cases.append(("u8-twovals", rng.choice(np.array([0, 255], np.uint8), (64, 64)), dict(max_z_err=0)))
    # ---- nDepth > 1 (slice difference encoding for integer lossless)
    base = terrain(70, 90, rng)
    cube = np.stack([base + 3 * k + rng.standard_normal(base.shape) for k in range(4)], axis=-1)
    cases.append(("u16-depth4", _cast(cube, np.uint16), dict(max_z_err=0, n_depth=4)))
    cases.append(("i32-depth4", _cast(cube * 1000, np.int32), dict(max_z_err=0, n_depth=4)))
    cases.append(("i16-depth2-lossy", _cast(cube[..., :2], np.int16), dict(max_z_err=3, n_depth=2)))
    cases.append(("f32-depth3", cube[..., :3].astype(f), dict(max_z_err=0.01, n_depth=3)))
    cube_same = np.repeat(_cast(base, np.uint16)[..., None], 3, axis=-1)
    cases.append(("u16-depth3-identical", cube_same, dict(max_z_err=0, n_depth=3)))
    # ---- masks
    m = np.ones((129, 257), np.uint8)
    m[::10, :] = 0
    m[:, ::13] = 0
    cases.append(("f32-mask-grid", terrain(129, 257, rng).astype(f), dict(max_z_err=0.01, mask=m)))
    mr = (rng.random((129, 257)) > 0.3).astype(np.uint8)
    cases.append(("u16-mask-random", _cast(terrain(129, 257, rng), np.uint16), dict(max_z_err=0, mask=mr)))
    cases.append(("u8-mask-random", _cast(smooth[:96, :128], np.uint8), dict(max_z_err=0, mask=mr[:96, :128].copy())))
    mh = np.ones((64, 64), np.uint8); mh[20:50, 10:60] = 0
    cases.append(("f32-mask-hole", terrain(64, 64, rng).astype(f), dict(max_z_err=0.1, mask=mh)))
    cases.append(("f32-mask-allinvalid", terrain(64, 64, rng).astype(f), dict(max_z_err=0.1, mask=np.zeros((64, 64), np.uint8))))
    cases.append(("u16-depth3-mask", _cast(cube[..., :3], np.uint16), dict(max_z_err=0, n_depth=3, mask=(rng.random((70, 90)) > 0.2).astype(np.uint8))))
    # ---- multi band
    bands = np.stack([terrain(50, 60, rng, base=1000 + 100 * b) for b in range(3)]).astype(f)
    cases.append(("f32-3bands", bands, dict(max_z_err=0.01, n_bands=3)))
    mb = np.stack([(rng.random((50, 60)) > 0.2).astype(np.uint8) for _ in range(3)])
    cases.append(("f32-3bands-3masks", bands, dict(max_z_err=0.01, n_bands=3, mask=mb)))
    cases.append(("f32-3bands-1mask", bands, dict(max_z_err=0.01, n_bands=3, mask=mb[0].copy())))
    cases.append(("u8-3bands", _cast(bands / 8, np.uint8), dict(max_z_err=0, n_bands=3)))
    return cases


def nodata_fuzz_cases(n_iter, seed=33):
    """Rasters with a noData value (lerc_encode_4D): inside / below / above the data range or absent from the data,
    single pixels and whole pixels (all depths), with NaNs, masks, nDepth 1..3.  -> [(name, arr, maxZErr, kw)]"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n_iter):
        dt = [np.float32, np.float64, np.uint8, np.int16, np.uint16, np.int32][rng.integers(0, 6)]
        nd = int(rng.choice([1, 1, 2, 3]))
        r, c = int(rng.integers(8, 60)), int(rng.integers(8, 60))
        kind = np.dtype(dt).kind
        x = terrain(r, c, rng, amp=float(rng.choice([5, 100])), base=float(rng.choice([0, 100, 1000])), sigma=float(rng.choice([0, 0.5, 3])))
        x = np.stack([x + k for k in range(nd)], axis=-1) if nd > 1 else x
        if kind == "f" and rng.random() < 0.4:
            x = np.round(x)
        x = _cast(x, dt)
        style = rng.integers(0, 5)
        flat = x.reshape(-1)
        if style == 0:
            nod = float(flat[rng.integers(0, flat.size)])
        elif style == 1:
            nod = float(-9999 if kind != "u" else 0)
        elif style == 2:
            nod = float(32000 if np.dtype(dt).itemsize > 1 else 250)
        elif style == 3:
            nod = float(np.float32(-3.4e38)) if kind == "f" else float(np.iinfo(dt).min)
        else:
            nod = float(np.iinfo(dt).max) if kind != "f" else 1e30
        sel = rng.random(x.shape) < rng.choice([0.0, 0.05, 0.3])
        x = x.copy()
        x[sel] = np.array(nod).astype(dt)
        if nd > 1 and rng.random() < 0.5:
            x[rng.random((r, c)) < 0.1] = np.array(nod).astype(dt)
        if kind == "f" and rng.random() < 0.3:
            x[rng.random(x.shape) < 0.02] = np.nan
        e = float(rng.choice([0, 0.01, 0.5, 2])) if kind == "f" else float(rng.choice([0, 1, 3]))
        kw = dict(n_depth=nd, no_data=nod)
        if rng.random() < 0.3:
            kw["mask"] = (rng.random((r, c)) > 0.2).astype(np.uint8)
        out.append((f"nodata{it}-{np.dtype(dt).name}-{r}x{c}x{nd}-e{e}-style{style}", x, e, kw))
    x = np.stack([terrain(20, 30, rng) for _ in range(2)])[..., None].repeat(2, axis=-1).astype(np.float32)
    x[0, 3:5, 4:9, 1] = -9999
    out.append(("nodata-two-bands", x, 0.1, dict(n_depth=2, n_bands=2, no_data=[-9999, None])))
    return out


def check_nodata_case(T, P, name, arr, e, kw, same):
    """T: trusted library (reference or oracle), P: the library under test.  (Float rasters whose error bound the
    noData filter drops to 0 go through the lossless float codec.)"""
    s1, s2 = T.compute_size(arr, e, **kw), P.compute_size(arr, e, **kw)
    r1, b1 = T.encode(arr, e, **kw)
    r2, b2 = P.encode(arr, e, **kw)
    assert s1 == s2 and r1 == r2 and len(b1) == len(b2), name
    if b1 != b2:    # a float band the filter made lossless: the reference leaves a few padding bytes to chance
        a1, a2 = bytearray(b1), bytearray(b2)
        for k in lossless_float_dont_care(b1, arr.dtype.itemsize):
            a1[k] = a2[k] = 0
        assert a1 == a2, name
    if r1 != 0:
        return
    d1, d2 = T.decode(b1, with_nodata=True), P.decode(b1, with_nodata=True)
    assert d1[0] == d2[0], name
    for a, b in zip(d1[1:], d2[1:]):
        assert same(a, b), name
    assert T.decode(b1)[0] == P.decode(b1)[0], name    # the plain entry point refuses blobs that carry a noData value


def old_codec_cases(n_iter, seed=71):
    """lerc_encodeForVersion with codec 2, 3, 4, 5 (Lerc::EncodeInternal_v5; codec 2: no checksum, old bit layout): all dtypes, lossless float included (raw
    blocks before codec 6), NaNs (-> mask), masks, nDepth (codec >= 4), several bands.  -> [(name, arr, version, e, kw)]"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n_iter):
        dt = ALL_DTYPES[rng.integers(0, 8)]
        ver = int(rng.choice([2, 3, 4, 5]))
        nd = int(rng.choice([1, 1, 2, 3])) if ver >= 4 else 1
        nb = int(rng.choice([1, 1, 2]))
        r, c = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        kind = np.dtype(dt).kind
        planes = []
        for _ in range(nb):
            x = terrain(r, c, rng, amp=float(rng.choice([5, 100, 500])), base=float(rng.choice([0, 100, 1000])), sigma=float(rng.choice([0, 0.5, 3])))
            x = np.stack([x + 3 * k for k in range(nd)], axis=-1) if nd > 1 else x
            style = rng.integers(0, 4)
            if style == 1:
                x = np.round(x)
            if style == 2:
                x = np.round(x, 1)
            if np.dtype(dt).itemsize == 1:
                x = x / 8
            planes.append(x)
        x = _cast(np.stack(planes) if nb > 1 else planes[0], dt).copy()
        if kind == "f" and rng.random() < 0.3:
            sel = rng.random((nb, r, c) if nb > 1 else (r, c)) < 0.05
            x[sel] = np.nan    # every depth of the pixel: it leaves the mask (a NaN in some depths only becomes -FLT_MAX)
        e = float(rng.choice([0, 0.001, 0.01, 0.5, 1, 3])) if kind == "f" else float(rng.choice([0, 0, 1, 4]))
        kw = dict(n_depth=nd, n_bands=nb)
        if rng.random() < 0.3:
            m = (rng.random((nb, r, c)) > 0.25).astype(np.uint8)
            kw["mask"] = m if (nb > 1 and rng.random() < 0.5) else m[0]
        out.append((f"old{it}-v{ver}-{np.dtype(dt).name}-{nb}x{r}x{c}x{nd}-e{e}", x, ver, e, kw))
    x = _cast(terrain(40, 50, rng), np.float32)
    out.append(("old-version-6", x, 6, 0.01, {}))
    out.append(("old-version-7", x, 7, 0.01, {}))
    out.append(("old-version-1", x, 1, 0.01, {}))
    out.append(("old-v3-depth2", np.stack([x, x], axis=-1), 3, 0.01, dict(n_depth=2)))
    return out


def check_old_codec_case(T, P, name, arr, ver, e, kw, same):
    """T: trusted library, P: library under test.  Codec 2 (pre-v3 bit layout) is not built: WrongParam there."""
    t = T.encode_for_version(arr, ver, e, **kw)
    p = P.encode_for_version(arr, ver, e, **kw)
    assert t == p, (name, t[:3], p[:3])
    if t[0] == 0:
        d1, d2 = T.decode(t[3]), P.decode(t[3])
        assert d1[0] == d2[0] == 0 and same(d1[1], d2[1]) and same(d1[2], d2[2]), name


def deep_pixel_cases(seed=17):
    """Hyperspectral-like rasters: more values per pixel than a workgroup has threads (the per-depth ranges of
    Lerc2 v4+, Lerc2.cpp:1095-1166, are then gathered without the LDS table), masks, lossy and lossless.
    -> [(name, arr, max_z_err, kw)]"""
    rng = np.random.default_rng(seed)
    out = []
    for dt, (r, c, nd), e in [(np.uint8, (8, 8, 300), 0), (np.float32, (9, 11, 300), 0.01), (np.int16, (16, 16, 257), 1),
                              (np.uint8, (1, 1, 300), 0), (np.float64, (3, 2, 1000), 0.5), (np.uint16, (24, 40, 256), 0),
                              (np.float32, (5, 7, 513), 0)]:
        x = np.stack([terrain(r, c, rng, amp=40, base=100, sigma=1.0) + 3 * k for k in range(nd)], axis=-1)
        x = _cast(x, dt)
        m = (rng.random((r, c)) > 0.2).astype(np.uint8)
        out.append((f"deep-{np.dtype(dt).name}-{r}x{c}x{nd}", x, float(e), dict(n_depth=nd)))
        out.append((f"deep-{np.dtype(dt).name}-{r}x{c}x{nd}-mask", x, float(e), dict(n_depth=nd, mask=m)))
    return out


def check_deep_pixel_case(T, P, name, arr, e, kw, same):
    """T: trusted library, P: library under test: same blob, same pixels / mask / ranges back"""
    r1, b1 = T.encode(arr, e, **kw)
    r2, b2 = P.encode(arr, e, **kw)
    lossless_float = arr.dtype.kind == "f" and e == 0
    assert r1 == r2 == 0 and len(b1) == len(b2), (name, r1, r2, len(b1), len(b2))
    if not lossless_float:    # (there the reference leaves pad bytes undefined)
        assert bytes(b1) == bytes(b2), name
    d1, d2 = T.decode(b1), P.decode(b1)
    assert d1[0] == d2[0] == 0 and same(d1[1], d2[1]) and same(d1[2], d2[2]), name
    d3 = P.decode(b2)
    assert d3[0] == 0 and same(d1[1], d3[1]), name
    assert T.blob_info(b1) == P.blob_info(b1), name
    nd = kw["n_depth"]
    assert T.data_ranges(b1, nd, 1) == P.data_ranges(b1, nd, 1), name


def lerc1_cases(seed=23):
    """Legacy Lerc1 blobs from tests/lerc1_writer.py (the reference has a Lerc1 decoder, no encoder, and one fixture):
    masks and none, tile grids with remainder rows / columns, raw / constant / zero / bit-stuffed tiles with float, int16 and
    int8 offsets, several bands.  -> [(name, blob, n_bands)]"""
    import lerc1_writer
    rng = np.random.default_rng(seed)
    out = []
    for it, (h, w, tiles, e, masked, nb, raw_every) in enumerate([
            (40, 56, (5, 7), 0.1, True, 1, 0), (33, 47, (4, 5), 0.01, True, 1, 3), (64, 64, (8, 8), 0.5, False, 1, 0),
            (17, 300, (2, 9), 1.0, True, 2, 5), (100, 90, (1, 1), 0.05, False, 1, 0), (57, 61, (7, 6), 0.25, True, 3, 4),
            (8, 8, (1, 1), 0.0, False, 1, 0), (129, 130, (16, 16), 0.1, True, 1, 7), (23, 5, (23, 5), 0.1, True, 1, 2)]):
        bands = []
        for b in range(nb):
            z = terrain(h, w, rng, amp=float(rng.choice([5, 300, 3000])), base=float(rng.choice([-50, 0, 1000])), sigma=0.3)
            if it % 3 == 1:
                z = np.floor(z)                      # integer offsets -> int8 / int16 offset types
            z[: h // 4, : w // 3] = 0.0              # all-zero tiles
            z[h // 2:, w // 2:] = 17.0 if it % 2 else -1234.5    # constant tiles
            bands.append(z.astype(np.float32))
        mask = None
        if masked:
            mask = (rng.random((h, w)) > 0.15).astype(np.uint8)
            mask[h // 3: h // 3 + 6, :] = 0
            mask[:, w // 5: w // 5 + 3] = 1
        out.append((f"lerc1-{h}x{w}-t{tiles[0]}x{tiles[1]}-e{e}-b{nb}-{'mask' if masked else 'full'}",
                    lerc1_writer.write(bands, mask, e, tiles, rng, raw_every), nb))
    return out


def check_lerc1_case(T, P, name, blob, n_bands, same):
    """T: trusted library, P: library under test: info, ranges, pixels (float and double) and mask"""
    ti, pi = T.blob_info(blob), P.blob_info(blob)
    assert ti == pi and ti[0] == 0 and ti[1][0] == 0 and ti[1][5] == n_bands, (name, ti, pi)
    assert T.data_ranges(blob, 1, n_bands) == P.data_ranges(blob, 1, n_bands), name
    for dbl in (False, True):
        d1, d2 = T.decode(blob, to_double=dbl), P.decode(blob, to_double=dbl)
        assert d1[0] == d2[0] == 0, (name, d1[0], d2[0])
        assert same(d1[1], d2[1]) and same(d1[2], d2[2]), name


def lossless_float_cases(n_iter, seed=91, max_side=150):
    """maxZErr == 0 on float / double rasters (Lerc2 IEM_DeltaDeltaHuffman, fpl_*): smooth, noisy, stepped and random
    data so that every predictor (none / rows / rows + columns), difference order and plane coding (Huffman, one value,
    stored, PackBits) shows up; NaNs, masks, nDepth, bands.  -> [(name, arr, kw)]"""
    rng = np.random.default_rng(seed)
    out = []
    for it in range(n_iter):
        dt = [np.float32, np.float64][rng.integers(0, 2)]
        nd = int(rng.choice([1, 1, 1, 2, 3]))
        nb = int(rng.choice([1, 1, 1, 2]))
        r, c = int(rng.integers(1, max_side)), int(rng.integers(1, max_side))
        style = int(rng.integers(0, 7))
        planes = []
        for _ in range(nb):
            if style == 0:
                x = terrain(r, c, rng, amp=float(rng.choice([5, 500])), base=float(rng.choice([0, 1000])), sigma=float(rng.choice([0, 0.01, 1])))
            elif style == 1:
                x = rng.random((r, c)) * float(rng.choice([1, 1e6, 1e-6]))
            elif style == 2:
                x = np.floor(terrain(r, c, rng, sigma=0) / 50) * 50 + 0.25
            elif style == 3:
                x = np.cumsum(np.cumsum(rng.standard_normal((r, c)), axis=0), axis=1) * 1e-3
            elif style == 4:
                x = np.where(rng.random((r, c)) < 0.9, 1.5, rng.random((r, c)))
            elif style == 5:
                x = np.add.outer(np.arange(r) * 0.125, np.arange(c) * 0.5) + 0.1
            else:
                x = terrain(r, c, rng, sigma=0.0) * 1e-3 + 1e5
            x = np.stack([x * (1 + 0.01 * k) + k for k in range(nd)], axis=-1) if nd > 1 else x
            planes.append(x)
        x = np.ascontiguousarray(np.stack(planes) if nb > 1 else planes[0]).astype(dt)
        if rng.random() < 0.2:
            sel = rng.random((nb, r, c) if nb > 1 else (r, c)) < 0.03
            x[sel] = np.nan
        kw = dict(n_depth=nd, n_bands=nb)
        if rng.random() < 0.25:
            m = (rng.random((nb, r, c)) > 0.2).astype(np.uint8)
            kw["mask"] = m if (nb > 1 and rng.random() < 0.5) else m[0]
        out.append((f"fpl{it}-{np.dtype(dt).name}-{nb}x{r}x{c}x{nd}-style{style}", x, kw))
    # dimensions the streaming kernels like (they have to leave such a band alone)
    out.append(("fpl-streaming-dims-float32", (terrain(64, 512, rng) + rng.standard_normal((64, 512))).astype(np.float32), {}))
    out.append(("fpl-streaming-dims-float64", (terrain(32, 512, rng, sigma=0.001)).astype(np.float64), {}))
    return out


def lossless_float_dont_care(blob, itemsize):
    """Byte positions of a codec-6 blob whose value the reference leaves to chance: the read-ahead word behind every
    Huffman coded byte plane of a lossless float band (fpl_EsriHuffman.cpp:383-437 mallocs the plane's buffer and
    Huffman::PushValue only clears words it starts, so that last word is whatever the heap held), and the checksum
    over them.  The decoder never looks at those bits."""
    import struct
    skip = []
    s = 0
    while s + 90 <= len(blob) and blob[s:s + 6] == b"Lerc2 ":
        version = struct.unpack_from("<i", blob, s + 6)[0]
        n_depth, n_valid, _mb, blob_size, dt, _more = struct.unpack_from("<6i", blob, s + 22)
        max_z_err, z_min, z_max = struct.unpack_from("<3d", blob, s + 50)
        if version == 6 and dt >= 6 and max_z_err == 0 and n_valid > 0 and z_min != z_max:
            at = s + 90
            at += 4 + struct.unpack_from("<i", blob, at)[0]
            rng = blob[at:at + 2 * n_depth * itemsize]
            at += 2 * n_depth * itemsize
            if rng[:n_depth * itemsize] != rng[n_depth * itemsize:] and blob[at] == 0 and blob[at + 1] == 3:
                at += 3    # one-sweep flag, image mode, predictor code
                pads = []
                for _ in range(itemsize):
                    size = struct.unpack_from("<I", blob, at + 2)[0]
                    if blob[at + 6] == 0:
                        pads += list(range(at + 6 + size - 4, at + 6 + size))
                    at += 6 + size
                if pads:
                    skip += pads + list(range(s + 10, s + 14))
        s += blob_size
    return skip


def check_lossless_float_case(T, P, name, arr, kw, same):
    """T: trusted library, P: library under test; blobs byte-identical up to the reference's uninitialised padding
    (lossless_float_dont_care), decodes bit-identical (every pixel, valid or not)."""
    s1, s2 = T.compute_size(arr, 0, **kw), P.compute_size(arr, 0, **kw)
    r1, b1 = T.encode(arr, 0, **kw)
    r2, b2 = P.encode(arr, 0, **kw)
    assert s1 == s2 and r1 == r2, (name, s1, s2, r1, r2)
    assert len(b1) == len(b2), (name, len(b1), len(b2))
    if b1 != b2:
        a1, a2 = bytearray(b1), bytearray(b2)
        for k in lossless_float_dont_care(b1, arr.dtype.itemsize):
            a1[k] = a2[k] = 0
        assert a1 == a2, (name, len(b1), [i for i in range(len(a1)) if a1[i] != a2[i]][:8])
        d3 = P.decode(b2)    # our own blob (with its zero padding and its own checksum) reads back the same
        assert d3[0] == 0 and same(d3[1], P.decode(b1)[1]), name
    if r1 == 0:
        d1, d2 = T.decode(b1), P.decode(b1)
        assert d1[0] == d2[0] == 0 and same(d1[1], d2[1]) and same(d1[2], d2[2]), name


def check_lossless_float_golden(P, vec, blob_dir, sha, n_iter=60, max_side=90):
    """P against tests/golden/fpl_vectors.json (made by the real reference): status, size, blob (up to the reference's
    uninitialised padding), decode of the reference's own blobs and of P's."""
    import os
    n_checked = 0
    for name, arr, kw in lossless_float_cases(n_iter, max_side=max_side):
        v = vec[name]
        assert P.compute_size(arr, 0, **kw) == (v["rc_size"], v["size"]), name
        rc, b = P.encode(arr, 0, **kw)
        assert rc == v["rc"], name
        if rc != 0:
            continue
        a = bytearray(b)
        for k in lossless_float_dont_care(b, arr.dtype.itemsize):
            a[k] = 0
        assert len(b) == v["blob_len"] and sha(a) == v["blob_sha_masked"], name
        d = P.decode(b)
        assert d[0] == v["dec_rc"] and sha(d[1].tobytes()) == v["dec_sha"], name
        assert (sha(d[2].tobytes()) if d[2] is not None else None) == v["mask_sha"], name
        if "blob_file" in v:
            ref_blob = open(os.path.join(blob_dir, v["blob_file"]), "rb").read()
            d = P.decode(ref_blob)
            assert d[0] == v["dec_rc"] and sha(d[1].tobytes()) == v["dec_sha"], name
            n_checked += 1
    assert n_checked >= 10


def damaged_blob_cases(T, n_damage, seed=123):
    """Blobs of every path (streaming, any-width streaming, masked, ragged, nDepth, 8-bit Huffman, lossless float, raw
    blocks, tiny) with one byte flipped / overwritten, a short stretch zeroed, or the end cut off.  -> [(name, blob, n_bands hint)]
    T encodes the intact blobs."""
    rng = np.random.default_rng(seed)
    blobs = []
    f = np.float32
    blobs.append(("stream-f32", T.encode(_cast(terrain(64, 512, rng, sigma=1.5), f), 0.01)[1]))
    blobs.append(("stream-u16", T.encode(_cast(terrain(32, 512, rng, sigma=1.5), np.uint16), 0)[1]))
    blobs.append(("anywidth-f32", T.encode(_cast(terrain(40, 328, rng, sigma=1.5), f), 0.01)[1]))
    blobs.append(("anywidth-f64", T.encode(_cast(terrain(24, 200, rng, sigma=1.5), np.float64), 0.001)[1]))
    m = (rng.random((70, 90)) > 0.2).astype(np.uint8)
    blobs.append(("masked-f32", T.encode(_cast(terrain(70, 90, rng), f), 0.01, mask=m)[1]))
    blobs.append(("ragged-i16", T.encode(_cast(terrain(37, 53, rng), np.int16), 0)[1]))
    blobs.append(("depth3-u16", T.encode(_cast(np.stack([terrain(40, 48, rng) + k for k in range(3)], -1), np.uint16), 0, n_depth=3)[1]))
    blobs.append(("huffman-u8", T.encode(_cast(terrain(80, 96, rng, amp=50, base=100, sigma=2), np.uint8), 0)[1]))
    blobs.append(("lossless-f32", T.encode(_cast(terrain(48, 64, rng, sigma=0.01), f), 0)[1]))
    zr = _cast(terrain(32, 512, rng), f)
    zr[::8, ::8] *= 1e20
    blobs.append(("raw-blocks-f32", T.encode(zr, 0.01)[1]))
    blobs.append(("two-bands-f32", T.encode(np.stack([_cast(terrain(24, 32, rng), f)] * 2), 0.01, n_bands=2)[1]))
    out = []
    for name, blob in blobs:
        assert len(blob) > 100, name
        for t in range(n_damage):
            b = bytearray(blob)
            how = int(rng.integers(0, 4))
            k = int(rng.integers(0, len(b)))
            if how == 0:
                b[k] ^= 1 << int(rng.integers(0, 8))
            elif how == 1:
                b[k] = int(rng.integers(0, 256))
            elif how == 2:
                n_zero = min(len(b) - k, int(rng.integers(2, 40)))
                b[k:k + n_zero] = bytes(n_zero)
            else:
                b = b[:max(24, k)]
            out.append((f"{name}-{t}-how{how}-at{k}", bytes(b)))
    return out


def check_damaged_blob(T, P, name, blob, same):
    """same verdict as the trusted decoder; where both accept the blob (damage in bytes nobody reads), the same pixels"""
    d1, d2 = T.decode(blob), P.decode(blob)
    assert (d1[0] == 0) == (d2[0] == 0), (name, d1[0], d2[0])
    if d1[0] == 0:
        assert same(d1[1], d2[1]) and same(d1[2], d2[2]), name


def huffman_stress_cases(scale=1):
    """8-bit rasters whose Huffman code books hold code words longer than the decoders' 12-bit look-up table, in streams
    of many speculative sub-sequences: a smooth signal with rare large steps (delta mode, long codes for the steps), a
    geometric value distribution (plain mode), three values per pixel, and stretches of a single very short code next to
    noise (sub-sequences that hold hundreds of symbols next to ones that hold few)."""
    rng = np.random.default_rng(77)
    out = []
    h, w = 192 * scale, 256 * scale
    walk = np.cumsum(rng.choice([-1, 0, 0, 0, 1], size=(h, w)), axis=1)
    jump = (rng.random((h, w)) < 0.02) * np.minimum(rng.geometric(0.07, (h, w)), 120) * rng.choice([-1, 1], (h, w))
    out.append(("huff-steps-u8", ((walk + np.cumsum(jump, axis=1)) & 255).astype(np.uint8), {}))
    out.append(("huff-steps-i8", (((walk + np.cumsum(jump, axis=1)) & 255) - 128).astype(np.int8), {}))
    geo = np.minimum(rng.geometric(0.45, (h, w)) - 1, 255)
    perm = rng.permutation(256)
    out.append(("huff-geometric-u8", perm[geo].astype(np.uint8), {}))
    x3 = np.stack([(walk + 7 * c + np.cumsum(jump, axis=1) * (c + 1)) & 255 for c in range(3)], axis=-1).astype(np.uint8)
    out.append(("huff-steps-u8-depth3", x3, dict(n_depth=3)))
    flat = np.full((h, w), 9, np.uint8)
    flat[:, w // 3: w // 2] = rng.integers(0, 256, (h, w // 2 - w // 3))
    flat[rng.random((h, w)) < 0.0005] = 200
    out.append(("huff-flat-and-noise-u8", flat, {}))
    return out


def byte_tiling_cases():
    """8-bit rasters of whole 8 x 8 blocks, every pixel valid, 1 .. 4 values per pixel: the encoder prices the tiling with a
    lane per block position (k_tile_sizes_bytes).  Content that makes blocks constant, bit-stuffed, LUT coded (long runs of
    equal values), raw, and -- for more than one value per pixel -- cheaper as a difference to the slice in front."""
    rng = np.random.default_rng(123)
    out = []
    for dt in (np.uint8, np.int8):
        for nd in (1, 2, 3, 4):
            h, w = 64, 96 + 8 * nd
            base = np.cumsum(rng.integers(-2, 3, (h, w)), axis=1) + np.cumsum(rng.integers(-1, 2, (h, 1)), axis=0) * 3
            planes = []
            for m in range(nd):
                kind = (m + (0 if dt is np.uint8 else 1)) % 4
                if kind == 0: x = base + 11 * m                                             # tracks the slice in front: differences are small
                elif kind == 1: x = np.repeat(rng.integers(0, 9, (h, w // 8)), 8, axis=1) * 20 + (rng.random((h, w)) < 0.03) * 7    # runs: LUT
                elif kind == 2: x = rng.integers(0, 256, (h, w))                              # noise: raw blocks
                else: x = base // 4 + (rng.random((h, w)) < 0.5)                               # few bits
                planes.append(x)
            a = np.stack(planes, axis=-1)
            a[:16, :32] = 5                                                                  # constant blocks
            a[16:24, :32] = 0                                                                # all zero blocks
            arr = (a & 255).astype(np.uint8).view(dt) if dt is np.int8 else (a & 255).astype(np.uint8)
            out.append((f"bytes-{np.dtype(dt).name}-depth{nd}", arr if nd > 1 else arr[:, :, 0], dict(n_depth=nd) if nd > 1 else {}))
            # regions with code books of their own: here the tiling beats one Huffman code for the whole raster, so the
            # sizes per block position end up as offsets in the blob
            h, w = 64, 128
            planes = []
            for m in range(nd):
                x = np.zeros((h, w), np.int64)
                x[:, :w // 4] = rng.integers(0, 4, (h, w // 4)) + 40 * m
                x[:, w // 4:w // 2] = rng.integers(0, 256, (h, w // 4))
                x[:, w // 2:3 * w // 4] = 77
                x[:, 3 * w // 4:] = np.repeat(rng.integers(0, 6, (h, w // 32)), 8, axis=1) * 37 + (rng.random((h, w // 4)) < 0.04) * 3
                if m % 2 == 1: x[:, w // 4:w // 2] = planes[m - 1][:, w // 4:w // 2] + rng.integers(0, 3, (h, w // 4))
                planes.append(x)
            a = np.stack(planes, axis=-1)
            arr = (a & 255).astype(np.uint8).view(dt) if dt is np.int8 else (a & 255).astype(np.uint8)
            out.append((f"bytes-regions-{np.dtype(dt).name}-depth{nd}", arr if nd > 1 else arr[:, :, 0], dict(n_depth=nd) if nd > 1 else {}))
    return out
