"""Differential test: oracle restatement vs the REAL reference build (oracle/_ref/libLercRef.so).
Skipped when the reference build is absent (it is compiled by `make -C oracle ref` /
__graft_entry__.build() wherever /root/reference exists, and travels to the GPU box as a .so)."""
import numpy as np
import pytest

import capi
import cases

R = capi.ref()
pytestmark = pytest.mark.skipif(R is None, reason="oracle/_ref/libLercRef.so not built")


@pytest.fixture(scope="module")
def O():
    return capi.oracle()


def _same(a, b):
    if a is None or b is None:
        return a is None and b is None
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def test_case_matrix(O):
    for name, arr, kw in cases.basic_cases():
        kw = dict(kw)
        e = kw.pop("max_z_err")
        assert R.compute_size(arr, e, **kw) == O.compute_size(arr, e, **kw), name
        r1, b1 = R.encode(arr, e, **kw)
        r2, b2 = O.encode(arr, e, **kw)
        assert r1 == r2 and b1 == b2, name
        if r1 == 0:
            d1, d2 = R.decode(b1), O.decode(b1)
            assert d1[0] == d2[0] and _same(d1[1], d2[1]) and _same(d1[2], d2[2]), name
            assert R.blob_info(b1) == O.blob_info(b1), name
            assert R.decode(b1, to_double=True)[0] == O.decode(b1, to_double=True)[0]
            assert _same(R.decode(b1, to_double=True)[1], O.decode(b1, to_double=True)[1]), name


def test_random_fuzz(O):
    """Random dtype / shape / error / mask / depth combinations."""
    rng = np.random.default_rng(20260926)
    for it in range(300):
        dt = cases.ALL_DTYPES[rng.integers(0, 8)]
        r, c = int(rng.integers(1, 70)), int(rng.integers(1, 70))
        nd = int(rng.choice([1, 1, 1, 2, 3]))
        kind = np.dtype(dt).kind
        style = rng.integers(0, 4)
        base = cases.terrain(r, c, rng, amp=float(rng.choice([5, 50, 500])), base=float(rng.choice([0, 100, 1000])),
                             sigma=float(rng.choice([0, 0.3, 3])))
        x = np.stack([base + k for k in range(nd)], axis=-1) if nd > 1 else base
        if style == 1:
            x = np.floor(x / 16) * 16
        if style == 2:
            x = np.round(x, 1)
        if np.dtype(dt).itemsize == 1:
            x = x / 8
        x = cases._cast(x, dt)
        e = float(rng.choice([0, 0.001, 0.01, 0.5, 1, 3])) if kind == "f" else float(rng.choice([0, 0, 1, 4]))
        if kind == "f" and e == 0:
            e = 0.01    # lossless float (fpl path) is out of scope
        kw = dict(n_depth=nd)
        if rng.random() < 0.3:
            kw["mask"] = (rng.random((r, c)) > rng.random() * 0.6).astype(np.uint8)
        tag = f"fuzz{it} {np.dtype(dt).name} {r}x{c}x{nd} e={e} style={style} mask={'mask' in kw}"
        assert R.compute_size(x, e, **kw) == O.compute_size(x, e, **kw), tag
        r1, b1 = R.encode(x, e, **kw)
        r2, b2 = O.encode(x, e, **kw)
        assert r1 == r2 and b1 == b2, tag
        if r1 == 0:
            d1, d2 = R.decode(b1), O.decode(b1)
            assert d1[0] == d2[0] and _same(d1[1], d2[1]) and _same(d1[2], d2[2]), tag


def test_nodata_4d(O):
    """_4D entry points with a noData value: nDepth > 1 mixes of valid / noData."""
    rng = np.random.default_rng(5)
    for dt, nd_val in ((np.float32, -9999.0), (np.int16, -9999), (np.uint8, 255), (np.float64, 1e30)):
        for e in (0, 0.01, 2):
            if np.dtype(dt).kind == "f" and e == 0:
                continue
            x = cases.terrain(40, 50, rng, amp=30, base=100, sigma=1)
            cube = np.stack([x, x + 1, x + 2], axis=-1)
            cube = cases._cast(cube, dt)
            cube[5:10, 5:10, :] = nd_val          # whole pixel noData -> moves into the mask
            cube[20:25, 20:25, 1] = nd_val        # mixed -> needs noData passed through
            kw = dict(n_depth=3, no_data=nd_val)
            assert R.compute_size(cube, e, **kw) == O.compute_size(cube, e, **kw)
            r1, b1 = R.encode(cube, e, **kw)
            r2, b2 = O.encode(cube, e, **kw)
            assert r1 == r2 and b1 == b2, (dt, e)
            if r1 == 0:
                d1 = R.decode(b1, with_nodata=True)
                d2 = O.decode(b1, with_nodata=True)
                assert d1[0] == d2[0] and _same(d1[1], d2[1]) and _same(d1[2], d2[2])
                assert np.array_equal(d1[3], d2[3]) and np.array_equal(d1[4], d2[4])
                assert R.decode(b1)[0] == O.decode(b1)[0]    # HasNoData(5) when the caller omits the arrays


def test_error_codes(O):
    a = np.zeros((4, 4), np.float32)
    for lib in (R, O):
        rc, _ = lib.encode(a, -1.0)
        assert rc == 2
        rc, blob = lib.encode(a + np.arange(4, dtype=np.float32), 0.01, buf_size=20)
        assert rc == 3 and blob == b""
    n = np.full((8, 8, 2), 1.0, np.float32)
    n[1, 1, 0] = np.nan
    assert R.encode(n, 0.01, n_depth=2)[0] == O.encode(n, 0.01, n_depth=2)[0] == 4


def test_fletcher32(O):
    import ctypes as ct
    O.lib.orc_fletcher32.restype = ct.c_uint
    O.lib.orc_fletcher32.argtypes = [ct.c_void_p, ct.c_int]
    rng = np.random.default_rng(1)
    for n in (1, 2, 3, 717, 718, 719, 100001):
        a = np.zeros((1, n), np.uint8)
        a[:] = rng.integers(0, 256, n)
        rc, blob = R.encode(a, 0)
        assert rc == 0
        b = np.frombuffer(blob, np.uint8)
        stored = int(np.frombuffer(blob[10:14], np.uint32)[0])
        assert O.lib.orc_fletcher32(b[14:].ctypes.data, len(blob) - 14) == stored


def test_many_values_per_pixel(O):
    """nDepth of several hundred: the restatement against the real reference"""
    for name, arr, e, kw in cases.deep_pixel_cases():
        cases.check_deep_pixel_case(R, O, name, arr, e, kw, _same)


def test_old_codec_versions(O):
    """lerc_encodeForVersion / lerc_computeCompressedSizeForVersion for codec 3..5 (Lerc.cpp:526-624)."""
    for name, arr, ver, e, kw in cases.old_codec_cases(250):
        cases.check_old_codec_case(R, O, name, arr, ver, e, kw, _same)


def test_lossless_float(O):
    """maxZErr == 0 on float / double: the restated fpl_* codec against the real one (blobs equal up to the bytes the
    reference leaves uninitialised), including the multi-band nDepth > 1 size query quirk."""
    for name, arr, kw in cases.lossless_float_cases(150, seed=94, max_side=200):
        cases.check_lossless_float_case(R, O, name, arr, kw, _same)


def test_nodata_fuzz(O):
    """the shared noData case generator (also used against the product), lossless float bands included"""
    for name, arr, e, kw in cases.nodata_fuzz_cases(250, seed=34):
        cases.check_nodata_case(R, O, name, arr, e, kw, _same)


def test_lerc1_world(O):
    """the legacy Lerc1 fixture: info, ranges, pixels (as float, double and int16) and mask as the real reference gives them"""
    import os
    blob = open(os.path.join(capi.ROOT, "tests", "golden", "world.lerc1"), "rb").read()
    assert R.blob_info(blob) == O.blob_info(blob)
    assert R.data_ranges(blob, 1, 1) == O.data_ranges(blob, 1, 1)
    for kw in ({}, {"to_double": True}):
        d1, d2 = R.decode(blob, **kw), O.decode(blob, **kw)
        assert d1[0] == d2[0] == 0 and _same(d1[1], d2[1]) and _same(d1[2], d2[2])


def test_lerc1_written_blobs(O):
    """Lerc1 blobs written by tests/lerc1_writer.py: the real reference reads them, and the restatement reads the same"""
    for name, blob, nb in cases.lerc1_cases():
        cases.check_lerc1_case(R, O, name, blob, nb, _same)
