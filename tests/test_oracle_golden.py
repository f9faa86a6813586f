"""Pins the CPU restatement (oracle/) against committed golden vectors -- runs with no GPU and
without /root/reference.  Sources of the vectors: tests/golden/make_golden.py."""
import hashlib
import json
import os

import numpy as np
import pytest

import capi
import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_FPL_VEC = json.load(open(os.path.join(GOLD, "fpl_vectors.json")))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def O():
    lib = capi.oracle()
    if lib is None:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(capi.ROOT, "oracle")])
        lib = capi.oracle()
    return lib


def test_more_md_worked_example(O):
    """doc/MORE.md:5-41: 4x4 block, 12 valid pixels; 12 bits -> 18+7 = 25 block bytes at 0.01,
    5 bits -> 8+7 = 15 block bytes at 1.0.  Blob bytes as emitted by the real reference."""
    kat = json.load(open(os.path.join(GOLD, "kat_more_md.json")))
    a = np.array(kat["values"], np.float32).reshape(4, 4)
    m = np.array(kat["mask"], np.uint8).reshape(4, 4)
    expect_q = {"0.01": [591, 979, 1699, 2250, 2929, 1300, 2523, 2854, 0, 851, 2134, 1390],
                "1.0": [6, 10, 17, 23, 29, 13, 25, 29, 0, 9, 21, 14]}
    for e, run in kat["runs"].items():
        rc, size = O.compute_size(a, float(e), mask=m)
        rc2, blob = O.encode(a, float(e), mask=m)
        assert rc == 0 and rc2 == 0
        assert size == run["size"] == len(blob)
        assert blob.hex() == run["blob_hex"]
        # block = last 25 / 15 bytes: flag, float offset, numBits byte, count 12, payload
        nb = {"0.01": 12, "1.0": 5}[e]
        blk = blob[-(7 + (12 * nb + 7) // 8):]
        assert blk[0] == 0x01 and blk[5] == (0x80 | nb) and blk[6] == 12
        bits = int.from_bytes(blk[7:], "little")
        q = [(bits >> (i * nb)) & ((1 << nb) - 1) for i in range(12)]
        assert q == expect_q[e]
        rc, dec, mask = O.decode(blob)
        assert rc == 0 and np.array_equal(mask[0], m)
        assert np.all(np.abs(dec[0, :, :, 0][m == 1] - a[m == 1]) <= float(e) * 1.0001)


def test_js_sanity_blob(O):
    """OtherLanguages/js/tests/sanity.mjs:5-39 -- v5 blob, 30x20, nDepth 3, U8."""
    blob = open(os.path.join(GOLD, "js_sanity_v5.lerc2"), "rb").read()
    rc, info, rng = O.blob_info(blob)
    assert rc == 0
    assert info[0] == 5 and info[1] == 1 and info[2] == 3 and info[3] == 30 and info[4] == 20 and info[5] == 1
    rc, dec, mask = O.decode(blob)
    assert rc == 0 and mask is None
    px = dec[0]
    assert px.reshape(-1)[:6].tolist() == [13, 57, 68, 14, 59, 80]
    assert [int(px[..., k].min()) for k in range(3)] == [0, 30, 60]
    assert int(px.max()) == 89
    rc, mins, maxs = O.data_ranges(blob, 3, 1)
    assert rc == 0 and mins == [0, 30, 60]


def test_california_decode(O):
    """BASELINE config 1 / SURVEY 8d C1: info, range and decoded digest of the reference's sample."""
    blob = open(os.path.join(GOLD, "california_400_400_1_float.lerc2"), "rb").read()
    rc, info, rng = O.blob_info(blob)
    assert rc == 0
    assert info == [3, 6, 1, 400, 400, 1, 58515, 176451, 1, 1, 0]
    assert rng == [-82.97209167480469, 4080.61376953125, 7.5e-05]
    rc, dec, mask = O.decode(blob)
    assert rc == 0
    assert int(mask.sum()) == 58515
    assert sha(dec.tobytes())[:16] == "61e4aa3ffeeccb06"


def test_bluemarble_decode(O):
    blob = open(os.path.join(GOLD, "bluemarble_256_256_3_byte.lerc2"), "rb").read()
    rc, info, rng = O.blob_info(blob)
    assert rc == 0 and info[:6] == [3, 1, 1, 256, 256, 3] and info[6] == 43008
    rc, dec, mask = O.decode(blob)
    assert rc == 0
    assert sha(dec.tobytes())[:16] == "4763435eb56d2b71"


_VEC = json.load(open(os.path.join(GOLD, "ref_vectors.json")))
_CASES = cases.basic_cases()


@pytest.mark.parametrize("idx", range(len(_CASES)), ids=[c[0] for c in _CASES])
def test_matches_reference_vectors(O, idx):
    """Byte identity with the real reference, via digests recorded by make_golden.py."""
    name, arr, kw = _CASES[idx]
    v = _VEC[name]
    kw = dict(kw)
    e = kw.pop("max_z_err")
    assert sha(np.ascontiguousarray(arr).tobytes()) == v["input_sha"], "case generator drifted; regenerate golden"
    rc, size = O.compute_size(arr, e, **kw)
    assert (rc, size) == (v["rc_size"], v["size"])
    rc, blob = O.encode(arr, e, **kw)
    assert rc == v["rc"]
    if rc != 0:
        return
    assert len(blob) == v["size"] and sha(blob) == v["blob_sha"]
    rc, dec, mask = O.decode(blob)
    assert rc == v["dec_rc"] and sha(dec.tobytes()) == v["dec_sha"]
    assert (sha(mask.tobytes()) if mask is not None else None) == v["mask_sha"]
    rc, info, rng = O.blob_info(blob)
    assert info == v["info"] and rng == v["range"]


def test_decode_committed_blobs(O):
    """Decode fixtures: blobs written by the real reference."""
    d = os.path.join(GOLD, "blobs")
    n = 0
    for f in sorted(os.listdir(d)):
        blob = open(os.path.join(d, f), "rb").read()
        rc, dec, mask = O.decode(blob)
        assert rc == 0, f
        key = f[:-6]
        vec = _VEC if key in _VEC else _FPL_VEC    # blobs/fpl-<case>.lerc2: lossless float fixtures
        assert sha(dec.tobytes()) == vec[key[4:] if key not in _VEC else key]["dec_sha"], f
        n += 1
    assert n >= 10


def test_reject_corruption(O):
    blob = bytearray(open(os.path.join(GOLD, "blobs", "mixed-float32.lerc2"), "rb").read())
    rc, _, _ = O.decode(bytes(blob))
    assert rc == 0
    blob[len(blob) // 2] ^= 0x40
    rc, _, _ = O.decode(bytes(blob))
    assert rc == 1    # Fletcher32 mismatch -> Failed
    rc, _, _ = O.decode(bytes(blob[:200]))
    assert rc != 0


def test_lossless_float_vectors(O):
    """tests/golden/fpl_vectors.json (made by the real reference): the oracle's lossless float / double codec"""
    import hashlib
    vec = json.load(open(os.path.join(GOLD, "fpl_vectors.json")))
    cases.check_lossless_float_golden(O, vec, os.path.join(GOLD, "blobs"), lambda b: hashlib.sha256(bytes(b)).hexdigest())


def test_lerc1_world(O):
    """testData/world.lerc1, the reference's legacy-format fixture: info array and ranges as the reference reports them,
    sha256 of the valid pixels as the reference decodes them (tests/golden/make_golden.py copies the file)"""
    blob = open(os.path.join(GOLD, "world.lerc1"), "rb").read()
    assert O.blob_info(blob) == (0, [0, 6, 1, 257, 257, 1, 65025, 63518, 1, 1, 0], [-27.458635330200195, 5474.1728515625, 0.1])
    rc, dec, mask = O.decode(blob)
    assert rc == 0
    m = mask.reshape(257, 257).astype(bool)
    assert int(m.sum()) == 65025
    assert sha(dec.reshape(257, 257)[m].tobytes()) == "74f626d1a4fcf78f1eae5b7cb07f7690a8a0bf76d5d77315b3737a2bae5aae09"


def test_lerc1_written_blobs(O):
    """tests/golden/lerc1_vectors.json: the real reference's reading of the Lerc1 blobs of cases.lerc1_cases"""
    vec = json.load(open(os.path.join(GOLD, "lerc1_vectors.json")))
    for name, blob, nb in cases.lerc1_cases():
        v = vec[name]
        assert sha(blob) == v["blob_sha"], "tests/lerc1_writer.py drifted; regenerate golden (make_golden.py lerc1)"
        rc, info, rng = O.blob_info(blob)
        assert rc == 0 and info == v["info"] and rng == v["range"], name
        assert json.loads(json.dumps(list(O.data_ranges(blob, 1, nb)))) == v["ranges"], name    # (tuples become lists)
        d, dd = O.decode(blob), O.decode(blob, to_double=True)
        assert d[0] == 0 and dd[0] == 0
        assert sha(d[1].tobytes()) == v["dec_sha"] and sha(dd[1].tobytes()) == v["dec_double_sha"], name
        assert (sha(d[2].tobytes()) if d[2] is not None else None) == v["mask_sha"], name
