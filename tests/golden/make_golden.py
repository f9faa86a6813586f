#!/usr/bin/env python3
"""Regenerates tests/golden/* from the REAL reference (needs /root/reference and oracle/_ref).

Run in the build container only:   make -C oracle ref && python tests/golden/make_golden.py

Outputs (data only -- inputs and expected outputs, never reference source text):
  js_sanity_v5.lerc2        the inline golden blob of OtherLanguages/js/tests/sanity.mjs:6 (number list -> bytes)
  california_400_400_1_float.lerc2, bluemarble_256_256_3_byte.lerc2, world.lerc1   the reference's own testData blobs
  kat_more_md.json          doc/MORE.md:5-41 worked 4x4 example: inputs + reference blobs (hex) at maxZErr 0.01 / 1.0
  ref_vectors.json          for every case of tests/cases.py: reference status, blob size, sha256(blob),
                            sha256(decoded bytes), sha256(mask), getBlobInfo arrays
  blobs/<case>.lerc2        full reference blobs for a handful of cases (decode fixtures)
  fpl_vectors.json, blobs/fpl-*.lerc2   the same for lossless float / double rasters (cases.lossless_float_cases)
  lerc1_vectors.json                    the reference's reading of the Lerc1 blobs of cases.lerc1_cases (tests/lerc1_writer.py)
"""
import hashlib
import json
import os
import re
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import capi    # noqa: E402
import cases   # noqa: E402

REF_ROOT = os.environ.get("LERC_REF_ROOT", "/root/reference")


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def main():
    R = capi.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    # 1. JS inline golden blob
    txt = open(os.path.join(REF_ROOT, "OtherLanguages/js/tests/sanity.mjs")).read()
    m = re.search(r'const data4D =\s*"([0-9,]+)"', txt)
    blob = bytes(int(x) for x in m.group(1).split(","))
    open(os.path.join(HERE, "js_sanity_v5.lerc2"), "wb").write(blob)
    # 2. testData blobs
    for f in ("california_400_400_1_float.lerc2", "bluemarble_256_256_3_byte.lerc2", "world.lerc1"):
        shutil.copyfile(os.path.join(REF_ROOT, "testData", f), os.path.join(HERE, f))
    # 3. MORE.md worked example
    vals = [1234.1234, 1241.8741, 1256.2759, 1267.2950, 1280.8725, 1248.2917, 1272.7511, 1279.3802,
            0, 1222.2943, 1239.3072, 0, 1264.9720, 1250.0852, 0, 0]
    msk = [1, 1, 1, 1, 1, 1, 1, 1, 0, 1, 1, 0, 1, 1, 0, 0]
    a = np.array(vals, np.float32).reshape(4, 4)
    mk = np.array(msk, np.uint8).reshape(4, 4)
    kat = {"values": vals, "mask": msk, "runs": {}}
    for e in (0.01, 1.0):
        rc, b = R.encode(a, e, mask=mk)
        assert rc == 0
        kat["runs"][str(e)] = {"blob_hex": b.hex(), "size": len(b)}
    json.dump(kat, open(os.path.join(HERE, "kat_more_md.json"), "w"), indent=1)
    # 4. reference vectors for the shared case matrix
    os.makedirs(os.path.join(HERE, "blobs"), exist_ok=True)
    keep = {"mixed-uint16", "mixed-float32", "f32-mb16", "f32-round1", "f32-allint", "u8-rgb-deltahuff",
            "u8-fewvals-huff", "u16-depth4", "f32-mask-grid", "u16-mask-random", "f32-3bands-3masks",
            "u8-random-onesweep", "f32-some-raw", "i8-smooth", "u8-mask-random"}
    vec = {}
    for name, arr, kw in cases.basic_cases():
        kw = dict(kw)
        e = kw.pop("max_z_err")
        rc_s, size = R.compute_size(arr, e, **kw)
        rc, b = R.encode(arr, e, **kw)
        ent = {"rc_size": rc_s, "size": size, "rc": rc, "input_sha": sha(np.ascontiguousarray(arr).tobytes())}
        if rc == 0:
            drc, dec, dm = R.decode(b)
            irc, info, rng = R.blob_info(b)
            ent.update(blob_sha=sha(b), dec_rc=drc, dec_sha=sha(dec.tobytes()),
                       mask_sha=sha(dm.tobytes()) if dm is not None else None, info=info, range=rng)
            if name in keep:
                open(os.path.join(HERE, "blobs", name + ".lerc2"), "wb").write(b)
        vec[name] = ent
    json.dump(vec, open(os.path.join(HERE, "ref_vectors.json"), "w"), indent=0, sort_keys=True)
    print("wrote", len(vec), "vectors")


def masked_sha(blob, itemsize):
    """sha256 of a blob with the bytes the reference leaves uninitialised (cases.lossless_float_dont_care) set to 0"""
    a = bytearray(blob)
    for k in cases.lossless_float_dont_care(blob, itemsize):
        a[k] = 0
    return sha(a)


def main_lossless_float():
    """fpl_vectors.json + blobs/fpl-*.lerc2: lossless float / double (maxZErr 0) cases of cases.lossless_float_cases"""
    R = capi.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    vec = {}
    kept = 0
    for i, (name, arr, kw) in enumerate(cases.lossless_float_cases(60, max_side=90)):
        rc_s, size = R.compute_size(arr, 0, **kw)
        rc, b = R.encode(arr, 0, **kw)
        ent = {"rc_size": rc_s, "size": size, "rc": rc}
        if rc == 0:
            drc, dec, dm = R.decode(b)
            ent.update(blob_len=len(b), blob_sha_masked=masked_sha(b, arr.dtype.itemsize), dec_rc=drc, dec_sha=sha(dec.tobytes()),
                       mask_sha=sha(dm.tobytes()) if dm is not None else None)
            if len(b) < 40000 and kept < 16 and i % 3 == 0:
                open(os.path.join(HERE, "blobs", "fpl-%s.lerc2" % name), "wb").write(b)
                ent["blob_file"] = "fpl-%s.lerc2" % name
                kept += 1
        vec[name] = ent
    json.dump(vec, open(os.path.join(HERE, "fpl_vectors.json"), "w"), indent=0, sort_keys=True)
    print("wrote", len(vec), "lossless float vectors,", kept, "blobs")


def main_lerc1():
    """lerc1_vectors.json: what the real reference makes of the Lerc1 blobs of cases.lerc1_cases (written by
    tests/lerc1_writer.py: the reference has a Lerc1 decoder and no encoder) -- blob digest (so that a drifting writer
    is noticed), info, ranges, digests of the decoded pixels (float and double; the harness presets the output, so pixels
    that are not valid are the same everywhere) and of the mask"""
    R = capi.ref()
    assert R is not None, "build oracle/_ref first (make -C oracle ref)"
    vec = {}
    for name, blob, nb in cases.lerc1_cases():
        rc, info, rng = R.blob_info(blob)
        d = R.decode(blob)
        dd = R.decode(blob, to_double=True)
        assert rc == 0 and d[0] == 0 and dd[0] == 0, name
        vec[name] = {"blob_sha": sha(blob), "info": info, "range": rng, "ranges": list(R.data_ranges(blob, 1, nb)),
                     "dec_sha": sha(d[1].tobytes()), "dec_double_sha": sha(dd[1].tobytes()),
                     "mask_sha": sha(d[2].tobytes()) if d[2] is not None else None}
    json.dump(vec, open(os.path.join(HERE, "lerc1_vectors.json"), "w"), indent=0, sort_keys=True)
    print("wrote", len(vec), "Lerc1 vectors")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "lossless-float":
        main_lossless_float()
    elif len(sys.argv) > 1 and sys.argv[1] == "lerc1":
        main_lerc1()
    else:
        main()
        main_lossless_float()
        main_lerc1()
