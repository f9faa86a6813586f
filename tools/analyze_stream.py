"""What does a band's block stream look like, and how selective is the byte pattern the scanning decoder looks for?

Encodes a window of one of the BASELINE rasters with the real reference (oracle/_ref, CPU), walks the block stream the way
Lerc2::ReadTiles does (Lerc2.cpp:1672-1713, :2025-2110) and prints: block modes, bits per value, offset types, look-up tables; then
runs the scanning decoder's candidate rules over the same bytes (tile_fast_decode_scan.hip) and counts what they find that is no block.

  python tools/analyze_stream.py [c2|c3] [rows cols]
"""
import collections
import struct
import sys

import numpy as np

sys.path.insert(0, ".")
from tests import capi                      # noqa: E402
from lerc_amd import synth                  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
if which == "c2":
    a = synth.c2_float32(rows, cols).numpy(); mz = 0.01; tb = 4
    off_bytes = {0: 4, 1: 2, 2: 1}
else:
    a = synth.c3_uint16(rows, cols).numpy().astype(np.uint16); mz = 0.0; tb = 2
    off_bytes = {0: 2, 1: 1}
ref = capi.ref()
rc, blob = ref.encode(a, mz)
assert rc == 0
b = np.frombuffer(blob, np.uint8)
version = struct.unpack_from("<i", blob, 6)[0]
hdr = 90 if version >= 6 else 66
data_begin = hdr + 4 + 2 * tb + 1
print("blob", len(blob), "version", version, "bytes/block", (len(blob) - data_begin) / ((rows // 8) * (cols // 8)))

def bitlen(n):
    return int(n).bit_length()

pos = data_begin
starts = []
modes = collections.Counter(); nbs = collections.Counter(); tcs = collections.Counter(); luts = 0
nblk = (rows // 8) * (cols // 8)
for k in range(nblk):
    flag = int(b[pos]); mode = flag & 3; tc = flag >> 6
    starts.append(pos)
    modes[mode] += 1
    if mode == 0:
        pos += 1 + 64 * tb
    elif mode == 2:
        pos += 1
    elif mode == 3:
        tcs[tc] += 1; pos += 1 + off_bytes[tc]
    else:
        tcs[tc] += 1
        ob = off_bytes[tc]
        t = int(b[pos + 1 + ob]); nb = t & 31; lut = (t >> 5) & 1
        assert (t >> 6) == 2 and int(b[pos + 2 + ob]) == 64, (pos, t)
        nbs[nb] += 1
        if lut:
            luts += 1
            nlut = int(b[pos + 3 + ob]) - 1
            pos += 4 + ob + (nlut * nb + 7) // 8 + (64 * bitlen(nlut) + 7) // 8
        else:
            pos += 3 + ob + 8 * nb
assert pos == len(blob), (pos, len(blob))
print("modes", dict(modes), "offset types", dict(tcs), "luts", luts)
print("bits", sorted(nbs.items()))

# the scanning decoder's rule: a byte 64 behind a byte 10?nnnnn (n != 0); flag byte 2 + offB in front with mode 1, that offset type
is_start = np.zeros(len(b) + 600, bool); is_start[starts] = True
cnt = (b[1:] == 64) & ((b[:-1] & 0xC0) == 0x80) & ((b[:-1] & 31) != 0)
qs = np.nonzero(cnt)[0] + 1
cand = collections.Counter()
START = np.zeros(len(b) + 600, bool); END = np.zeros(len(b) + 600, bool)
n_false = 0
for q in qs:
    if q < data_begin:
        continue
    t = int(b[q - 1]); nb = t & 31; lut = (t >> 5) & 1
    for tc, ob in off_bytes.items():
        p = q - 2 - ob
        if p < data_begin:
            continue
        f = int(b[p])
        if (f & 3) != 1 or (f >> 6) != tc or (version >= 5 and (f & 4)):
            continue
        if lut:
            nlut = (int(b[q + 1]) - 1) & 0xFF if q + 1 < len(b) else 0
            if not (1 <= nlut - 0 and nlut - 1 < 254):
                continue
            ln = 4 + ob + (nlut * nb + 7) // 8 + (64 * bitlen(nlut) + 7) // 8
        else:
            ln = 3 + ob + 8 * nb
        if ln > 1 + 64 * tb or p + ln > len(b):
            continue
        START[p] = True; END[p + ln] = True
        if not is_start[p]:
            n_false += 1
START[data_begin] = True; END[data_begin] = True
S = START & END
true_stuffed = sum(1 for s in starts if (int(b[s]) & 3) == 1)
print("count bytes found", len(qs), "candidates", int(START.sum()), "false candidates", n_false, "per 32 KiB", n_false * 32768 / len(b))
print("survivors", int(S.sum()), "false survivors", int((S & ~is_start[:len(S)]).sum()), "true blocks missing from survivors",
      int((is_start[:len(S)] & ~S).sum()), "of", nblk)
