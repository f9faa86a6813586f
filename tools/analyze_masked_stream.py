"""A masked band's block stream under the scanning decoder's rules (tile_fast_decode_scan.hip, MODE 1): candidates with any count byte,
the flag byte behind a candidate's end, one-byte blocks found by flooding runs from block ends -- how many false block starts survive?

  python tools/analyze_masked_stream.py [rows cols]
"""
import struct
import sys

import numpy as np

sys.path.insert(0, ".")
from tests import capi                      # noqa: E402
from lerc_amd import synth                  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
a = synth.c2_float32(rows, cols).numpy()
ii = np.arange(rows).reshape(-1, 1); jj = np.arange(cols).reshape(1, -1)
kind = sys.argv[3] if len(sys.argv) > 3 else "stripes"
if kind == "stripes":
    mk = (((ii // 97) + (jj // 131)) % 10 != 0).astype(np.uint8)
elif kind == "half":
    mk = ((jj < cols // 2) | (ii % 200 < 40)).astype(np.uint8) * np.ones((rows, 1), np.uint8)
else:
    r = np.random.default_rng(5); mk = (r.random((rows // 16, cols // 16)) < 0.7).astype(np.uint8).repeat(16, 0).repeat(16, 1)
mk = np.ascontiguousarray(np.broadcast_to(mk, (rows, cols)))
ref = capi.ref()
rc, blob = ref.encode(a, 0.01, mask=mk)
assert rc == 0
b = np.frombuffer(blob, np.uint8)
version = struct.unpack_from("<i", blob, 6)[0]
hdr = 90 if version >= 6 else 66
nmask = struct.unpack_from("<i", blob, hdr)[0]
data_begin = hdr + 4 + nmask + 8 + 1
off_bytes = {0: 4, 1: 2, 2: 1}
pattern = 14 if version >= 5 else 15
step = 2 if version >= 5 else 1

def bitlen(n):
    return int(n).bit_length()

def block_len(pos):
    f = int(b[pos]); mode = f & 3
    if mode == 2:
        return 1
    if mode == 3:
        return 1 + off_bytes[f >> 6]
    if mode == 0:
        return None
    ob = off_bytes[f >> 6]
    t = int(b[pos + 1 + ob]); nb = t & 31; lut = (t >> 5) & 1; cnt = int(b[pos + 2 + ob])
    if lut:
        nlut = int(b[pos + 3 + ob]) - 1
        return 4 + ob + (nlut * nb + 7) // 8 + (cnt * bitlen(nlut) + 7) // 8
    return 3 + ob + (cnt * nb + 7) // 8

pos = data_begin; starts = []
nblk = (rows // 8) * (cols // 8)
for k in range(nblk):
    starts.append(pos)
    ln = block_len(pos)
    assert ln, "raw block"
    pos += ln
assert pos == len(b), (pos, len(b))
N = len(b) + 700
is_start = np.zeros(N, bool); is_start[starts] = True
modes = np.bincount(b[starts] & 3, minlength=4)
print("blob", len(b), "mask bytes", nmask, "blocks", nblk, "modes", modes.tolist(), "pieces", len(b) / 32768)

def sig_ok(prev, cur):
    return cur == prev or cur == ((prev + step) & pattern) or cur == 0

t = b[:-1]; c = b[1:]
cnt_ok = ((t & 0xC0) == 0x80) & ((t & 31) != 0) & (c >= 1) & (c <= 64)
qs = np.nonzero(cnt_ok)[0] + 1
START = np.zeros(N, bool); END = np.zeros(N, bool)
n_cand = n_false = n_false_sig = 0
for q in qs:
    if q < data_begin:
        continue
    tt = int(b[q - 1]); nb = tt & 31; lut = (tt >> 5) & 1; cnt = int(b[q])
    for tc, ob in off_bytes.items():
        p = q - 2 - ob
        if p < data_begin:
            continue
        f = int(b[p])
        if (f & 3) != 1 or (f >> 6) != tc or (version >= 5 and (f & 4)):
            continue
        if lut:
            nlut = (int(b[q + 1]) - 1) & 0xFF if q + 1 < len(b) else 0
            if not (1 <= nlut <= 254):
                continue
            ln = 4 + ob + (nlut * nb + 7) // 8 + (cnt * bitlen(nlut) + 7) // 8
        else:
            ln = 3 + ob + (cnt * nb + 7) // 8
        if ln > 1 + cnt * 4 or p + ln > len(b):
            continue
        n_cand += 1
        if not is_start[p]:
            n_false += 1
        e = p + ln
        if e < len(b):
            nf = int(b[e])
            if not sig_ok((f >> 2) & pattern, (nf >> 2) & pattern) or (version >= 5 and (nf & 4)):
                continue
        if not is_start[p]:
            n_false_sig += 1
        START[p] = True; END[e] = True
START[data_begin] = True; END[data_begin] = True
per = 32768 / len(b)
print("candidates", n_cand, "false", n_false, "false after the flag byte check", n_false_sig, "per piece", n_false_sig * per)
S0 = START & END
print("survivors", int(S0.sum()), "false", int((S0 & ~is_start).sum()), "per piece", (S0 & ~is_start).sum() * per, "true blocks not among them", int((is_start & ~S0).sum()))

# the flood: one-byte blocks (mode 2, bit 2 clear from codec 5 on); a run goes on while the signature does
bb = np.concatenate([b, np.zeros(N - len(b), np.uint8)])
m2 = ((bb & 3) == 2) & (((bb & 4) == 0) | (version < 5))
m2[len(b):] = False; m2[:data_begin] = False
sg = (bb >> 2) & pattern
cont = np.zeros(N, bool)
cont[1:] = m2[1:] & m2[:-1] & ((sg[1:] == sg[:-1]) | (sg[1:] == ((sg[:-1] + step) & pattern)) | (sg[1:] == 0))
F = END & m2
while True:
    G = F.copy(); G[1:] |= F[:-1] & cont[1:]
    if (G == F).all():
        break
    F = G
S1 = (START | F) & (END | np.concatenate([[False], F[:-1]]))
print("after the flood: survivors", int(S1.sum()), "false", int((S1 & ~is_start).sum()), "per piece", (S1 & ~is_start).sum() * per,
      "true blocks not among them", int((is_start & ~S1).sum()), "per piece", (is_start & ~S1).sum() * per)
miss = np.nonzero(is_start & ~S1)[0]
print("   modes of the missing", np.bincount(b[miss] & 3, minlength=4).tolist(), "; modes of their predecessors",
      np.bincount(b[[starts[np.searchsorted(starts, m) - 1] for m in miss if m != starts[0]]] & 3, minlength=4).tolist())
# a second round: a block behind a run has its END now; what begins where IT ends?  (its END was set when it was a candidate: all candidates' are)
pieces = int(np.ceil(len(b) / 32768))
bad_pieces = sum(1 for k in range(pieces) if ((S1 ^ is_start)[k * 32768:(k + 1) * 32768]).any())
print("pieces whose survivors are not exactly their blocks:", bad_pieces, "of", pieces)

# the flood seeded by SURVIVORS' ends only (a candidate's end is kept with its start), round after round
cand_end = {}
for q in qs:
    if q < data_begin:
        continue
    tt = int(b[q - 1]); nb = tt & 31; lut = (tt >> 5) & 1; cnt = int(b[q])
    for tc, ob in off_bytes.items():
        p = q - 2 - ob
        if p < data_begin or not START[p]:
            continue
        f = int(b[p])
        if (f & 3) != 1 or (f >> 6) != tc:
            continue
        ln = block_len(p) if not lut or q + 1 < len(b) else None
        if ln:
            cand_end.setdefault(p, p + ln)
S = S0.copy()
for rnd in range(1, 6):
    seeds = np.zeros(N, bool)
    for p in np.nonzero(S)[0]:
        e = cand_end.get(int(p), int(p) + 1 if m2[p] else None)
        if p == data_begin and e is None:
            e = p + (block_len(p) or 0)
        if e is not None and e < N:
            seeds[e] = True
    F = seeds & m2
    while True:
        G = F.copy(); G[1:] |= F[:-1] & cont[1:]
        if (G == F).all():
            break
        F = G
    S = S | ((START | F) & (END | np.concatenate([[False], F[:-1]])) & (F | np.concatenate([[False], F[:-1]]) | S))
    bad_pieces = sum(1 for k in range(pieces) if ((S ^ is_start)[k * 32768:(k + 1) * 32768]).any())
    print("round", rnd, ": false", int((S & ~is_start).sum()), "missing", int((is_start & ~S).sum()), "pieces not exact", bad_pieces)
