"""A writer for the legacy Lerc1 ("CntZImage") format -- test infrastructure only.

The reference ships a Lerc1 decoder and no encoder (src/LercLib/Lerc1Decode/), and one Lerc1 fixture
(testData/world.lerc1).  To pin the oracle's and the product's Lerc1 decoders on more than that one blob, this module
writes blobs from the format as the decoder reads it (CntZImage.cpp:74-480, BitStuffer.cpp:32-112):

  "CntZImage " | int32 version = 11 | int32 type = 8 | int32 height | int32 width | float64 maxZError
  count part: int32 nTilesVert = 0 | int32 nTilesHori = 0 | int32 numBytes | float32 maxValInImg | RLE of the bit mask
              (numBytes = 0: every count is maxValInImg)
  z part:     int32 nTilesVert | int32 nTilesHori | int32 numBytes | float32 maxValInImg | tiles, row major; a tile row /
              column of the remainder (height % nTilesVert, width % nTilesHori) follows the regular ones
  a z tile:   flag byte: low 6 bits 0 raw floats of the valid pixels / 1 offset + bit-stuffed / 2 all zero / 3 constant;
              high 2 bits: type of the offset (0 float32, 1 int16, 2 int8)
  bit stuffer: byte = numBits | (n's type << 6: 0 uint32, 1 uint16, 2 uint8) | n | values MSB first in 32-bit words, the
              last word shifted down by the bytes it does not need and cut short by them
  further bands: header + z part

The tests hand the blobs to the real reference (tests/test_oracle_vs_reference.py), to the oracle and to the product.
"""
import struct

import numpy as np


def rle_of(bits: bytes, rng) -> bytes:
    """A valid stream for the RLE decoder (RLE.cpp:255-318): int16 count > 0 + that many bytes, or -count + one byte to
    repeat; -32768 ends it.  Runs are cut at random so that both kinds and odd lengths show up."""
    out = bytearray()
    i, n = 0, len(bits)
    while i < n:
        run = 1
        while i + run < n and bits[i + run] == bits[i] and run < 32767:
            run += 1
        if run >= 3 and rng.random() < 0.8:
            out += struct.pack("<h", -run) + bits[i:i + 1]
            i += run
        else:
            k = int(min(n - i, rng.integers(1, 200)))
            out += struct.pack("<h", k) + bits[i:i + k]
            i += k
    return bytes(out + struct.pack("<h", -32768))


def stuff_bits(values, nb: int) -> bytes:
    """BitStuffer::write's layout (BitStuffer.cpp:114-157 as the reader :32-112 undoes it)"""
    n = len(values)
    byte = nb | (2 << 6 if n < 256 else (1 << 6 if n < 65536 else 0))
    out = bytearray([byte]) + (struct.pack("<B", n) if n < 256 else struct.pack("<H", n) if n < 65536 else struct.pack("<I", n))
    if nb == 0 or n == 0:
        return bytes(out)
    acc = 0
    for v in values:
        acc = (acc << nb) | int(v)
    total = n * nb
    pad = (-total) % 32
    acc <<= pad
    words = [(acc >> (32 * k)) & 0xFFFFFFFF for k in range((total + pad) // 32 - 1, -1, -1)]
    tail_bits = total & 31
    tail_bytes = (tail_bits + 7) >> 3
    drop = 4 - tail_bytes if tail_bytes else 0
    words[-1] >>= 8 * drop
    raw = b"".join(struct.pack("<I", w) for w in words)
    return bytes(out) + raw[:len(raw) - drop]


def z_tile(z, valid, max_z_err: float, mode: str) -> bytes:
    """z, valid: 2-D arrays of the tile.  mode: 'auto' | 'raw'"""
    v = z[valid > 0].astype(np.float32)
    if v.size == 0 or np.all(v == 0):
        return bytes([2])
    lo, hi = float(v.min()), float(v.max())
    if lo == hi:
        return _with_offset(3, lo) if mode != "raw" else bytes([0]) + v.tobytes()
    if mode == "raw" or max_z_err == 0:
        return bytes([0]) + v.tobytes()
    q = np.floor((v.astype(np.float64) - lo) / (2 * max_z_err) + 0.5)
    if q.max() >= 2 ** 30:
        return bytes([0]) + v.tobytes()
    nb = max(1, int(q.max()).bit_length())
    return _with_offset(1, lo) + stuff_bits(q.astype(np.int64), nb)


def _with_offset(flag: int, off: float) -> bytes:
    if off == int(off) and -128 <= off <= 127:
        return bytes([flag | (2 << 6)]) + struct.pack("<b", int(off))
    if off == int(off) and -32768 <= off <= 32767:
        return bytes([flag | (1 << 6)]) + struct.pack("<h", int(off))
    return bytes([flag]) + struct.pack("<f", off)


def write(bands, mask, max_z_err: float, n_tiles, rng, raw_every: int = 0) -> bytes:
    """bands: list of 2-D float32 arrays (same shape), mask: 2-D uint8 or None (all valid), n_tiles: (vert, hori)"""
    h, w = bands[0].shape
    valid = np.ones((h, w), np.uint8) if mask is None else (mask != 0).astype(np.uint8)
    out = bytearray()
    for ib, z in enumerate(bands):
        out += b"CntZImage " + struct.pack("<iiiid", 11, 8, h, w, float(max_z_err))
        if ib == 0:
            if mask is None:
                out += struct.pack("<iiif", 0, 0, 0, 1.0)
            else:
                rle = rle_of(np.packbits(valid.reshape(-1)).tobytes(), rng)
                out += struct.pack("<iiif", 0, 0, len(rle), 1.0) + rle
        tv, th = n_tiles
        tiles = bytearray()
        k = 0
        for it in range(tv + 1):
            th_rows = h // tv if it < tv else h % tv
            i0 = it * (h // tv)
            if th_rows == 0:
                continue
            for jt in range(th + 1):
                tw = w // th if jt < th else w % th
                j0 = jt * (w // th)
                if tw == 0:
                    continue
                k += 1
                mode = "raw" if raw_every and k % raw_every == 0 else "auto"
                tiles += z_tile(z[i0:i0 + th_rows, j0:j0 + tw], valid[i0:i0 + th_rows, j0:j0 + tw], max_z_err, mode)
        zmax = float(z[valid > 0].max()) if valid.any() else 0.0
        out += struct.pack("<iiif", tv, th, len(tiles), zmax) + tiles
    return bytes(out)
