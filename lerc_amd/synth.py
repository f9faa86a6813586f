"""Deterministic, libm-free synthetic rasters for the BASELINE.json configs.

Everything is integer hashing plus exactly-rounded IEEE add / mul / div / floor, evaluated op by op,
so the same bits come out of torch-on-CPU (tests, CPU baseline) and torch-on-MI355X (bench.py
generates inputs directly in HBM).  No sin/cos/exp: a triangle wave stands in for terrain.

  c2_float32(rows, cols, row0, col0)  8192^2 f32 terrain + sigma~1 noise   (seed 1234)
  c3_uint16(rows, cols)               16384^2 u16 DEM-like                 (seed 1235)
  c4_rgb_u8(rows, cols)               4096^2 x 3 u8 smooth + sigma~4 noise  (seed 1236)
  C5 tiles are windows of the C2 generator on a 65536^2 virtual raster.
"""
import torch

_M32 = 0xFFFFFFFF


def _hash32(x):
    """'lowbias32' integer hash evaluated in int64 lanes (wrapping products, masked to 32 bits)."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _noise(idx, seed):
    """Sum of four uniform 16-bit draws, centred, unit variance (Irwin-Hall n=4 scaled by sqrt(3))."""
    hi = _hash32((idx >> 31) + seed * 0x9E3779B1)    # keeps 65536^2 virtual rasters collision-free
    h0 = _hash32(((idx * 2) & _M32) ^ hi)
    h1 = _hash32(((idx * 2 + 1) & _M32) ^ hi)
    s = (h0 & 0xFFFF) + (h0 >> 16) + (h1 & 0xFFFF) + (h1 >> 16)
    return (s.to(torch.float64) / 65536.0 - 2.0) * 1.7320508075688772


def _tri(t):
    """Triangle wave, period 1, range [-1, 1], exact piecewise-linear arithmetic in f64."""
    f = t - torch.floor(t)
    return 1.0 - 4.0 * torch.abs(f - 0.5)


def _grid(rows, cols, row0, col0, virt_cols, device):
    i = torch.arange(row0, row0 + rows, dtype=torch.int64, device=device).view(-1, 1)
    j = torch.arange(col0, col0 + cols, dtype=torch.int64, device=device).view(1, -1)
    return i, j, i * virt_cols + j


def c2_float32(rows=8192, cols=8192, row0=0, col0=0, virt_cols=None, device="cpu", seed=1234):
    virt_cols = cols if virt_cols is None else virt_cols
    i, j, idx = _grid(rows, cols, row0, col0, virt_cols, device)
    terr = _tri(j.to(torch.float64) / 300.0) * _tri(i.to(torch.float64) / 211.0)
    v = 1000.0 + 500.0 * terr + _noise(idx, seed)
    return v.to(torch.float32).contiguous()


def c3_uint16(rows=16384, cols=16384, device="cpu", seed=1235):
    i, j, idx = _grid(rows, cols, 0, 0, cols, device)
    terr = _tri(j.to(torch.float64) / 300.0) * _tri(i.to(torch.float64) / 211.0)
    v = 1500.0 + 1200.0 * terr + 3.0 * _noise(idx, seed)
    v = torch.clamp(torch.floor(v + 0.5), 0.0, 65535.0)
    return v.to(torch.int32).to(torch.uint16).contiguous() if hasattr(torch, "uint16") else v.to(torch.int32)


def c4_rgb_u8(rows=4096, cols=4096, device="cpu", seed=1236):
    i, j, idx = _grid(rows, cols, 0, 0, cols, device)
    chans = []
    for k in range(3):
        terr = _tri(j.to(torch.float64) / 300.0 + 0.31 * k) * _tri(i.to(torch.float64) / 211.0 + 0.17 * k)
        v = 128.0 + 100.0 * terr + 4.0 * _noise(idx * 3 + k, seed)
        chans.append(torch.clamp(torch.floor(v + 0.5), 0.0, 255.0).to(torch.uint8))
    return torch.stack(chans, dim=-1).contiguous()    # [rows][cols][3], pixel-interleaved (nDepth = 3)


def c5_tile(tile_row, tile_col, tile=256, virt=65536, device="cpu"):
    return c2_float32(tile, tile, tile_row * tile, tile_col * tile, virt_cols=virt, device=device)
