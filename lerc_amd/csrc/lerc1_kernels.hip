// lerc1_kernels.hip -- device side of the legacy Lerc1 ("CntZImage") decoder.
//
// Reference: CntZImage::readTiles / readZTile (Lerc1Decode/CntZImage.cpp:219-258, :341-438), BitStuffer::read
// (Lerc1Decode/BitStuffer.cpp:32-112), Lerc::Convert (Lerc.cpp:795-845).  The z part is a row-major sequence of tiles
// (numTilesVert x numTilesHori plus a remainder row / column), each a flag byte and then nothing (all 0), a constant, raw
// floats of the valid pixels, or offset + integers bit-stuffed MSB first into 32-bit words (the layout codec 2 kept).
// Where a tile starts follows from the lengths of all tiles before it, and a raw tile's length from the number of valid
// pixels under it: one lane walks the tiles (a legacy format, and its rasters are small), a wave per tile decodes.
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

// little-endian bit field read with a hard upper bound on the bytes touched
__device__ __forceinline__ u32 readBitsBounded(const u8* __restrict__ p, u64 bitPos, int nbits, u32 end)
{
  const u64 byte = bitPos >> 3;
  const int sh = (int)(bitPos & 7);
  const int need = (sh + nbits + 7) >> 3;    // <= 5
  u64 v = 0;
  for (int i = 0; i < need; i++)
    if (byte + i < end) v |= (u64)p[byte + i] << (8 * i);
  return (u32)((v >> sh) & ((nbits >= 32) ? 0xFFFFFFFFull : ((1ull << nbits) - 1)));
}

// tile t of the list: rows [i0, i1), columns [j0, j1)
__device__ __forceinline__ void lerc1TileRect(const Lerc1Geom& g, u32 t, int& i0, int& i1, int& j0, int& j1)
{
  const u32 a = t / g.tilesAcross, b = t - a * g.tilesAcross;
  const int tileH = g.height / g.nTV, tileW = g.width / g.nTH;
  i0 = (int)a * tileH; i1 = ((int)a < g.nTV) ? i0 + tileH : g.height;
  j0 = (int)b * tileW; j1 = ((int)b < g.nTH) ? j0 + tileW : g.width;
}

__global__ void __launch_bounds__(256) k_lerc1_tile_valid(Lerc1Geom g, const u8* __restrict__ maskBits, u32* __restrict__ nValid)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= g.nTiles) return;
  int i0, i1, j0, j1;
  lerc1TileRect(g, t, i0, i1, j0, j1);
  u32 n = 0;
  for (int i = i0; i < i1; i++)
    for (int j = j0; j < j1; j++) n += (!maskBits || maskBit(maskBits, (i64)i * g.width + j)) ? 1u : 0u;
  nValid[t] = n;
}

struct Lerc1Tile { u32 flag, nOff, nb, nCount, numElem, payload, len; float offset; };

// parses the tile at `pos` (stream ends at `end`); false: damaged
__device__ __forceinline__ bool lerc1ParseTile(const u8* __restrict__ p, u32 pos, u32 end, u32 nValid, Lerc1Tile& t)
{
  if (pos >= end) return false;
  const u32 f = p[pos];
  const u32 bits67 = f >> 6;
  t.flag = f & 63u; t.nOff = 0; t.nb = 0; t.nCount = 0; t.numElem = 0; t.payload = 1; t.offset = 0;
  if (t.flag == 2u) { t.len = 1; return true; }
  if (t.flag > 3u) return false;
  if (t.flag == 0u) { t.len = 1u + 4u * nValid; return (u64)pos + t.len <= end; }
  t.nOff = (bits67 == 0u) ? 4u : 3u - bits67;
  if (t.nOff == 0u || (u64)pos + 1u + t.nOff > end) return false;
  if (t.nOff == 1u) t.offset = (float)(signed char)p[pos + 1];
  else if (t.nOff == 2u) t.offset = (float)(short)(p[pos + 1] | (p[pos + 2] << 8));
  else { u32 bits = 0; for (int k = 0; k < 4; k++) bits |= (u32)p[pos + 1 + k] << (8 * k); memcpy(&t.offset, &bits, 4); }
  if (t.flag == 3u) { t.len = 1u + t.nOff; return true; }
  u32 at = pos + 1u + t.nOff;
  if (at >= end) return false;
  const u32 b0 = p[at];
  const u32 c67 = b0 >> 6;
  t.nb = b0 & 63u;
  t.nCount = (c67 == 0u) ? 4u : 3u - c67;
  if (t.nCount == 0u || t.nb >= 32u || (u64)at + 1u + t.nCount > end) return false;
  for (u32 k = 0; k < t.nCount; k++) t.numElem |= (u32)p[at + 1 + k] << (8 * k);
  if (t.numElem < nValid || t.numElem > (1u << 26)) return false;    // (the reference would read past its buffer)
  t.payload = 1u + t.nOff + 1u + t.nCount;
  const u64 len = (u64)t.payload + (((u64)t.numElem * t.nb + 7) >> 3);
  if ((u64)pos + len > end) return false;
  t.len = (u32)len;
  return true;
}

__global__ void __launch_bounds__(64) k_lerc1_walk(Lerc1Geom g, const u8* __restrict__ part, u32 partBytes, const u32* __restrict__ nValid,
                                                   u32* __restrict__ tileOff, DeviceStatus* st)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 cur = 0;
  for (u32 t = 0; t < g.nTiles; t++)
  {
    Lerc1Tile tl;
    if (!lerc1ParseTile(part, cur, partBytes, g.allValid ? nValid[t] : nValid[t], tl)) { raiseError(st, kFailed, t); return; }
    tileOff[t] = cur;
    cur += tl.len;
  }
  tileOff[g.nTiles] = cur;
}

// float -> caller's type the way Lerc::Convert does it (Lerc.cpp:814, :830)
template<class T> __device__ __forceinline__ T lerc1Cast(float z) { return (T)floor((double)z + 0.5); }
template<> __device__ __forceinline__ float lerc1Cast<float>(float z) { return z; }
template<> __device__ __forceinline__ double lerc1Cast<double>(float z) { return (double)z; }

template<class T>
__global__ void __launch_bounds__(256) k_lerc1_decode(Lerc1Geom g, const u8* __restrict__ part, u32 partBytes, const u8* __restrict__ maskBits,
                                                      const u32* __restrict__ nValid, const u32* __restrict__ tileOff, T* __restrict__ out,
                                                      DeviceStatus* st)
{
  const int w = waveId(), lane = laneId();
  const u32 t = blockIdx.x * 4u + (u32)w;
  if (t >= g.nTiles) return;    // whole wave leaves together
  int i0, i1, j0, j1;
  lerc1TileRect(g, t, i0, i1, j0, j1);
  const int tileW = j1 - j0, nElem = (i1 - i0) * tileW;
  Lerc1Tile tl;
  if (!lerc1ParseTile(part, tileOff[t], partBytes, nValid[t], tl)) { if (lane == 0) raiseError(st, kFailed, t); return; }
  const u32 pos = tileOff[t];
  const double invScale = 2 * g.maxZErr;
  const u64 lt = laneMaskLt();
  int base = 0;
  for (int e0 = 0; e0 < nElem; e0 += 64)
  {
    const int e = e0 + lane;
    const bool inb = e < nElem;
    const int r = inb ? e / tileW : 0, c = inb ? e - r * tileW : 0;
    const i64 px = (i64)(i0 + r) * g.width + (j0 + c);
    const bool valid = inb && (!maskBits || maskBit(maskBits, px));
    const u64 bal = __ballot(valid);
    const u32 rank = (u32)(base + __popcll(bal & lt));
    base += __popcll(bal);
    if (!valid) continue;    // pixels that are not valid keep what the caller's buffer held
    float z = 0;
    if (tl.flag == 0u)
    {
      u32 bits = 0;
      for (int k = 0; k < 4; k++) bits |= (u32)part[pos + 1u + 4u * rank + (u32)k] << (8 * k);
      memcpy(&z, &bits, 4);
    }
    else if (tl.flag == 3u) z = tl.offset;
    else if (tl.flag == 1u)
    {
      u32 q = 0;
      if (tl.nb > 0)
      {
        const OldBitLayout o = oldBitLayout(rank, (int)tl.nb, tl.numElem);
        const u64 at = 8ull * (pos + tl.payload);
        q = readBitsBounded(part, at + o.pos0, (int)o.n0, partBytes) << o.n1;
        if (o.n1) q |= readBitsBounded(part, at + o.pos1, (int)o.n1, partBytes);
      }
      const float zz = (float)((double)tl.offset + (double)q * invScale);
      z = zz < g.maxZInImg ? zz : g.maxZInImg;    // std::min(z, maxZInImg)
    }
    out[px] = lerc1Cast<T>(z);
  }
}

void launchLerc1TileValid(const Lerc1Geom& g, const u8* maskBits, u32* nValid, hipStream_t st)
{
  hipLaunchKernelGGL(k_lerc1_tile_valid, dim3((g.nTiles + 255) / 256), dim3(256), 0, st, g, maskBits, nValid);
}

void launchLerc1Walk(const Lerc1Geom& g, const u8* part, u32 partBytes, const u32* nValid, u32* tileOff, DeviceStatus* status, hipStream_t st)
{
  hipLaunchKernelGGL(k_lerc1_walk, dim3(1), dim3(64), 0, st, g, part, partBytes, nValid, tileOff, status);
}

void launchLerc1Decode(int dt, const Lerc1Geom& g, const u8* part, u32 partBytes, const u8* maskBits, const u32* nValid, const u32* tileOff,
                       void* out, DeviceStatus* status, hipStream_t st)
{
  const dim3 grid((g.nTiles + 3) / 4), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_lerc1_decode<signed char>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (signed char*)out, status); break;
    case DT_Byte:   hipLaunchKernelGGL(k_lerc1_decode<unsigned char>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (unsigned char*)out, status); break;
    case DT_Short:  hipLaunchKernelGGL(k_lerc1_decode<short>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (short*)out, status); break;
    case DT_UShort: hipLaunchKernelGGL(k_lerc1_decode<unsigned short>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (unsigned short*)out, status); break;
    case DT_Int:    hipLaunchKernelGGL(k_lerc1_decode<int>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (int*)out, status); break;
    case DT_UInt:   hipLaunchKernelGGL(k_lerc1_decode<unsigned int>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (unsigned int*)out, status); break;
    case DT_Float:  hipLaunchKernelGGL(k_lerc1_decode<float>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (float*)out, status); break;
    case DT_Double: hipLaunchKernelGGL(k_lerc1_decode<double>, grid, block, 0, st, g, part, partBytes, maskBits, nValid, tileOff, (double*)out, status); break;
    default: break;
  }
}

}    // namespace lerc
