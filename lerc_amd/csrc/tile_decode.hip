// tile_decode.hip -- Lerc2 "tiling" decoder on CDNA4.
//
// Replaces Lerc2::ReadTiles / ReadTile (Lerc2.cpp:1672-1713, :2025-2230) and BitStuffer2::Decode /
// BitUnStuff (BitStuffer2.cpp:159-258, :476-540).
//
// 1. Block-offset discovery.  The stream stores no block offsets: block k+1 starts where block k
//    ends and a block's length is only known from its own first bytes.  The blob is cut into chunks;
//    for every chunk all positions of its first `window` bytes (window = longest possible block)
//    are tried as block starts and walked to the chunk end, dropping walks that hit an impossible
//    block header or an inconsistent column signature.  The true first block of the chunk is always
//    among the survivors; if all survivors leave the chunk at the same offset, that offset IS the
//    first block of the next chunk, whatever this chunk's own entry was.  Chunks therefore resolve
//    independently (self-synchronisation); the rare disagreeing chunk is re-walked from its
//    resolved predecessor, and blobs whose raw blocks have position dependent lengths (masks /
//    partial edge blocks) fall back to a single sequential walk on the device.
// 2. Decode: one wave64 per micro-block position, lane = element (rank by __ballot / __popcll for
//    masked blocks), bit extraction straight from the blob, dequantise in double precision in the
//    reference's expression order (compile with -ffp-contract=off), clamp, cast, store.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

struct BlkInfo
{
  u32 len;        // total bytes of the block
  u32 cnt;        // element count stored in the bit stuffer header
  u32 nLut;       // LUT entries without the implicit 0
  u32 payload;    // offset of the first payload byte relative to the block start
  u8 flag, mode, diff, tc, offBytes, nb, lut, dtUsed;
};

// The first 16 bytes of a block, which hold every header field there is (flag, offset of up to 8 bytes, the bit stuffer's
// first byte, a count of up to 4 bytes, the LUT size: 15 bytes) -- fetched with five aligned 32-bit loads that are all in
// flight at once, where a parse that goes byte by byte waits for memory half a dozen times in a row (flag -> offset type
// -> bit width -> count -> LUT size).  Bytes at or behind `end` read as zero; no load touches a 4-byte unit that lies
// entirely outside [blob, blob + end).
struct Win16
{
  u64 lo, hi;
  __device__ __forceinline__ u32 byteAt(u32 k) const { return (u32)((k < 8u ? lo >> (8u * k) : hi >> (8u * (k - 8u))) & 255u); }
};

__device__ __forceinline__ Win16 funnel16(const u32 (&d)[5], u32 sh, u32 have)
{
  u32 x[4];
#pragma unroll
  for (int k = 0; k < 4; k++) x[k] = (u32)((((u64)d[k + 1] << 32) | d[k]) >> sh);
  Win16 w;
  w.lo = ((u64)x[1] << 32) | x[0];
  w.hi = ((u64)x[3] << 32) | x[2];
  // blank what lies at or behind the end (have = bytes of the window inside the blob)
  if (have < 16u)
  {
    if (have <= 8u) { w.hi = 0; w.lo = have == 0u ? 0ull : (w.lo & (~0ull >> (64u - 8u * have))); }
    else w.hi &= ~0ull >> (64u - 8u * (have - 8u));
  }
  return w;
}

// bytes pos ... pos + 15 of blob (any memory, any alignment)
__device__ __forceinline__ Win16 loadWin16(const u8* __restrict__ blob, u32 pos, u32 end)
{
  const uintptr_t A = (uintptr_t)blob + pos, E = (uintptr_t)blob + end;
  const u32* wp = reinterpret_cast<const u32*>(A & ~(uintptr_t)3);
  u32 d[5];
#pragma unroll
  for (int k = 0; k < 5; k++) d[k] = ((uintptr_t)(wp + k) < E) ? wp[k] : 0u;
  return funnel16(d, (u32)(A & 3u) * 8u, end > pos ? min(16u, end - pos) : 0u);
}

// the same out of an LDS array that starts on a 4-byte boundary (kept apart so that the loads stay LDS loads)
__device__ __forceinline__ Win16 loadWin16Words(const u32* words, u32 pos, u32 end)
{
  const u32 w0 = pos >> 2;
  u32 d[5];
#pragma unroll
  for (int k = 0; k < 5; k++) d[k] = (4u * (w0 + k) < end) ? words[w0 + k] : 0u;
  return funnel16(d, (pos & 3u) * 8u, end > pos ? min(16u, end - pos) : 0u);
}

// ... where the array is known to reach 20 bytes beyond `pos` (what lies behind `end` is blanked all the same): five loads
// without a condition, which the compiler can issue together
__device__ __forceinline__ Win16 loadWin16WordsRoomy(const u32* words, u32 pos, u32 end)
{
  const u32 w0 = pos >> 2;
  u32 d[5];
#pragma unroll
  for (int k = 0; k < 5; k++) d[k] = words[w0 + k];
  return funnel16(d, (pos & 3u) * 8u, end > pos ? min(16u, end - pos) : 0u);
}

// 0 = ok, 1 = not a valid block here, 2 = raw block whose valid count is not known to the caller.
// Mirrors the checks of Lerc2::ReadTile and BitStuffer2::Decode; additionally refuses element counts
// that differ from the block's valid pixel count (the reference would read past its buffer there).
// `h` holds the block's first 16 bytes, pos / end say where it lies in its stream.
template<int TBYTES>
__device__ __forceinline__ int parseWindow(const Win16& h, u32 pos, u32 end, const BandParams& p, int nValid, u32 maxCount, BlkInfo& b)
{
  b.len = 0;
  if (pos >= end) return 1;
  const u32 flag = h.byteAt(0);
  b.flag = (u8)flag;
  b.diff = (p.version >= 5 && (flag & 4u)) ? 1 : 0;
  b.mode = (u8)(flag & 3u);
  b.tc = (u8)(flag >> 6);
  b.offBytes = 0; b.nb = 0; b.lut = 0; b.cnt = 0; b.nLut = 0; b.payload = 1; b.dtUsed = (u8)p.dt;
  u64 len = 1;
  if (b.diff && p.nDepth == 1) return 1;    // (difference to the slice before: there is none, Lerc2.cpp ReadTile refuses)
  if (b.mode == 2) { b.len = 1; return 0; }
  if (b.mode == 0)
  {
    if (b.diff) return 1;
    if (nValid < 0) return 2;
    len = 1 + (u64)nValid * TBYTES;
  }
  else
  {
    const int dtU = typeUsed((b.diff && p.dt < DT_Float) ? (int)DT_Int : p.dt, b.tc);
    if (dtU == DT_Undefined) return 1;
    b.dtUsed = (u8)dtU;
    b.offBytes = (u8)dtSize(dtU);
    len = 1 + b.offBytes;
    if (b.mode == 1)
    {
      const u32 at = (u32)len;                      // (relative to the block's start from here on: <= 9)
      if ((u64)pos + at >= end) return 1;
      const u32 b0 = h.byteAt(at);
      const u32 code = b0 >> 6;
      const int cb = (code == 0) ? 4 : 3 - (int)code;
      if (cb == 0) return 1;
      b.lut = (b0 & 32u) ? 1 : 0;
      b.nb = (u8)(b0 & 31u);
      if ((u64)pos + at + 1 + cb > end) return 1;
      // the count's cb bytes start at at + 1 <= 10: inside the window
      const u32 sh = 8u * (at + 1u);
      const u64 two = sh < 64u ? ((h.lo >> sh) | (sh ? h.hi << (64u - sh) : 0ull)) : (h.hi >> (sh - 64u));
      const u32 cnt = (u32)two & (cb == 4 ? 0xFFFFFFFFu : ((1u << (8 * cb)) - 1u));
      b.cnt = cnt;
      if (cnt == 0 || cnt > maxCount || b.nb == 0) return 1;
      if (nValid >= 0 && cnt != (u32)nValid) return 1;
      len += 1 + cb;
      if (!b.lut) { b.payload = (u32)len; len += ((u64)cnt * b.nb + 7) >> 3; }
      else
      {
        if ((u64)pos + len >= end) return 1;
        const int nLut = (int)h.byteAt((u32)len) - 1;    // (at most byte 14)
        if (nLut < 1) return 1;
        b.nLut = (u32)nLut;
        len += 1;
        b.payload = (u32)len;
        len += ((u64)nLut * b.nb + 7) >> 3;
        len += ((u64)cnt * bitLen((u32)nLut) + 7) >> 3;
      }
    }
  }
  if ((u64)pos + len > end) return 1;
  b.len = (u32)len;
  return 0;
}

// What a walk needs of parseWindow -- the block's length, or that there is none -- without a branch: every lane of a wave looks
// at another position, so every branch of the parse is taken by some lane, and the bookkeeping of who is in which costs more
// than the arithmetic (k_rank_chunks parses every position of the stream: 480 instructions a position that way, R this way).
// offPack: bytes of the offset for type code tc and difference flag d, 4 bits at (tc * 2 + d) * 4, 0 = no such type
// (typeUsed, Lerc2.h:528-542), see offsetBytesPack.  Returns the length, 0 = no block, kLenRawUnknown = a raw block whose
// valid count the caller does not know (nValid < 0).
static const u32 kLenRawUnknown = 0xFFFFFFFFu;

__device__ __forceinline__ u32 offsetBytesPack(const BandParams& p)
{
  u32 pack = 0;
  for (int tc = 0; tc < 4; tc++)
    for (int d = 0; d < 2; d++)
    {
      const int dtU = typeUsed((d && p.dt < DT_Float) ? (int)DT_Int : p.dt, tc);
      const u32 n = (dtU == DT_Undefined) ? 0u : (u32)dtSize(dtU);
      pack |= (n > 8u ? 0u : n) << ((tc * 2 + d) * 4);    // (8 fits 4 bits)
    }
  return pack;
}

template<int TBYTES>
__device__ __forceinline__ u32 blockLength(const Win16& h, u32 pos, u32 end, const BandParams& p, u32 offPack, int nValid, u32 maxCount)
{
  const u32 flag = (u32)h.lo & 255u;
  const u32 mode = flag & 3u, tc = flag >> 6;
  const u32 diff = (p.version >= 5) ? ((flag >> 2) & 1u) : 0u;
  const u32 offB = (offPack >> ((tc * 2u + diff) * 4u)) & 15u;
  // the bit stuffer's header behind the offset: first byte, count (1, 2 or 4 bytes), LUT size -- bytes at + 0 ... at + 5
  const u32 at = 1u + offB;                         // <= 9
  const u32 sh = 8u * at;
  const u64 six = sh < 64u ? ((h.lo >> sh) | ((h.hi << 1) << (63u - sh))) : h.hi >> (sh - 64u);    // (sh >= 8)
  const u32 b0 = (u32)six & 255u;
  const u32 code = b0 >> 6;
  const u32 cb = (code == 0u) ? 4u : 3u - code;     // 0: no such code
  const u32 nb = b0 & 31u, lut = (b0 >> 5) & 1u;
  const u32 cnt = (u32)(six >> 8) & (cb == 4u ? 0xFFFFFFFFu : ((1u << (8u * cb)) - 1u));
  const u32 hdr = at + 1u + cb;
  const u32 nLut = ((u32)(six >> (8u * (1u + cb))) & 255u) - 1u;    // (the byte behind the count; 0xFFFFFFFF for a zero byte)
  const u32 nbIdx = (u32)bitLen(nLut & 255u);
  const u32 plain = hdr + ((cnt * nb + 7u) >> 3);
  const u32 withLut = hdr + 1u + (((nLut & 255u) * nb + 7u) >> 3) + ((cnt * nbIdx + 7u) >> 3);
  const bool okStuffed = (offB != 0u) & (cb != 0u) & (cnt != 0u) & (cnt <= maxCount) & (nb != 0u) & ((nValid < 0) | (cnt == (u32)nValid))
                       & ((lut == 0u) | ((nLut >= 1u) & (nLut < 255u)));
  u32 len = lut ? withLut : plain;
  len = okStuffed ? len : 0u;
  len = (mode == 3u) ? (offB ? 1u + offB : 0u) : len;
  len = (mode == 2u) ? 1u : len;
  const u32 raw = diff ? 0u : (nValid < 0 ? kLenRawUnknown : 1u + (u32)nValid * (u32)TBYTES);
  len = (mode == 0u) ? raw : len;
  if (diff && p.nDepth == 1) len = 0u;
  if (len != kLenRawUnknown && ((u64)pos + len > end || pos >= end)) len = 0u;
  return len;
}

template<int TBYTES>
__device__ __forceinline__ int parseBlock(const u8* __restrict__ blob, u32 pos, u32 end, const BandParams& p, int nValid,
                                          u32 maxCount, BlkInfo& b)
{
  return parseWindow<TBYTES>(loadWin16(blob, pos, end), pos, end, p, nValid, maxCount, b);
}

// the block at byte `pos` of an LDS array given as words
template<int TBYTES>
__device__ __forceinline__ int parseBlockWords(const u32* words, u32 pos, u32 end, const BandParams& p, int nValid,
                                               u32 maxCount, BlkInfo& b)
{
  return parseWindow<TBYTES>(loadWin16Words(words, pos, end), pos, end, p, nValid, maxCount, b);
}

// little-endian bit field read with a hard upper bound on the bytes touched
__device__ __forceinline__ u32 readBits(const u8* __restrict__ blob, u64 bitPos, int nbits, u32 end)
{
  const u64 byte = bitPos >> 3;
  const int sh = (int)(bitPos & 7);
  const int need = (sh + nbits + 7) >> 3;    // <= 5
  u64 v = 0;
  for (int i = 0; i < need; i++)
    if (byte + i < end) v |= (u64)blob[byte + i] << (8 * i);
  return (u32)((v >> sh) & ((nbits >= 32) ? 0xFFFFFFFFull : ((1ull << nbits) - 1)));
}

// element i of a bit-stuffed field of n elements, nb bits each, that starts at bit `at` of the blob
// (BitStuffer2::BitUnStuff, BitStuffer2.cpp:476-540; codec 2: BitUnStuff_Before_Lerc2v3, :355-425)
__device__ __forceinline__ u32 unstuffElement(const u8* __restrict__ blob, u64 at, u32 i, int nb, u32 n, u32 end, int version)
{
  if (version >= 3) return readBits(blob, at + (u64)i * nb, nb, end);
  const OldBitLayout o = oldBitLayout(i, nb, n);
  u32 v = readBits(blob, at + o.pos0, (int)o.n0, end) << o.n1;
  if (o.n1) v |= readBits(blob, at + o.pos1, (int)o.n1, end);
  return v;
}

// ------------------------------------------------------------------------------------------------
// decode kernel
// ------------------------------------------------------------------------------------------------
template<class T>
__global__ void __launch_bounds__(256) k_decode_tiles(BandParams p, DecodeArgs a, DeviceStatus* st)
{
  __shared__ u32 s_lut[4][256];
  __shared__ __align__(16) u8 s_head[4][64];
  const int w = waveId(), lane = laneId();
  const int pos = (int)blockIdx.x * 4 + w;
  if (pos >= p.nTV * p.nTH) return;
  const int mb = p.mb, nD = p.nDepth;
  const int it = pos / p.nTH, jt = pos - it * p.nTH;
  const int i0 = it * mb, j0 = jt * mb;
  const int tileH = min(mb, p.nRows - i0), tileW = min(mb, p.nCols - j0);
  const int nElem = tileH * tileW;
  const int E = (nElem + 63) >> 6;
  const u64 lt = laneMaskLt();
  T* __restrict__ out = (T*)a.out;
  const u8* __restrict__ blob = a.blob;

  int nValid = 0;
  for (int k = 0; k < E; k++)
  {
    const int e = k * 64 + lane;
    const bool inb = e < nElem;
    const int r = inb ? e / tileW : 0, c = inb ? e - r * tileW : 0;
    const i64 px = (i64)(i0 + r) * p.nCols + (j0 + c);
    const bool valid = inb && (p.allValid || maskBit(a.maskBits, px));
    nValid += __popcll(__ballot(valid));
  }

  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  bool failed = false;

  for (int iD = 0; iD < nD; iD++)
  {
    const u32 off = a.blockOff[(i64)pos * nD + iD];
    // the block's first 64 bytes in one coalesced load: parsed lane by lane from global memory, the header's fields
    // (flag -> offset type -> bit width -> count) are half a dozen dependent round trips.  parseBlock looks at no byte
    // beyond the 15th of a block.
    s_head[w][lane] = ((u64)off + (u64)lane < (u64)a.blobEnd) ? blob[(u64)off + lane] : (u8)0;
    waveSync();
    BlkInfo b;    // (parsed relative to the block's start: position 0, the blob's end as seen from there)
    const int rc = (off < a.blobEnd) ? parseBlockWords<(int)sizeof(T)>(reinterpret_cast<const u32*>(s_head[w]), 0u, a.blobEnd - off, p, nValid, (u32)nElem, b) : 1;
    if (rc != 0 || (((u32)b.flag >> 2) & pattern) != (((u32)j0 >> 3) & pattern) || (b.diff && iD == 0))
    {
      failed = true;
      break;
    }
    double offset = 0;
    if (b.mode == 1 || b.mode == 3) offset = typedFromBits(getBytes(s_head[w] + 1, b.offBytes), b.dtUsed);
    const double zMax = (p.version >= 4 && nD > 1) ? a.zMaxVec[iD] : p.zMaxHdr;
    const u64 payloadBit = 8ull * ((u64)off + b.payload);
    const int nbIdx = b.lut ? bitLen(b.nLut) : 0;
    u64 idxBit = 0;
    if (b.mode == 1 && b.lut)
    {
      s_lut[w][0] = 0;
      for (u32 i = (u32)lane; i < b.nLut; i += 64) s_lut[w][i + 1] = unstuffElement(blob, payloadBit, i, b.nb, b.nLut, a.blobEnd, p.version);
      idxBit = payloadBit + 8ull * (((u64)b.nLut * b.nb + 7) >> 3);
      waveSync();
    }

    int base = 0;
    bool badIdx = false;
    for (int k = 0; k < E; k++)
    {
      const int e = k * 64 + lane;
      const bool inb = e < nElem;
      const int r = inb ? e / tileW : 0, c = inb ? e - r * tileW : 0;
      const i64 px = (i64)(i0 + r) * p.nCols + (j0 + c);
      const bool valid = inb && (p.allValid || maskBit(a.maskBits, px));
      const u64 bal = __ballot(valid);
      const int rank = base + __popcll(bal & lt);
      base += __popcll(bal);
      if (!inb) continue;
      const i64 m = px * nD + iD;
      T val = T(0);
      if (valid)
      {
        if (b.mode == 2) val = b.diff ? out[m - 1] : T(0);
        else if (b.mode == 0)
        {
          const u64 bits = getBytes(blob + off + 1 + (u64)rank * sizeof(T), (int)sizeof(T));
          memcpy(&val, &bits, sizeof(T));
        }
        else if (b.mode == 3)
        {
          if (!b.diff) val = (T)offset;
          else { const double z = offset + (double)out[m - 1]; val = (T)(z < zMax ? z : zMax); }
        }
        else
        {
          u32 q;
          if (!b.lut) q = unstuffElement(blob, payloadBit, (u32)rank, b.nb, b.cnt, a.blobEnd, p.version);
          else
          {
            const u32 ix = unstuffElement(blob, idxBit, (u32)rank, nbIdx, b.cnt, a.blobEnd, p.version);
            if (ix > b.nLut) { badIdx = true; q = 0; } else q = s_lut[w][ix];
          }
          double z = offset + (double)q * p.invScale;
          if (b.diff) z = z + (double)out[m - 1];
          val = (T)(z < zMax ? z : zMax);    // std::min(z, zMax)
        }
      }
      out[m] = val;
    }
    if (__any(badIdx)) { failed = true; break; }
    waveSync();
  }
  if (failed && lane == 0) raiseError(st, kFailed, (u32)pos);
}

// ------------------------------------------------------------------------------------------------
// The same for 8 x 8 blocks of single-depth bands (every block one sub-block), codec >= 3: the shape of the masked and the
// ragged rasters that the streaming kernels do not take.  k_decode_tiles gives a block a wave that waits three times on
// global memory (offset -> header -> payload) and stores 64 scattered elements; here a workgroup takes kGroup blocks that
// follow each other in the stream, stages their bytes -- one contiguous range -- in LDS, parses one header per thread,
// and a wave then decodes V blocks side by side, a lane V pixels of a row, so that a row of the V blocks is one
// contiguous store of up to 128 bytes.  Checks and arithmetic are k_decode_tiles' (the parse is the same function).
// ------------------------------------------------------------------------------------------------
static const u32 kGroup = 64;

// nbits (<= 32) at bit position bitPos of an LDS word stream
__device__ __forceinline__ u32 wordBits(const u32* words, u32 bitPos, u32 nbits)
{
  const u32 w = bitPos >> 5, sh = bitPos & 31u;
  const u64 x = ((u64)words[w + 1] << 32) | words[w];
  return (u32)(x >> sh) & (nbits >= 32u ? 0xFFFFFFFFu : ((1u << nbits) - 1u));
}

template<class T>
__global__ void __launch_bounds__(256) k_decode_blocks8(BandParams p, DecodeArgs a, DeviceStatus* st)
{
  constexpr u32 TB = (u32)sizeof(T);
  constexpr u32 V = (16u / TB) < 8u ? (16u / TB) : 8u;    // pixels of a row per lane
  constexpr u32 LPB = 64u / V;                             // lanes per block
  constexpr u32 kCap = kGroup * (64u * TB + 1u);           // no block is longer than a raw one
  __shared__ __align__(16) u8 s_bytes[kCap + 48];
  __shared__ double s_offset[kGroup];
  __shared__ u32 s_info[kGroup];      // payload (relative to the staged range, 16 bits) | nb << 16 | mode << 22 | lut << 24 | ok << 31
  __shared__ u32 s_lutBits[kGroup];   // LUT blocks: bits per index | entries << 8
  __shared__ u32 s_i0[kGroup], s_j0[kGroup];
  const u32 nPos = (u32)p.nTV * (u32)p.nTH;
  const u32 pos0 = blockIdx.x * kGroup;
  const u32 nHere = min(kGroup, nPos - pos0);
  const u32 from = a.blockOff[pos0];
  const u32 to = (pos0 + nHere < nPos) ? a.blockOff[pos0 + nHere] : a.blobEnd;
  if (to < from || to - from > kCap || to > a.blobEnd)
  {
    if (threadIdx.x == 0) raiseError(st, kFailed, pos0);
    return;
  }
  // stage with 16-byte loads from the aligned-down start; LDS byte i + shift <-> blob byte from + i
  const u32 a0 = from & ~15u, shift = from - a0;
  const u8* __restrict__ blob = a.blob;
  for (u32 v = threadIdx.x; a0 + 16u * v < to; v += 256u)
  {
    const u32 g = a0 + 16u * v;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (g + 16u <= a.blobEnd && ((uintptr_t)(blob + g) & 15u) == 0u) x = *reinterpret_cast<const uint4*>(blob + g);
    else for (u32 k2 = 0; k2 < 16u && g + k2 < a.blobEnd; k2++) (&x.x)[k2 >> 2] |= (u32)blob[g + k2] << (8u * (k2 & 3u));
    *reinterpret_cast<uint4*>(&s_bytes[16u * v]) = x;
  }
  __syncthreads();
  const u8* s_range = s_bytes + shift;
  const u32 pattern = (p.version >= 5) ? 14u : 15u;

  if (threadIdx.x < nHere)
  {
    const u32 t = threadIdx.x, pos = pos0 + t;
    const u32 it = pos / (u32)p.nTH, jt = pos - it * (u32)p.nTH;
    const u32 i0 = it * 8u, j0 = jt * 8u;
    s_i0[t] = i0; s_j0[t] = j0;
    const u32 tileH = min(8u, (u32)p.nRows - i0), tileW = min(8u, (u32)p.nCols - j0);
    const int nValid = a.nValidBlk ? (int)a.nValidBlk[pos] : (int)(tileH * tileW);
    const u32 off = a.blockOff[pos];
    const u32 next = (t + 1 < nHere) ? a.blockOff[pos + 1] : to;
    BlkInfo b;
    u32 info = 0;
    double offset = 0;
    int rc = 1;
    if (off >= from && off < to && next > off && next <= to)
      rc = parseBlockWords<(int)TB>(reinterpret_cast<const u32*>(s_bytes), shift + off - from, shift + next - from, p, nValid, 64u, b);
    const bool good = rc == 0 && b.len == next - off && (((u32)b.flag >> 2) & pattern) == ((j0 >> 3) & pattern) && !b.diff;
    if (good)
    {
      if (b.mode == 1 || b.mode == 3) offset = typedFromBits(getBytes(s_range + (off - from) + 1, b.offBytes), b.dtUsed);
      info = (off - from + b.payload) | ((u32)b.nb << 16) | ((u32)b.mode << 22) | ((u32)b.lut << 24) | (1u << 31);
      s_lutBits[t] = b.lut ? ((u32)bitLen(b.nLut) | (b.nLut << 8)) : 0u;
    }
    else raiseError(st, kFailed, pos);
    s_info[t] = info;
    s_offset[t] = offset;
  }
  __syncthreads();

  const u32 lane = (u32)laneId(), w = (u32)waveId();
  const u32 sub = lane / LPB, li = lane - sub * LPB;     // block of the wave's V, lane in the block
  const u32 r = li / (8u / V), h = li - r * (8u / V);    // row, segment of the row
  const u64 groupLt = laneMaskLt() & (((LPB == 64u) ? ~0ull : ((1ull << LPB) - 1ull)) << (sub * LPB));
  const u32* s_words = reinterpret_cast<const u32*>(s_bytes);
  T* __restrict__ out = (T*)a.out;
  const double zMax = p.zMaxHdr, invScale = p.invScale;
  const u64 maskBytes = ((u64)p.nRows * (u64)p.nCols + 7ull) >> 3;
  bool badIdx = false;
  for (u32 t0 = w * V; t0 < nHere; t0 += 4u * V)
  {
    const u32 t = t0 + sub;
    const bool have = t < nHere;
    const u32 info = have ? s_info[t] : 0u;
    const u32 i = (have ? s_i0[t] : 0u) + r, j = (have ? s_j0[t] : 0u) + h * V;
    const bool rowIn = have && (info >> 31) && i < (u32)p.nRows;
    // validity of the lane's V pixels, pixel k in bit k
    u32 inb = 0u, vb = 0u;
    if (rowIn)
    {
      const u32 nIn = j < (u32)p.nCols ? min(V, (u32)p.nCols - j) : 0u;
      inb = (1u << nIn) - 1u;
      vb = inb;
      if (!p.allValid && nIn)
      {
        const u64 px = (u64)i * (u64)p.nCols + j;
        const u64 by = px >> 3;
        const u32 sh = (u32)(px & 7u);
        u32 win = (u32)a.maskBits[by] << 8;
        if (sh + nIn > 8u && by + 1 < maskBytes) win |= a.maskBits[by + 1];
        const u32 msb = (win << sh) >> (16u - V) & ((1u << V) - 1u);    // pixel 0 in bit V - 1 ... (the window's bits 15 - sh downward)
        const u32 field = msb & ((1u << V) - 1u);
        u32 rev = 0u;
#pragma unroll
        for (u32 k = 0; k < V; k++) rev |= ((field >> (V - 1u - k)) & 1u) << k;
        vb = rev & inb;
      }
    }
    // rank of the lane's first valid pixel among the block's
    u32 e0 = 0u;
#pragma unroll
    for (u32 k = 0; k < V; k++) e0 += (u32)__popcll(__ballot((vb >> k) & 1u) & groupLt);
    if (!rowIn || !inb) continue;
    const u32 mode = (info >> 22) & 3u, nb = (info >> 16) & 63u, payload = info & 0xFFFFu;
    const double offset = s_offset[t];
    T vals[V];
    u32 e = e0;
#pragma unroll
    for (u32 k = 0; k < V; k++)
    {
      T val = T(0);
      if ((vb >> k) & 1u)
      {
        if (mode == 0u)
        {
          u64 bits = 0;
          for (u32 q = 0; q < TB; q++) bits |= (u64)s_range[payload + e * TB + q] << (8u * q);
          memcpy(&val, &bits, sizeof(T));
        }
        else if (mode == 3u) val = (T)offset;
        else if (mode == 1u)
        {
          const u32 bit0 = (shift + payload) * 8u;
          u32 q;
          if (!((info >> 24) & 1u)) q = wordBits(s_words, bit0 + e * nb, nb);
          else
          {
            const u32 lb = s_lutBits[t], nbIdx = lb & 255u, nLut = lb >> 8;
            const u32 idxBit = bit0 + 8u * ((nLut * nb + 7u) >> 3);
            const u32 ix = wordBits(s_words, idxBit + e * nbIdx, nbIdx);
            if (ix > nLut) { badIdx = true; q = 0u; }
            else q = ix ? wordBits(s_words, bit0 + (ix - 1u) * nb, nb) : 0u;
          }
          const double z = offset + (double)q * invScale;
          val = (T)(z < zMax ? z : zMax);    // std::min(z, zMax)
        }
        e++;
      }
      vals[k] = val;
    }
    T* dst = out + ((u64)i * (u64)p.nCols + j);
    if (inb == (1u << V) - 1u && ((uintptr_t)dst & (V * TB - 1u)) == 0u)
    {
      if (V * TB == 16u) { uint4 x; memcpy(&x, vals, 16); *reinterpret_cast<uint4*>(dst) = x; }
      else { uint2 x; memcpy(&x, vals, 8); *reinterpret_cast<uint2*>(dst) = x; }
    }
    else
    {
#pragma unroll
      for (u32 k = 0; k < V; k++) if ((inb >> k) & 1u) dst[k] = vals[k];
    }
  }
  if (__any(badIdx) && laneId() == 0) raiseError(st, kFailed, pos0);
}

void launchTileDecode(int dt, const BandParams& p, const DecodeArgs& a, DeviceStatus* st, hipStream_t stream)
{
  const int nPos = p.nTV * p.nTH;
  if (p.mb == 8 && p.nDepth == 1 && p.version >= 3 && nPos > 0)
  {
    const dim3 grid8((nPos + (int)kGroup - 1) / (int)kGroup), block8(256);
    switch (dt)
    {
      case DT_Char:   hipLaunchKernelGGL(k_decode_blocks8<signed char>, grid8, block8, 0, stream, p, a, st); break;
      case DT_Byte:   hipLaunchKernelGGL(k_decode_blocks8<unsigned char>, grid8, block8, 0, stream, p, a, st); break;
      case DT_Short:  hipLaunchKernelGGL(k_decode_blocks8<short>, grid8, block8, 0, stream, p, a, st); break;
      case DT_UShort: hipLaunchKernelGGL(k_decode_blocks8<unsigned short>, grid8, block8, 0, stream, p, a, st); break;
      case DT_Int:    hipLaunchKernelGGL(k_decode_blocks8<int>, grid8, block8, 0, stream, p, a, st); break;
      case DT_UInt:   hipLaunchKernelGGL(k_decode_blocks8<unsigned int>, grid8, block8, 0, stream, p, a, st); break;
      case DT_Float:  hipLaunchKernelGGL(k_decode_blocks8<float>, grid8, block8, 0, stream, p, a, st); break;
      case DT_Double: hipLaunchKernelGGL(k_decode_blocks8<double>, grid8, block8, 0, stream, p, a, st); break;
      default: break;
    }
    return;
  }
  const dim3 grid((nPos + 3) / 4), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_decode_tiles<signed char>, grid, block, 0, stream, p, a, st); break;
    case DT_Byte:   hipLaunchKernelGGL(k_decode_tiles<unsigned char>, grid, block, 0, stream, p, a, st); break;
    case DT_Short:  hipLaunchKernelGGL(k_decode_tiles<short>, grid, block, 0, stream, p, a, st); break;
    case DT_UShort: hipLaunchKernelGGL(k_decode_tiles<unsigned short>, grid, block, 0, stream, p, a, st); break;
    case DT_Int:    hipLaunchKernelGGL(k_decode_tiles<int>, grid, block, 0, stream, p, a, st); break;
    case DT_UInt:   hipLaunchKernelGGL(k_decode_tiles<unsigned int>, grid, block, 0, stream, p, a, st); break;
    case DT_Float:  hipLaunchKernelGGL(k_decode_tiles<float>, grid, block, 0, stream, p, a, st); break;
    case DT_Double: hipLaunchKernelGGL(k_decode_tiles<double>, grid, block, 0, stream, p, a, st); break;
    default: break;
  }
}

// ================================================================================================
// block-offset discovery
// ================================================================================================
static const u32 kNone = 0xFFFFFFFFu;

WalkPlan makeWalkPlan(const BandParams& p, u32 dataBegin, u32 blobEnd, int numValid)
{
  WalkPlan wp;
  const u32 n = (u32)p.mb * (u32)p.mb;
  const u32 tb = (u32)dtSize(p.dt);
  const u32 raw = 1 + n * tb;
  const u32 simple = 1 + 8 + 1 + 4 + ((n * 31 + 7) >> 3);
  const u32 lut = 1 + 8 + 1 + 4 + 1 + ((254u * 31 + 7) >> 3) + n;
  u32 win = raw > simple ? raw : simple;
  win = lut > win ? lut : win;
  wp.window = win + 1;
  u32 cb = 4096;
  while (cb < 2 * wp.window) cb <<= 1;
  wp.chunkBytes = cb;
  const u32 span = blobEnd > dataBegin ? blobEnd - dataBegin : 0;
  wp.nChunks = span ? (span + cb - 1) / cb : 1;
  wp.nSub = (u32)p.nTV * (u32)p.nTH * (u32)p.nDepth;
  const bool uniform = (numValid == p.nRows * p.nCols) && (p.nRows % p.mb == 0) && (p.nCols % p.mb == 0);
  wp.uniformN = uniform ? (int)n : 0;
  wp.candWindow = std::min(wp.window, 2u + n * tb);
  wp.tabled = (wp.chunkBytes <= 4096u && wp.window <= 1100u) ? 1u : 0u;    // (kMemoChunk, kMemoWindowMax below)
  wp.test = 0u;
  return wp;
}

// Column-signature progression between two consecutive sub-blocks of the stream (flag bits 2-5,
// Lerc2.cpp:1955-1958 / :2042-2046): same position (next depth slice), next column, or column 0 of
// the next block row.  Only checked for the standard block sizes.
__device__ __forceinline__ bool sigFollows(u32 prev, u32 cur, int mb, u32 pattern)
{
  if (mb != 8 && mb != 16 && mb != 32) return true;
  u32 step = (u32)mb >> 3;
  if (step == 1 && pattern == 14u) step = 2;
  return cur == prev || cur == ((prev + step) & pattern) || cur == 0;
}

// D1: one wave per chunk.
template<int TBYTES>
__global__ void __launch_bounds__(64) k_walk_chunks(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin,
                                                    u32 blobEnd, u32* __restrict__ chunkExit)
{
  const u32 c = blockIdx.x;
  const int lane = laneId();
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
  const u32 winEnd = (c == 0) ? chunkStart + 1 : min(chunkStart + wp.window, chunkEnd);
  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  const u32 kUnknown = 0xFFFFFFFEu;

  u32 agreed = kNone;
  bool conflict = false;
  for (u32 r0 = chunkStart; r0 < winEnd; r0 += 64)
  {
    u32 cur = r0 + (u32)lane;
    bool alive = cur < winEnd;
    bool unknown = false;
    u32 prevSig = kNone;
    while (__any(alive && !unknown && cur < chunkEnd))
    {
      if (alive && !unknown && cur < chunkEnd)
      {
        BlkInfo b;
        const int rc = parseBlock<TBYTES>(blob, cur, blobEnd, p, wp.uniformN > 0 ? wp.uniformN : -1, maxCount, b);
        if (rc == 1) alive = false;
        else if (rc == 2)
        {
          // a raw block of a masked / ragged band: its length is the block's valid pixel count, which takes the block
          // index.  The candidate drops out -- one random byte in eight looks like such a block, so counting it as
          // "undecided" would leave no chunk decided.  If the true path goes with it, what the others agree on may be
          // wrong; the sweep (D3) only ever takes over walks that started where it arrives, so that costs time, not truth.
          if (wp.uniformN == 0) alive = false; else unknown = true;
        }
        else
        {
          const u32 sig = ((u32)b.flag >> 2) & pattern;
          if (prevSig != kNone && !sigFollows(prevSig, sig, p.mb, pattern)) alive = false;
          else { prevSig = sig; cur += b.len; }
        }
      }
    }
    const u32 e = unknown ? kUnknown : cur;
    const u32 lo = waveMin(alive ? e : kNone);
    const u32 hi = waveMax(alive ? e : 0u);
    if (lo != kNone)    // at least one survivor in this round
    {
      if (lo != hi) conflict = true;
      else if (agreed == kNone) agreed = lo;
      else if (agreed != lo) conflict = true;
    }
  }
  if (lane == 0) chunkExit[c] = (!conflict && agreed != kNone && agreed != kUnknown) ? agreed : kNone;
}

// D1 for chunks of 4 KiB (every 8 x 8 raster, 16 x 16 up to 32-bit types), without a walk.  A walk is a chain of dependent
// steps, one per block, a few hundred to a chunk, and all a workgroup's other lanes do is wait for it.  Instead: EVERY
// position of the chunk is parsed as if a block started there (one look at 16 bytes of LDS; most positions fail), which
// gives every position its successor, and pointer doubling turns successors into "where the stream leaves the chunk
// coming through here, and how many blocks that takes" in log2(blocks) rounds over all positions at once.  What the walks
// agreed on falls out of the table -- the candidates of the first window that leave the chunk have to name one exit --, and
// so does what D2 walked for: the number of blocks on the way from the chunk's entry, which is one of those candidates.
// Per candidate the table goes to candTab (8192^2 float with a 10 % mask: walks 0.98 ms, four sub-chunk walks joined by
// look-ups 0.55 ms, this R ms, and D2's 0.09 ms become a look-up).
static const u32 kMemoChunk = 4096, kMemoWindowMax = 1100;
static const u32 kRankSub = 1024;    // the KiB of a chunk that D4 gives a lane each

template<int TBYTES>
__global__ void __launch_bounds__(256) k_rank_chunks(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin,
                                                    u32 blobEnd, u32* __restrict__ chunkExit, u32* __restrict__ candTab, u32* __restrict__ chunkSub)
{
  __shared__ __align__(16) u8 s_bytes[kMemoChunk + kMemoWindowMax + 48];
  __shared__ u32 s_nc[kMemoChunk];    // successor relative to the chunk's start (16 bits; >= the chunk's length: outside; kDead) | blocks from here to there << 16
  __shared__ u32 s_flag[2], s_lo, s_hi, s_lowR;
  constexpr u32 E = kMemoChunk / 256u;
  const u32 kDead = 0xFFFFu;
  const u32 c = blockIdx.x;
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
  const u32 len = chunkEnd - chunkStart;
  const u32 stageEnd = min(chunkEnd + wp.window, blobEnd);
  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  // stage with 16-byte loads from the aligned-down start (never past the blob's end); LDS byte i + shift <-> blob byte chunkStart + i
  const u32 a0 = chunkStart & ~15u, shift = chunkStart - a0;
  for (u32 v = threadIdx.x; a0 + 16u * v < stageEnd; v += 256u)
  {
    const u32 g = a0 + 16u * v;
    uint4 x = make_uint4(0, 0, 0, 0);
    if (g + 16u <= blobEnd && ((uintptr_t)(blob + g) & 15u) == 0u) x = *reinterpret_cast<const uint4*>(blob + g);
    else for (u32 k2 = 0; k2 < 16u && g + k2 < blobEnd; k2++) (&x.x)[k2 >> 2] |= (u32)blob[g + k2] << (8u * (k2 & 3u));
    *reinterpret_cast<uint4*>(&s_bytes[16u * v]) = x;
  }
  if (threadIdx.x == 0) { s_lo = kNone; s_hi = 0u; s_lowR = kNone; s_flag[0] = 0u; s_flag[1] = 0u; }
  __syncthreads();
  const u32* s_words = reinterpret_cast<const u32*>(s_bytes);
  const u32 endRel = stageEnd - chunkStart;    // (what is staged of the stream, as seen from the chunk's start)

  // ---- every position's block, if it is one.  A raw block of a masked / ragged band (rc 2) has the length of its valid
  // pixel count, which takes the block index: no successor (the sweep, D3, walks such a chunk with the index in hand).
  // The column signature has to go on from block to block (sigFollows); checked on the links inside the chunk.
  // Most positions fall to a look at their first byte: a difference flag where there is one slice only, a raw block of unknown
  // length -- no block; a constant block or (every block full) a raw one -- its length is known.  Only a bit-stuffed block
  // (mode 1: an eighth of all byte values, nearly all true blocks) needs its header, and those positions are collected in a
  // queue per wave and parsed 64 at a time, every lane busy, instead of one lane in eight of a wave that parses all 64.
  __shared__ u16 s_queue[4][128];
  u32 nc[E];    // (a thread's own positions stay in registers; LDS holds what the others look up)
  const u32 offPack = offsetBytesPack(p);
  const bool sigChecked = p.mb == 8 || p.mb == 16 || p.mb == 32;
  const u32 sigStep = (p.mb == 8 && pattern == 14u) ? 2u : (u32)p.mb >> 3;
  const int nValidAll = wp.uniformN > 0 ? wp.uniformN : -1;
  const int wv = waveId();
  const u64 lt = laneMaskLt();
  // successor of position r whose block is bl bytes long (0: none) and whose flag byte is `flag`, as a packed word
  auto linked = [&](u32 r, u32 bl, u32 flag) -> u32
  {
    u32 nx = kDead;
    if (bl != 0u && bl < 4094u)    // (no block of a chunk this size is that long; a raw block of unknown length: no successor)
    {
      nx = r + bl;
      const u32 sgA = (flag >> 2) & pattern, sgB = ((u32)s_bytes[min(nx, len - 1u) + shift] >> 2) & pattern;
      const bool follows = !sigChecked | (sgB == sgA) | (sgB == ((sgA + sigStep) & pattern)) | (sgB == 0u);    // sigFollows, its constants taken out of the loop
      nx = (nx < len && !follows) ? kDead : nx;
    }
    return nx | (1u << 16);
  };
  auto parsed = [&](u32 r) -> u32
  {
    const Win16 h = loadWin16WordsRoomy(s_words, r + shift, endRel + shift);    // (r + shift < 4112, the array holds 5244 bytes)
    const u32 bl = blockLength<TBYTES>(h, r + shift, endRel + shift, p, offPack, nValidAll, maxCount);
    return linked(r, bl == kLenRawUnknown ? 0u : bl, (u32)h.lo & 255u);
  };
  u32 nQueued = 0;    // (the same in all lanes of the wave)
#pragma unroll 1
  for (u32 q = 0; q < E; q++)
  {
    const u32 r = q * 256u + threadIdx.x;
    const u32 flag = s_bytes[min(r, len - 1u) + shift];
    const u32 mode = flag & 3u, tc = flag >> 6;
    const u32 diff = (p.version >= 5) ? ((flag >> 2) & 1u) : 0u;
    const bool in = r < len && !(diff && p.nDepth == 1);
    const bool toQueue = in && mode == 1u;
    if (!toQueue)
    {
      const u32 offB = (offPack >> ((tc * 2u + diff) * 4u)) & 15u;
      u32 bl = (mode == 2u) ? 1u : (mode == 3u) ? (offB ? 1u + offB : 0u) : ((diff || nValidAll < 0) ? 0u : 1u + (u32)nValidAll * (u32)TBYTES);
      if (!in || r + bl > endRel) bl = 0u;
      s_nc[r] = linked(r, bl, flag);
    }
    const u64 bal = __ballot(toQueue);
    if (toQueue) s_queue[wv][nQueued + (u32)__popcll(bal & lt)] = (u16)r;
    nQueued += (u32)__popcll(bal);
    waveSync();
    if (nQueued >= 64u)
    {
      nQueued -= 64u;
      const u32 r2 = s_queue[wv][nQueued + (u32)laneId()];
      s_nc[r2] = parsed(r2);
      waveSync();
    }
  }
  if ((u32)laneId() < nQueued)
  {
    const u32 r2 = s_queue[wv][laneId()];
    s_nc[r2] = parsed(r2);
  }
  __syncthreads();
#pragma unroll
  for (u32 q = 0; q < E; q++) nc[q] = s_nc[q * 256u + threadIdx.x];
#ifdef HIPSIM
  for (u32 q = 0; q < E; q++)    // (emulator builds: the ordinary parser's verdict on every position)
  {
    const u32 r = q * 256u + threadIdx.x;
    u32 want = kDead | (1u << 16);
    if (r < len)
    {
      const Win16 h = loadWin16Words(s_words, r + shift, endRel + shift);
      BlkInfo b;
      const int rc = parseWindow<TBYTES>(h, r + shift, endRel + shift, p, nValidAll, maxCount, b);
      want = linked(r, rc == 0 ? b.len : 0u, (u32)h.lo & 255u);
    }
    if (want != nc[q]) { fprintf(stderr, "k_rank_chunks: position %u of chunk %u: %08x, the ordinary parser says %08x\n", r, c, nc[q], want); abort(); }
  }
#endif
  // ---- doubling, synchronous rounds: a hop that ends inside its KiB of the chunk is extended by the hop that starts where it
  // ends -- up to the KiB's end, not the chunk's: the few positions that have to know how the chunk is left (its candidates)
  // get there in four look-ups, and where the stream enters every KiB is what lets D4 walk the KiBs side by side.
  // Most positions are no block start or end in one after a hop or two; `live` says which of a thread's sixteen still go on.
  u32 live = 0u;
#pragma unroll
  for (u32 q = 0; q < E; q++) live |= ((nc[q] & 0xFFFFu) < min(len, ((q >> 2) + 1u) * kRankSub) ? 1u : 0u) << q;    // (position q * 256 + tid lies in KiB q / 4)
  for (u32 k = 0; ; k++)
  {
    u32 ch = 0u;
#pragma unroll
    for (u32 q = 0; q < E; q++)
      if ((live >> q) & 1u)
      {
        const u32 v = s_nc[nc[q] & 0xFFFFu];
        nc[q] = (v & 0xFFFFu) | ((nc[q] & 0xFFFF0000u) + (v & 0xFFFF0000u));
        ch |= 1u << q;
        if ((v & 0xFFFFu) >= min(len, ((q >> 2) + 1u) * kRankSub)) live &= ~(1u << q);
      }
    __syncthreads();    // (everybody has read what it wanted of the old table)
    if (ch)
    {
      s_flag[k & 1u] = 1u;
#pragma unroll
      for (u32 q = 0; q < E; q++) if ((ch >> q) & 1u) s_nc[q * 256u + threadIdx.x] = nc[q];
    }
    if (threadIdx.x == 0) s_flag[(k + 1u) & 1u] = 0u;    // (the next round's; whoever read it last did so before this round's first barrier)
    __syncthreads();
    if (!s_flag[k & 1u]) break;
  }
  // ---- the candidates: every byte up to one raw block + 1 behind the chunk's start (the stream's first block starts at
  // dataBegin).  Encoders never write a longer block (they fall back to raw); should a blob hold one, the exits agreed on
  // here may be wrong or missing, which D3 notices (it only takes over walks that started where it arrives) and pays for
  // with its own walk.
  const u32 nCand = (c == 0) ? 1u : min(wp.candWindow, len);
  u32 lo = kNone, hi = 0u;
  u32 myR = kNone, mySub[3] = { kNone, kNone, kNone };    // (a thread holds two candidates at most; the lower one that gets out is kept)
  for (u32 r = threadIdx.x; r < wp.candWindow; r += 256u)
  {
    u32 e = 0u;
    if (r < nCand)
    {
      // from KiB to KiB: where the chain enters each (landing | blocks before it << 16), where it leaves the chunk
      u32 at = r, cnt = 0u, land[3] = { kNone, kNone, kNone };
      bool out = false;
      for (u32 hop = 0; hop < kMemoChunk / kRankSub && !out; hop++)
      {
        const u32 v = s_nc[at];
        const u32 nx = v & 0xFFFFu;
        if (nx == kDead) break;
        cnt += v >> 16;
        if (nx >= len) { e = nx | (cnt << 16); out = true; }
        else { land[(nx / kRankSub) - 1u] = nx | (cnt << 16); at = nx; }    // (nx lies in a later KiB than `at`: 1 ... 3)
      }
      if (out)
      {
        const u32 x = e & 0xFFFFu;
        lo = min(lo, x); hi = max(hi, x);
        if (myR == kNone)
        {
          myR = r;
#pragma unroll
          for (int k = 0; k < 3; k++) mySub[k] = land[k] == kNone ? kNone : ((land[k] & 0xFFFFu) | (((e >> 16) - (land[k] >> 16)) << 16));    // landing | blocks from it to the exit << 16
        }
      }
    }
    candTab[(size_t)c * wp.candWindow + r] = e;    // (a block is a byte at least: x >= 1, e != 0)
  }
  lo = waveMin(lo); hi = waveMax(hi);
  const u32 lowR = waveMin(myR);
  if (laneId() == 0 && lo != kNone) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); atomicMin(&s_lowR, lowR); }
  __syncthreads();
  const bool agreed = s_lo != kNone && s_lo == s_hi;
  if (threadIdx.x == 0) chunkExit[c] = agreed ? chunkStart + s_lo : kNone;
  // what D4's four lanes per chunk start from: the lowest candidate's landings (every candidate that gets out runs into the
  // same chain sooner or later; D4's first lane checks that it arrives where this one did)
  if (agreed ? (myR != kNone && myR == s_lowR) : threadIdx.x == 0)
  {
#pragma unroll
    for (int k = 0; k < 3; k++) chunkSub[3u * c + k] = agreed ? mySub[k] : kNone;
  }
}

// D2 for chunks that k_rank_chunks has tabled: the entry is a candidate, its count and exit are in the table
__global__ void __launch_bounds__(256) k_walk_counts_tab(WalkPlan wp, u32 dataBegin, const u32* __restrict__ chunkExit,
                                                         const u32* __restrict__ candTab, u32* __restrict__ chunkCount, u32* __restrict__ exitOut)
{
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c >= wp.nChunks) return;
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 cur = (c == 0) ? dataBegin : chunkExit[c - 1];
  u32 e = 0u;
  if (cur != kNone && cur >= chunkStart && cur - chunkStart < wp.candWindow) e = candTab[(size_t)c * wp.candWindow + (cur - chunkStart)];
  chunkCount[c] = e ? (e >> 16) : kNone;
  exitOut[c] = e ? chunkStart + (e & 0xFFFFu) : kNone;
}

// D2: one lane per chunk whose entry is known (= the agreed exit of its predecessor) walks it: number of sub-blocks
// that start in the chunk and the true exit.  A raw block of a masked / ragged band cannot be sized without the block's
// index (its length is the block's valid pixel count), such a chunk is left to D3.
template<int TBYTES>
__global__ void __launch_bounds__(256) k_walk_counts(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                     const u32* __restrict__ chunkExit, u32* __restrict__ chunkCount, u32* __restrict__ exitOut)
{
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c >= wp.nChunks) return;
  const u32 chunkEnd = min(dataBegin + (c + 1) * wp.chunkBytes, blobEnd);
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  u32 cur = (c == 0) ? dataBegin : chunkExit[c - 1];
  u32 n = 0;
  bool ok = cur != kNone;
  while (ok && cur < chunkEnd)
  {
    BlkInfo b;
    if (parseBlock<TBYTES>(blob, cur, blobEnd, p, wp.uniformN > 0 ? wp.uniformN : -1, maxCount, b) != 0) ok = false;
    else { n++; cur += b.len; }
  }
  chunkCount[c] = ok ? n : kNone;
  exitOut[c] = ok ? cur : kNone;
}

// D2 / D4 out of LDS: the same two walks for streams whose chunks fit the table below.  A lane that steps through global
// memory pays three to four dependent round trips per block (flag -> offset type -> bit width -> count); here a
// workgroup stages kLdsWalkChunks chunks (+ the window a block may reach into behind each) with 16-byte loads and
// eight of its lanes walk them (the reference's 400 x 400 `california` blob, 43 chunks: call 0.45 -> 0.40 ms).
static const u32 kLdsWalkChunks = 8;
static const u32 kLdsWalkStride = ((kMemoChunk + kMemoWindowMax + 48 + 16 + 15) / 16) * 16;    // bytes per staged chunk
static_assert(kLdsWalkStride % 16 == 0, "16-byte staging");

// stages chunk c0 + k, k < kLdsWalkChunks, at s_bytes + k * kLdsWalkStride (LDS byte i + shift[k] <-> blob byte chunkStart + i)
__device__ __forceinline__ void stageWalkChunks(u8* s_bytes, u32* s_shift, const u8* __restrict__ blob, const WalkPlan& wp, u32 c0, u32 dataBegin, u32 blobEnd)
{
  for (u32 k = 0; k < kLdsWalkChunks; k++)
  {
    const u32 c = c0 + k;
    if (c >= wp.nChunks) break;
    const u32 chunkStart = dataBegin + c * wp.chunkBytes;
    const u32 stageEnd = min(min(chunkStart + wp.chunkBytes, blobEnd) + wp.window + 16u, blobEnd);
    const u32 a0 = chunkStart & ~15u;
    if (threadIdx.x == 0) s_shift[k] = chunkStart - a0;
    for (u32 v = threadIdx.x; a0 + 16u * v < stageEnd && 16u * v + 16u <= kLdsWalkStride; v += 256u)
    {
      const u32 g = a0 + 16u * v;
      uint4 x = make_uint4(0, 0, 0, 0);
      if (g + 16u <= blobEnd && ((uintptr_t)(blob + g) & 15u) == 0u) x = *reinterpret_cast<const uint4*>(blob + g);
      else for (u32 k2 = 0; k2 < 16u && g + k2 < blobEnd; k2++) (&x.x)[k2 >> 2] |= (u32)blob[g + k2] << (8u * (k2 & 3u));
      *reinterpret_cast<uint4*>(&s_bytes[k * kLdsWalkStride + 16u * v]) = x;
    }
  }
}

template<int TBYTES>
__global__ void __launch_bounds__(256) k_walk_counts_lds(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                         const u32* __restrict__ chunkExit, u32* __restrict__ chunkCount, u32* __restrict__ exitOut)
{
  __shared__ __align__(16) u8 s_bytes[kLdsWalkChunks * kLdsWalkStride];
  __shared__ u32 s_shift[kLdsWalkChunks];
  const u32 c0 = blockIdx.x * kLdsWalkChunks;
  stageWalkChunks(s_bytes, s_shift, blob, wp, c0, dataBegin, blobEnd);
  __syncthreads();
  const u32 c = c0 + threadIdx.x;
  if (threadIdx.x >= kLdsWalkChunks || c >= wp.nChunks) return;
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  u32 cur = (c == 0) ? dataBegin : chunkExit[c - 1];
  const u32* s_mine = reinterpret_cast<const u32*>(s_bytes + threadIdx.x * kLdsWalkStride);
  const u32 shift = s_shift[threadIdx.x];
  u32 n = 0;
  bool ok = cur != kNone;
  while (ok && cur < chunkEnd)
  {
    BlkInfo b;
    // (a block in front of the chunk -- an entry there has never been seen -- is read where it lies)
    const int rc = cur >= chunkStart ? parseBlockWords<TBYTES>(s_mine, cur - chunkStart + shift, blobEnd - chunkStart + shift, p, wp.uniformN > 0 ? wp.uniformN : -1, maxCount, b)
                                     : parseBlock<TBYTES>(blob, cur, blobEnd, p, wp.uniformN > 0 ? wp.uniformN : -1, maxCount, b);
    if (rc != 0) ok = false;
    else { n++; cur += b.len; }
  }
  chunkCount[c] = ok ? n : kNone;
  exitOut[c] = ok ? cur : kNone;
}

template<int TBYTES>
__global__ void __launch_bounds__(256) k_walk_emit_lds(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                       const u32* __restrict__ chunkEntry, const u32* __restrict__ chunkBase,
                                                       const u16* __restrict__ nValidBlk, u32* __restrict__ blockOff, DeviceStatus* st)
{
  __shared__ __align__(16) u8 s_bytes[kLdsWalkChunks * kLdsWalkStride];
  __shared__ u32 s_shift[kLdsWalkChunks];
  if (st->error) return;    // raised by the sweep: the entries cannot be trusted
  const u32 c0 = blockIdx.x * kLdsWalkChunks;
  stageWalkChunks(s_bytes, s_shift, blob, wp, c0, dataBegin, blobEnd);
  __syncthreads();
  const u32 c = c0 + threadIdx.x;
  if (threadIdx.x >= kLdsWalkChunks || c >= wp.nChunks) return;
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  const u32 nPos = (u32)p.nTV * (u32)p.nTH;
  u32 cur = chunkEntry[c];
  u32 pos = chunkBase[c];
  const u32 posEnd = chunkBase[c + 1];
  const u32* s_mine = reinterpret_cast<const u32*>(s_bytes + threadIdx.x * kLdsWalkStride);
  const u32 shift = s_shift[threadIdx.x];
  // (the valid count of the NEXT block is asked for while this one is parsed: it is the one load left on the way)
  const bool table = wp.uniformN <= 0 && nValidBlk != nullptr;
  u32 nvNext = (table && pos / (u32)p.nDepth < nPos) ? (u32)nValidBlk[pos / (u32)p.nDepth] : 0u;
  while (cur < chunkEnd && pos < posEnd)
  {
    const u32 blk = pos / (u32)p.nDepth;
    const int nValid = wp.uniformN > 0 ? wp.uniformN : ((table && blk < nPos) ? (int)nvNext : -1);
    const u32 blkNext = (pos + 1u) / (u32)p.nDepth;
    if (table && blkNext < nPos) nvNext = (u32)nValidBlk[blkNext];
    BlkInfo b;
    // (a block in front of the chunk -- an entry there has never been seen -- is read where it lies)
    const int rc = cur >= chunkStart ? parseBlockWords<TBYTES>(s_mine, cur - chunkStart + shift, blobEnd - chunkStart + shift, p, nValid, maxCount, b)
                                     : parseBlock<TBYTES>(blob, cur, blobEnd, p, nValid, maxCount, b);
    if (rc != 0) { raiseError(st, kFailed, 0x80000000u | c); break; }
    if (pos < wp.nSub) blockOff[pos] = cur;
    pos++;
    cur += b.len;
  }
}

// D3: one workgroup sweeps over the chunks in order and fixes, for every chunk, where its first block starts and which
// sub-block that is.  Chunks that D2 walked from the position the sweep arrives at are skipped in one step (their count
// and exit are taken over); the others -- raw blocks of a masked band, chunks behind a chunk whose candidates did not
// agree -- are staged in LDS by all threads and walked by one, now that the block index (and with it the valid pixel
// count of every block) is known.  entryExit[] holds D2's exits on entry and the chunks' entries on exit.
static const u32 kSweepChunkMax = 16384, kSweepWindowMax = 8208;
static const u32 kSweepThreads = 1024;    // chunks looked at per step (8192^2 float: 23 500 chunks, 0.13 ms with 256, R with 1024)

template<int TBYTES>
__global__ void __launch_bounds__(kSweepThreads) k_walk_sweep(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                    const u32* __restrict__ chunkExit, const u32* __restrict__ chunkCount,
                                                    u32* __restrict__ entryExit, const u16* __restrict__ nValidBlk,
                                                    u32* __restrict__ chunkBase, DeviceStatus* st)
{
  __shared__ u32 s_from[kSweepThreads], s_cnt[kSweepThreads], s_exit[kSweepThreads];
  __shared__ u32 s_cur, s_pos, s_todo, s_bad, s_run[kSweepThreads / 64], s_wsum[kSweepThreads / 64];
  __shared__ __align__(16) u8 s_chunk[kSweepChunkMax + kSweepWindowMax + 16];
  __shared__ u16 s_nv[kSweepChunkMax];
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  const u32 nPos = (u32)p.nTV * (u32)p.nTH;
  if (threadIdx.x == 0) { s_cur = dataBegin; s_pos = 0; s_bad = 0; }
  if (wp.chunkBytes > kSweepChunkMax || wp.window > kSweepWindowMax) { if (threadIdx.x == 0) raiseError(st, kFailed, 0x20000000u); return; }
  __syncthreads();
  // (a step's three loads are asked for one step ahead: a step is a handful of barriers, and the loads' way from memory was
  // as long as all of them together)
  u32 nextFrom = kNone, nextCnt = kNone, nextExit = kNone;
  auto ask = [&](u32 base)
  {
    const u32 c = base + threadIdx.x;
    const bool in = c < wp.nChunks;
    nextFrom = !in ? kNone : (c == 0 ? dataBegin : chunkExit[c - 1]);
    nextCnt = in ? chunkCount[c] : kNone;
    nextExit = in ? entryExit[c] : kNone;    // (D2's exits; this kernel writes the entries of chunks it has placed only)
  };
  ask(0u);
  for (u32 base = 0; base < wp.nChunks; base += kSweepThreads)
  {
    const u32 batchEnd = min(base + kSweepThreads, wp.nChunks);
    s_from[threadIdx.x] = nextFrom; s_cnt[threadIdx.x] = nextCnt; s_exit[threadIdx.x] = nextExit;
    if (base + kSweepThreads < wp.nChunks) ask(base + kSweepThreads);
    __syncthreads();
    u32 c = base;    // first chunk of the batch that is not placed yet (same in every thread)
    for (;;)
    {
      // chunks c, c + 1, ... can be taken over from D2 as long as each was walked from where its predecessor ended
      // (the first one: from where the sweep stands); their bases are a running sum of the counts
      const u32 t = threadIdx.x, k = c + t;    // this thread looks at chunk k
      const bool in = k < batchEnd;
      const u32 cntK = in ? s_cnt[k - base] : kNone;
      const u32 prevEnd = (t == 0) ? s_cur : ((k - 1 < batchEnd) ? s_exit[k - 1 - base] : kNone);
      const bool ok = in && cntK != kNone && s_from[k - base] == prevEnd;
      // length of the run of ok's from thread 0, and the counts' prefix sums over it
      const u64 bad = __ballot(!ok);
      if (laneId() == 0) s_run[waveId()] = bad ? (u32)(__ffsll((long long)bad) - 1) : 64u;
      u32 inc = ok ? cntK : 0u;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (laneId() >= d) inc += o; }
      if (laneId() == 63) s_wsum[waveId()] = inc;
      __syncthreads();
      u32 run = 0, before = 0;
      for (int w2 = 0; w2 < (int)(kSweepThreads / 64); w2++) { run += s_run[w2]; if (s_run[w2] < 64u) break; }
      for (int w2 = 0; w2 < waveId(); w2++) before += s_wsum[w2];
      const u32 pos0 = s_pos, cur0 = s_cur;
      if (t < run) { entryExit[k] = (t == 0) ? cur0 : prevEnd; chunkBase[k] = pos0 + before + inc - cntK; }
      __syncthreads();
      if (run > 0 && t == run - 1) { s_cur = s_exit[k - base]; s_pos = pos0 + before + inc; }
      if (t == 0) s_todo = c + run;
      c += run;
      __syncthreads();
      const u32 todo = s_todo;
      if (todo >= batchEnd || s_bad) break;
      // stage chunk `todo` (from its start: the entry lies at or behind it) and the valid counts of the blocks from s_pos on
      const u32 chunkStart = dataBegin + todo * wp.chunkBytes;
      const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
      const u32 stageEnd = min(chunkEnd + wp.window, blobEnd);
      for (u32 i = threadIdx.x; chunkStart + i < stageEnd; i += kSweepThreads) s_chunk[i] = blob[chunkStart + i];
      if (nValidBlk)
      {
        const u32 pos0 = s_pos / (u32)p.nDepth;
        for (u32 i = threadIdx.x; i < wp.chunkBytes && pos0 + i < nPos; i += kSweepThreads) s_nv[i] = nValidBlk[pos0 + i];
      }
      __syncthreads();
      if (threadIdx.x == 0)
      {
        u32 cur = s_cur, pos = s_pos;
        const u32 pos0 = pos / (u32)p.nDepth;
        entryExit[todo] = cur; chunkBase[todo] = pos;
        bool ok = cur >= chunkStart;
        while (ok && cur < chunkEnd)
        {
          const u32 blk = pos / (u32)p.nDepth;
          const int nValid = wp.uniformN > 0 ? wp.uniformN : ((nValidBlk && blk < nPos) ? (int)s_nv[blk - pos0] : -1);
          BlkInfo b;
          if (blk >= nPos || parseBlockWords<TBYTES>(reinterpret_cast<const u32*>(s_chunk), cur - chunkStart, stageEnd - chunkStart, p, nValid, maxCount, b) != 0) ok = false;
          else { pos++; cur += b.len; }
        }
        if (!ok) { raiseError(st, kFailed, 0x10000000u | todo); s_bad = 1; }
        s_cur = cur; s_pos = pos;
      }
      c = todo + 1;
      __syncthreads();
      if (s_bad) break;
    }
    if (s_bad) break;
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    entryExit[wp.nChunks] = s_cur; chunkBase[wp.nChunks] = s_pos;
    if (!s_bad && (s_pos != wp.nSub || s_cur != blobEnd)) raiseError(st, kFailed, 0x40000000u);
  }
}

// D4: one lane per chunk walks from the resolved entry and writes the offsets of the sub-blocks that start in it
template<int TBYTES>
__global__ void __launch_bounds__(256) k_walk_emit(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                   const u32* __restrict__ chunkEntry, const u32* __restrict__ chunkBase,
                                                   const u16* __restrict__ nValidBlk, u32* __restrict__ blockOff, DeviceStatus* st)
{
  if (st->error) return;    // raised by the sweep: the entries cannot be trusted
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c >= wp.nChunks) return;
  const u32 chunkEnd = min(dataBegin + (c + 1) * wp.chunkBytes, blobEnd);
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  const u32 nPos = (u32)p.nTV * (u32)p.nTH;
  u32 cur = chunkEntry[c];
  u32 pos = chunkBase[c];
  const u32 posEnd = chunkBase[c + 1];
  while (cur < chunkEnd && pos < posEnd)
  {
    const u32 blk = pos / (u32)p.nDepth;
    const int nValid = wp.uniformN > 0 ? wp.uniformN : ((nValidBlk && blk < nPos) ? (int)nValidBlk[blk] : -1);
    BlkInfo b;
    if (parseBlock<TBYTES>(blob, cur, blobEnd, p, nValid, maxCount, b) != 0) { raiseError(st, kFailed, 0x80000000u | c); break; }
    if (pos < wp.nSub) blockOff[pos] = cur;
    pos++;
    cur += b.len;
  }
}

// D4 for chunks that k_rank_chunks has tabled: four lanes a chunk, a KiB each.  Lane 0 starts at the chunk's entry, the
// others where the lowest candidate's chain entered their KiB, with the block index counted back from the chunk's end
// (chunkSub: landing | blocks from there to the exit << 16) -- the entry's chain is that chain from wherever the two meet.
// Whether they have met by the first landing is what lane 0 finds out first: it walks its KiB, and only if it arrives at
// that landing with that block index do the other lanes walk theirs (side by side: a chain of two KiB instead of four);
// if not (a candidate that joins the true chain late: never seen), lane 0 walks on to the chunk's end alone.
template<int TBYTES>
__global__ void __launch_bounds__(256) k_walk_emit_sub(BandParams p, WalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
                                                       const u32* __restrict__ chunkEntry, const u32* __restrict__ chunkBase,
                                                       const u32* __restrict__ chunkSub, const u16* __restrict__ nValidBlk,
                                                       u32* __restrict__ blockOff, DeviceStatus* st)
{
  if (st->error) return;    // raised by the sweep: the entries cannot be trusted
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  const u32 c = min(t >> 2, wp.nChunks - 1u), k = t & 3u;
  const bool mine = (t >> 2) < wp.nChunks;
  const u32 chunkStart = dataBegin + c * wp.chunkBytes;
  const u32 chunkEnd = min(chunkStart + wp.chunkBytes, blobEnd);
  const u32 maxCount = (u32)p.mb * (u32)p.mb;
  const u32 nPos = (u32)p.nTV * (u32)p.nTH;
  const u32 posBegin = chunkBase[c], posEnd = chunkBase[c + 1];
  const u32 total = posEnd - posBegin;
  u32 sub[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
  {
    sub[i] = chunkSub[3u * c + i];
    if (sub[i] != kNone && (wp.test & 16u)) sub[i] += 1u;    // (test: a landing that is none)
    if (sub[i] != kNone && ((sub[i] & 0xFFFFu) >= chunkEnd - chunkStart || (sub[i] >> 16) >= total)) sub[i] = kNone;    // (never: an entry that cannot be; a block lies in front of a landing)
  }
  const bool table = wp.uniformN <= 0 && nValidBlk != nullptr;
  // the stretch from `cur` (block index `pos`) to `stop` (block index `posStop`); false: something on the way is no block
  auto walk = [&](u32& cur, u32& pos, u32 stop, u32 posStop) -> bool
  {
    while (cur < stop && pos < posStop)
    {
      const u32 blk = pos / (u32)p.nDepth;
      const int nValid = wp.uniformN > 0 ? wp.uniformN : ((table && blk < nPos) ? (int)nValidBlk[blk] : -1);
      BlkInfo b;
      if (parseBlock<TBYTES>(blob, cur, blobEnd, p, nValid, maxCount, b) != 0) return false;
      if (pos < wp.nSub) blockOff[pos] = cur;
      pos++;
      cur += b.len;
    }
    return true;
  };
  // the first landing there is: lane 0's first stop
  u32 firstStop = chunkEnd, firstPos = posEnd;
#pragma unroll
  for (int i = 2; i >= 0; i--) if (sub[i] != kNone) { firstStop = chunkStart + (sub[i] & 0xFFFFu); firstPos = posEnd - (sub[i] >> 16); }
  u32 cur0 = chunkEntry[c], pos0 = posBegin;
  u32 verdict = 0u;    // 0: lane 0 goes on alone; 1: the landings hold, the other lanes walk; 2: the chunk is done; 3: no block on the way
  if (mine && k == 0u)
  {
    if (!walk(cur0, pos0, firstStop, firstPos)) { raiseError(st, kFailed, 0x80000000u | c); verdict = 3u; }
    else verdict = (firstStop == chunkEnd) ? 2u : ((cur0 == firstStop && pos0 == firstPos && !(wp.test & 8u)) ? 1u : 0u);
  }
  verdict = (u32)__shfl((int)verdict, laneId() & ~3);
  if (!mine) return;
  if (k == 0u)
  {
    if (verdict == 0u && !walk(cur0, pos0, chunkEnd, posEnd)) raiseError(st, kFailed, 0x80000000u | c);
    return;
  }
  if (verdict != 1u || sub[k - 1u] == kNone) return;
  u32 cur = chunkStart + (sub[k - 1u] & 0xFFFFu), pos = posEnd - (sub[k - 1u] >> 16), stop = chunkEnd, posStop = posEnd;
#pragma unroll
  for (int i = 2; i >= 0; i--)
    if ((u32)i + 1u > k && sub[i] != kNone) { stop = chunkStart + (sub[i] & 0xFFFFu); posStop = posEnd - (sub[i] >> 16); }
  if (!walk(cur, pos, stop, posStop)) raiseError(st, kFailed, 0x80000000u | c);    // (on the entry's own chain: what is no block here is none)
}

// candidates per chunk -> exits the candidates agree on -> counts where the entry is known -> one sweep that closes
// the gaps (and sizes the raw blocks of masked / ragged bands, whose length hangs on the block index) -> offsets.
// The first step needs neither the mask nor the blocks' valid counts and is launched on its own.
template<int TBYTES>
static void launchWalkChunksT(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, hipStream_t stream)
{
  if (wp.tabled && wb.candTab && wb.chunkSub)
    hipLaunchKernelGGL(k_rank_chunks<TBYTES>, dim3(wp.nChunks), dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, wb.chunkExit, wb.candTab, wb.chunkSub);
  else
    hipLaunchKernelGGL(k_walk_chunks<TBYTES>, dim3(wp.nChunks), dim3(64), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, wb.chunkExit);
}

template<int TBYTES>
static void launchWalkRestT(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, DeviceStatus* st,
                            hipStream_t stream)
{
  const dim3 gridC((wp.nChunks + 255) / 256), gridL((wp.nChunks + kLdsWalkChunks - 1) / kLdsWalkChunks);
  // Small streams walk out of LDS (eight lanes per workgroup, 0.3 us a step instead of 2); large ones have tens of thousands
  // of lanes in flight to hide the round trips and are faster one lane per chunk (8192^2 with a 10 % mask, same box: 1.52 against 1.84 ms)
  const bool lds = wp.chunkBytes <= kMemoChunk && wp.window <= kMemoWindowMax && wp.nChunks <= 1024u;
  if (wp.tabled && wb.candTab)
    hipLaunchKernelGGL(k_walk_counts_tab, gridC, dim3(256), 0, stream, wp, a.dataBegin, (const u32*)wb.chunkExit, (const u32*)wb.candTab,
                       wb.chunkCount, wb.chunkEntry);
  else if (lds)
    hipLaunchKernelGGL(k_walk_counts_lds<TBYTES>, gridL, dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, (const u32*)wb.chunkExit,
                       wb.chunkCount, wb.chunkEntry);
  else
    hipLaunchKernelGGL(k_walk_counts<TBYTES>, gridC, dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, (const u32*)wb.chunkExit,
                       wb.chunkCount, wb.chunkEntry);
  hipLaunchKernelGGL(k_walk_sweep<TBYTES>, dim3(1), dim3(kSweepThreads), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, (const u32*)wb.chunkExit,
                     (const u32*)wb.chunkCount, wb.chunkEntry, wb.nValidBlk, wb.chunkBase, st);
#ifdef HIPSIM
  if (getenv("LERC_DEBUG_WALK"))
  {
    hipStreamSynchronize(stream);
    u32 noExit = 0, noCount = 0, mism = 0;
    for (u32 c = 0; c < wp.nChunks; c++)
    {
      if (wb.chunkExit[c] == kNone) { noExit++; fprintf(stderr, "  no exit: chunk %u (bytes %u .. %u)\n", c, a.dataBegin + c * wp.chunkBytes, a.dataBegin + (c + 1) * wp.chunkBytes); }
      if (wb.chunkCount[c] == kNone) noCount++;
    }
    fprintf(stderr, "walk debug: %u chunks, %u without agreed exit, %u without count\n", wp.nChunks, noExit, noCount);
  }
#endif
  if (wp.tabled && wb.candTab)
    hipLaunchKernelGGL(k_walk_emit_sub<TBYTES>, dim3((4u * wp.nChunks + 255u) / 256u), dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd,
                       (const u32*)wb.chunkEntry, (const u32*)wb.chunkBase, (const u32*)wb.chunkSub, wb.nValidBlk, wb.blockOff, st);
  else if (lds)
    hipLaunchKernelGGL(k_walk_emit_lds<TBYTES>, gridL, dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, (const u32*)wb.chunkEntry,
                       (const u32*)wb.chunkBase, wb.nValidBlk, wb.blockOff, st);
  else
    hipLaunchKernelGGL(k_walk_emit<TBYTES>, gridC, dim3(256), 0, stream, p, wp, a.blob, a.dataBegin, a.blobEnd, (const u32*)wb.chunkEntry,
                       (const u32*)wb.chunkBase, wb.nValidBlk, wb.blockOff, st);
}

void launchWalkChunks(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, hipStream_t stream)
{
  switch (dtSize(p.dt))
  {
    case 1: launchWalkChunksT<1>(p, wp, a, wb, stream); break;
    case 2: launchWalkChunksT<2>(p, wp, a, wb, stream); break;
    case 4: launchWalkChunksT<4>(p, wp, a, wb, stream); break;
    default: launchWalkChunksT<8>(p, wp, a, wb, stream); break;
  }
}

void launchWalkRest(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, DeviceStatus* st,
                    hipStream_t stream)
{
  switch (dtSize(p.dt))
  {
    case 1: launchWalkRestT<1>(p, wp, a, wb, st, stream); break;
    case 2: launchWalkRestT<2>(p, wp, a, wb, st, stream); break;
    case 4: launchWalkRestT<4>(p, wp, a, wb, st, stream); break;
    default: launchWalkRestT<8>(p, wp, a, wb, st, stream); break;
  }
}

void launchWalk(const BandParams& p, const WalkPlan& wp, const DecodeArgs& a, const WalkBuffers& wb, DeviceStatus* st,
                hipStream_t stream)
{
  launchWalkChunks(p, wp, a, wb, stream);
  launchWalkRest(p, wp, a, wb, st, stream);
}

}    // namespace lerc
