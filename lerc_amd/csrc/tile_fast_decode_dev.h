// tile_fast_decode_dev.h -- device helpers shared by the streaming decoder kernels (tile_fast_decode.hip: discovery +
// resolve / decode in two launches; tile_fast_decode_one.hip: everything in one launch): branch-free block header
// parsing (Lerc2::ReadTile, Lerc2.cpp:2025-2110; BitStuffer2::Decode, BitStuffer2.cpp:159-258), the band header
// (Lerc2::ReadHeader, Lerc2.cpp:790-917), the dequantiser (Lerc2.cpp:2159-2160).
#pragma once
#include "tile_fast.h"
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

static const u32 kNoOffset = 0xFFFFFFFFu;
// raises flag k of the call (tile_fast.h): in the device cell, and -- one band, whose verdict the host reads straight out of
// pinned memory -- in the host's
__device__ __forceinline__ void raiseFlag(const FastDecodeBuffers& b, int k)
{
  b.fallback[k] = b.epoch;
  if (b.hostFallback) b.hostFallback[k] = b.epoch;
}

// The decoded pixels leave with the non-temporal hint: nothing in the call reads them again, and written the ordinary
// way they sit dirty in L2 / the Infinity Cache until the NEXT kernel's traffic pushes them out (measured on C2: this
// kernel 79 -> 71 us, the statistics pass of the following encode 77 -> 53 us).  The same hint on the blob loads here or in
// k_fast_discover costs 5-8 us: those bytes were just written by the kernel in front and are still on the die.
#ifdef LERC_DECODE_PLAIN_STORES    // (tuning: the pixels written the ordinary way)
#define DECODE_STORE(ptr, val) (*(ptr) = (val))
#else
#define DECODE_STORE(ptr, val) storeStreaming(ptr, val)
#endif

// ------------------------------------------------------------------------------------------------
// block header parsing, branch free
// ------------------------------------------------------------------------------------------------
// code word of one block: len (10) | mode (2) << 10 | lut << 12 | numBits (5) << 13 | nLut (8) << 18 | offBytes (4) << 26
__device__ __forceinline__ u32 codeLen(u32 c) { return c & 1023u; }
__device__ __forceinline__ u32 codeMode(u32 c) { return (c >> 10) & 3u; }
__device__ __forceinline__ u32 codeLut(u32 c) { return (c >> 12) & 1u; }
__device__ __forceinline__ u32 codeBits(u32 c) { return (c >> 13) & 31u; }
__device__ __forceinline__ u32 codeNLut(u32 c) { return (c >> 18) & 255u; }
__device__ __forceinline__ u32 codeOffBytes(u32 c) { return (c >> 26) & 15u; }

// bytes of the block offset for each of the 4 type codes, a nibble each (0 = no such type); Lerc2.h:528-542
template<int DT> __device__ __forceinline__ u32 offBytesTable()
{
  u32 t = 0;
#pragma unroll
  for (int tc = 0; tc < 4; tc++)
  {
    const int dtU = typeUsed(DT, tc);
    t |= (u32)(dtU == DT_Undefined ? 0 : dtSize(dtU)) << (4 * tc);
  }
  return t;
}

// Rasters whose rows / columns are no multiples of 8 (RAG): the blocks of the last block column are wl = nCols % 8 pixels wide,
// those of the last block row hl = nRows % 8 high (Lerc2.cpp:1504-1519), so a block holds 64, 8 wl, 8 hl or wl hl elements --
// its count byte says which; the decode kernel, which knows where a block lies, checks that it is the right one.
struct RagCounts
{
  u32 cR = 64u, cB = 64u, cC = 64u;    // right edge, bottom edge, corner (64 where there is no such edge)
  __device__ __forceinline__ bool allowed(u32 c) const { return (c == 64u) | (c == cR) | (c == cB) | (c == cC); }
};
__device__ __forceinline__ RagCounts ragCounts(int nRows, int nCols)
{
  const u32 wl = (u32)nCols & 7u, hl = (u32)nRows & 7u;
  RagCounts rc;
  rc.cR = wl ? 8u * wl : 64u; rc.cB = hl ? 8u * hl : 64u; rc.cC = (wl ? wl : 8u) * (hl ? hl : 8u);
  return rc;
}

// the first 12 bytes at LDS byte offset rel: one round of aligned word reads + funnel shifts
template<int DT>
__device__ __forceinline__ void ldsHeader(const u32* words, u32 rel, u32& h0, u32& h1, u32& h2)
{
  const u32 w = rel >> 2, sh = 8u * (rel & 3u);
  const u32 x0 = words[w], x1 = words[w + 1], x2 = words[w + 2];
  h0 = (u32)((((u64)x1 << 32) | x0) >> sh);
  h1 = (u32)((((u64)x2 << 32) | x1) >> sh);
  h2 = 0;
  if (DT == DT_Double) { const u32 x3 = words[w + 3]; h2 = (u32)((((u64)x3 << 32) | x2) >> sh); }
}

// Code word of the all-valid 8 x 8 block (64 elements) of data type DT whose first 12 bytes are h0 h1 h2, or 0
// if no valid block starts there.  The caller checks that the block ends inside the stream.  Mirrors
// Lerc2::ReadTile (Lerc2.cpp:2025-2110) and BitStuffer2::Decode (BitStuffer2.cpp:159-258); blocks longer than
// the raw form are refused (the reference encoder never writes one; the general kernels take such blobs).
template<int DT>
__device__ __forceinline__ u32 parseCode(u32 h0, u32 h1, u32 h2, int version, u32 n = 64u)    // n: elements of the block (64 unless the raster is ragged)
{
  constexpr u32 TB = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  const u32 RAW = 1 + n * TB;
  const u32 flag = h0 & 0xFFu, mode = flag & 3u;
  const u32 offB = (offBytesTable<DT>() >> ((flag >> 4) & 12u)) & 15u;
  const u64 hdr = ((u64)h1 << 32) | h0;
  u32 t = (u32)(hdr >> ((8u + 8u * offB) & 63u));    // bytes 1 + offB ...: numBits byte, count, LUT size
  if (DT == DT_Double) t = (offB == 8u) ? (h2 >> 8) : t;
  const u32 nb = t & 31u, lut = (t >> 5) & 1u;
  const u32 nLut = ((t >> 16) & 0xFFu) - 1u;                           // valid: 1 ... 254
  const bool okBits = ((t & 0xFFC0u) == ((n << 8) | 0x80u)) & (nb != 0u);    // one-byte count field == n
  const bool okLut = (nLut - 1u) < 254u;
  const u32 lenSimple = 3u + offB + ((n * nb + 7u) >> 3);
  const u32 lenLut = 4u + offB + (((nLut & 0xFFu) * nb + 7u) >> 3) + ((n * (u32)bitLen(nLut & 0xFFu) + 7u) >> 3);
  const u32 len = (mode == 0u) ? RAW : (mode == 2u) ? 1u : (mode == 3u) ? 1u + offB : (lut ? lenLut : lenSimple);
  bool ok = (mode == 0u) | (mode == 2u) | ((offB != 0u) & ((mode == 3u) | (okBits & ((lut == 0u) | okLut))));
  ok = ok & !((version >= 5) & ((flag & 4u) != 0u)) & (len <= RAW);    // slice difference needs nDepth > 1
  const u32 code = len | (mode << 10) | (lut << 12) | (nb << 13) | ((nLut & 0xFFu) << 18) | (offB << 26);
  return ok ? code : 0u;
}

__device__ __forceinline__ bool sigOk(u32 prev, u32 cur, u32 pattern)
{
  const u32 step = (pattern == 14u) ? 2u : 1u;    // 8 x 8 blocks: signature = (j0 >> 3) & pattern
  return (cur == prev) | (cur == ((prev + step) & pattern)) | (cur == 0u);
}

// one step of a walk: the block at `cur` (absolute), or 0
template<int DT>
__device__ __forceinline__ u32 stepAt(const u32* words, u32 a0, u32 cur, u32 end, int version, u32& sig, u32 pattern)
{
  u32 h0, h1, h2;
  ldsHeader<DT>(words, cur - a0, h0, h1, h2);
  const u32 code = parseCode<DT>(h0, h1, h2, version);
  const u32 sg = (h0 >> 2) & pattern;
  const bool ok = (code != 0u) & (cur + codeLen(code) <= end) & ((sig == kNoOffset) | sigOk(sig, sg, pattern));
  sig = sg;
  return ok ? code : 0u;
}

// The same step written for the walks' critical path, in three parts: the LDS words a block's first bytes lie in, the
// block's length from them (a dozen dependent operations, no branch), and whether it is a block at all with a signature
// that follows the previous one's (off the critical path: a walk fetches the words of the NEXT block before it looks at
// that).  UNIFORM: all lanes of the wave are here (the look-up table arithmetic is then skipped unless some lane has
// such a block).
template<int DT> struct LeanWords { u32 x0, x1, x2, x3; };
template<int DT> __device__ __forceinline__ LeanWords<DT> leanFetch(const u32* words, u32 rel)    // rel: inside the staged bytes
{
  const u32 wi = rel >> 2;
  LeanWords<DT> v;
  v.x0 = words[wi]; v.x1 = words[wi + 1]; v.x2 = words[wi + 2];
  v.x3 = (DT == DT_Double) ? words[wi + 3] : 0u;
  return v;
}
struct LeanBlock { u32 h0, t, offB, len, okLut; };
// (RAG: toEnd = bytes from the block's start to the end of the blob.  A raw block's length hangs on where the block lies, which
// a walk does not know: it takes raw blocks for whole ones -- except the raster's very last block, the corner, which is
// one when it ends the blob exactly: a corner of one pixel is ALWAYS raw, 5 bytes either way and raw wins ties)
template<int DT, bool UNIFORM, bool RAG = false>
__device__ __forceinline__ LeanBlock leanLength(const LeanWords<DT>& v, u32 rel, u32 toEnd = 0u, const RagCounts& rc = RagCounts())
{
  constexpr u32 TB = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 RAW = 1 + 64 * TB;
  const u32 sh = 8u * rel;    // (alignbit takes the shift mod 32)
  LeanBlock k;
  const u32 h0 = __builtin_amdgcn_alignbit(v.x1, v.x0, sh), h1 = __builtin_amdgcn_alignbit(v.x2, v.x1, sh);
  const u32 mode = h0 & 3u;
  const u32 offB = (offBytesTable<DT>() >> ((h0 >> 4) & 12u)) & 15u;
  u32 t;    // bytes 1 + offB ...: numBits byte, count, LUT size
  if (DT == DT_Double)
  {
    const u32 h2 = __builtin_amdgcn_alignbit(v.x3, v.x2, sh);
    t = (offB == 8u) ? (h2 >> 8) : (u32)((((u64)h1 << 32) | h0) >> ((8u + 8u * offB) & 63u));
  }
  else t = (u32)((((u64)h1 << 32) | h0) >> (8u + 8u * offB));    // (offsets of at most 4 bytes: no shift beyond 40)
  const u32 nb = t & 31u, lut = (t >> 5) & 1u;
  const u32 cnt = RAG ? ((t >> 8) & 0xFFu) : 64u;                          // (RAG: the stream says how many elements the block holds)
  u32 lenStuffed = 3u + offB + (RAG ? ((cnt * nb + 7u) >> 3) : 8u * nb);
  k.okLut = 1u;
  if (!UNIFORM || __builtin_amdgcn_ballot_w64((t & 32u) != 0u && mode == 1u) != 0ull)
  {
    const u32 nLut = (((t >> 16) & 0xFFu) - 1u) & 0xFFu;                   // valid: 1 ... 254
    k.okLut = (lut ^ 1u) | (u32)((nLut - 1u) < 254u);
    const u32 lenLut = 4u + offB + ((nLut * nb + 7u) >> 3) + (RAG ? ((cnt * (u32)bitLen(nLut) + 7u) >> 3) : 8u * (u32)bitLen(nLut));
    lenStuffed = lut ? lenLut : lenStuffed;
  }
  u32 lenRaw = RAW;
  if (RAG) lenRaw = (toEnd == 1u + rc.cC * TB) ? toEnd : RAW;
  const u32 lenOther = (mode == 0u) ? lenRaw : (mode == 2u) ? 1u : 1u + offB;
  k.h0 = h0; k.t = t; k.offB = offB;
  k.len = (mode == 1u) ? lenStuffed : lenOther;
  return k;
}
template<int DT, bool RAG = false>
__device__ __forceinline__ bool leanValid(const LeanBlock& k, u32 remaining, bool v5, u32 prevSig, u32 pattern, u32& sigOut, const RagCounts& rc = RagCounts())
{
  constexpr u32 TB = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 RAW = 1 + 64 * TB;
  const u32 mode = k.h0 & 3u, nb = k.t & 31u;
  const u32 okBits = RAG ? ((u32)((k.t & 0xC0u) == 0x80u) & (u32)(nb != 0u) & (u32)rc.allowed((k.t >> 8) & 0xFFu))
                         : ((u32)((k.t & 0xFFC0u) == 0x4080u) & (u32)(nb != 0u));  // one-byte count field == 64 (RAG: one of the block sizes)
  const u32 okStuffed = okBits & (u32)(k.offB != 0u) & k.okLut;
  const u32 okOther = (u32)(mode != 3u) | (u32)(k.offB != 0u);
  u32 ok = (mode == 1u) ? okStuffed : okOther;
  ok &= (u32)!(v5 && (k.h0 & 4u)) & (u32)(k.len <= remaining);             // slice difference needs nDepth > 1
  if (3u + 8u + 8u * 31u > RAW) ok &= (u32)(k.len <= RAW);                 // (32-bit and wider types: no stuffed block is that long)
  const u32 sg = (k.h0 >> 2) & pattern;
  ok &= (u32)(prevSig == kNoOffset) | (u32)sigOk(prevSig, sg, pattern);
  sigOut = sg;
  return ok != 0u;
}
// What a walk needs to know of that: enough to end a walk that is not on the path within a few steps (the signature
// sequence alone ends 5 of 8 per step; "bit-stuffed" with anything but a header byte and the count 64 behind the offset
// nearly all the rest) and to stay inside the blob.  Everything else about a block is checked, in full, by the decode
// kernel (parseCode, contiguity, the column signature), which sends a damaged blob to the general path.
__device__ __forceinline__ bool leanPlausible(const LeanBlock& k, u32 remaining, u32 prevSig, u32 pattern, u32& sigOut)
{
  const u32 mode = k.h0 & 3u;
  const u32 okBits = (u32)((k.t & 0xFFC0u) == 0x4080u) & (u32)((k.t & 31u) != 0u);
  u32 ok = (u32)(mode != 1u) | okBits;
  ok &= (u32)(k.len <= remaining);
  const u32 sg = (k.h0 >> 2) & pattern;
  ok &= (u32)(prevSig == kNoOffset) | (u32)sigOk(prevSig, sg, pattern);
  sigOut = sg;
  return ok != 0u;
}
// all three at once: the block's length or 0
template<int DT, bool UNIFORM, bool RAG = false>
__device__ __forceinline__ u32 stepLean(const u32* words, u32 rel, u32 remaining, bool v5, u32 prevSig, u32 pattern, u32& sigOut, const RagCounts& rc = RagCounts())
{
  const LeanBlock k = leanLength<DT, UNIFORM, RAG>(leanFetch<DT>(words, rel), rel, remaining, rc);
  return leanValid<DT, RAG>(k, remaining, v5, prevSig, pattern, sigOut, rc) ? k.len : 0u;
}

// ------------------------------------------------------------------------------------------------
// header
// ------------------------------------------------------------------------------------------------
// The band header is read the way Lerc2::ReadHeader / ReadMask / ReadMinMaxRanges do (Lerc2.cpp:790-1008,
// :2642-2677) to decide whether the streaming kernels may take the band.  Every discovery wave does it for itself
// (all lanes alike: the same 128 bytes, no divergence, so it costs what one lane would); the first one leaves the
// result in *P for the later kernels and the host, which can therefore enqueue the whole decode without having seen a
// single byte of the blob; it checks P->ok (and the fallback flags) when it reads the results back.
// The first 128 bytes of the band sit in 32 registers; all field offsets are compile-time constants per codec
// version, so the parse is a handful of funnel shifts instead of a chain of byte loads.
struct Head128
{
  u32 w[32];
  __device__ __forceinline__ u32 u32At(u32 at) const    // at: constant after inlining
  {
    const u32 i = at >> 2, sh = 8u * (at & 3u);
    return sh ? ((w[i] >> sh) | (w[i + 1] << (32u - sh))) : w[i];
  }
  __device__ __forceinline__ u32 byteAt(u32 at) const { return (w[at >> 2] >> (8u * (at & 3u))) & 0xFFu; }
  __device__ __forceinline__ double f64At(u32 at) const
  {
    const u64 v = (u64)u32At(at) | ((u64)u32At(at + 4) << 32);
    double d; memcpy(&d, &v, 8);
    return d;
  }
};

template<int DT, int VER>
__device__ __forceinline__ void parseHead(const Head128& h, u32 sizeGiven, int nRows, int nCols, FastDecodeParams& hp)
{
  constexpr u32 TB = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  // byte offsets of the fields for this codec version (Lerc2.cpp:790-917)
  constexpr u32 oRows = 14, oCols = 18, oDepth = 22;
  constexpr u32 oValid = (VER >= 4) ? 26 : 22, oMb = oValid + 4, oSize = oValid + 8, oDt = oValid + 12;
  constexpr u32 oFlags = oValid + 20;                              // v6: nBlobsMore at oValid + 16, then 4 flag bytes
  constexpr u32 oDbl = (VER >= 6) ? oValid + 24 : oValid + 16;     // maxZError, zMin, zMax
  constexpr u32 oMask = oDbl + 24 + ((VER >= 6) ? 16 : 0);         // numBytesMask
  constexpr u32 oRanges = oMask + 4;
  constexpr u32 oSweep = oRanges + ((VER >= 4) ? 2 * TB : 0);
  constexpr u32 oData = oSweep + 1;
  static_assert(oData + 4 <= 124, "the parsed part of the header fits the 128 bytes read");
  const u32 nDepth = (VER >= 4) ? h.u32At(oDepth) : 1u;
  const u32 blobSize = h.u32At(oSize);
  const double maxZErr = h.f64At(oDbl), zMin = h.f64At(oDbl + 8), zMax = h.f64At(oDbl + 16);
  bool ok = h.u32At(oRows) == (u32)nRows && h.u32At(oCols) == (u32)nCols && nDepth == 1u
    && h.u32At(oValid) == (u32)nRows * (u32)nCols && h.u32At(oMb) == 8u && h.u32At(oDt) == (u32)DT
    && ((VER < 6) || h.byteAt(oFlags) == 0u) && h.u32At(oMask) == 0u && blobSize <= sizeGiven && zMin != zMax
    && maxZErr > 0 && maxZErr == maxZErr;
  if (VER >= 4)                                                    // ranges: min then max, raw T (nDepth == 1)
  {
    bool differ = false;
#pragma unroll
    for (u32 i = 0; i < TB; i += 4) differ = differ || (h.u32At(oRanges + i) != h.u32At(oRanges + TB + i));
    if (TB == 2) differ = (h.u32At(oRanges) & 0xFFFFu) != (h.u32At(oRanges + 2) & 0xFFFFu);
    ok = ok && differ;
  }
  ok = ok && h.byteAt(oSweep) == 0u && oData < blobSize;          // not the one-sweep raw form
  hp.dataBegin = oData;
  hp.blobEnd = blobSize;
  hp.nChunks = ok ? (blobSize + kFastChunkBytes - 1) / kFastChunkBytes : 0u;    // chunk c = blob bytes [c * kFastChunkBytes, (c + 1) * kFastChunkBytes)
  hp.invScale = 2 * maxZErr;
  hp.zMaxHdr = zMax;
  hp.ok = ok ? 1u : 0u;
}

template<int DT>
__device__ __forceinline__ FastDecodeParams parseBandHeader(const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols)
{
  FastDecodeParams hp;
  memset(&hp, 0, sizeof(hp));
  Head128 h;
  const uint4* src = reinterpret_cast<const uint4*>(blob);    // the band is 16-byte aligned and at least 70 bytes long
#pragma unroll
  for (int i = 0; i < 8; i++)
  {
    uint4 x = make_uint4(0, 0, 0, 0);
    if ((u32)(16 * i + 16) <= sizeGiven) x = src[i];
    else
    {
      u32 t4[4] = { 0, 0, 0, 0 };
#pragma unroll
      for (u32 k = 0; k < 16u; k++) if (16u * i + k < sizeGiven) t4[k >> 2] |= (u32)blob[16 * i + k] << (8 * (k & 3));
      x = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
    h.w[4 * i] = x.x; h.w[4 * i + 1] = x.y; h.w[4 * i + 2] = x.z; h.w[4 * i + 3] = x.w;
  }
  const u32 version = h.u32At(6);
  const bool magic = sizeGiven >= 70u && h.u32At(0) == 0x6372654Cu && (h.u32At(4) & 0xFFFFu) == 0x2032u;    // "Lerc2 "
  hp.version = version;
  hp.expectChecksum = h.u32At(10);
  hp.nBlocks = (u32)((nRows + 7) / 8) * (u32)((nCols + 7) / 8);    // (the blocks of the last block row / column may be smaller)
  hp.nTH = ((u32)nCols + 7u) / 8u;
  hp.nCols = (u32)nCols;
  hp.nRows = (u32)nRows;
  if (magic)
  {
    if (version == 6u) parseHead<DT, 6>(h, sizeGiven, nRows, nCols, hp);
    else if (version == 5u || version == 4u) parseHead<DT, 4>(h, sizeGiven, nRows, nCols, hp);
    else if (version == 3u) parseHead<DT, 3>(h, sizeGiven, nRows, nCols, hp);
  }
  return hp;
}

__device__ __forceinline__ bool fastRaised(const u32* __restrict__ fallback, u32 epoch)
{
  return fallback[0] == epoch || fallback[1] == epoch || fallback[2] == epoch || fallback[3] == epoch;
}

// (THROUGH: the one-launch decoder, where the checksum verdict is written into the same 64 bytes later in the launch by a
// workgroup on another XCD -- both as write-through stores, so that neither L2 holds a dirty copy of the other's half)
template<bool THROUGH> __device__ __forceinline__ void storeParams(FastDecodeParams* dst, const FastDecodeParams& hp)
{
  static_assert(sizeof(FastDecodeParams) == 64, "eight 8-byte stores");
  if (!THROUGH) { *dst = hp; return; }
  u64 wds[8];
  memcpy(wds, &hp, 64);
#pragma unroll
  for (int i = 0; i < 8; i++) publish64(reinterpret_cast<u64*>(dst) + i, wds[i]);
}

// ------------------------------------------------------------------------------------------------
// discovery
// ------------------------------------------------------------------------------------------------
// What a discovery workgroup needs of the band header; every workgroup reads it for itself (three 16-byte loads of
// the same address in all lanes).  The full check is done once, by parseBandHeader in workgroup 0: if that one says
// "not ours" nobody looks at what the others did.
struct HeadLite { u32 ok, version, dataBegin, blobEnd; };
// the band's first 64 bytes (a band is 70 bytes at least), asked for in one go: what parseHeadLite / headLiteEligible look at
struct Head64 { uint4 a, c, d, e; };
__device__ __forceinline__ Head64 loadHead64(const u8* __restrict__ blob, u32 sizeGiven)
{
  Head64 h;
  h.a = h.c = h.d = h.e = make_uint4(0, 0, 0, 0);
  if (sizeGiven < 70u) return h;
  const uint4* src = reinterpret_cast<const uint4*>(blob);
  h.a = src[0]; h.c = src[1]; h.d = src[2]; h.e = src[3];
  return h;
}
template<int DT>
__device__ __forceinline__ HeadLite parseHeadLite(const Head64& hd, u32 sizeGiven)
{
  constexpr u32 TB = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  HeadLite h = { 0u, 0u, 0u, 0u };
  if (sizeGiven < 70u) return h;
  const uint4 a = hd.a, c = hd.c, d = hd.d;
  const u32 version = (a.y >> 16) | (a.z << 16);                    // bytes 6 .. 9
  const u32 size3 = (c.w >> 16) | (d.x << 16), size4 = (d.x >> 16) | (d.y << 16);
  h.version = version;
  const u32 hdr = (version >= 6u) ? 90u : (version >= 4u) ? 66u : 62u;
  h.dataBegin = hdr + 4u + ((version >= 4u) ? 2u * TB : 0u) + 1u;   // mask byte count, ranges, one-sweep flag
  h.blobEnd = min((version >= 4u) ? size4 : size3, sizeGiven);
  h.ok = (version >= 3u && version <= 6u && h.blobEnd > h.dataBegin) ? 1u : 0u;
  return h;
}
template<int DT>
__device__ __forceinline__ bool headLiteEligible(const Head64& hd, u32 version, int nRows, int nCols)
{
  const uint4 c = hd.c, d = hd.d, e = hd.e;    // bytes 16 .. 63
  const u32 numValid = (version >= 4u) ? ((c.z >> 16) | (c.w << 16)) : ((c.y >> 16) | (c.z << 16));    // bytes 26 .. 29 / 22 .. 25
  const u64 z6 = ((u64)(e.x >> 16)) | ((u64)e.y << 16) | ((u64)(e.z & 0xFFFFu) << 48);
  const u64 z4 = ((u64)(d.z >> 16)) | ((u64)d.w << 16) | ((u64)(e.x & 0xFFFFu) << 48);
  const u64 z3 = ((u64)(d.y >> 16)) | ((u64)d.z << 16) | ((u64)(d.w & 0xFFFFu) << 48);
  const u64 z = (version >= 6u) ? z6 : (version >= 4u) ? z4 : z3;
  return numValid == (u32)nRows * (u32)nCols && z != 0ull && (z >> 63) == 0ull;
}
template<int DT>
__device__ __forceinline__ HeadLite parseHeadLite(const u8* __restrict__ blob, u32 sizeGiven)
{
  constexpr u32 TB = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  HeadLite h = { 0u, 0u, 0u, 0u };
  if (sizeGiven < 70u) return h;
  const uint4* src = reinterpret_cast<const uint4*>(blob);
  const uint4 a = src[0], c = src[1], d = src[2];
  const u32 version = (a.y >> 16) | (a.z << 16);                    // bytes 6 .. 9
  // blob size: byte 30 (codec 3) or 34 (codec >= 4), Lerc2.cpp:790-917
  const u32 size3 = (c.w >> 16) | (d.x << 16), size4 = (d.x >> 16) | (d.y << 16);
  h.version = version;
  const u32 hdr = (version >= 6u) ? 90u : (version >= 4u) ? 66u : 62u;
  h.dataBegin = hdr + 4u + ((version >= 4u) ? 2u * TB : 0u) + 1u;   // mask byte count, ranges, one-sweep flag
  h.blobEnd = min((version >= 4u) ? size4 : size3, sizeGiven);
  h.ok = (version >= 3u && version <= 6u && h.blobEnd > h.dataBegin) ? 1u : 0u;
  return h;
}

// Two more things of the header that every workgroup can see for itself in the bytes it has read anyway (+ the 16 behind them):
// a band with a mask (fewer valid pixels than pixels) and a lossless one (maxZError 0) are none for the streaming kernels.  The
// full parse says so too, but only after a workgroup has asked for its 32 KiB of blob; with this the 4 800 workgroups of a
// 150 MB lossless-float blob leave before they do.
template<int DT>
__device__ __forceinline__ bool headLiteEligible(const u8* __restrict__ blob, u32 version, int nRows, int nCols)
{
  const uint4* src = reinterpret_cast<const uint4*>(blob);
  const uint4 c = src[1], d = src[2], e = src[3];    // bytes 16 .. 63 (the band is 70 bytes at least)
  const u32 numValid = (version >= 4u) ? ((c.z >> 16) | (c.w << 16)) : ((c.y >> 16) | (c.z << 16));    // bytes 26 .. 29 / 22 .. 25
  // maxZError: bytes 50 .. 57 (codec 6), 42 .. 49 (4, 5), 38 .. 45 (3)
  const u64 z6 = ((u64)(e.x >> 16)) | ((u64)e.y << 16) | ((u64)(e.z & 0xFFFFu) << 48);
  const u64 z4 = ((u64)(d.z >> 16)) | ((u64)d.w << 16) | ((u64)(e.x & 0xFFFFu) << 48);
  const u64 z3 = ((u64)(d.y >> 16)) | ((u64)d.z << 16) | ((u64)(d.w & 0xFFFFu) << 48);
  const u64 z = (version >= 6u) ? z6 : (version >= 4u) ? z4 : z3;
  return numValid == (u32)nRows * (u32)nCols && z != 0ull && (z >> 63) == 0ull;
}

// 0x80 in every byte of v that is zero (exact per byte, unlike the borrow trick)
__device__ __forceinline__ u32 zeroBytes(u32 v) { return ~(((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v | 0x7F7F7F7Fu); }

template<class T> struct DCfg
{
  static constexpr int V = (sizeof(T) >= 4) ? 16 / (int)sizeof(T) : 8;
  static constexpr int LPR = 8 / V;
  static constexpr int BPW = 8 / LPR;
  static constexpr int TILE_COLS = 8 * V;
  static constexpr int IT = kFastBlocksPerWG / (4 * BPW);
};

// nbits (<= 32) at bit position bitPos of the LDS word stream
__device__ __forceinline__ u32 ldsBits(const u32* words, u32 bitPos, int nbits)
{
  const u32 w = bitPos >> 5, sh = bitPos & 31;
  const u64 x = ((u64)words[w + 1] << 32) | words[w];
  return (u32)(x >> sh) & (nbits >= 32 ? 0xFFFFFFFFu : ((1u << nbits) - 1u));
}

template<class T> __device__ __forceinline__ T dequant(double offset, u32 q, double invScale, double zMax, i64 offI, i64 invI, i64 zMaxI)
{
  if (DtOf<T>::v >= DT_Float)
  {
    const double z = offset + (double)q * invScale;    // Lerc2.cpp:2159-2160, no contraction
    return (T)(z < zMax ? z : zMax);
  }
  // integer types: offset, 2 * maxZError and zMax are integers, the double expression is exact
  const i64 z = offI + (i64)q * invI;
  return (T)(z < zMaxI ? z : zMaxI);
}

}    // namespace lerc
