// tile_fast_decode.hip -- streaming decoder kernels for the common case (one band, nDepth == 1, every
// pixel valid, 8 x 8 blocks, rows and columns multiples of 8).  Same results as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends), so decoding starts with a
// discovery pass over 4 KiB chunks of the blob (tile_fast.h).  Everything serial about it runs out of LDS:
//   k_fast_discover    one wave = kDiscChunks chunks staged with 16-byte loads (their Fletcher32 terms are summed
//                      on the way, so the decode kernel does not look at the checksum at all).  Every wave parses
//                      the band header itself (the host has not seen a byte of the blob when it enqueues the four
//                      kernels).  Each chunk belongs to kDiscLanes lanes: they try every position of the chunk's
//                      first `window` bytes as a block start, keep the ones that stay valid for kFilterSteps blocks
//                      (compacting LDS queues: most die at once) and walk the survivors to the chunk's end in
//                      lockstep, writing down the block starts they pass.  The true first block of a chunk is
//                      always among the survivors.
//   k_fast_resolve     whatever ALL live walks of a chunk agree on is true without knowing which one is real:
//                      entry of chunk c = agreed exit of chunk c-1; the walk that starts exactly there is the true
//                      path and its length the chunk's block count.  Also folds the checksum terms.
//   k_fast_gather      block offsets = the true walks' lists, placed by a scan of the counts.
//   k_fast_decode      a workgroup owns 64 consecutive blocks (8 rows x 512 columns where a block row is that
//                      long): it stages their byte span in LDS, parses the 64 block headers once (lane = block;
//                      signature and contiguity checks = ReadTile's integrity checks), then every lane extracts V
//                      consecutive pixels of one raster row, dequantises (double precision in the reference's
//                      expression order for float types, exact integer arithmetic for integer types) and stores
//                      one 16-byte vector.
// Whenever a precondition fails (a block longer than its raw size, disagreeing walks, too many survivors,
// ...) the kernels raise an epoch tagged flag, and the host repeats the band with the general kernels.
// Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540; Lerc2.cpp:1037-1064 (checksum).
#include "tile_fast.h"
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

static const u32 kNoOffset = 0xFFFFFFFFu;
PROBE_DEFINE(fast_decode)

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
__device__ __forceinline__ u32 parseCode(u32 h0, u32 h1, u32 h2, int version)
{
  constexpr u32 TB = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 RAW = 1 + 64 * TB;
  const u32 flag = h0 & 0xFFu, mode = flag & 3u;
  const u32 offB = (offBytesTable<DT>() >> ((flag >> 4) & 12u)) & 15u;
  const u64 hdr = ((u64)h1 << 32) | h0;
  u32 t = (u32)(hdr >> ((8u + 8u * offB) & 63u));    // bytes 1 + offB ...: numBits byte, count, LUT size
  if (DT == DT_Double) t = (offB == 8u) ? (h2 >> 8) : t;
  const u32 nb = t & 31u, lut = (t >> 5) & 1u;
  const u32 nLut = ((t >> 16) & 0xFFu) - 1u;                           // valid: 1 ... 254
  const bool okBits = ((t & 0xFFC0u) == 0x4080u) & (nb != 0u);         // 64 elements: one-byte count field == 64
  const bool okLut = (nLut - 1u) < 254u;
  const u32 lenSimple = 3u + offB + 8u * nb;
  const u32 lenLut = 4u + offB + (((nLut & 0xFFu) * nb + 7u) >> 3) + 8u * (u32)bitLen(nLut & 0xFFu);
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

// The same step written for the walks' critical path: everything is computed for every lane and selected at the end,
// so a step is one LDS round trip and a dozen dependent VALU operations, with no branch.  `rel` must lie inside the
// staged bytes (the caller clamps it for lanes that are not walking); returns the block's length or 0.
template<int DT>
__device__ __forceinline__ u32 stepLean(const u32* words, u32 rel, u32 remaining, bool v5, u32 prevSig, u32 pattern, u32& sigOut)
{
  constexpr u32 TB = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 RAW = 1 + 64 * TB;
  const u32 wi = rel >> 2, sh = 8u * rel;    // (alignbit takes the shift mod 32)
  const u32 x0 = words[wi], x1 = words[wi + 1], x2 = words[wi + 2];
  const u32 h0 = __builtin_amdgcn_alignbit(x1, x0, sh), h1 = __builtin_amdgcn_alignbit(x2, x1, sh);
  const u32 mode = h0 & 3u;
  const u32 offB = (offBytesTable<DT>() >> ((h0 >> 4) & 12u)) & 15u;
  u32 t;    // bytes 1 + offB ...: numBits byte, count, LUT size
  if (DT == DT_Double)
  {
    const u32 x3 = words[wi + 3];
    const u32 h2 = __builtin_amdgcn_alignbit(x3, x2, sh);
    t = (offB == 8u) ? (h2 >> 8) : (u32)((((u64)h1 << 32) | h0) >> ((8u + 8u * offB) & 63u));
  }
  else t = (u32)((((u64)h1 << 32) | h0) >> ((8u + 8u * offB) & 63u));
  const u32 nb = t & 31u, lut = (t >> 5) & 1u;
  const u32 nLut = (((t >> 16) & 0xFFu) - 1u) & 0xFFu;                     // valid: 1 ... 254
  const u32 okBits = (u32)((t & 0xFFC0u) == 0x4080u) & (u32)(nb != 0u);    // 64 elements: one-byte count field == 64
  const u32 okLut = (u32)((nLut - 1u) < 254u);
  const u32 lenSimple = 3u + offB + 8u * nb;
  const u32 lenLut = 4u + offB + ((nLut * nb + 7u) >> 3) + 8u * (u32)bitLen(nLut);
  const u32 lenStuffed = lut ? lenLut : lenSimple;
  const u32 lenOther = (mode == 0u) ? RAW : (mode == 2u) ? 1u : 1u + offB;
  const u32 len = (mode == 1u) ? lenStuffed : lenOther;
  const u32 okStuffed = okBits & (u32)(offB != 0u) & ((lut ^ 1u) | okLut);
  const u32 okOther = (u32)(mode != 3u) | (u32)(offB != 0u);
  u32 ok = (mode == 1u) ? okStuffed : okOther;
  ok &= (u32)!(v5 && (h0 & 4u)) & (u32)(len <= RAW) & (u32)(len <= remaining);    // slice difference needs nDepth > 1
  const u32 sg = (h0 >> 2) & pattern;
  ok &= (u32)(prevSig == kNoOffset) | (u32)sigOk(prevSig, sg, pattern);
  sigOut = sg;
  return ok ? len : 0u;
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
  hp.nChunks = ok ? (blobSize + kFastChunkBytes - 1) / kFastChunkBytes : 0u;    // chunk c = blob bytes [c * 4096, (c + 1) * 4096)
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
  hp.nBlocks = (u32)(nRows / 8) * (u32)(nCols / 8);
  hp.nTH = (u32)nCols / 8u;
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

// ------------------------------------------------------------------------------------------------
// discovery
// ------------------------------------------------------------------------------------------------
// Fletcher32 terms (Lerc2.cpp:1037-1064) of one 16-byte unit whose first byte is byte 2 * k0 of the checksummed range
// blob[14 ..): the checksum works on big-endian 16-bit words w, A = sum w, B = sum index * w.  Bytes at even positions
// weigh 256: four byte dot products per dword, the word index inside the unit (0 .. 7) rides in the weights.
__device__ __forceinline__ void fletcherUnit(const uint4& x, u64 k0, u32& A, u64& B)
{
  u32 ae = 0, ao = 0, be = 0, bo = 0;
  ae = __builtin_amdgcn_udot4(x.x, 0x00010001u, ae, false); ao = __builtin_amdgcn_udot4(x.x, 0x01000100u, ao, false);
  be = __builtin_amdgcn_udot4(x.x, 0x00010000u, be, false); bo = __builtin_amdgcn_udot4(x.x, 0x01000000u, bo, false);
  ae = __builtin_amdgcn_udot4(x.y, 0x00010001u, ae, false); ao = __builtin_amdgcn_udot4(x.y, 0x01000100u, ao, false);
  be = __builtin_amdgcn_udot4(x.y, 0x00030002u, be, false); bo = __builtin_amdgcn_udot4(x.y, 0x03000200u, bo, false);
  ae = __builtin_amdgcn_udot4(x.z, 0x00010001u, ae, false); ao = __builtin_amdgcn_udot4(x.z, 0x01000100u, ao, false);
  be = __builtin_amdgcn_udot4(x.z, 0x00050004u, be, false); bo = __builtin_amdgcn_udot4(x.z, 0x05000400u, bo, false);
  ae = __builtin_amdgcn_udot4(x.w, 0x00010001u, ae, false); ao = __builtin_amdgcn_udot4(x.w, 0x01000100u, ao, false);
  be = __builtin_amdgcn_udot4(x.w, 0x00070006u, be, false); bo = __builtin_amdgcn_udot4(x.w, 0x07000600u, bo, false);
  const u32 a = 256u * ae + ao;    // < 2^19
  A += a;
  B += k0 * a + (256u * be + bo);
}

// queue entry of the candidate filter: start in the window (10) | current position relative to the chunk (13) << 10 |
// signature of the last block (4) << 23 | valid blocks so far (3) << 27
__device__ __forceinline__ u32 qMake(u32 start, u32 rel, u32 sig, u32 steps) { return start | (rel << 10) | ((sig & 15u) << 23) | (steps << 27); }
__device__ __forceinline__ u32 qStart(u32 e) { return e & 0x3FFu; }
__device__ __forceinline__ u32 qRel(u32 e) { return (e >> 10) & 0x1FFFu; }
__device__ __forceinline__ u32 qSig(u32 e) { return (e >> 23) & 15u; }
__device__ __forceinline__ u32 qSteps(u32 e) { return e >> 27; }

// What a discovery workgroup needs of the band header; every workgroup reads it for itself (three 16-byte loads of
// the same address in all lanes).  The full check is done once, by parseBandHeader in workgroup 0: if that one says
// "not ours" nobody looks at what the others did.
struct HeadLite { u32 ok, version, dataBegin, blobEnd; };
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

template<int DT>
__device__ __forceinline__ void
fastDiscoverBody(const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols, const FastDecodeBuffers& b)
{
  constexpr int TBYTES = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 W = kFastWindow(TBYTES);
  constexpr u32 CH = kFastChunkBytes, NCH = (u32)kDiscChunks, LPC = (u32)kDiscLanes, NW = (u32)kDiscWalks;
  constexpr u32 kUnits = NCH * CH / 16;                    // 16-byte units a workgroup owns
  constexpr u32 kStageUnits = kUnits + 1;                  // + the first bytes of the block that may start right before the end
  constexpr u32 QCAP = (TBYTES == 8 || W < 200u) ? W : 200u;    // positions of a chunk's window whose flag byte passes (7 of 16 on noise; all four offset types of float64 are valid: half)
  constexpr u32 kBitWords = (W + 31) / 32;
  static_assert(NCH * LPC == 256 && NW == 8 && W < 1024 && CH + 64 * TBYTES + 1 < 8192, "lane layout / queue entry fields");
  __shared__ __align__(16) u32 s_in[kStageUnits * 4];
  __shared__ u32 s_q[NCH][QCAP];
  __shared__ u32 s_bits[NCH][kBitWords];
  __shared__ u16 s_final[NCH][NW];
  __shared__ u32 s_nFinal[NCH];
  __shared__ u32 s_exit[NCH][NW];
  __shared__ u64 s_fa[4], s_fb[4];
  __shared__ u32 s_over;

  PROBE_BEGIN;
  const int lane = laneId(), w = waveId();
  const u32 c0 = blockIdx.x * NCH;                         // first chunk of this workgroup
  const u32 r0 = c0 * CH;                                  // blob offset of LDS byte 0

  // ---- all loads in flight first: the workgroup's chunks (clipped to what the caller says is readable) ...
  constexpr int kRounds = (int)((kStageUnits + 255) / 256);
  uint4 x[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * 256u + threadIdx.x;
    const u64 a = (u64)r0 + 16ull * i;
    x[k] = make_uint4(0, 0, 0, 0);
    if (i < kStageUnits)
    {
      if (a + 16 <= sizeGiven) x[k] = *reinterpret_cast<const uint4*>(blob + a);
      else if (a < sizeGiven)    // never read past the blob
      {
        u32 t4[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (u32 q = 0; q < 16; q++) if (a + q < sizeGiven) t4[q >> 2] |= (u32)blob[a + q] << (8 * (q & 3));
        x[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
      }
    }
  }
  // ... and the band header
  HeadLite hl;
  if (blockIdx.x == 0)
  {
    const FastDecodeParams hp = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    if (threadIdx.x == 0) *b.params = hp;
    hl.ok = hp.ok; hl.version = hp.version; hl.dataBegin = hp.dataBegin; hl.blobEnd = hp.blobEnd;
  }
  else hl = parseHeadLite<DT>(blob, sizeGiven);
  const u32 nChunks = (hl.blobEnd + CH - 1) / CH;
  if (!hl.ok || c0 >= nChunks) return;    // (the grid is sized for the largest stream the blob could hold)
  const int version = (int)hl.version;
  const u32 dataBegin = hl.dataBegin, blobEnd = hl.blobEnd;

  // ---- stage + Fletcher terms of the units this workgroup owns (bytes 14 ... blobEnd - 1 of the blob are checksummed)
  u32 fA = 0;
  u64 fB = 0;
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * 256u + threadIdx.x;
    if (i < kStageUnits) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
    const u64 a = (u64)r0 + 16ull * i;
    if (i < kUnits && a < blobEnd)
    {
      uint4 y = x[k];
      if (a == 0 || a + 16 > blobEnd)    // blank what is not checksummed: the first 14 bytes, whatever lies behind the blob
      {
        u32 wd[4] = { y.x, y.y, y.z, y.w };
#pragma unroll
        for (u32 q = 0; q < 16; q++)
          if (a + q < 14u || a + q >= blobEnd) wd[q >> 2] &= ~(0xFFu << (8 * (q & 3)));
        y = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
      // unit at blob offset a holds words (a - 14) / 2 ...; the first unit's index -7 as its residue mod 65535
      fletcherUnit(y, a ? (a - 14u) / 2u : 65528ull, fA, fB);
    }
  }
  {
    const u64 A = waveSum((u64)fA % 65535u), B = waveSum(fB % 65535u);
    if (lane == 0) { s_fa[w] = A; s_fb[w] = B; }
  }
  if (threadIdx.x == 0) s_over = 0u;
  __syncthreads();
  PROBE(16);
  if (threadIdx.x == 0)
  {
    b.waveFletcher[2 * (size_t)blockIdx.x] = (s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3]) % 65535u;
    b.waveFletcher[2 * (size_t)blockIdx.x + 1] = (s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]) % 65535u;
  }

  // ---- candidates.  Lanes g * LPC ... of wave w own chunk c0 + 4 w + g.
  const u32 g = (u32)lane / LPC, sl = (u32)lane % LPC;
  const u32 cl = (u32)w * 4u + g;                                         // chunk inside the workgroup
  const u32 chunk = c0 + cl;
  const u32 chunkStart = chunk * CH;                                      // (no overflow: chunk < nChunks)
  const bool chunkLive = chunk < nChunks;
  const u32 chunkEnd = chunkLive ? min(chunkStart + CH, blobEnd) : chunkStart;
  const u32 pattern = (version >= 5) ? 14u : 15u;
  const u32 groupShift = g * LPC;
  const u32 belowMe = (1u << sl) - 1u;
  bool overflow = false;
  u32* __restrict__ q = s_q[cl];

  // step 0: window positions whose first byte can be a block's flag byte (Lerc2.cpp:1961-1973: bits 0-1 how the block
  // is coded, bit 2 -- codec >= 5 -- "difference to the previous slice", never set with nDepth == 1, bits 6-7 the type of
  // the offset, which only bit-stuffed and constant blocks have); the chunk that holds the first block: that block only.
  // Before codec 5 bit 2 belongs to the signature and 7 of 8 bytes pass, more than the queue holds: those take their
  // first two blocks at once.
  u32 nq = 0;                                                             // entries in this chunk's queue (the same in all its lanes)
  for (u32 o0 = 0; o0 < W; o0 += LPC)
  {
    const u32 o = o0 + sl;
    const u32 cur = chunkStart + o;
    bool live = chunkLive && o < W && cur < chunkEnd && (chunkStart <= dataBegin ? cur == dataBegin : true);
    u32 entry = qMake(o, o, kNoOffset, 0u);
    if (live && version >= 5)
    {
      const u32 rel = cur - r0;
      const u32 flag = (s_in[rel >> 2] >> (8u * (rel & 3u))) & 0xFFu;
      const bool hasOffset = (flag & 1u) != 0u;                           // bit-stuffed or constant
      live = !(flag & 4u) && !(hasOffset && ((offBytesTable<DT>() >> ((flag >> 4) & 12u)) & 15u) == 0u);
    }
    else if (live)
    {
      u32 sig = kNoOffset, rel = o, steps = 0;
      for (int k = 0; k < 2 && live && chunkStart + rel < chunkEnd; k++)
      {
        const u32 code = stepAt<DT>(s_in, r0, chunkStart + rel, blobEnd, version, sig, pattern);
        live = code != 0u;
        rel += codeLen(code); steps++;
      }
      entry = qMake(o, rel, sig, steps);
    }
    const u32 gm = (u32)(__ballot(live) >> groupShift) & ((1u << LPC) - 1u);
    const u32 at = nq + (u32)__popc(gm & belowMe);
    if (live) { if (at < QCAP) q[at] = entry; else overflow = true; }
    nq = min(nq + (u32)__popc(gm), QCAP);
  }
  waveSync();
  PROBE(17);

  // further blocks through the queues, compacted in place (a round reads LPC entries before it writes at most LPC at
  // or before them); who is still valid after kFilterSteps blocks (or has reached the chunk's end) is a survivor
  const bool v5 = version >= 5;
  constexpr u32 kMaxRel = NCH * CH - 1;                                   // last staged byte a block may start at
  for (int pass = 0; pass < kFilterSteps; pass++)
  {
    u32 nOut = 0;
    bool pending = false;
    for (u32 i0 = 0; __any(i0 < nq); i0 += LPC)
    {
      const u32 i = i0 + sl;
      const bool have = i < nq;
      const u32 e = have ? q[i] : 0u;
      const u32 rel = qRel(e), steps = qSteps(e);
      const u32 cur = chunkStart + rel;
      const bool doStep = have && steps < (u32)kFilterSteps && cur < chunkEnd;
      u32 sg;
      const u32 len = stepLean<DT>(s_in, min(cur - r0, kMaxRel), blobEnd - min(cur, blobEnd), v5, steps ? qSig(e) : kNoOffset, pattern, sg);
      const bool live = have && (!doStep || len != 0u);
      const u32 rel2 = doStep ? rel + len : rel, steps2 = doStep ? steps + 1u : steps, sig2 = doStep ? sg : qSig(e);
      pending = pending || (live && steps2 < (u32)kFilterSteps && chunkStart + rel2 < chunkEnd);
      waveSync();    // every lane has read its entry
      const u32 gm = (u32)(__ballot(live) >> groupShift) & ((1u << LPC) - 1u);
      if (live) q[nOut + (u32)__popc(gm & belowMe)] = qMake(qStart(e), rel2, sig2, steps2);
      nOut += (u32)__popc(gm);
      waveSync();
    }
    nq = nOut;
    if (!__any(pending)) break;
  }
  PROBE(18);
  // Survivors that sit on one path (the true path crosses the window in several blocks, each of them a survivor; a run
  // of constant blocks makes every byte one) need one walk, from the first of them: a survivor that is the block
  // right behind another survivor is dropped (its predecessor is valid for kFilterSteps blocks too, hence a survivor).
  for (u32 i = sl; i < kBitWords; i += LPC) s_bits[cl][i] = 0u;
  waveSync();
  for (u32 i0 = 0; __any(i0 < nq); i0 += LPC)
    if (i0 + sl < nq) { const u32 st = qStart(q[i0 + sl]); atomicOr(&s_bits[cl][st >> 5], 1u << (st & 31u)); }
  waveSync();
  for (u32 i0 = 0; __any(i0 < nq); i0 += LPC)
  {
    if (i0 + sl < nq)
    {
      const u32 st = qStart(q[i0 + sl]);
      u32 sg;
      const u32 nx = st + stepLean<DT>(s_in, chunkStart + st - r0, blobEnd - (chunkStart + st), v5, kNoOffset, pattern, sg);    // valid: it was a moment ago
      if (nx < W) atomicAnd(&s_bits[cl][nx >> 5], ~(1u << (nx & 31u)));
    }
  }
  waveSync();
  u32 nFinal = 0;
  for (u32 i0 = 0; __any(i0 < nq); i0 += LPC)
  {
    const u32 st = (i0 + sl < nq) ? qStart(q[i0 + sl]) : 0u;
    const bool head = i0 + sl < nq && ((s_bits[cl][st >> 5] >> (st & 31u)) & 1u) != 0u;
    const u32 gd = (u32)(__ballot(head) >> groupShift) & ((1u << LPC) - 1u);
    if (head)
    {
      const u32 at = nFinal + (u32)__popc(gd & belowMe);
      if (at < NW) s_final[cl][at] = (u16)st; else overflow = true;
    }
    nFinal = min(nFinal + (u32)__popc(gd), NW);
  }
  if (sl == 0) s_nFinal[cl] = chunkLive ? nFinal : 0u;
  if (__any(overflow) && lane == 0) s_over = 1u;
  PROBE(19);
  __syncthreads();
  PROBE(20);

  // ---- walks: the path heads of all 16 chunks, lane = (chunk, head); waves 0 and 1 take heads 0-3 and 4-7 (there are
  // seldom more than two), the other waves are done
  if (w < 2)
  {
    const u32 wc = (u32)lane >> 2, slot = ((u32)lane & 3u) + 4u * (u32)w;    // chunk inside the workgroup, head
    const u32 wChunk = c0 + wc;
    const u32 wStart = wChunk * CH;
    const bool wLive = wChunk < nChunks;
    const u32 wEnd = wLive ? min(wStart + CH, blobEnd) : wStart;
    const bool walker = wLive && slot < s_nFinal[wc];
    u32 cur = wStart + (walker ? (u32)s_final[wc][slot] : 0u);
    u32 sig = kNoOffset, count = 0;
    bool alive = walker, tooMany = false;
    u16 first[kRecPrefix];
#pragma unroll
    for (int k = 0; k < kRecPrefix; k++) first[k] = (u16)0xFFFFu;
    u16* __restrict__ list = b.lists + ((size_t)wChunk * NW + slot) * kFastListCap;
    bool active = alive && cur < wEnd;
    while (__any(active))
    {
      u32 sg;
      const u32 len = stepLean<DT>(s_in, min(cur - r0, kMaxRel), blobEnd - min(cur, blobEnd), v5, sig, pattern, sg);
      const bool room = count < (u32)kFastListCap;
      const bool ok = active && len != 0u && room;
      tooMany = tooMany || (active && len != 0u && !room);
      const u16 at = (u16)(cur - wStart);
      if (ok) list[count] = at;
#pragma unroll
      for (int k = 0; k < kRecPrefix; k++) first[k] = (ok && count == (u32)k) ? at : first[k];
      alive = alive && (!active || ok);
      cur += ok ? len : 0u;
      count += ok ? 1u : 0u;
      sig = ok ? sg : sig;
      active = alive && cur < wEnd;
    }
    s_exit[wc][slot] = alive ? cur : kNoOffset;
    if (wLive)
    {
      FastChunkRec* rec = b.recs + wChunk;
#pragma unroll
      for (int k = 0; k < kRecPrefix; k++) rec->first[slot][k] = first[k];
      rec->count[slot] = alive ? (u16)count : (u16)0xFFFFu;
    }
    if (__any(tooMany) && lane == 0) s_over = 1u;
  }
  PROBE(21);
  __syncthreads();
  // what all live walks of a chunk agree on
  if (threadIdx.x < NCH && c0 + threadIdx.x < nChunks)
  {
    u32 lo = kNoOffset, hi = 0u, n = 0u;
#pragma unroll
    for (u32 k = 0; k < NW; k++)
    {
      const u32 e = s_exit[threadIdx.x][k];
      if (e != kNoOffset) { lo = min(lo, e); hi = max(hi, e); n++; }
    }
    FastChunkRec* rec = b.recs + c0 + threadIdx.x;
    rec->exit = (n != 0u && lo == hi) ? lo : kNoOffset;
    rec->nLive = n;
  }
  if (threadIdx.x == 0 && s_over) b.fallback[0] = b.epoch;
}

// ------------------------------------------------------------------------------------------------
// resolve
// ------------------------------------------------------------------------------------------------
// One thread per chunk.  Only the exits need agreement (they break the chunk-to-chunk dependency): once the entry of
// a chunk is known, the walk that starts there IS the true path.  The counts are scanned inside the workgroup; the
// gather step adds the sums of the workgroups before its own.  Workgroup 0 also folds the checksum terms.
__device__ __forceinline__ void fastResolveBody(const FastDecodeBuffers& b, u32 nWavesBound)
{
  __shared__ u32 s_w[kResolveWG / 64];
  __shared__ u64 s_a[kResolveWG / 64], s_b[kResolveWG / 64];
  const FastDecodeParams hp = *b.params;
  if (!hp.ok) return;
  const u32 blobEnd = hp.blobEnd;
  const u32 c = blockIdx.x * kResolveWG + threadIdx.x;
  const int lane = laneId(), w = waveId();
  u32 count = 0, laneOfPath = kNoOffset;
  bool bad = false;
  if (c < hp.nChunks)
  {
    const u32 chunkStart = c * kFastChunkBytes, chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
    const u32 e = (chunkStart <= hp.dataBegin) ? hp.dataBegin : b.recs[c - 1].exit;
    if (e == kNoOffset || e < chunkStart) bad = true;
    else if (e >= chunkEnd) bad = (e != blobEnd);    // the last block may begin before the last chunk and end with it
    else
    {
      const FastChunkRec rec = b.recs[c];
      const u32 rel = e - chunkStart;
      // the walk the entry lies on: normally a walk starts there; else it is one of the first blocks of a walk that
      // began a little earlier (a stray byte in front of the entry that looks like a one byte block), or -- long runs
      // of tiny blocks -- somewhere in its list
#pragma unroll
      for (int l = 0; l < kDiscWalks; l++)
#pragma unroll
        for (int k = 0; k < kRecPrefix; k++)
          if ((u32)rec.first[l][k] == rel && rec.count[l] != 0xFFFFu && laneOfPath == kNoOffset) { laneOfPath = (u32)l | ((u32)k << 8); count = rec.count[l] - (u32)k; }
      if (laneOfPath == kNoOffset && rec.exit != kNoOffset)
      {
#pragma unroll
        for (int l = 0; l < kDiscWalks; l++)
        {
          const u32 n = rec.count[l];
          if (laneOfPath != kNoOffset || n == 0xFFFFu || n <= (u32)kRecPrefix || (u32)rec.first[l][0] > rel) continue;
          const u16* __restrict__ list = b.lists + ((size_t)c * kDiscWalks + l) * kFastListCap;
          u32 lo = kRecPrefix, hi = n;    // first index with list[i] >= rel
          while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u32)list[mid] < rel) lo = mid + 1; else hi = mid; }
          if (lo < n && (u32)list[lo] == rel) { laneOfPath = (u32)l | (lo << 8); count = n - lo; }
        }
      }
      if (laneOfPath == kNoOffset || rec.exit == kNoOffset) bad = true;    // (no agreement on the exit: the next chunk says so too)
    }
    if (bad) count = 0;
    b.chunkCount[c] = count;
    b.chunkLane[c] = laneOfPath;
  }
  if (__any(bad) && lane == 0) b.fallback[1] = b.epoch;
  // exclusive scan of the counts inside the workgroup
  u32 inc = count;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  u32 before = 0;
  for (int i = 0; i < w; i++) before += s_w[i];
  if (c < hp.nChunks) b.chunkLocal[c] = before + inc - count;
  if (threadIdx.x == kResolveWG - 1) b.groupSum[blockIdx.x] = before + inc;

  if (blockIdx.x != 0) return;
  // checksum: Fletcher32 over blob[14 ..) from the discovery waves' partial sums (Lerc2.cpp:1037-1064)
  const u32 nWaves = min((hp.nChunks + (u32)kDiscChunks - 1u) / (u32)kDiscChunks, nWavesBound);
  u64 A = 0, B = 0;
  for (u32 i = threadIdx.x; i < nWaves; i += kResolveWG) { A += b.waveFletcher[2 * (size_t)i]; B += b.waveFletcher[2 * (size_t)i + 1]; }    // each < 65535
  A = waveSum(A % 65535u); B = waveSum(B % 65535u);
  if (lane == 0) { s_a[w] = A; s_b[w] = B; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  A = 0; B = 0;
  for (u32 i = 0; i < kResolveWG / 64; i++) { A += s_a[i]; B += s_b[i]; }
  A %= 65535u; B %= 65535u;
  const u64 N = ((u64)(blobEnd - 14u) + 1) / 2;
  u64 s1 = A, s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
  if (s1 == 0) s1 = 0xffff;
  if (s2 == 0) s2 = 0xffff;
  b.params->checksumOk = ((u32)((s2 << 16) | s1) == hp.expectChecksum) ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------------
// block offsets
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void fastGatherBody(const FastDecodeBuffers& b)
{
  static_assert(kFastListCap == 256, "one thread per list entry");
  __shared__ u32 s_part[4], s_n[kGatherChunks], s_loc[kGatherChunks], s_lane[kGatherChunks];
  const FastDecodeParams hp = *b.params;
  if (!hp.ok) return;
  const u32 cFirst = blockIdx.x * kGatherChunks;
  if (cFirst >= hp.nChunks) return;          // the grid is sized for the largest stream the blob could hold
  const int lane = laneId(), w = waveId();
  if (threadIdx.x < kGatherChunks)
  {
    const u32 c = cFirst + threadIdx.x;
    const bool have = c < hp.nChunks;
    s_n[threadIdx.x] = have ? b.chunkCount[c] : 0u;
    s_loc[threadIdx.x] = have ? b.chunkLocal[c] : 0u;
    s_lane[threadIdx.x] = have ? b.chunkLane[c] : 0u;
  }
  // blocks before this workgroup's resolve group
  const u32 grp = cFirst / kResolveWG;
  u32 sum = 0;
  for (u32 i = threadIdx.x; i < grp; i += 256u) sum += b.groupSum[i];
  sum = waveSum(sum);
  if (lane == 0) s_part[w] = sum;
  __syncthreads();
  const u32 before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  if (cFirst + kGatherChunks >= hp.nChunks && threadIdx.x == 0)
  {
    // the workgroup that holds the last chunk: sentinel + total
    const u32 last = hp.nChunks - 1u - cFirst;
    const u32 total = before + s_loc[last] + s_n[last];
    b.blockOff[hp.nBlocks] = hp.blobEnd;     // end of the last block
    if (total != hp.nBlocks) b.fallback[2] = b.epoch;
  }
  if (fastRaised(b.fallback, b.epoch)) return;    // raised by an earlier kernel: nothing below can be trusted
  const u32 i = threadIdx.x;                       // entry of each chunk's list
  bool bad = false;
#pragma unroll
  for (u32 k = 0; k < kGatherChunks; k++)
  {
    const u32 c = cFirst + k, ls = s_lane[k];      // walk | index of the chunk's first block in its list << 8
    const u32 src = min((ls >> 8) + i, (u32)kFastListCap - 1u);
    const u32 v = b.lists[((size_t)c * kDiscWalks + (ls & 7u)) * kFastListCap + src];    // (chunks behind the last one: inside the buffer's slack)
    const u32 at = before + s_loc[k] + i;
    if (i < s_n[k]) { if (at < hp.nBlocks) b.blockOff[at] = c * kFastChunkBytes + v; else bad = true; }
  }
  if (__any(bad) && lane == 0) b.fallback[2] = b.epoch;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
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

template<class T, bool WIDE>
__device__ __forceinline__ void
fastDecodeBody(const FastDecodeParams* __restrict__ P, const u8* __restrict__ blob, const u32* __restrict__ blockOff,
              T* __restrict__ outPix, u32* __restrict__ fallback, u32 epoch)
{
  const FastDecodeParams hp = *P;
  if (!hp.ok) return;
  const u32 blobEnd = hp.blobEnd;
  const struct { int nCols, version; double invScale, zMaxHdr; } p = { (int)hp.nCols, (int)hp.version, hp.invScale, hp.zMaxHdr };
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW, IT = C::IT;
  constexpr int DT = DtOf<T>::v;
  constexpr int kSpanWords = (kFastBlocksPerWG * (1 + 64 * (int)sizeof(T)) + 32) / 4 + 8;
  __shared__ __align__(16) u32 s_in[kSpanWords];
  __shared__ u32 s_off[kFastBlocksPerWG + 1];
  __shared__ u32 s_code[kFastBlocksPerWG];     // parseCode of the block, 0 = bad
  __shared__ double s_offs[kFastBlocksPerWG];
  if (fallback[0] == epoch || fallback[1] == epoch || fallback[2] == epoch) return;    // raised by an earlier kernel

  PROBE_BEGIN;
  const int w = waveId(), lane = laneId();
  const int r = lane >> 3, c = lane & 7, b = c / LPR, h = c % LPR;
  const FastSpan span = fastSpanOf(blockIdx.x, hp.nTH, hp.nRows / 8u);
  const u32 firstBlk = blockIdx.x * kFastBlocksPerWG;

  if (threadIdx.x <= kFastBlocksPerWG) s_off[threadIdx.x] = blockOff[min(firstBlk + threadIdx.x, hp.nBlocks)];    // [nBlocks] = end of the stream
  __syncthreads();
  const u32 g0 = s_off[0], g1 = s_off[kFastBlocksPerWG];
  const u32 spanLen = g1 - g0;
  if (g1 < g0 || spanLen > (u32)(kFastBlocksPerWG * (1 + 64 * (int)sizeof(T))) || g1 > blobEnd)
  {
    if (threadIdx.x == 0) fallback[3] = epoch;
    return;
  }
  PROBE(8);
  // ---- stage the span (16-byte loads from the aligned-down start; the checksum was taken care of by the discovery waves)
  const u32 a0 = g0 & ~15u;
  const u32 shift = g0 - a0;
  const u32 nChunks = (shift + spanLen + 15) >> 4;
  for (u32 ch = threadIdx.x; ch < nChunks; ch += 256)
  {
    uint4 x;
    if (a0 + ch * 16 + 16 <= blobEnd) x = *reinterpret_cast<const uint4*>(blob + a0 + ch * 16);
    else
    {
      u32 t4[4] = { 0, 0, 0, 0 };
      for (u32 k = 0; a0 + ch * 16 + k < blobEnd; k++) t4[k >> 2] |= (u32)blob[a0 + ch * 16 + k] << (8 * (k & 3));    // never read past the blob
      x = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
    *reinterpret_cast<uint4*>(&s_in[ch * 4]) = x;
  }
  __syncthreads();
  PROBE(9);

  // ---- parse the 64 block headers once: lane = block
  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  if (w == (int)((blockIdx.x * 2654435761u) >> 30))    // rotates over the waves (= SIMDs) from workgroup to workgroup
  {
    const u32 off = s_off[lane];
    const u32 jt = fastSpanCol(span, (u32)lane);
    u32 h0, h1, h2;
    ldsHeader<DT>(s_in, off - a0, h0, h1, h2);
    u32 code = parseCode<DT>(h0, h1, h2, p.version);
    const bool exists = fastSpanHas(span, (u32)lane);    // (the last workgroup may hold fewer than 64 blocks)
    if (off + codeLen(code) != s_off[lane + 1]) code = 0;
    if (((h0 >> 2) & pattern) != (jt & pattern)) code = 0;    // signature = (j0 >> 3) & pattern, j0 = 8 jt
    if (!exists) code = 0;
    double offset = 0;
    const u32 mode = codeMode(code);
    if (code && (mode == 1 || mode == 3))
    {
      const u32 offB = codeOffBytes(code);
      u64 bits = (((u64)h1 << 32) | h0) >> 8;
      if (DT == DT_Double) bits |= (u64)h2 << 56;
      if (offB < 8) bits &= (1ull << (8 * offB)) - 1;
      offset = typedFromBits(bits, typeUsed(DT, (int)((h0 >> 6) & 3u)));
    }
    s_offs[lane] = offset;
    s_code[lane] = code;
    if (__any(code == 0u && exists) && lane == 0) fallback[3] = epoch;
  }
  __syncthreads();
  PROBE(10);

  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  bool bad = false;
#pragma unroll
  for (int t = 0; t < IT; t++)
  {
    const int tile = t * 4 + w;
    const int blk = tile * BPW + b;
    const u32 code = s_code[blk];
    const double offset = s_offs[blk];
    const u32 mode = codeMode(code), lut = codeLut(code), offB = codeOffBytes(code);
    const u32 pbit = 8u * (s_off[blk] - a0 + ((mode == 1u) ? 3u + offB + lut : 1u));    // payload / first raw value
    const int e0 = r * 8 + h * V;
    T v[V];
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = T(0);
    if (code)
    {
      if (mode == 0)
      {
#pragma unroll
        for (int k = 0; k < V; k++)
        {
          const u32 bp = pbit + (u32)(e0 + k) * 8u * (u32)sizeof(T);
          u64 bits = ldsBits(s_in, bp, 32);
          if (sizeof(T) == 8) bits |= (u64)ldsBits(s_in, bp + 32, 32) << 32;
          else if (sizeof(T) < 4) bits &= (1ull << (8 * sizeof(T))) - 1;
          memcpy(&v[k], &bits, sizeof(T));
        }
      }
      else if (mode == 3)
      {
#pragma unroll
        for (int k = 0; k < V; k++) v[k] = (T)offset;
      }
      else if (mode == 1)
      {
        const int nb = (int)codeBits(code);
        const i64 offI = (i64)offset;
        if (!lut)
        {
#pragma unroll
          for (int k = 0; k < V; k++)
            v[k] = dequant<T>(offset, ldsBits(s_in, pbit + (u32)(e0 + k) * (u32)nb, nb), p.invScale, p.zMaxHdr, offI, invI, zMaxI);
        }
        else
        {
          const u32 nLut = codeNLut(code);
          const int nbIdx = bitLen(nLut);
          const u32 idxBit = pbit + 8u * ((nLut * (u32)nb + 7) >> 3);
#pragma unroll
          for (int k = 0; k < V; k++)
          {
            u32 ix = ldsBits(s_in, idxBit + (u32)(e0 + k) * (u32)nbIdx, nbIdx);
            if (ix > nLut) { ix = 0; bad = true; }    // the reference would read outside its table here
            const u32 q = ix ? ldsBits(s_in, pbit + (ix - 1) * (u32)nb, nb) : 0u;
            v[k] = dequant<T>(offset, q, p.invScale, p.zMaxHdr, offI, invI, zMaxI);
          }
        }
      }
    }
    struct alignas(sizeof(T) * V) Vec { T e[V]; };
    Vec o;
#pragma unroll
    for (int k = 0; k < V; k++) o.e[k] = v[k];
    i64 at;
    if (WIDE) at = (i64)(span.it0 * 8u + (u32)r) * p.nCols + (i64)span.jt0 * 8 + tile * (BPW * 8) + c * V;    // one block row: constant stride
    else { const u32 j = (u32)blk; at = (i64)(fastSpanRow(span, j) * 8u + (u32)r) * p.nCols + (i64)fastSpanCol(span, j) * 8 + h * V; }
    if (fastSpanHas(span, (u32)blk)) *reinterpret_cast<Vec*>(outPix + at) = o;
  }
  PROBE(11);
  if (__any(bad) && lane == 0) fallback[3] = epoch;
}

// ------------------------------------------------------------------------------------------------
bool fastDecodeEligible(int dt, int version, int mb, int nRows, int nCols, int nDepth, bool allValid)
{
  if (!allValid || nDepth != 1 || mb != 8 || version < 3) return false;
  if (dt == DT_Char || dt == DT_Byte) return false;
  if (!fastDimsOk(dt, nRows, nCols)) return false;
  return true;
}

FastWalkPlan makeFastWalkPlan(int nRows, int nCols, u32 sizeGiven)
{
  // upper bounds from what the caller knows without reading the blob
  FastWalkPlan wp;
  wp.nChunks = ((sizeGiven ? sizeGiven : 1u) + kFastChunkBytes - 1) / kFastChunkBytes;
  wp.nBlocks = (u32)(nRows / 8) * (u32)(nCols / 8);
  wp.nWaves = (wp.nChunks + (u32)kDiscChunks - 1) / (u32)kDiscChunks;
  return wp;
}

// ------------------------------------------------------------------------------------------------
// kernels: blockIdx.y = tile of a batch (one raster: a batch of 1).  Each tile has its own slice of every buffer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tileSlice(FastDecodeBuffers& b, const FastDecodeBatch& t, const u8*& blob, u32& sizeGiven)
{
  const size_t tile = blockIdx.y;
  const size_t sChunk = fastChunkStride(t.nChunks);
  b.params += tile; b.fallback += 4 * tile;
  b.recs += tile * t.nChunks; b.lists += tile * t.nChunks * (size_t)(kDiscWalks * kFastListCap);
  b.chunkCount += tile * sChunk; b.chunkLane += tile * sChunk; b.chunkLocal += tile * sChunk;
  b.groupSum += tile * ((t.nChunks + kResolveWG - 1) / kResolveWG + 1);
  b.blockOff += tile * ((size_t)t.nBlocks + 4); b.waveFletcher += tile * 2 * (size_t)t.nWaves;
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
}

template<int DT>
__global__ void __launch_bounds__(256)
k_fast_discover(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, int nRows, int nCols)
{
  tileSlice(b, t, blob, sizeGiven);
  fastDiscoverBody<DT>(blob, sizeGiven, nRows, nCols, b);
}
__global__ void __launch_bounds__(kResolveWG) k_fast_resolve(FastDecodeBuffers b, FastDecodeBatch t)
{
  const u8* blob = nullptr;
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastResolveBody(b, t.nWaves);
}
__global__ void __launch_bounds__(256) k_fast_gather(FastDecodeBuffers b, FastDecodeBatch t)
{
  const u8* blob = nullptr;
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastGatherBody(b);
}
template<class T, bool WIDE>
__global__ void __launch_bounds__(256)
k_fast_decode(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, T* __restrict__ outPix)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastDecodeBody<T, WIDE>(b.params, blob, b.blockOff, outPix + (size_t)blockIdx.y * t.tileElems, b.fallback, b.epoch);
}

template<class T>
static void launchFastDecodeT(int stage, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                              const FastDecodeBuffers& b, void* out, hipStream_t st)
{
  constexpr int DT = DtOf<T>::v;
  const u32 nT = t.nTiles;
  switch (stage)
  {
    case 0:
      hipLaunchKernelGGL(k_fast_discover<DT>, dim3(t.nWaves, nT), dim3(256), 0, st, b, t, blob, sizeGiven, nRows, nCols);
      break;
    case 1:
      hipLaunchKernelGGL(k_fast_resolve, dim3((t.nChunks + kResolveWG - 1) / kResolveWG, nT), dim3(kResolveWG), 0, st, b, t);
      break;
    case 2:
      hipLaunchKernelGGL(k_fast_gather, dim3((t.nChunks + kGatherChunks - 1) / kGatherChunks, nT), dim3(256), 0, st, b, t);
      break;
    default:
      if ((nCols / 8) % 64 == 0)
        hipLaunchKernelGGL((k_fast_decode<T, true>), dim3((t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG, nT), dim3(256), 0, st, b, t, blob, (T*)out);
      else
        hipLaunchKernelGGL((k_fast_decode<T, false>), dim3((t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG, nT), dim3(256), 0, st, b, t, blob, (T*)out);
      break;
  }
}

void launchFastDecode(int stage, int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                      const FastDecodeBuffers& b, void* out, hipStream_t st)
{
  switch (dt)
  {
    case DT_Short:  launchFastDecodeT<short>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UShort: launchFastDecodeT<unsigned short>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Int:    launchFastDecodeT<int>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UInt:   launchFastDecodeT<unsigned int>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Float:  launchFastDecodeT<float>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Double: launchFastDecodeT<double>(stage, nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    default: break;
  }
}

}    // namespace lerc
