// tile_fast_decode.hip -- streaming decoder kernels for the common case (one band, nDepth == 1, every
// pixel valid, 8 x 8 blocks, nRows % 8 == 0, nCols % 512 == 0).  Same results as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends), so decoding starts with a
// discovery pass over 4 KiB chunks of the stream.  Serial walking is latency bound, so the serial parts run
// as plain one-lane-per-chain kernels straight from global memory (L2 / MALL hits) with every chain of the
// whole stream in flight at once, and only the wide, throughput bound part is staged through LDS:
//   k_fast_candidates  LDS staged.  A workgroup takes a few chunks and tries every position of each chunk's
//                      first `window` bytes as a block start.  Candidates are filtered for a few steps through
//                      compacting queues (most die at once) and the survivors are merged into distinct
//                      chains (a hash on position + signature).  The true first block of a chunk is always
//                      among the survivors.
//   k_fast_chains      one lane per chain walks to the end of its chunk: exit, #blocks, and the first block
//                      start behind every 512-byte sub-chunk boundary it passes.
//   k_fast_resolve     whatever ALL live chains of a chunk agree on is true without knowing which one is
//                      real: entry of chunk c = agreed exit of chunk c-1; #blocks of chunk c = steps + count
//                      of the survivor that starts exactly there; agreed sub-chunk entries.  An exclusive
//                      scan turns the counts into block indices.
//   k_fast_emit        one lane per sub-chunk walks from its agreed entry to the next one and writes the
//                      block offsets.
//   k_fast_decode      a workgroup owns 64 consecutive blocks (8 rows x 512 columns): it stages their byte
//                      span in LDS (accumulating the Fletcher32 sums of those bytes word-wise), parses the 64
//                      block headers once (lane = block), then every lane extracts V consecutive pixels of one
//                      raster row, dequantises (double precision in the reference's expression order for
//                      float types, exact integer arithmetic for integer types) and stores one 16-byte vector.
// Whenever a precondition fails (a block longer than its raw size, disagreeing chains, too many survivors,
// ...) the kernels raise `fallback`, and the host repeats the band with the general kernels.
// Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540.
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

// the first 12 bytes at blob offset pos, straight from global memory.  Never touches a word behind the one that
// holds the last blob byte (device allocations are at least 4-byte granular; blob itself is 16-byte aligned).
struct __attribute__((packed, aligned(4))) Words4 { u32 x0, x1, x2, x3; };    // a 16-byte load that is only 4-byte aligned

template<int DT>
__device__ __forceinline__ void globalHeader(const u8* __restrict__ blob, u32 pos, u32 blobEnd, u32& h0, u32& h1, u32& h2)
{
  const u32 w = pos & ~3u, sh = 8u * (pos & 3u), last = (blobEnd - 1u) & ~3u;
  u32 x0, x1, x2, x3;
  if (w + 12u <= last)    // one request per lane
  {
    const Words4 v = *reinterpret_cast<const Words4*>(blob + w);
    x0 = v.x0; x1 = v.x1; x2 = v.x2; x3 = v.x3;
  }
  else                    // the last bytes of the blob
  {
    x0 = *reinterpret_cast<const u32*>(blob + w);
    x1 = *reinterpret_cast<const u32*>(blob + min(w + 4u, last));
    x2 = *reinterpret_cast<const u32*>(blob + min(w + 8u, last));
    x3 = *reinterpret_cast<const u32*>(blob + min(w + 12u, last));
  }
  h0 = (u32)((((u64)x1 << 32) | x0) >> sh);
  h1 = (u32)((((u64)x2 << 32) | x1) >> sh);
  h2 = (DT == DT_Double) ? (u32)((((u64)x3 << 32) | x2) >> sh) : 0u;
}

// one step of a walk through global memory: the block at `cur`, or 0
template<int DT>
__device__ __forceinline__ u32 stepGlobal(const u8* __restrict__ blob, u32 cur, u32 blobEnd, int version, u32& sig, u32 pattern)
{
  u32 h0, h1, h2;
  globalHeader<DT>(blob, cur, blobEnd, h0, h1, h2);
  const u32 code = parseCode<DT>(h0, h1, h2, version);
  const u32 sg = (h0 >> 2) & pattern;
  const bool ok = (code != 0u) & (cur + codeLen(code) <= blobEnd) & ((sig == kNoOffset) | sigOk(sig, sg, pattern));
  sig = sg;
  return ok ? code : 0u;
}

// ------------------------------------------------------------------------------------------------
// header
// ------------------------------------------------------------------------------------------------
// One lane reads the band header the way Lerc2::ReadHeader / ReadMask / ReadMinMaxRanges do (Lerc2.cpp:790-1008,
// :2642-2677) and decides whether the streaming kernels may take the band.  Everything the later kernels need is
// left in *P, so the host can enqueue the whole decode without having seen a single byte of the blob; it checks
// P->ok (and the fallback bits) when it reads the results back.
// The first 128 bytes of the band sit in 32 registers of the parsing lane; all field offsets are compile-time
// constants per codec version, so the parse is a handful of funnel shifts instead of a chain of byte loads.
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
  hp.nChunks = ok ? (blobSize - oData + kFastChunkBytes - 1) / kFastChunkBytes : 0u;
  hp.invScale = 2 * maxZErr;
  hp.zMaxHdr = zMax;
  // Fletcher terms of the bytes in front of the first block (Lerc2.cpp:1037-1064; word k of blob[14 ..))
  u64 A = 0, B = 0;
#pragma unroll
  for (u32 pos = 0; pos + 14u < oData; pos++)
  {
    const u32 cw = h.byteAt(14 + pos) << ((pos & 1u) ? 0 : 8);
    A += cw; B += (u64)(pos >> 1) * cw;
  }
  hp.prefixA = A; hp.prefixB = B;
  hp.ok = ok ? 1u : 0u;
}

template<int DT>
__device__ __forceinline__ void
fastHeaderBody(const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols, FastDecodeParams* __restrict__ P,
              u32* __restrict__ clearStatus, u32* __restrict__ clearFallback)
{
  // first kernel of a decode: it also clears the cells the later kernels raise flags in (saves a memset launch)
  if (threadIdx.x < 4 && clearStatus) clearStatus[threadIdx.x] = 0u;
  if (threadIdx.x >= 4 && threadIdx.x < 8 && clearFallback) clearFallback[threadIdx.x - 4] = 0u;
  if (threadIdx.x != 0) return;
  FastDecodeParams hp;
  memset(&hp, 0, sizeof(hp));
  Head128 h;
  const uint4* src = reinterpret_cast<const uint4*>(blob);    // the band is 16-byte aligned and at least 70 bytes long
#pragma unroll
  for (int i = 0; i < 8; i++)
  {
    uint4 x = make_uint4(0, 0, 0, 0);
    if ((u32)(16 * i + 16) <= sizeGiven) x = src[i];
    else for (u32 k = 0; 16u * i + k < sizeGiven && k < 16u; k++) (&x.x)[k >> 2] |= (u32)blob[16 * i + k] << (8 * (k & 3));
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
  *P = hp;
}

// ------------------------------------------------------------------------------------------------
// candidates
// ------------------------------------------------------------------------------------------------
static const int kWalkG = kFastCandChunks;    // chunks per workgroup
static const int kMinSteps = 4;               // every candidate is filtered for at least this many blocks
static const int kMaxRounds = 40;            // steps a candidate may need to leave the window; more -> its chain stays apart
static const u32 kSurvivorCap = 512;         // survivors of a workgroup after the last filter step
static const u32 kHashSize = 1024;
static const u32 kChainCap = kFastCandChunks * kFastChainsPerChunk;    // distinct chains of a workgroup (its slice of the chain array)

// appends e for the lanes with p; call in wave-uniform control flow
__device__ __forceinline__ void queuePush(bool p, u32 e, u32* q, u32* qn, u32 cap, u32* over)
{
  const u64 m = __ballot(p);
  if (!m) return;
  const int lane = laneId(), leader = __ffsll((long long)m) - 1;
  u32 base = 0;
  if (lane == leader) base = atomicAdd(qn, (u32)__popcll(m));
  base = __shfl(base, leader);
  if (p)
  {
    const u32 idx = base + (u32)__popcll(m & laneMaskLt());
    if (idx < cap) q[idx] = e; else *over = 1;
  }
}

// queue entry: candidate index g * W + o (13) | position relative to the chunk start (12) << 13 | signature (4) << 25
__device__ __forceinline__ u32 qMake(u32 f, u32 rel, u32 sig) { return f | (rel << 13) | ((sig & 15u) << 25); }
__device__ __forceinline__ u32 qCand(u32 e) { return e & 0x1FFFu; }
__device__ __forceinline__ u32 qRel(u32 e) { return (e >> 13) & 0xFFFu; }
__device__ __forceinline__ u32 qSig(u32 e) { return (e >> 25) & 15u; }

// Every candidate walks until it has left the window (and for at least kMinSteps blocks).  Candidates that sit on one
// path (the true path crosses the window in several blocks, each of them a candidate) arrive at the same block start
// there and merge into one chain, however short the blocks are -- except the last few, which stop up to kMinSteps - 1
// blocks further.  Only the head of every chunk is needed for that: window + kMinSteps blocks + one header.
template<int DT>
__device__ __forceinline__ void
fastCandidatesBody(const FastDecodeParams* __restrict__ P, const u8* __restrict__ blob,
                  u32* __restrict__ chunkListN, u64* __restrict__ chunkList, FastChain* __restrict__ chains, u32* __restrict__ chainCount,
                  u32* __restrict__ fallback)
{
  const FastDecodeParams hp = *P;
  if (!hp.ok) return;
  const int version = (int)hp.version;
  const u32 dataBegin = hp.dataBegin, blobEnd = hp.blobEnd;
  const FastWalkPlan wp = { hp.nChunks, hp.nBlocks, 0u };
  if (blockIdx.x * kWalkG >= wp.nChunks)    // the grid is sized for the largest stream the blob could hold
  {
    if (threadIdx.x == 0) chainCount[blockIdx.x] = 0u;
    return;
  }
  constexpr int TBYTES = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 W = kFastWindow(TBYTES), RAW = 1 + 64 * TBYTES;
  constexpr u32 kHead = (W + kMinSteps * RAW + 16 + 15) & ~15u;             // staged bytes per chunk, from its 16-byte aligned start
  constexpr u32 kSlice = kHead + 32;                                        // LDS bytes per chunk
  constexpr u32 kQueueCap = (kWalkG * W * 3) / 4;    // live candidates after the first step (noise: 3/8 with codec >= 5, 3/4 before)
  constexpr int kRounds = (int)((kWalkG * (kSlice / 16) + 255) / 256);
  static_assert(kWalkG * W <= 0x2000 && W - 1 + (kMinSteps + 1) * RAW <= 0xFFF, "queue entry fields");
  __shared__ __align__(16) u32 s_in[kWalkG * kSlice / 4];
  __shared__ u32 s_qa[kQueueCap], s_qb[kQueueCap];
  __shared__ u32 s_hkey[kHashSize];
  __shared__ u16 s_hval[kHashSize];
  __shared__ u16 s_slot[kSurvivorCap];       // hash slot | owner << 15 of every survivor
  __shared__ u32 s_done[kSurvivorCap];       // survivors: queue entry ...
  __shared__ u8 s_doneSteps[kSurvivorCap];   // ... and the steps it took
  __shared__ u32 s_nq[3], s_nDone, s_nChains, s_over;    // three queue counters in rotation: one barrier per round
  __shared__ u32 s_listN[kWalkG];

  PROBE_BEGIN;
  const u32 c0 = blockIdx.x * kWalkG;
  const u32 nChunksHere = min((u32)kWalkG, wp.nChunks - c0);
  {
    uint4 x[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; k++)    // all loads in flight before the first LDS store
    {
      const u32 i = (u32)k * 256u + threadIdx.x;
      const u32 g = i / (kSlice / 16), q = i - g * (kSlice / 16);
      const u32 src = ((dataBegin + (c0 + g) * kFastChunkBytes) & ~15u) + q * 16u;
      x[k] = make_uint4(0, 0, 0, 0);
      if (g < nChunksHere)
      {
        if (src + 16 <= blobEnd) x[k] = *reinterpret_cast<const uint4*>(blob + src);
        else if (src < blobEnd)
        {
          u32 t4[4] = { 0, 0, 0, 0 };
          for (u32 b = 0; src + b < blobEnd; b++) t4[b >> 2] |= (u32)blob[src + b] << (8 * (b & 3));    // never read past the blob
          x[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < kRounds; k++)
    {
      const u32 i = (u32)k * 256u + threadIdx.x;
      if (i < kWalkG * (kSlice / 16)) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
    }
  }
  for (u32 i = threadIdx.x; i < kHashSize; i += 256) s_hkey[i] = 0;
  if (threadIdx.x == 0) { s_nq[0] = 0; s_nq[1] = 0; s_nq[2] = 0; s_nDone = 0; s_nChains = 0; s_over = 0; }
  if (threadIdx.x < kWalkG) s_listN[threadIdx.x] = 0;
  __syncthreads();
  PROBE(0);

  const u32 pattern = (version >= 5) ? 14u : 15u;

  // ---- step 1: every window position of every chunk
  const u32 nCand = nChunksHere * W;
  for (u32 base = 0; base < nCand; base += 256)
  {
    const u32 f = base + threadIdx.x;
    const u32 g = f / W, o = f - g * W;
    const u32 chunkStart = dataBegin + (c0 + g) * kFastChunkBytes;
    const u32 chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
    const u32 cur = chunkStart + o;
    bool live = (f < nCand) & (cur < chunkEnd) & ((c0 + g != 0) | (o == 0));    // the very first block of the stream is known
    u32 sig = kNoOffset, code = 0;
    if (live) code = stepAt<DT>(s_in + g * (kSlice / 4), chunkStart & ~15u, cur, blobEnd, version, sig, pattern);
    live = live & (code != 0u);
    queuePush(live, qMake(f, o + codeLen(code), sig), s_qa, &s_nq[0], kQueueCap, &s_over);
  }
  __syncthreads();

  // ---- further steps through compacting queues (most candidates die at once); who has left the window (or the
  // chunk) is a survivor
  u32* qIn = s_qa;
  u32* qOut = s_qb;
  for (int round = 1; ; round++)
  {
    const u32 nIn = min(s_nq[(round - 1) % 3], kQueueCap);
    if (nIn == 0) break;
    if (threadIdx.x == 0) s_nq[(round + 1) % 3] = 0;    // read for the last time two barriers ago
    const bool lastRound = round >= kMaxRounds;
    for (u32 base = 0; base < nIn; base += 256)
    {
      const u32 i = base + threadIdx.x;
      const u32 e = (i < nIn) ? qIn[i] : 0;
      const u32 g = qCand(e) / W;
      const u32 chunkStart = dataBegin + (c0 + g) * kFastChunkBytes;
      const u32 chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
      const u32 cur = chunkStart + qRel(e);
      const bool have = i < nIn;
      const bool done = have & (((qRel(e) >= W) & (round >= kMinSteps)) | (cur >= chunkEnd) | lastRound);
      bool live = have & !done;
      u32 out = e;
      if (live)
      {
        u32 sig = qSig(e);
        const u32 code = stepAt<DT>(s_in + g * (kSlice / 4), chunkStart & ~15u, cur, blobEnd, version, sig, pattern);
        live = code != 0u;
        out = qMake(qCand(e), qRel(e) + codeLen(code), sig);
      }
      queuePush(live, out, qOut, &s_nq[round % 3], kQueueCap, &s_over);
      // survivors: wave-aggregated append, like queuePush
      const u64 m = __ballot(done);
      if (m)
      {
        const int lane = laneId(), leader = __ffsll((long long)m) - 1;
        u32 at = 0;
        if (lane == leader) at = atomicAdd(&s_nDone, (u32)__popcll(m));
        at = __shfl(at, leader) + (u32)__popcll(m & laneMaskLt());
        if (done) { if (at < kSurvivorCap) { s_done[at] = e; s_doneSteps[at] = (u8)round; } else s_over = 1; }
      }
    }
    __syncthreads();
    u32* t = qIn; qIn = qOut; qOut = t;
  }
  const u32* qs = s_done;                                          // the survivors
  const u32 nSvAll = s_nDone;
  const u32 nSv = min(nSvAll, kSurvivorCap);
  PROBE(1);

  // ---- merge the survivors into distinct chains: same chunk, position and signature behave alike from here on
  for (u32 i = threadIdx.x; i < nSv; i += 256)
  {
    const u32 e = qs[i];
    const u32 key = (e >> 13 & 0xFFFFu) | ((qCand(e) / W) << 16) | 0x80000000u;    // position + signature, chunk
    u32 h = (key * 2654435761u) >> 22;                             // 10 bits
    u32 owner = 0;
    for (;;)
    {
      const u32 old = atomicCAS(&s_hkey[h], 0u, key);
      if (old == 0u) { s_hval[h] = (u16)atomicAdd(&s_nChains, 1u); owner = 1; break; }
      if (old == key) break;
      h = (h + 1) & (kHashSize - 1);
    }
    s_slot[i] = (u16)(h | (owner << 15));
  }
  __syncthreads();
  const u32 nChains = s_nChains;
  const u32 chainBase = blockIdx.x * kChainCap;
  const bool over = (s_over != 0u) | (nSvAll > kSurvivorCap) | (nChains > kChainCap);
  if (threadIdx.x == 0) chainCount[blockIdx.x] = over ? 0u : nChains;

  // ---- hand the chains and the survivor lists (start, steps so far, chain) of every chunk over
  if (!over)
  {
    for (u32 i = threadIdx.x; i < nSv; i += 256)
    {
      const u32 e = qs[i];
      const u32 f = qCand(e), g = f / W, o = f - g * W;
      const u32 slot16 = s_slot[i];
      const u32 chain = chainBase + s_hval[slot16 & 0x7FFFu];
      const u32 slot = atomicAdd(&s_listN[g], 1u);
      if (slot < (u32)kFastListCap) chunkList[(size_t)(c0 + g) * kFastListCap + slot] = (u64)(o | ((u32)s_doneSteps[i] << 16)) | ((u64)chain << 32);
      if (slot16 >> 15)
      {
        FastChain ch;
        ch.cur = dataBegin + (c0 + g) * kFastChunkBytes + qRel(e);
        ch.chunkSig = (c0 + g) | (qSig(e) << 28);
        ch.exit = 0; ch.count = 0; ch.alive = 0;
        for (int j = 0; j < kFastSubPerChunk; j++) { ch.marks[j] = 0; ch.markCount[j] = 0; }
        chains[chain] = ch;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nChunksHere)
  {
    // (no survivor at all is fine for the last chunk of a stream when the last block begins before it: k_fast_resolve
    // tells a chunk without its true entry from an empty one)
    const bool ok = !over && s_listN[threadIdx.x] <= (u32)kFastListCap;
    chunkListN[c0 + threadIdx.x] = ok ? s_listN[threadIdx.x] : 0u;
    if (!ok) atomicOr(fallback, 1u);
  }
  PROBE(2);
}

// ------------------------------------------------------------------------------------------------
// chains
// ------------------------------------------------------------------------------------------------
template<int DT>
__device__ __forceinline__ void
fastChainsBody(const FastDecodeParams* __restrict__ P, const u8* __restrict__ blob, FastChain* __restrict__ chains,
              const u32* __restrict__ chainCount, u32 chainCap)
{
  const FastDecodeParams hp = *P;
  if (!hp.ok) return;
  const int version = (int)hp.version;
  const u32 dataBegin = hp.dataBegin, blobEnd = hp.blobEnd;
  const u32 t = blockIdx.x * 256u + threadIdx.x;    // chain slot: kChainCap per candidates workgroup
  if (t >= chainCap || t % kChainCap >= chainCount[t / kChainCap]) return;
  constexpr int NS = kFastSubPerChunk;
  const u32 pattern = (version >= 5) ? 14u : 15u;
  const FastChain ch = chains[t];
  const u32 chunk = ch.chunkSig & 0x0FFFFFFFu;
  const u32 chunkStart = dataBegin + chunk * kFastChunkBytes;
  const u32 chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
  u32 cur = ch.cur, sig = ch.chunkSig >> 28, count = 0;
  u32 nextSub = (cur - chunkStart) / kFastSubBytes + 1;    // boundaries at or before the chain start are nobody's
  u32 nextBoundary = chunkStart + nextSub * kFastSubBytes;
  u16 marks[NS], markCount[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) { marks[j] = 0; markCount[j] = 0; }
  bool alive = true;
  while (cur < chunkEnd)
  {
    const u32 code = stepGlobal<DT>(blob, cur, blobEnd, version, sig, pattern);
    if (code == 0u) { alive = false; break; }
    cur += codeLen(code); count++;
    while (nextSub < (u32)NS && cur >= nextBoundary)
    {
#pragma unroll
      for (int j = 1; j < NS; j++) if ((u32)j == nextSub) { marks[j] = (u16)(cur - chunkStart); markCount[j] = (u16)count; }
      nextSub++; nextBoundary += kFastSubBytes;
    }
  }
  FastChain out = ch;
  out.exit = cur; out.count = (u16)count; out.alive = alive ? 1 : 0;
#pragma unroll
  for (int j = 0; j < NS; j++) { out.marks[j] = marks[j]; out.markCount[j] = markCount[j]; }
  chains[t] = out;
}

// ------------------------------------------------------------------------------------------------
// resolve
// ------------------------------------------------------------------------------------------------
// the exit every live chain of chunk c agrees on, or kNoOffset
__device__ __forceinline__ u32 agreedExit(u32 c, const u32* __restrict__ chunkListN, const u64* __restrict__ chunkList,
                                          const FastChain* __restrict__ chains)
{
  const u32 n = min(chunkListN[c], (u32)kFastListCap);
  u32 ex = kNoOffset;
  bool any = false, same = true;
  for (u32 i = 0; i < n; i++)
  {
    const FastChain& ch = chains[(u32)(chunkList[(size_t)c * kFastListCap + i] >> 32)];
    if (!ch.alive) continue;
    if (!any) { ex = ch.exit; any = true; }
    else if (ch.exit != ex) same = false;
  }
  return (any && same) ? ex : kNoOffset;
}

template<int DT>
__device__ __forceinline__ void
fastResolveBody(const FastDecodeParams* __restrict__ P, const u8* __restrict__ blob, const u32* __restrict__ chunkListN,
               const u64* __restrict__ chunkList, const FastChain* __restrict__ chains, u32* __restrict__ chunkEntry,
               u32* __restrict__ chunkCount, u32* __restrict__ subEntry, u32* __restrict__ subIndex, u32* __restrict__ fallback)
{
  const FastDecodeParams hp = *P;
  if (!hp.ok) return;
  const int version = (int)hp.version;
  const u32 dataBegin = hp.dataBegin, blobEnd = hp.blobEnd;
  const FastWalkPlan wp = { hp.nChunks, hp.nBlocks, 0u };
  constexpr int NS = kFastSubPerChunk;
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c > wp.nChunks) return;
  if (c == wp.nChunks) { chunkEntry[c] = blobEnd; return; }
  // Only the exits need agreement (they break the chunk-to-chunk dependency).  Once the entry of this chunk is
  // known, the survivor that starts there IS the true path, and so is the chain it merged into.
  const u32 e = (c == 0) ? dataBegin : agreedExit(c - 1, chunkListN, chunkList, chains);
  chunkEntry[c] = e;
  const u32 chunkStart = dataBegin + c * kFastChunkBytes;
  const u32 n = min(chunkListN[c], (u32)kFastListCap);
  u32 steps = 0, chainIdx = kNoOffset;
  for (u32 i = 0; i < n; i++)
  {
    const u64 rec = chunkList[(size_t)c * kFastListCap + i];
    if (e != kNoOffset && ((u32)rec & 0xFFFFu) == e - chunkStart) { steps = ((u32)rec >> 16) & 0xFFFFu; chainIdx = (u32)(rec >> 32); }
  }
  u32 pos[NS], idx[NS];
#pragma unroll
  for (int j = 0; j < NS; j++) { pos[j] = kNoOffset; idx[j] = 0; }
  u32 count = 0;
  bool ok = chainIdx != kNoOffset;
  // the last block of the stream may begin before the last chunk and end with it: nothing starts in that chunk
  const bool emptyTail = e != kNoOffset && e >= min(chunkStart + kFastChunkBytes, blobEnd);
  if (emptyTail) ok = (e == blobEnd);
  else if (ok)
  {
    const FastChain ch = chains[chainIdx];
    ok = ch.alive != 0;
    // the first steps, before the survivor joined its chain
    const u32 pattern = (version >= 5) ? 14u : 15u;
    u32 cur = e, sig = kNoOffset, nextSub = 1;
    pos[0] = e;
    for (u32 s = 0; s < steps && ok; s++)
    {
      const u32 code = stepGlobal<DT>(blob, cur, blobEnd, version, sig, pattern);
      if (code == 0u) { ok = false; break; }
      cur += codeLen(code);
      while (nextSub < (u32)NS && cur >= chunkStart + nextSub * kFastSubBytes)
      {
#pragma unroll
        for (int j = 1; j < NS; j++) if ((u32)j == nextSub) { pos[j] = cur; idx[j] = s + 1; }
        nextSub++;
      }
    }
    ok = ok && cur == ch.cur;
#pragma unroll
    for (int j = 1; j < NS; j++)
      if ((u32)j >= nextSub && ch.marks[j] != 0) { pos[j] = chunkStart + ch.marks[j]; idx[j] = steps + ch.markCount[j]; }    // 0: behind the end of the blob
    count = steps + ch.count;
  }
  if (!ok) { atomicOr(fallback, 2u); count = 0; }
  chunkCount[c] = count;
#pragma unroll
  for (int j = 0; j < NS; j++) { subEntry[(size_t)c * NS + j] = (ok && !emptyTail) ? pos[j] : kNoOffset; subIndex[(size_t)c * NS + j] = idx[j]; }
}

// ------------------------------------------------------------------------------------------------
// block offsets
// ------------------------------------------------------------------------------------------------
template<int DT>
__device__ __forceinline__ void
fastEmitBody(const FastDecodeParams* __restrict__ P, const u8* __restrict__ blob, const u32* __restrict__ chunkEntry,
            const u32* __restrict__ chunkCount, const u32* __restrict__ subEntry, const u32* __restrict__ subIndex,
            u32* __restrict__ blockOff, u32* __restrict__ fallback)
{
  constexpr int NS = kFastSubPerChunk;
  constexpr u32 kChunksPerWG = 256 / NS;
  __shared__ u32 s_part[4], s_cbase[kChunksPerWG];
  const FastDecodeParams hp = *P;
  if (!hp.ok) return;
  const int version = (int)hp.version;
  const u32 blobEnd = hp.blobEnd;
  const FastWalkPlan wp = { hp.nChunks, hp.nBlocks, 0u };
  const u32 c0 = blockIdx.x * kChunksPerWG;
  if (c0 >= wp.nChunks) return;              // the grid is sized for the largest stream the blob could hold
  const int lane = laneId(), w = waveId();

  // index of this workgroup's first block: every workgroup adds up the counts of the chunks before its own
  // (about 100 workgroups x 100 KB out of L2 -- cheaper than a scan kernel in between)
  u32 sum = 0;
  for (u32 i = threadIdx.x; i < c0; i += 256u) sum += chunkCount[i];
  sum = waveSum(sum);
  if (lane == 0) s_part[w] = sum;
  __syncthreads();
  if (w == 0)
  {
    const u32 cc = c0 + (u32)lane;
    const u32 cnt = (lane < (int)kChunksPerWG && cc < wp.nChunks) ? chunkCount[cc] : 0u;
    u32 inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
    const u32 before = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    if (lane < (int)kChunksPerWG) s_cbase[lane] = before + inc - cnt;
    const u32 total = before + __shfl(inc, 63);
    if (lane == 0 && c0 + kChunksPerWG >= wp.nChunks)
    {
      blockOff[wp.nBlocks] = blobEnd;        // sentinel: end of the last block
      if (total != wp.nBlocks) atomicOr(fallback, 4u);
    }
  }
  __syncthreads();

  const u32 t = blockIdx.x * 256u + threadIdx.x;
  const u32 c = t / NS, j = t % NS;
  if (c >= wp.nChunks || *fallback != 0u) return;    // raised by an earlier kernel: nothing below can be trusted
  const u32 start = subEntry[(size_t)c * NS + j];
  if (start == kNoOffset) return;
  // this lane walks [start, limit): up to the next sub-chunk entry, or the entry of the next chunk.  Entries never
  // decrease from lane to lane (positions on one path); equal ones (a block spanning two boundaries) leave nothing.
  u32 limit = chunkEntry[c + 1], endIdx = chunkCount[c];
  if (j + 1 < (u32)NS)
  {
    const u32 nx = subEntry[(size_t)c * NS + j + 1];
    if (nx != kNoOffset) { limit = nx; endIdx = subIndex[(size_t)c * NS + j + 1]; }
  }
  const u32 base = s_cbase[c - c0];
  u32 at = base + subIndex[(size_t)c * NS + j];
  u32 cur = start, sig = kNoOffset;
  bool bad = false;
  while (cur < limit)
  {
    const u32 code = stepGlobal<DT>(blob, cur, blobEnd, version, sig, 0u);    // signatures are checked by the decoder
    if (code == 0u || at >= wp.nBlocks) { bad = true; break; }
    blockOff[at++] = cur;
    cur += codeLen(code);
  }
  if (bad || cur != limit || at != base + endIdx) atomicOr(fallback, 8u);
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

__device__ __forceinline__ void fletcherWordD(u32 x, u32 pos, u64& A, u64& B)
{
  const u32 w0 = ((x & 0xFFu) << 8) | ((x >> 8) & 0xFFu), w1 = ((x >> 8) & 0xFF00u) | (x >> 24);
  const u32 k = pos >> 1;
  A += w0 + w1;
  B += (u64)k * w0 + (u64)(k + 1) * w1;
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
              T* __restrict__ outPix, u64* __restrict__ wgFletcher, const u32* __restrict__ fallback, DeviceStatus* st)
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
  __shared__ u64 s_fa[4], s_fb[4];
  if (*fallback) return;    // only earlier kernels raise it

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
    if (threadIdx.x == 0) raiseError(st, kFailed, blockIdx.x);
    return;
  }
  PROBE(8);
  // ---- stage the span (16-byte loads from the aligned-down start) + Fletcher sums of the owned bytes
  const u32 a0 = g0 & ~15u;
  const u32 shift = g0 - a0;
  const u32 nChunks = (shift + spanLen + 15) >> 4;
  u64 A = 0, B = 0;
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
    const u32 lo = ch * 16;
    if (lo < shift || lo + 16 > shift + spanLen)
    {
      // first / last chunk: blank the neighbours' bytes before summing
      u32 wd[4] = { x.x, x.y, x.z, x.w };
      for (u32 i = lo; i < lo + 16; i++)
        if (i < shift || i >= shift + spanLen) wd[(i - lo) >> 2] &= ~(0xFFu << (8 * ((i - lo) & 3)));
      x = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
    const u32 pos = a0 + lo - 14;    // a0 + lo is a multiple of 16 and >= 16: even position inside blob[14 ..)
    fletcherWordD(x.x, pos, A, B);
    fletcherWordD(x.y, pos + 4, A, B);
    fletcherWordD(x.z, pos + 8, A, B);
    fletcherWordD(x.w, pos + 12, A, B);
  }
  A %= 65535u; B %= 65535u;
  A = waveSum(A); B = waveSum(B);
  if (lane == 0) { s_fa[w] = A; s_fb[w] = B; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    wgFletcher[2 * (size_t)blockIdx.x] = (s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3]) % 65535u;    // folded by k_fast_fletcher_sum
    wgFletcher[2 * (size_t)blockIdx.x + 1] = (s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]) % 65535u;
  }
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
    if (__any(code == 0u && exists) && lane == 0) raiseError(st, kFailed, blockIdx.x);
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
  if (__any(bad) && lane == 0) raiseError(st, kFailed, blockIdx.x);
}

// folds the workgroups' partial sums and the prefix bytes into the checksum and compares it with the header's
// (Lerc2.cpp:1037-1064)
__device__ __forceinline__ void fastFletcherSumBody(FastDecodeParams* __restrict__ P, const u64* __restrict__ wgFletcher, u32 nWG)
{
  __shared__ u64 s_a[16], s_b[16];
  if (!P->ok) return;
  u64 A = 0, B = 0;
  for (u32 i = threadIdx.x; i < nWG; i += blockDim.x) { A += wgFletcher[2 * (size_t)i]; B += wgFletcher[2 * (size_t)i + 1]; }    // each < 65535
  A = waveSum(A % 65535u); B = waveSum(B % 65535u);
  if (threadIdx.x < 16) { s_a[threadIdx.x] = 0; s_b[threadIdx.x] = 0; }
  __syncthreads();
  if (laneId() == 0) { s_a[waveId()] = A; s_b[waveId()] = B; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  A = P->prefixA; B = P->prefixB;
  for (int i = 0; i < 16; i++) { A += s_a[i]; B += s_b[i]; }
  A %= 65535u; B %= 65535u;
  const u64 N = ((u64)(P->blobEnd - 14u) + 1) / 2;
  u64 s1 = A, s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
  if (s1 == 0) s1 = 0xffff;
  if (s2 == 0) s2 = 0xffff;
  P->checksumOk = ((u32)((s2 << 16) | s1) == P->expectChecksum) ? 1u : 0u;
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
  // upper bounds from what the caller knows without reading the blob: the shortest header in front of a block
  // stream is 67 bytes (codec 3: 62-byte header, mask length, one-sweep flag)
  FastWalkPlan wp;
  const u32 span = sizeGiven > 67u ? sizeGiven - 67u : 1u;
  wp.nChunks = (span + kFastChunkBytes - 1) / kFastChunkBytes;
  wp.nBlocks = (u32)(nRows / 8) * (u32)(nCols / 8);
  wp.chainCap = ((wp.nChunks + kFastCandChunks - 1) / kFastCandChunks) * (u32)(kFastCandChunks * kFastChainsPerChunk);
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
  b.chunkListN += tile * sChunk; b.chunkList += tile * t.nChunks * kFastListCap;
  b.chains += tile * t.chainCap; b.chainCount += tile * ((t.nChunks + kFastCandChunks - 1) / kFastCandChunks);
  b.chunkEntry += tile * sChunk; b.chunkCount += tile * sChunk;
  b.subEntry += tile * t.nChunks * kFastSubPerChunk; b.subIndex += tile * t.nChunks * kFastSubPerChunk;
  b.blockOff += tile * ((size_t)t.nBlocks + 4); b.wgFletcher += tile * 2 * ((t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG);
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
}

template<int DT>
__global__ void __launch_bounds__(64)
k_fast_header(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, int nRows, int nCols, u32* clearStatus)
{
  tileSlice(b, t, blob, sizeGiven);
  fastHeaderBody<DT>(blob, sizeGiven, nRows, nCols, b.params, (b.clearCells && blockIdx.y == 0) ? clearStatus : nullptr,
                     b.clearCells ? b.fallback : nullptr);
}
template<int DT>
__global__ void __launch_bounds__(256)
k_fast_candidates(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastCandidatesBody<DT>(b.params, blob, b.chunkListN, b.chunkList, b.chains, b.chainCount, b.fallback);
}
template<int DT>
__global__ void __launch_bounds__(256)
k_fast_chains(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastChainsBody<DT>(b.params, blob, b.chains, b.chainCount, t.chainCap);
}
template<int DT>
__global__ void __launch_bounds__(256)
k_fast_resolve(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastResolveBody<DT>(b.params, blob, b.chunkListN, b.chunkList, b.chains, b.chunkEntry, b.chunkCount, b.subEntry, b.subIndex, b.fallback);
}
template<int DT>
__global__ void __launch_bounds__(256)
k_fast_emit(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastEmitBody<DT>(b.params, blob, b.chunkEntry, b.chunkCount, b.subEntry, b.subIndex, b.blockOff, b.fallback);
}
template<class T, bool WIDE>
__global__ void __launch_bounds__(256)
k_fast_decode(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, T* __restrict__ outPix, DeviceStatus* st)
{
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastDecodeBody<T, WIDE>(b.params, blob, b.blockOff, outPix + (size_t)blockIdx.y * t.tileElems, b.wgFletcher, b.fallback, st);
}
__global__ void __launch_bounds__(1024) k_fast_fletcher_sum(FastDecodeBuffers b, FastDecodeBatch t)
{
  const u8* blob = nullptr;
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven);
  fastFletcherSumBody(b.params, b.wgFletcher, (t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG);
}

template<class T>
static void launchFastDecodeT(int stage, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                              const FastDecodeBuffers& b, void* out, DeviceStatus* status, hipStream_t st)
{
  constexpr int DT = DtOf<T>::v;
  const u32 nT = t.nTiles;
  switch (stage)
  {
    case 0:
      hipLaunchKernelGGL(k_fast_header<DT>, dim3(1, nT), dim3(64), 0, st, b, t, blob, sizeGiven, nRows, nCols, reinterpret_cast<u32*>(status));
      break;
    case 1:
      hipLaunchKernelGGL(k_fast_candidates<DT>, dim3((t.nChunks + kWalkG - 1) / kWalkG, nT), dim3(256), 0, st, b, t, blob);
      break;
    case 2:
      hipLaunchKernelGGL(k_fast_chains<DT>, dim3((t.chainCap + 255) / 256, nT), dim3(256), 0, st, b, t, blob);
      break;
    case 3:
      hipLaunchKernelGGL(k_fast_resolve<DT>, dim3((t.nChunks + 256) / 256, nT), dim3(256), 0, st, b, t, blob);
      break;
    case 4:
      hipLaunchKernelGGL(k_fast_emit<DT>, dim3((t.nChunks * kFastSubPerChunk + 255) / 256, nT), dim3(256), 0, st, b, t, blob);
      break;
    case 5:
      if ((nCols / 8) % 64 == 0)
        hipLaunchKernelGGL((k_fast_decode<T, true>), dim3((t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG, nT), dim3(256), 0, st, b, t, blob, (T*)out, status);
      else
        hipLaunchKernelGGL((k_fast_decode<T, false>), dim3((t.nBlocks + kFastBlocksPerWG - 1) / kFastBlocksPerWG, nT), dim3(256), 0, st, b, t, blob, (T*)out, status);
      break;
    default:
      hipLaunchKernelGGL(k_fast_fletcher_sum, dim3(1, nT), dim3(nT > 1 ? 256 : 1024), 0, st, b, t);
      break;
  }
}

void launchFastDecode(int stage, int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                      const FastDecodeBuffers& b, void* out, DeviceStatus* status, hipStream_t st)
{
  switch (dt)
  {
    case DT_Short:  launchFastDecodeT<short>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    case DT_UShort: launchFastDecodeT<unsigned short>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    case DT_Int:    launchFastDecodeT<int>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    case DT_UInt:   launchFastDecodeT<unsigned int>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    case DT_Float:  launchFastDecodeT<float>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    case DT_Double: launchFastDecodeT<double>(stage, nRows, nCols, t, blob, sizeGiven, b, out, status, st); break;
    default: break;
  }
}

}    // namespace lerc
