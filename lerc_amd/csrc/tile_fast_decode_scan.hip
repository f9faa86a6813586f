// tile_fast_decode_scan.hip -- the streaming decoder's first tier: ONE launch, and no walk at all.
//
// The block stream stores no offsets (block k + 1 starts where block k ends; Lerc2::ReadTiles, Lerc2.cpp:1672-1713).  The two tiers
// behind this one (tile_fast_decode_one.hip, tile_fast_decode.hip) find the block starts by letting block headers found in a few
// windows WALK the stream -- chains of dependent steps, a dozen barriers, lists to settle which walk is the path.  Here every byte
// of the stream is looked at once, in the registers it arrives in, and the chain is never followed:
//   1. a workgroup of 512 threads takes a PIECE of 32 KiB of the blob (+ one block's length in front of it and behind it), every
//      lane four or five 16-byte units; on the way to LDS the Fletcher32 terms of the piece, and the SCAN: "a byte 64 behind a byte
//      10?nnnnn" -- the count byte of a bit-stuffed block of 64 values behind its bits byte (BitStuffer2.cpp:35-77) -- by exact
//      byte-parallel arithmetic, 13 instructions a dword; units with a hit go to a queue (a ballot and one LDS atomic a wave);
//   2. CANDIDATES, lane = unit with hits: the flag byte 2 + (bytes of the offset) in front of the count byte has to read "bit
//      stuffed, that offset type" (Lerc2.cpp:1961-2021); the block's length follows from the bits byte (and the table size).  A
//      candidate sets two bits: START[where it begins], END[where it ends];
//   3. SURVIVORS = START & END: a block that begins where another one ends.  Every block of the path is one (its predecessor
//      ends there: hence the block's length of bytes in front of the piece); of what the scan finds that is no block -- offset
//      bytes that look like a flag byte in front of the same header, payload bytes: 15 to 30 a piece -- almost nothing is (it
//      would have to begin where another false candidate ends: one piece in 60 of the uint16 raster, none of the float one);
//      popcounts and one scan turn the bitmap into the list of the piece's block starts;
//   4. CHECK, lane = block: the header in full (parseCode: ReadTile's and BitStuffer2::Decode's checks), and "this block ends
//      where the next one of the list begins" -- the last one behind the piece.  A list that TILES its stretch of the stream and
//      begins where the piece in front says its last block ends is the path, whatever the bitmaps looked like: the piece in
//      front is right by the same argument, the first piece begins with the stream's first block, the last one has to end
//      with the blob, and the blocks have to be as many as the raster has;
//   5. where the list does not tile (a false survivor; blocks that are not bit-stuffed -- constant, all zero, raw -- which the
//      scan does not see, and the block behind each of them, which has no END) the piece's first WAVE mends it (a masked band: one
//      thread): survivors inside a good block's extent are struck, gaps are walked and what is found there entered -- a RUN of
//      constant / all-zero blocks (a flat stretch of the raster: hundreds on end) 64 blocks a step, lane i looking i block lengths
//      on --, and a piece that begins inside such a run (no anchor in the bytes in front of it) takes its first block's place from
//      the piece in front.  A piece that needs more than its tables hold gives up; the host then takes the band to the next tier
//      (and keeps to it for a while: codec_decode.cpp);
//   6. count out (an epoch-tagged cell: blocks of the piece, where its last block ends -- EARLY, form 4: the count leaves as soon as
//      the survivors are counted, before steps 4 - 5; a piece whose final count differs says so and the band is decoded once more with
//      late counts), the cells of the pieces in front added up -- the only wait --, then the pixels as in tile_fast_decode_one.hip:
//      lane = V pixels of one block row (float: of two).
// Two more rules keep false candidates rare: a candidate is no longer than its raw form would be, and the byte where it ends has
// to read like the flag byte of the next block -- the column signature going on, by a step or none, or beginning again with a
// block row (four in five of anything else do not).  The ANCHOR -- where the piece's first block begins -- comes from the last
// survivors in FRONT of the piece's own bytes, not from the own bytes' first survivor (a false one there may tile with what follows).
//
// MODE 1 cuts the block stream of a band WITH A MASK into blocks (k_fast_scan_offsets; the general kernels decode the pixels,
// which need the mask): count bytes of 1 ... 64; RUNS of one-byte blocks (no valid pixel) found by a FLOOD over the bitmaps, seeded
// where the survivors end, its carries between threads and waves by two ballots; false survivors struck by the lane that sees a
// block end at the entry after next; the hits laid out one behind the other and taken lane = hit (a unit holds up to eight);
// the one thread's mending left with RAW blocks, whose length only the mask knows (counts are tried against a chain of blocks
// that parse, the signature going on in pairs, up to a known block -- a guess the decode kernels, which have the mask, may refuse:
// the general discovery then takes the band).  A piece owns the blocks that begin in its bytes.  Limits: 2048 blocks a piece (streams
// of mostly one-byte blocks go to the general discovery), 8 x 8 blocks, one value a pixel, 16-bit and wider types.
// Rasters whose rows / columns are no multiples of 8 take the RAG instantiation: the edge blocks' count bytes (8 x rows mod 8, 8 x columns
// mod 8) in the filter beside 64, a block's size checked against its place, raw edge blocks sized by where the next block begins,
// partial rows stored pixel by pixel (Lerc2.cpp:1504-1519).
// Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540; Lerc2.cpp:1037-1064 (checksum).
#include "tile_fast_decode_dev.h"
#include <cstddef>

namespace lerc {

#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time lines (constant-rate counter), read by tools/trace_decode_one.py
static __device__ unsigned long long g_traceS[16 * 8192];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace_decode_scan(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceS), sizeof(unsigned long long) * (size_t)n); }
#define TRACES(slot) do { if (threadIdx.x == 0 && wg < 8192u) g_traceS[16 * wg + (slot)] = wall_clock64(); } while (0)
#define TRACEV(slot, v) do { if (wg < 8192u) g_traceS[16 * wg + (slot)] = (unsigned long long)(v); } while (0)    // (one thread's)
#else
#define TRACES(slot)
#define TRACEV(slot, v)
#endif

// 16-byte vectors of pixels a lane takes out of the stream while the cells of the pieces in front travel: as many as leave the
// kernel at 80 vector registers (six waves a SIMD: three workgroups a CU)
#ifndef LERC_SCAN_SLEEP
#define LERC_SCAN_SLEEP 4
#endif
#ifndef LERC_SCAN_WIDE
#define LERC_SCAN_WIDE 1
#endif
#ifndef LERC_SCAN_HELD
#define LERC_SCAN_HELD (LERC_SCAN_WIDE ? 4 : 6)    // (a lane of eight pixels has more in flight while it decodes: two wide vectors are what 80 registers hold)
#endif
#ifndef LERC_SCAN_HELD32
#define LERC_SCAN_HELD32 4
#endif
#ifndef LERC_SCAN_HELD16
#define LERC_SCAN_HELD16 2
#endif
// tuning: LERC_DEC_EXIT=n builds a decoder that leaves at mark n (profiles/r06_notes.md: instruction counts per phase); results are invalid
#ifndef LERC_SCAN_DIRECT
#define LERC_SCAN_DIRECT 0
#endif
#ifndef LERC_SCAN_STAGGER
#define LERC_SCAN_STAGGER 0
#endif
#ifndef LERC_SCAN_EARLY
#define LERC_SCAN_EARLY 1
#endif
#ifndef LERC_DEC_EXIT
#define LERC_DEC_EXIT 99
#endif
#define DEC_EXIT(n) do { if (LERC_DEC_EXIT == (n)) return; } while (0)
static const u32 kScanBadCap = 64, kScanFalseCap = 64, kScanInsCap = 128, kScanRunCap = 64;    // (the one thread's mending: broken links, entries struck, blocks entered -- a masked band enters into END's bitmap, 2048)

template<class T> struct ScanGeom
{
  static constexpr int DT = DtOf<T>::v;
  static constexpr u32 TB = (u32)sizeof(T), W = kFastWindow((int)sizeof(T));
  static constexpr u32 P = kScanPiece, NT = kScanThreads;
  static constexpr u32 PRE = scanPre((int)sizeof(T)), POST = scanPost((int)sizeof(T));
  static constexpr u32 kBytes = PRE + P + POST, kUnits = kBytes / 16u;
  static constexpr int kRounds = (int)((kUnits + NT - 1u) / NT);
  static constexpr u32 kScanUnits = (PRE + P) / 16u + 1u;        // units the count byte of a block that begins in front of the piece's end can lie in
  static constexpr u32 kMapWords = (kBytes + 31u) / 32u, kMapVecs = (kMapWords + 3u + 3u) / 4u;    // bitmap words; 16-byte vectors that hold them (+ 3 of slack)
  static constexpr u32 kOwnWord0 = PRE / 32u;
  static constexpr u32 R = NT;                                    // blocks per decode round: one header per thread
  static constexpr u32 kListCap = P / (P < 32768u ? 4u : 16u);    // (the emulator's small pieces hold runs as long as the full-size ones)
  static constexpr u32 kQueueSeg = (u32)kRounds * 64u, kQueueCap = (NT / 64u) * kQueueSeg;    // a wave's stretch of the queue holds all its units (a masked band: two units in three have a candidate)
  static_assert(PRE % 32u == 0u && POST % 16u == 0u && P % 2048u == 0u && NT % 64u == 0u, "units, bitmap words, waves");
  static_assert(kBytes + 64u < 65535u, "16-bit positions");
  static_assert(kUnits < 65536u && kScanUnits <= kUnits, "queue entries");
};

// what a piece's verdict is made of (see fastScanBody: verdict)
struct ScanVerdictIn { u32 total, base, dataRel, blobRel, dataBegin, nWanted; };

template<class T> struct ScanShared
{
  typedef ScanGeom<T> G;
  alignas(16) u32 inAll[4 + G::kBytes / 4 + 4];      // [4 ...): the staged bytes; the word in front of them reads 0
  alignas(16) u32 sb[4 * G::kMapVecs];              // START; from step 3 on: START & END of the piece's own bytes
  union U
  {
    alignas(16) u32 end[4 * G::kMapVecs];           // steps 2 - 3
    struct X                                         // the pixels, one round
    {
      double offs[G::R];
      u32 code[G::R];                                // what the pixel loop wants to know of the block, 0 = bad
      u32 at[G::R];                                  // raster offset (pixels) of the block's first pixel
      u8 dims[G::R];                                 // RAG: the block's columns | rows << 4 (the raster's last blocks are smaller)
    } x;
  } u;
  union L
  {
    u16 queue[G::kQueueCap];                         // steps 1 - 2: units that may hold a count byte
    u16 list[G::kListCap + 8];                       // from step 3 on: the block starts, relative to the staged bytes
  } l;
  u16 badIdx[kScanBadCap], falseIdx[kScanFalseCap], insPos[kScanInsCap];
  u16 runPos[kScanRunCap], runLen[kScanRunCap], runCnt[kScanRunCap];    // an unmasked band's mending: blocks entered, run by run (where, a block's bytes, how many)
  u32 wsum[G::NT / 64], qn[G::NT / 64];               // a wave's survivors; units a wave queued
  u64 fa[G::NT / 64], fb[G::NT / 64];
  u64 part;                                          // sum of the cells: this group's in the low half, the groups' in front in the high half
  u32 t0, frontBad;                                  // where the anchor ends (0: no anchor); the list's first entries lie in front of that
  u32 nEnt, nBad[4], nStruck[4], nFalse, nIns, over, bad, lost, exitRel, prevExit, mended;    // nBad: broken links found by the first / the second / the third check
  u32 earlyCount, vDefer;                            // EARLY: what the piece said it holds before the check; the verdict is given behind the pixels
  ScanVerdictIn vIn;
  u32 mendExit;                                      // MODE 1: where the piece's last block ends, if the mending had to find out (a raw block: see there)
#ifdef LERC_PROBE
  u32 dbg[4];
#endif
  u32 preCarry, fco[G::NT / 64];                     // MODE 1, the flood of one-byte blocks: a run goes on from the bytes in front into the piece's own; a wave's carry out (bit 0: without, bit 1: with a carry in)
  FastDecodeParams hp;                               // the band header, parsed in full by the first wave
};

// ANYCOUNT: the count byte of a bit-stuffed block of a MASKED band -- any number of valid pixels, 1 ... 64 -- instead of 64.
// 0x80 clear in every byte of the result where cur4's byte is such a count byte (and set elsewhere)
template<bool ANYCOUNT> __device__ __forceinline__ u32 countByteTest(u32 cur4)
{
  if (!ANYCOUNT) return cur4 ^ 0x40404040u;                                     // a zero byte: 64
  // 1 ... 64: bit 7 clear, and the low seven bits + 63 come to 01......
  return ((((cur4 & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) ^ 0x40404040u) & 0xC0C0C0C0u) | (cur4 & 0x80808080u);
}
// 0x80 in every byte of cur4 that is such a count byte behind a byte 10?nnnnn, n != 0 (prev4: the dword in front of cur4)
template<bool ANYCOUNT> __device__ __forceinline__ u32 countByteHits(u32 cur4, u32 prev4)
{
  const u32 hdr4 = __builtin_amdgcn_alignbit(cur4, prev4, 24);                   // the bytes in front of cur4's
  const u32 t = countByteTest<ANYCOUNT>(cur4) | ((hdr4 & 0xC0C0C0C0u) ^ 0x80808080u);      // a zero byte: a count byte behind 10......
  const u32 z = ((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu;               // 0x80 clear in the zero bytes, and only there
  const u32 nz = ((hdr4 & 0x1F1F1F1Fu) + 0x1F1F1F1Fu) << 2;                         // 0x80 set where n != 0
  return ~z & nz;
}
// the same for ONE count value v (v4 = v in every byte): 0x80 in every byte of cur4 that is v behind a byte 10?nnnnn, n != 0
__device__ __forceinline__ u32 countByteHitsOf(u32 cur4, u32 prev4, u32 v4)
{
  const u32 hdr4 = __builtin_amdgcn_alignbit(cur4, prev4, 24);
  const u32 t = (cur4 ^ v4) | ((hdr4 & 0xC0C0C0C0u) ^ 0x80808080u);
  const u32 z = ((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t | 0x7F7F7F7Fu;
  const u32 nz = ((hdr4 & 0x1F1F1F1Fu) + 0x1F1F1F1Fu) << 2;
  return ~z & nz;
}
// bit 7 of a byte of the result is CLEAR where cur4 reads such a count byte behind a byte 10...... (a filter: n is not looked at)
template<bool ANYCOUNT> __device__ __forceinline__ u32 countByteMaybe(u32 cur4, u32 prev4)
{
  const u32 hdr4 = __builtin_amdgcn_alignbit(cur4, prev4, 24);
  const u32 t = countByteTest<ANYCOUNT>(cur4) | ((hdr4 & 0xC0C0C0C0u) ^ 0x80808080u);
  return ((t & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | t;
}

// bits of a bitmap word (positions base ... base + 31) for the positions [lo, hi)
__device__ __forceinline__ u32 scanRangeMask(u32 base, u32 lo, u32 hi)
{
  const u32 nLo = lo > base ? lo - base : 0u, nHi = hi > base ? hi - base : 0u;
  const u32 under = nLo >= 32u ? 0xFFFFFFFFu : ((1u << nLo) - 1u);
  const u32 below = nHi >= 32u ? 0xFFFFFFFFu : ((1u << nHi) - 1u);
  return below & ~under;
}

// MODE 1 (tile_decode.hip's kernels decode the pixels): the block stream of a band with a MASK -- a block holds 1 ... 64 pixels and its
// count byte says how many; a block without a valid pixel is one byte, "all zero" (Lerc2.h:422) -- is only cut into blocks: where
// block k of the stream begins goes to blockOff[k].  A piece's blocks are those that begin in its own bytes; the piece in front says
// where its last block ends, and this piece's first block has to begin there.
struct ScanOffsetsJob
{
  u32 version, dataBegin, blobEnd;    // of the band (the host has read its header and mask)
  u32* blockOff;                       // [nPos] out: offset of every block of the raster, in raster order
  u32 nPos;
};

template<class T, int MODE, bool RAG = false>
__device__ __forceinline__ void
fastScanBody(ScanShared<T>& S, const FastDecodeBuffers& b, const u8* __restrict__ blob, u32 sizeGiven, u32 specEnd, int nRows, int nCols,
             T* __restrict__ outPix, u32 wg, const ScanOffsetsJob& job)
{
  constexpr bool OFFS = MODE == 1;
  static_assert(!(RAG && OFFS), "ragged rasters with a mask keep to the general discovery");
  // RAG: a raster whose rows / columns are no multiples of 8 -- the last block of a block row holds 8 x wl pixels, the blocks of the last
  // block row hl x 8 (the corner wl x hl), and their count bytes say so (Lerc2.cpp:1504-1519).  The filter also takes the count byte of the
  // LAST BLOCK ROW's blocks (a whole block row of them on end); the one edge block a block row comes in through the mending.
  const u32 ragWl = RAG ? (u32)nCols & 7u : 0u, ragHl = RAG ? (u32)nRows & 7u : 0u;
  const u32 ragC2 = ragHl * 8u;                      // count byte of the last block row's blocks (0: no such row)
  const u32 ragC3 = (ragWl != ragHl) ? ragWl * 8u : 0u;    // ... and of a block row's last block (0: no such column, or the same value): one block in nTH -- left to the mending it is a gap every few blocks of a narrow raster
  // a RAW block's length is 1 + n sizeof(T), n the pixels of the block -- which the stream does not say: a whole block's 64 or an edge
  // block's; where the next block begins says which (the mending looks, the last check takes it from the list)
  auto ragRawOk = [&](u32 bytes) -> bool
  {
    if (bytes < 1u + (u32)sizeof(T) || (bytes - 1u) % (u32)sizeof(T) != 0u) return false;
    const u32 n = (bytes - 1u) / (u32)sizeof(T);
    return n == 64u || (ragWl != 0u && n == 8u * ragWl) || (ragHl != 0u && n == 8u * ragHl) || (ragWl != 0u && ragHl != 0u && n == ragWl * ragHl);
  };
  typedef ScanGeom<T> G;
  constexpr int DT = G::DT;
  constexpr u32 W = G::W, P = G::P, NT = G::NT, PRE = G::PRE, kUnits = G::kUnits;
  constexpr u32 kWaves = NT / 64, kListCap = G::kListCap, kQueueCap = G::kQueueCap, R = G::R;
  constexpr u32 RAW = 1u + 64u * G::TB;
  u32* const s_in = S.inAll + 4;
  auto& s_sb = S.sb; auto& s_end = S.u.end; auto& s_queue = S.l.queue; auto& s_list = S.l.list;
  const int lane = laneId(), w = waveId();

  // ---- the band header: every wave asks for its first 64 bytes; the first wave reads it in full (Lerc2::ReadHeader's checks)
  // while the staged bytes are on their way, and leaves the result in LDS (workgroup 0: also where the host wants it).
  // The piece's own bytes are asked for BEHIND the header's, without waiting for it, where the host expects the blob to reach
  // that far (specEnd: the blob's size where it is known, else what the context's last band of this shape had -- the bands
  // of one job are alike): the header's two microseconds are off the workgroup's critical path.  Pieces beyond that wait.
  const u32 pieceStart = wg * P;                   // blob offset of the piece's first own byte = of LDS byte PRE
  if (LERC_SCAN_STAGGER != 0 && !OFFS && wg < 768u)    // (tuning: the launch's first pieces -- all resident at once -- start one after the other instead of together)
    for (u32 i = 0; i < wg * (u32)LERC_SCAN_STAGGER / 16u; i++) __builtin_amdgcn_s_sleep(1);
  const Head64 h64 = loadHead64(blob, sizeGiven);
  constexpr int kRounds = G::kRounds;
  uint4 x[kRounds];
  const u32 aMine = pieceStart + 16u * threadIdx.x - PRE;    // (wraps for the first piece's units in front of the blob: not loaded)
  // all loads in flight at once (clipped to what the caller says is readable; 32-bit offsets from the blob's first byte: a blob
  // is less than 4 GB; the bytes in front of the first piece do not exist).  A round in which a wave has no unit -- the last
  // one, for all waves but the first -- is skipped by that wave, here and below.
  auto issueLoads = [&]()
  {
#pragma unroll
    for (int k = 0; k < kRounds; k++)
    {
      x[k] = make_uint4(0, 0, 0, 0);
      if ((u32)k * NT + 64u * (u32)w >= kUnits) continue;
      const u32 i = (u32)k * NT + threadIdx.x;
      const u32 a = aMine + (u32)k * NT * 16u;
      if (i < kUnits && (wg != 0u || i >= PRE / 16u))
      {
        if (a <= sizeGiven && sizeGiven - a >= 16u) x[k] = *reinterpret_cast<const uint4*>(blob + a);
        else if (a < sizeGiven)    // never read past the blob
        {
          u32 t4[4] = { 0, 0, 0, 0 };
#pragma unroll
          for (u32 q = 0; q < 16; q++) if (q < sizeGiven - a) t4[q >> 2] |= (u32)blob[a + q] << (8 * (q & 3));
          x[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
        }
      }
    }
  };
  const bool earlyLoads = pieceStart < specEnd && pieceStart < sizeGiven;
  if (earlyLoads) issueLoads();
  HeadLite hl = parseHeadLite<DT>(h64, sizeGiven);
  if (OFFS) { hl.ok = 1u; hl.version = job.version; hl.dataBegin = job.dataBegin; hl.blobEnd = job.blobEnd; }    // (the host has read the header)
  const u32 blobEnd = hl.blobEnd;
  const bool ours = OFFS || (hl.ok && headLiteEligible<DT>(h64, hl.version, nRows, nCols));
  if (!OFFS && wg == 0u && !ours && threadIdx.x == 0)      // (not a band of ours: say so)
  {
    const FastDecodeParams hp0 = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    storeParams<true>(b.params, hp0); if (b.hostParams) *b.hostParams = hp0;
  }
  const u32 nWG = fastScanNumWG(blobEnd);
  // (the grid is sized for the largest stream the blob could hold, or by the host's guess.  A guess that was too small: nobody decodes, the
  // first workgroup hands the band on with its header -- flag 2, below)
  const bool gridShort = !OFFS && nWG > gridDim.x;
  if (!ours || wg >= nWG || (gridShort && wg != 0u)) return;
  if (LERC_DEC_EXIT == 0)    // a workgroup of the same shape (threads, LDS) that loads the header and stores one vector
  {
    S.inAll[threadIdx.x] = blobEnd + threadIdx.x;
    __syncthreads();
    if (!OFFS && threadIdx.x < 4u) outPix[(size_t)wg * 1024u + threadIdx.x] = (T)S.inAll[(threadIdx.x + 1u) & 511u];
    return;
  }
  TRACES(0);
  if (!earlyLoads) issueLoads();
  const int version = (int)hl.version;
  const bool v5 = version >= 5;
  const u32 pattern = v5 ? 14u : 15u;
  const u32 epoch = b.epoch;
  const u64 tag = (u64)b.publishEpoch << 32;
  const bool lastPiece = wg == nWG - 1u;
  // positions are relative to the staged bytes: LDS byte r is blob byte pieceStart + r - PRE
  const u32 blobRel = blobEnd - pieceStart + PRE;                                                    // the blob's end
  const u32 dataRel = hl.dataBegin + PRE > pieceStart ? hl.dataBegin + PRE - pieceStart : 0u;        // the stream's first block (or 0: in front of all this)
  const u32 pieceEndRel = PRE + P;

  if (!OFFS && w == (int)kWaves - 1)    // (the last wave: the first one has a round of staging more)
  {
    const FastDecodeParams hpFull = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    if (lane == 0)
    {
      S.hp = hpFull;
      if (wg == 0u)
      {
        // (the verdict on the checksum comes from the launch's last workgroup, microseconds later for a small blob: one writer
        // per byte -- the host's copy, which travels over PCIe, gets everything BUT that word here (the host has zeroed it))
        storeParams<true>(b.params, hpFull);
        if (b.hostParams)
        {
          u64 wds[8];
          memcpy(wds, &hpFull, 64);
          static_assert(offsetof(FastDecodeParams, checksumOk) == 40 && sizeof(FastDecodeParams) == 64, "word 5 holds the verdict");
#pragma unroll
          for (int i = 0; i < 8; i++) if (i != 5) reinterpret_cast<volatile u64*>(b.hostParams)[i] = wds[i];
        }
      }
    }
  }
  for (u32 i = threadIdx.x; i < G::kMapVecs; i += NT)
  {
    reinterpret_cast<uint4*>(s_sb)[i] = make_uint4(0, 0, 0, 0);
    reinterpret_cast<uint4*>(s_end)[i] = make_uint4(0, 0, 0, 0);
  }
  if (threadIdx.x < 4u) S.inAll[threadIdx.x] = 0u;
  if (threadIdx.x == 0)
  {
    S.nEnt = 0u; S.frontBad = 0u; S.t0 = 0u; S.nBad[0] = 0u; S.nBad[1] = 0u; S.nBad[2] = 0u; S.nBad[3] = 0u; S.nStruck[0] = 0u; S.nStruck[1] = 0u; S.nStruck[2] = 0u; S.nStruck[3] = 0u; S.preCarry = 0u; S.mendExit = 0u; S.nFalse = 0u; S.nIns = 0u; S.over = 0u; S.bad = 0u; S.lost = 0u; S.exitRel = 0u;
#ifdef LERC_PROBE
    S.dbg[0] = S.dbg[1] = S.dbg[2] = S.dbg[3] = 0u;
#endif
    S.prevExit = kNoOffset; S.mended = 0u; S.part = 0ull; S.vDefer = 0u;
  }
  // (no barrier here: nothing below reads what was written above before the barrier behind the staging -- the queue is a
  // wave's own, the bitmaps and the counters are for the steps behind that barrier)

  // ---- stage; Fletcher terms of the piece's own units (bytes 14 ... blobEnd - 1 of the blob are checksummed); scan
  u32 fA = 0, nMine = 0;
  u64 fB = 0;
  constexpr u32 kQueueSeg = G::kQueueSeg;          // a wave's stretch of the queue
  constexpr u32 ownUnit0 = PRE / 16u, ownUnit1 = (PRE + P) / 16u;
  const bool inner = pieceStart != 0u && (u64)pieceStart + P <= blobEnd;    // no unit of this piece needs blanking
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    if ((u32)k * NT + 64u * (u32)w >= kUnits) continue;    // (the same for all lanes of the wave)
    const u32 i = (u32)k * NT + threadIdx.x;
    if (i < kUnits) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
    const u32 a = pieceStart + 16u * i - PRE;                                 // (own units: >= 0 and < 2^32)
    if (OFFS) { }    // (the general path sums the band's checksum in a kernel of its own)
    else if (inner)
    {
      if (i >= ownUnit0 && i < ownUnit1) fletcherUnit(x[k], (a - 14u) / 2u, fA, fB);    // unit at blob offset a holds words (a - 14) / 2 ...
    }
    else if (i >= ownUnit0 && i < ownUnit1 && a < blobEnd)
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
      fletcherUnit(y, a ? (a - 14u) / 2u : 65528ull, fA, fB);             // (the first unit's index -7 as its residue mod 65535)
    }
    // the scan, first half: may the unit hold a count byte at all?  Those that may go to the wave's own stretch of the queue (a ballot,
    // no atomic), and the step below looks at them byte by byte.  An unmasked band: "holds a byte 64" -- three instructions a dword
    // (x ^ 0x40..., the zero byte test (t - 0x01...) & ~t, which may also flag the byte above a zero byte: a filter), one unit in five
    // passes, a wave's queued units still fit one round of lanes as good as always; that the byte in front reads 10...... is left
    // to the exact test (in the filter it cost three times as much as it saved).  A masked band: a byte 1 ... 64 behind a byte
    // 10...... (the byte in front of a wave's first unit belongs to another wave: taken for 10......).
    {
      bool has;
      if (!OFFS)
      {
        u32 acc = 0u;
        { const u32 t = x[k].x ^ 0x40404040u; acc |= (t - 0x01010101u) & ~t; }
        { const u32 t = x[k].y ^ 0x40404040u; acc |= (t - 0x01010101u) & ~t; }
        { const u32 t = x[k].z ^ 0x40404040u; acc |= (t - 0x01010101u) & ~t; }
        { const u32 t = x[k].w ^ 0x40404040u; acc |= (t - 0x01010101u) & ~t; }
        if (RAG && ragC2 != 0u)
        {
          const u32 c4 = ragC2 * 0x01010101u;
          { const u32 t = x[k].x ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].y ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].z ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].w ^ c4; acc |= (t - 0x01010101u) & ~t; }
        }
        if (RAG && ragC3 != 0u)
        {
          const u32 c4 = ragC3 * 0x01010101u;
          { const u32 t = x[k].x ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].y ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].z ^ c4; acc |= (t - 0x01010101u) & ~t; }
          { const u32 t = x[k].w ^ c4; acc |= (t - 0x01010101u) & ~t; }
        }
        has = (acc & 0x80808080u) != 0u;
      }
      else
      {
        u32 pv = dppMov<kDppWaveShr1>(x[k].w);
        if (lane == 0) pv = 0x80000000u;
        const u32 z = countByteMaybe<OFFS>(x[k].x, pv) & countByteMaybe<OFFS>(x[k].y, x[k].x) & countByteMaybe<OFFS>(x[k].z, x[k].y) & countByteMaybe<OFFS>(x[k].w, x[k].z);
        has = (z & 0x80808080u) != 0x80808080u;
      }
      has = has && i < G::kScanUnits && 16u * i < blobRel;    // (what lies behind the blob in the caller's buffer is not looked at)
      const u64 bal = __builtin_amdgcn_ballot_w64(has);
      const u32 slot = nMine + (u32)__popcll(bal & laneMaskLt());
      if (has && slot < kQueueSeg) s_queue[(u32)w * kQueueSeg + slot] = (u16)i;
      nMine += (u32)__popcll(bal);
    }
  }
  {
    // (a lane holds 5 units, A < 2^23 and B < 2^54 per lane; the sums are wanted mod 65535 and 2^16 = 1 there: folded to 32 bits
    // first -- a wave's sum of the folds stays below 2^24 -- the reductions are DPP adds, and nobody divides)
    const u32 A = waveSum(fold65535(fA)), B = waveSum(fold65535(fB));
    if (lane == 0) { S.fa[w] = A; S.fb[w] = B; S.qn[w] = nMine; }
  }
  __syncthreads();
  TRACES(1);
  DEC_EXIT(1);
  if (!OFFS && !S.hp.ok) return;    // (not a band the streaming kernels take: the header says so in full only)
  if (gridShort)
  {
    if (threadIdx.x == 0) raiseFlag(b, 2);
    return;
  }
  if (!OFFS && threadIdx.x == 0)
  {
    // this workgroup's checksum terms: one atomic nobody waits for (the launch's last workgroup folds the accumulators)
    u32 A32 = 0, B32 = 0;
#pragma unroll
    for (u32 k = 0; k < kWaves; k++) { A32 += (u32)S.fa[k]; B32 += (u32)S.fb[k]; }
    const u64 A = fold65535(A32), B = fold65535(B32);    // (congruent, below 2^17: a group's 64 terms fit the accumulator's 24-bit fields)
    if (wg == 0u) drainVmem();    // (the band's parameters have arrived)
    __hip_atomic_fetch_add(b.wgAcc + wg / kOneGroup, A | (B << 24) | (1ull << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- candidates: lane = unit with hits.  A bit-stuffed block reads: flag byte (bits 0-1 == 1, bits 6-7 the type of the
  // offset, bit 2 clear from codec 5 on), the offset in that type, the bits byte 10?nnnnn (bit 5: look-up table), the count 64,
  // [table size + 1, table,] payload (Lerc2.cpp:1961-2021, BitStuffer2.cpp:35-153).  The count byte stands 2 + (bytes of the
  // offset) behind the flag byte: each offset type is tried.  The twelve bytes in front of the count byte's successor come out
  // of four LDS words, shifted so that the byte 10 in front of the count byte is byte 0.
  {
    // (a wave takes the units it queued itself: no counter to share -- but the bytes around them are other waves')
    const u32 nQ = S.qn[w];
    if (nQ > kQueueSeg && lane == 0) S.over = 1u;
    // what stands around the count byte at q: every offset type's flag byte, the block's length, the flag byte behind it -- and its START / END bits
    auto candidate = [&](u32 q)
    {
      const u32 wi = (q + 6u) >> 2, sh = 8u * ((q + 6u) & 3u);             // (word index into inAll: 16 bytes of zeros in front)
      const u32 w0 = S.inAll[wi], w1 = S.inAll[wi + 1], w2 = S.inAll[wi + 2], w3 = S.inAll[wi + 3];
      const u32 a0 = __builtin_amdgcn_alignbit(w1, w0, sh), a1 = __builtin_amdgcn_alignbit(w2, w1, sh), a2 = __builtin_amdgcn_alignbit(w3, w2, sh);
      const u32 tB = (a2 >> 8) & 0xFFu, nb = tB & 31u, lut = (tB >> 5) & 1u;
      const u32 nLut = ((a2 >> 24) - 1u) & 0xFFu;                          // valid: 1 ... 254
      const bool okLut = (lut == 0u) | ((nLut - 1u) < 254u);
      const u32 cnt = (OFFS || RAG) ? ((a2 >> 16) & 0xFFu) : 64u;              // elements of the block: the count byte (RAG: 64, or the last block row's)
      const u32 payload = lut ? 1u + ((nLut * nb + 7u) >> 3) + ((cnt * (u32)bitLen(nLut) + 7u) >> 3) : ((cnt * nb + 7u) >> 3);
#pragma unroll
      for (u32 tc = 0; tc < 4; tc++)
      {
        const u32 offB = (offBytesTable<DT>() >> (4u * tc)) & 15u;
        if (offB == 0u) continue;
        const u32 m = 8u - offB;                                           // the flag byte is byte m of the twelve
        const u32 flag = ((m < 4u ? a0 : m < 8u ? a1 : a2) >> (8u * (m & 3u))) & 0xFFu;
        const u32 len = 3u + offB + payload;
        const bool ok = (flag & 3u) == 1u && (flag >> 6) == tc && !(v5 && (flag & 4u)) && okLut && len <= ((OFFS || RAG) ? 1u + cnt * G::TB : RAW) && q >= 2u + offB;    // (no longer than the raw form of so many values)
        const u32 p = q - 2u - offB, e = p + len;
        // (what stands where the candidate ends has to read like the flag byte of the block behind it: the column signature goes
        // on, by a step or none, or begins again with a block row -- four bytes in five of anything else do not)
        bool follows = true;
        if (ok && e < blobRel && e < G::kBytes)
        {
          const u32 nf = (s_in[e >> 2] >> (8u * (e & 3u))) & 0xFFu;
          follows = sigOk((flag >> 2) & pattern, (nf >> 2) & pattern, pattern) && !(v5 && (nf & 4u));
        }
        if (ok && follows && p >= dataRel && p < pieceEndRel && e <= blobRel)
        {
          atomicOr(&s_sb[p >> 5], 1u << (p & 31u));
          atomicOr(&s_end[e >> 5], 1u << (e & 31u));                      // (e < pieceEndRel + W: inside the bitmap)
        }
      }
    };
    // the scan, second half: which bytes of a unit -- exactly: 64 (a masked band: 1 ... 64) behind 10?nnnnn, n != 0; a bit per byte,
    // bit k: byte 4 (k & 3) + (k >> 2)
    auto unitHits = [&](u32 unit) -> u32
    {
      const uint4 xu = *reinterpret_cast<const uint4*>(&s_in[4u * unit]);
      const u32 pvu = S.inAll[4u * unit + 3u];                               // (the dword in front of the unit; in front of the staged bytes: 0)
      u32 m0 = countByteHits<OFFS>(xu.x, pvu), m1 = countByteHits<OFFS>(xu.y, xu.x), m2 = countByteHits<OFFS>(xu.z, xu.y), m3 = countByteHits<OFFS>(xu.w, xu.z);
      if (RAG && ragC2 != 0u)
      {
        const u32 c4 = ragC2 * 0x01010101u;
        m0 |= countByteHitsOf(xu.x, pvu, c4); m1 |= countByteHitsOf(xu.y, xu.x, c4); m2 |= countByteHitsOf(xu.z, xu.y, c4); m3 |= countByteHitsOf(xu.w, xu.z, c4);
      }
      if (RAG && ragC3 != 0u)
      {
        const u32 c4 = ragC3 * 0x01010101u;
        m0 |= countByteHitsOf(xu.x, pvu, c4); m1 |= countByteHitsOf(xu.y, xu.x, c4); m2 |= countByteHitsOf(xu.z, xu.y, c4); m3 |= countByteHitsOf(xu.w, xu.z, c4);
      }
      const u32 zb = (m0 >> 7) | ((m1 >> 7) << 1) | ((m2 >> 7) << 2) | ((m3 >> 7) << 3);
      const u32 tb = zb | (zb >> 4);
      return (tb & 0xFFu) | ((tb >> 8) & 0xFF00u);
    };
    if (!OFFS)
    {
      // (a unit with a hit has one, as good as always: lane = unit)
      for (u32 hq = (u32)lane; hq < min(nQ, kQueueSeg); hq += 64u)
      {
        const u32 unit = (u32)s_queue[(u32)w * kQueueSeg + hq];
        u32 hits = unitHits(unit);
        while (hits)
        {
          const u32 k = (u32)__ffs((int)hits) - 1u;
          hits &= hits - 1u;
          candidate(16u * unit + 4u * (k & 3u) + (k >> 2));
        }
      }
    }
    else
    {
      // A masked band: any byte 1 ... 64 behind 10?nnnnn is a hit -- two thousand a piece, two in three of them no block, a unit holds up
      // to eight.  Lane = unit would have the wave go round as often as its busiest lane has hits: the hits are laid out first, one
      // behind the other (the wave's stretch of the queue, read by then, holds them), and taken 64 at a time, lane = hit.
      u32 hm[kRounds], un[kRounds];
#pragma unroll
      for (int j = 0; j < kRounds; j++)
      {
        const u32 hq = (u32)lane + 64u * (u32)j;
        hm[j] = 0u; un[j] = 0u;
        if (hq < min(nQ, kQueueSeg)) { un[j] = (u32)s_queue[(u32)w * kQueueSeg + hq]; hm[j] = unitHits(un[j]); }
      }
      __builtin_amdgcn_wave_barrier();    // (every lane has read its queue entries: the stretch is free)
      u16* const buf = &s_queue[(u32)w * kQueueSeg];
      static_assert(kQueueSeg >= 128u, "room for a wave's worth of hits and what is left of the last");
      u32 fill = 0u;
#pragma unroll
      for (int j = 0; j < kRounds; j++)
      {
        u32 m = hm[j];
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull)
        {
          const bool has = m != 0u;
          u32 q = 0u;
          if (has) { const u32 k = (u32)__ffs((int)m) - 1u; m &= m - 1u; q = 16u * un[j] + 4u * (k & 3u) + (k >> 2); }
          const u64 bal = __builtin_amdgcn_ballot_w64(has);
          if (has) buf[fill + (u32)__popcll(bal & laneMaskLt())] = (u16)q;
          fill += (u32)__popcll(bal);
          __builtin_amdgcn_wave_barrier();
          if (fill >= 64u)
          {
            candidate((u32)buf[lane]);
            const u32 rest = fill - 64u;
            const u32 mv = (u32)lane < rest ? (u32)buf[64u + (u32)lane] : 0u;
            __builtin_amdgcn_wave_barrier();
            if ((u32)lane < rest) buf[lane] = (u16)mv;
            __builtin_amdgcn_wave_barrier();
            fill = rest;
          }
        }
      }
      if ((u32)lane < fill) candidate((u32)buf[lane]);
    }
    if (threadIdx.x == 0 && dataRel >= PRE && dataRel < pieceEndRel)    // the stream's first block, whatever it is
    {
      atomicOr(&s_sb[dataRel >> 5], 1u << (dataRel & 31u));
      atomicOr(&s_end[dataRel >> 5], 1u << (dataRel & 31u));
    }
  }
  __syncthreads();
  TRACES(2);
  DEC_EXIT(2);

  // ---- survivors of the piece's own bytes: a thread's two bitmap words; their list by popcounts and one scan
  const u32 myWord = G::kOwnWord0 + 2u * threadIdx.x;
  { s_sb[myWord] &= s_end[myWord]; s_sb[myWord + 1u] &= s_end[myWord + 1u]; }
  // (the bytes in front of the piece's own: their survivors say where the piece's first block begins -- the ANCHOR, below)
  static_assert(G::kOwnWord0 <= NT, "a thread per bitmap word in front of the own bytes");
  if (threadIdx.x < G::kOwnWord0) s_sb[threadIdx.x] &= s_end[threadIdx.x];
  if (OFFS)
  {
    // (a masked band: END is done with; the bitmap takes where the SURVIVORS end -- the seeds of the flood below.  A thread clears
    // the words it alone has read, and those behind the piece's own, which nobody has)
    s_end[myWord] = 0u; s_end[myWord + 1u] = 0u;
    if (threadIdx.x < G::kOwnWord0) s_end[threadIdx.x] = 0u;
    constexpr u32 kBehind = G::kOwnWord0 + 2u * NT;
    if (kBehind + threadIdx.x < 4u * G::kMapVecs) s_end[kBehind + threadIdx.x] = 0u;
  }
  // EARLY: the piece's count leaves as soon as the survivors are counted -- before the list is written, before anybody has looked at a
  // header: the pieces behind need it for their blocks' places and wait for the slowest of up to 63 pieces in front; what the list
  // and the check below cost a piece is what every piece behind it waits less.  Where the piece's last block ends is not known
  // yet (0xFFFE; only the piece right behind wants it, and only for its verdict: it looks again at the very end).  A piece whose
  // check or mending comes to ANOTHER count has told the pieces behind a wrong one: it says so (flag 1) and the band is decoded
  // once more without early counts.
  const bool early = !OFFS && LERC_SCAN_EARLY != 0 && b.scanEarly != 0u;
  auto buildList = [&](bool first)
  {
    const u32 s0 = s_sb[myWord], s1 = s_sb[myWord + 1u];
    const u32 c = (u32)__popc(s0) + (u32)__popc(s1);
    const u32 inc = waveInclusiveScan(c);
    if (lane == 63) S.wsum[w] = inc;
    __syncthreads();                                  // (and: the queue, which the list lies on, has been read by everybody)
    u32 idx = inc - c;
    for (int k = 0; k < w; k++) idx += S.wsum[k];
    if (first && early && threadIdx.x == NT - 1u)
    {
      const u32 n = min(min(idx + c, kListCap), 0xFFFFu);
      S.earlyCount = n;
      publish64(b.wgCell + wg, tag | (0xFFFEull << 16) | (u64)n);
    }
    const u32 pos0 = PRE + 64u * threadIdx.x;
    u32 m = s0;
    while (m) { const u32 bt = (u32)__ffs((int)m) - 1u; m &= m - 1u; if (idx < kListCap) s_list[idx] = (u16)(pos0 + bt); idx++; }
    m = s1;
    while (m) { const u32 bt = (u32)__ffs((int)m) - 1u; m &= m - 1u; if (idx < kListCap) s_list[idx] = (u16)(pos0 + 32u + bt); idx++; }
    if (threadIdx.x == NT - 1u) { S.nEnt = min(idx, kListCap); if (idx > kListCap) S.over = 1u; }
    __syncthreads();
  };
  buildList(true);
  TRACES(3);
  DEC_EXIT(3);

  // ---- every block's header in full, lane = block: length, mode, bits, offset (ReadTile's and BitStuffer2::Decode's checks), and
  // "the blocks tile the stream": a block ends where the next one of the list begins, the last one behind the piece (the
  // blob's last piece: with the blob).  What the pixel loop wants to know of the first R blocks is kept.
  FastDecodeParams hp = S.hp;
  if (OFFS) { memset(&hp, 0, sizeof(hp)); hp.version = job.version; hp.nCols = (u32)nCols; }
  const struct { int nCols, version; double invScale, zMaxHdr; } p = { (int)hp.nCols, (int)hp.version, hp.invScale, hp.zMaxHdr };
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  auto& s_offs = S.u.x.offs; auto& s_code = S.u.x.code; auto& s_at = S.u.x.at; auto& s_dims = S.u.x.dims;
  // the block at list entry f: its length, or 0 if it is none; keep: what the pixel loop needs goes to slot t
  auto parseBlock = [&](u32 pos, bool keep, u32 t) -> u32
  {
    u32 h0, h1, h2;
    ldsHeader<DT>(s_in, pos, h0, h1, h2);
    u32 nEl = 64u;
    if (OFFS)
    {
      // a masked band's block holds 1 ... 64 pixels, and its count byte says how many (the decode kernel, which knows the block's
      // place, checks the number); a raw block's length hangs on it without saying it: such a stream goes the long way
      const u32 offB = (offBytesTable<DT>() >> ((h0 >> 4) & 12u)) & 15u;
      u32 tt = (u32)((((u64)h1 << 32) | h0) >> ((8u + 8u * offB) & 63u));
      if (DT == DT_Double && offB == 8u) tt = h2 >> 8;
      const u32 mode = h0 & 3u;
      nEl = mode == 1u ? ((tt >> 8) & 0xFFu) : 64u;       // (constant blocks have no count, and their length does not hang on it)
      if (mode == 0u || (nEl - 1u) >= 64u) nEl = 0u;
    }
    if (RAG)
    {
      // the count byte of a bit-stuffed block says how many pixels it holds: a whole block's 64, or what an edge block may hold (which of
      // them it has to be is checked when the block's place is known); raw blocks are taken for whole ones (a raw EDGE block is
      // shorter: such a stream does not tile and goes down a tier), constant blocks have no count
      const u32 offB = (offBytesTable<DT>() >> ((h0 >> 4) & 12u)) & 15u;
      u32 tt = (u32)((((u64)h1 << 32) | h0) >> ((8u + 8u * offB) & 63u));
      if (DT == DT_Double && offB == 8u) tt = h2 >> 8;
      if ((h0 & 3u) == 1u)
      {
        const u32 c = (tt >> 8) & 0xFFu;
        nEl = (c == 64u || (ragWl != 0u && c == 8u * ragWl) || (ragHl != 0u && c == 8u * ragHl) || (ragWl != 0u && ragHl != 0u && c == ragWl * ragHl)) ? c : 0u;
      }
    }
    u32 code = ((OFFS || RAG) && nEl == 0u) ? 0u : parseCode<DT>(h0, h1, h2, p.version, nEl);
    if (pos + codeLen(code) > blobRel) code = 0u;
    const u32 len = codeLen(code);
    if (!OFFS && keep)
    {
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
      // in one word: where the payload begins (16: byte among the staged ones; the first raw value of a raw block), bits per
      // value (5) << 16, mode (2) << 21, look-up table << 23, "plain" << 24 -- bit-stuffed without a table, a lane's V values
      // inside 64 bits, and not even the largest value nb bits can hold reaches the header's zMax, so the pixels need no clamp
      // -- and bit 31 (0: no such block)
      u32 word = 0u;
      if (code)
      {
        const u32 nb = codeBits(code), lutB = codeLut(code);
        bool plainB = false;
        if (mode == 1)
        {
          const u32 qTop = nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
          const bool below = (DT >= DT_Float) ? (offset + (double)qTop * p.invScale < p.zMaxHdr)
                                              : ((i64)offset + (i64)qTop * (i64)p.invScale < (i64)p.zMaxHdr);
          plainB = below && !lutB && (u32)V * nb <= 64u && (!RAG || nEl == 64u);
        }
        const u32 pay = pos + ((mode == 1u) ? 3u + codeOffBytes(code) + lutB : 1u);
        word = (pay & 0xFFFFu) | (nb << 16) | (mode << 21) | (lutB << 23) | ((plainB ? 1u : 0u) << 24) | 0x80000000u;
        if (RAG) word |= ((nEl - 1u) & 63u) << 25;    // (how many pixels the block says it holds, until its place is known)
      }
      // (integer types: the offset as the integer the pixel loop adds to -- converted here, once a block, not once a lane and block row)
      if (DT < DT_Float) { const i64 oi = (i64)offset; double od; memcpy(&od, &oi, 8); s_offs[t] = od; }
      else s_offs[t] = offset;
      s_code[t] = word;
      s_at[t] = (h0 >> 2) & pattern;     // (the signature, until the block's place is known)
    }
    return len;
  };
  // The ANCHOR: where the piece's first block begins, told by the survivors in front of the piece's own bytes -- whatever the own bytes'
  // first survivors say (a false one there may tile with the true ones behind it: nothing in the piece would tell).  The last
  // two survivors in front: if the one ends where the other begins they are blocks, as good as certainly, and the piece's first
  // block begins where the last one ends; if the last one lies inside the other, where that one ends.  0: nothing to go by.
  auto anchorEnd = [&]() -> u32
  {
    if (!OFFS)
    {
      // (the unmasked bands' kernel: the last two survivors in front)
      u32 a1 = 0xFFFFu, a2 = 0xFFFFu;
      for (u32 wd = G::kOwnWord0; wd-- > 0u && a2 == 0xFFFFu; )
      {
        u32 v = s_sb[wd];
        while (v && a2 == 0xFFFFu)
        {
          const u32 bit = 31u - (u32)__clz((int)v);
          v &= ~(1u << bit);
          if (a1 == 0xFFFFu) a1 = 32u * wd + bit; else a2 = 32u * wd + bit;
        }
      }
      if (a1 == 0xFFFFu) return 0u;
      const u32 l1 = parseBlock(a1, false, 0u);
      u32 t0 = l1 ? a1 + l1 : 0u;
      if (a2 != 0xFFFFu)
      {
        const u32 l2 = parseBlock(a2, false, 0u);
        if (l2 == 0u || a2 + l2 < a1) t0 = 0u;          // (not one behind the other: one of them is no block)
        else if (a2 + l2 > a1) t0 = a2 + l2;            // (the last one lies inside the one in front)
      }
      return t0 >= PRE ? t0 : 0u;
    }
    else
    {
      u32 a1 = 0xFFFFu, a2 = 0xFFFFu, a3 = 0xFFFFu;    // the last survivors in front of the own bytes, the last one first
      u32 nA = 0u;
      for (u32 wd = G::kOwnWord0; wd-- > 0u && nA < 3u; )
      {
        u32 v = s_sb[wd];
        while (v && nA < 3u)
        {
          const u32 bit = 31u - (u32)__clz((int)v);
          v &= ~(1u << bit);
          const u32 pos = 32u * wd + bit;
          if (nA == 0u) a1 = pos; else if (nA == 1u) a2 = pos; else a3 = pos;
          nA++;
        }
      }
      if (nA == 0u) return 0u;
      const u32 l1 = parseBlock(a1, false, 0u);
      u32 t0 = l1 ? a1 + l1 : 0u;
      if (nA >= 2u)
      {
        const u32 l2 = parseBlock(a2, false, 0u);
        if (l2 == 0u || a2 + l2 < a1) t0 = 0u;          // (not one behind the other: one of them is no block)
        else if (a2 + l2 > a1)
        {
          // the two overlap: one of them is no block.  The survivor in front of both says which -- the one that begins where IT ends
          t0 = 0u;
          if (nA >= 3u)
          {
            const u32 l3 = parseBlock(a3, false, 0u);
            if (l3 != 0u && a3 + l3 == a1) t0 = l1 ? a1 + l1 : 0u;
            else if (l3 != 0u && a3 + l3 == a2) t0 = a2 + l2;
          }
        }
      }
      // (a masked band: blocks the scan does not see may lie between there and the piece's own bytes -- the mending walks them)
      return (OFFS || t0 >= PRE) ? t0 : 0u;
    }
  };
  auto tilePass = [&](u32 pass, bool final)
  {
    const u32 nEnt = S.nEnt;
    for (u32 f = threadIdx.x; f < nEnt; f += NT)
    {
      const u32 pos = (u32)s_list[f];
      u32 len = parseBlock(pos, f < R, f);
      const bool last = f + 1u == nEnt;
      const u32 nxt = last ? 0u : (u32)s_list[f + 1u];
      if (OFFS && final && len == 0u)
      {
        // (a raw block of a masked band, whose length the stream does not say: the mending has found where the next block begins)
        const u32 b0 = (s_in[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu;
        const u32 e = last ? S.mendExit : nxt;
        if ((b0 & 3u) == 0u && (b0 >> 6) == 0u && !(v5 && (b0 & 4u)) && e > pos + 1u && (e - pos - 1u) % G::TB == 0u && (e - pos - 1u) / G::TB <= 64u && e <= blobRel) len = e - pos;
      }
      if (RAG && final)
      {
        // (a raw EDGE block is shorter than parseBlock takes it to be: it ends where the next entry begins -- the mending has looked)
        const u32 b0 = (s_in[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu;
        const u32 e = last ? S.mendExit : nxt;
        if ((b0 & 3u) == 0u && (b0 >> 6) == 0u && !(v5 && (b0 & 4u)) && e > pos && e != pos + len && e <= blobRel && ragRawOk(e - pos)) len = e - pos;
      }
      const u32 ext = pos + len;
      const bool ok = len != 0u && (last ? (lastPiece ? ext == blobRel : ext >= pieceEndRel) : ext == nxt);
      if (!ok) { const u32 at = atomicAdd(&S.nBad[pass], 1u); if (at < kScanBadCap) S.badIdx[at] = (u16)f; }
      if (OFFS && !final && !ok && len != 0u)
      {
        // a block that ends where the entry after next begins (or the one after that): what lies between is no block -- false
        // survivors, struck here by the lane that sees it (the list is built again before anybody trusts it)
        for (u32 d = 2u; d <= 3u && f + d < nEnt; d++)
          if ((u32)s_list[f + d] == ext)
          {
            for (u32 j = 1u; j < d; j++) { const u32 q = (u32)s_list[f + j]; atomicAnd(&s_sb[q >> 5], ~(1u << (q & 31u))); }
            atomicAdd(&S.nStruck[pass], 1u);
            break;
          }
      }
      if (OFFS && pass == 0u && len != 0u && ext < G::kBytes) atomicOr(&s_end[ext >> 5], 1u << (ext & 31u));    // (a seed of the flood)
      if (last) S.exitRel = ext;
      if (f == 0u && (OFFS || final))    // (entries in front of where the anchor ends are none; a masked band: blocks between there and the first entry are missing)
      {
        if (!final) S.t0 = anchorEnd();
        const u32 t0 = S.t0;
        if (t0 > pos || ((OFFS || final) && t0 != 0u && t0 < pos)) { if (!final) S.frontBad = 1u; else atomicAdd(&S.nBad[pass], 1u); }
      }
    }
    // An unmasked band's first check: the anchor -- one thread going backwards through the survivors in front of the piece's own bytes, two
    // headers parsed one after the other -- is the LAST wave's, which has no list entry to check unless the piece holds more than
    // NT - 64 blocks: beside the other waves' headers instead of behind the first wave's (0.5 us of every piece's life)
    if (!OFFS && !final && threadIdx.x == NT - 64u && nEnt != 0u)
    {
      const u32 t0 = anchorEnd();
      S.t0 = t0;
      if (t0 > (u32)s_list[0]) S.frontBad = 1u;
    }
    if (OFFS && nEnt == 0u && threadIdx.x == 0)    // (no entry: all of the piece is missing, if the bytes in front say where it begins)
    {
      if (!final) S.t0 = anchorEnd();
      if (S.t0 != 0u && S.t0 < pieceEndRel && S.t0 < blobRel) { if (!final) S.frontBad = 1u; else atomicAdd(&S.nBad[pass], 1u); }
    }
    if (OFFS && pass == 0u && threadIdx.x < G::kOwnWord0)
    {
      // the survivors in front of the piece's own bytes: where they end (a run of one-byte blocks that begins there may reach the own bytes)
      u32 v = s_sb[threadIdx.x];
      while (v)
      {
        const u32 bit = (u32)__ffs((int)v) - 1u; v &= v - 1u;
        const u32 pos = 32u * threadIdx.x + bit, len = parseBlock(pos, false, 0u), ext = pos + len;
        if (len != 0u && ext < G::kBytes) atomicOr(&s_end[ext >> 5], 1u << (ext & 31u));
      }
    }
    __syncthreads();
  };
  tilePass(0u, false);
  TRACES(4);
  DEC_EXIT(4);

  // ---- A masked band has RUNS of one-byte blocks (pixels all invalid, Lerc2.h:422) which the scan cannot see, and behind each run a
  // block without an END.  They are found by a FLOOD over the bitmaps: a byte that reads like such a block (mode 2; bit 2 clear from
  // codec 5 on) IS one if a block ends in front of it -- a survivor (the seeds: tilePass has left the survivors' ends in END's
  // place) or the one-byte block in front of it, as long as the column signature goes on from byte to byte -- and what begins behind
  // the run's last byte is a block as well.  "Goes on while the bytes allow" is the carry chain of an addition: a thread adds
  // its 64 positions' seeds to their go-on bits, the carries between threads and waves come from two ballots (generate /
  // propagate), and the bits a carry went INTO are the run (a & the go-on bits) and the block behind it (the one bit where it stopped).
  u32 cp = 0u;    // the pass whose broken links stand
  if (OFFS && dataRel < pieceEndRel &&
      (S.nBad[0] != 0u || S.frontBad != 0u || S.nEnt == 0u || (S.t0 == 0u && !(dataRel >= PRE))))
  {
    auto rangeMask = [&](u32 base, u32 lo, u32 hi) -> u32 { return scanRangeMask(base, lo, hi); };
    // bitmap word wd: which of its 32 bytes read like a one-byte block (m2w), and which of those go on from the byte in front (contw)
    auto wordMasks = [&](u32 wd, u32& m2w, u32& contw)
    {
      // (a one-byte block as the reference writes it: mode 2, the signature, nothing in bits 6 - 7, Lerc2.cpp:1955-1962; a writer that
      // leaves something there is followed by the mending, or the general discovery)
      const u32 mBits = v5 ? 0xC7C7C7C7u : 0xC3C3C3C3u, pat4 = pattern * 0x01010101u, step4 = v5 ? 0x02020202u : 0x01010101u;
      u32 prev = S.inAll[3u + 8u * wd];    // (the word in front; in front of the staged bytes: 0)
      m2w = 0u; contw = 0u;
#pragma unroll
      for (u32 j = 0; j < 8u; j++)
      {
        const u32 x = s_in[8u * wd + j];
        const u32 px = __builtin_amdgcn_alignbit(x, prev, 24);                    // the bytes in front of x's
        // (0x80 in the bytes of y that are zero; y's bytes are below 0x80)
        auto zeroBytes = [](u32 y) { return ~(y + 0x7F7F7F7Fu) & 0x80808080u; };
        // (0x80 in the bytes of y that are zero, any y)
        auto zeroBytesAny = [](u32 y) { return ~(((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y) & 0x80808080u; };
        const u32 a = zeroBytesAny((x ^ 0x02020202u) & mBits), ap = zeroBytesAny((px ^ 0x02020202u) & mBits);
        const u32 sg = (x >> 2) & pat4, sp = (px >> 2) & pat4;
        const u32 any = zeroBytes(sg ^ sp) | zeroBytes(sg ^ ((sp + step4) & pat4)) | zeroBytes(sg);    // sigOk, four bytes at a time
        const u32 c = a & ap & any;
        m2w |= ((((a >> 7) * 0x01020408u) >> 24) & 15u) << (4u * j);
        contw |= ((((c >> 7) * 0x01020408u) >> 24) & 15u) << (4u * j);
        prev = x;
      }
      m2w &= rangeMask(32u * wd, dataRel, blobRel);
      contw &= m2w;    // (in range; the stream's first block goes on from nothing)
      if (dataRel >= 32u * wd && dataRel < 32u * wd + 32u) contw &= ~(1u << (dataRel - 32u * wd));
    };
    u32 m2a, ca, m2b, cb;
    wordMasks(myWord, m2a, ca); wordMasks(myWord + 1u, m2b, cb);
    const u64 X = (u64)ca | ((u64)cb << 32), M2 = (u64)m2a | ((u64)m2b << 32);
    const u64 seed = ((u64)s_end[myWord] | ((u64)s_end[myWord + 1u] << 32)) & M2;
    const u64 a = X | seed;
    const u64 s0 = a + seed;
    const bool gen = s0 < a, prop = !gen && s0 == ~0ull;
    if (w == 0)
    {
      // the bytes in front of the piece's own, a bitmap word a lane.  If ALL of them read like a run and no block is known to end among
      // them, the run began further in front: its first byte here is taken for a block (the piece in front, which says where its
      // last block ends, will confirm or refute it)
      u32 pm2 = 0u, pc = 0u, pseed = 0u;
      const bool mine = (u32)lane < G::kOwnWord0;
      if (mine) { wordMasks((u32)lane, pm2, pc); pseed = s_end[lane] & pm2; }
      const bool likeRun = !mine || (pseed == 0u && (pc | (lane == 0 ? 1u : 0u)) == 0xFFFFFFFFu && (pm2 & 1u) != 0u);
      if (__all(likeRun) && dataRel == 0u && lane == 0) pseed |= 1u;
      const u32 a32 = pc | pseed;
      const u64 s32 = (u64)a32 + (u64)pseed;
      const bool g32 = mine && (s32 >> 32) != 0u, p32 = mine && !g32 && (u32)s32 == 0xFFFFFFFFu;
      const u64 Gm = __builtin_amdgcn_ballot_w64(g32), Am = __builtin_amdgcn_ballot_w64(p32) | Gm;
      const u64 into = (Am + Gm) ^ Am ^ Gm;                                        // lanes a carry goes into
      const u32 cin = (u32)(into >> lane) & 1u;
      const u32 cinb = (u32)((u64)a32 + (u64)pseed + cin) ^ a32 ^ pseed;           // bits a carry goes into
      const u32 F = pseed | (cinb & pc), behind = cinb & ~F & rangeMask(32u * (u32)lane, dataRel, blobRel);
      if (mine) s_sb[lane] |= F | behind;
      if (lane == 0) S.preCarry = (u32)(into >> G::kOwnWord0) & 1u;
    }
    const u64 Gm = __builtin_amdgcn_ballot_w64(gen), Am = __builtin_amdgcn_ballot_w64(prop) | Gm;
    {
      const u64 t = Am + Gm;
      const u32 co0 = t < Am ? 1u : 0u, co1 = (co0 || t + 1ull == 0ull) ? 1u : 0u;    // the wave's carry out without / with a carry in
      if (lane == 0) S.fco[w] = co0 | (co1 << 1);
    }
    __syncthreads();
    u32 cw = S.preCarry;
    for (int k = 0; k < w; k++) cw = (S.fco[k] >> cw) & 1u;
    const u64 into = (Am + Gm + (u64)cw) ^ Am ^ Gm;
    const u64 cin = (into >> lane) & 1ull;
    const u64 cinb = (a + seed + cin) ^ a ^ seed;
    const u64 F = seed | (cinb & X);
    const u64 lim = (u64)rangeMask(32u * myWord, dataRel, blobRel) | ((u64)rangeMask(32u * (myWord + 1u), dataRel, blobRel) << 32);
    const u64 FT = F | (cinb & ~F & lim);
    s_sb[myWord] |= (u32)FT; s_sb[myWord + 1u] |= (u32)(FT >> 32);
#ifdef LERC_PROBE
    atomicAdd(&S.dbg[0], (u32)__popcll(M2)); atomicAdd(&S.dbg[1], (u32)__popcll(X)); atomicAdd(&S.dbg[2], (u32)__popcll(seed)); atomicAdd(&S.dbg[3], (u32)__popcll(FT));
#endif
    if (threadIdx.x == 0) S.frontBad = 0u;    // (everybody has read it; the check below says it again)
    __syncthreads();
    buildList(false);
    cp = 1u;
    tilePass(1u, false);
  }

  // (a masked band: the lanes have struck false survivors, and links are still broken -- maybe theirs: look again)
  if (OFFS && S.nBad[cp] != 0u && S.nStruck[cp] != 0u)
  {
    buildList(false);
    cp++;
    tilePass(cp, false);
  }

  // ---- a list that does not tile is mended by one thread (see the head of the file): false survivors struck, gaps walked.
  // Nothing is written until the whole list has been gone through; then the survivors' bitmap is corrected and the list built
  // again from it.
  TRACEV(7, cp | (S.frontBad << 4) | (S.nBad[cp] << 8) | (S.nEnt << 16));
#ifdef LERC_PROBE
  TRACEV(6, (u64)S.dbg[0] | ((u64)S.dbg[1] << 16) | ((u64)S.dbg[2] << 32) | ((u64)S.dbg[3] << 48));
#endif
  // An unmasked band's piece is mended by its first WAVE: the walk through a gap takes RUNS of constant / all-zero blocks -- a flat stretch
  // of the raster: a lake, the sea, a fill value; blocks of 1 ... 9 bytes the scan does not see, hundreds on end -- 64 blocks a step
  // (lane i looks at the byte i block lengths on: the same flag byte but for the column signature is the same block again), and where
  // the bytes in front hold no anchor (they lie in such a run) the piece's first block begins where the piece in front says its last
  // one ends: that piece has said so before it waits for anybody (a chain only through pieces that begin inside a run).
  const bool firstOfStream = dataRel >= PRE && dataRel < pieceEndRel;
  bool mendIt = S.nBad[cp] != 0u || S.frontBad != 0u;
  if (!OFFS && !firstOfStream && dataRel < PRE) mendIt = mendIt || S.t0 == 0u || S.nEnt == 0u || S.t0 < (u32)s_list[0];    // (no anchor; blocks missing in front of the first survivor)
  if (mendIt)
  {
    if (!OFFS)
    {
      if (w == 0)
      {
        const u32 n = S.nEnt, nBad = S.nBad[cp];
        const u32 endTarget = lastPiece ? blobRel : pieceEndRel;
        u32 t0 = firstOfStream ? dataRel : S.t0;
        if (t0 == 0u && n == 0u && !firstOfStream) t0 = anchorEnd();    // (nobody has looked: the check looks for an anchor beside its first entry)
        if (t0 == 0u && !firstOfStream && wg != 0u)
        {
          // no anchor: the piece in front knows (its cell's upper half, once its own list is checked and mended)
          const u64* pc = b.wgCell + (wg - 1u);
          u64 c = observe64(pc);
          for (u32 spin = 0; ((u32)(c >> 32) != epoch || (((u32)c >> 16) & 0xFFFFu) == 0xFFFEu) && spin < b.spinLimit; spin++)
          {
            __builtin_amdgcn_s_sleep(LERC_SCAN_SLEEP);
            c = observe64(pc);
          }
          const u32 pe = ((u32)c >> 16) & 0xFFFFu;
          if ((u32)(c >> 32) == epoch && pe < 0xFFFEu) t0 = PRE + pe;
        }
        bool fail = t0 == 0u || S.over != 0u || nBad > kScanBadCap;
        bool good = false;
        u32 nFalse = 0u, nRuns = 0u;
        u32 start = 0u;
        while (start < n && (u32)s_list[start] < t0) start++;
        if (start > kScanFalseCap) fail = true;
        for (u32 j = 0; j < start && !fail; j++) S.falseIdx[nFalse++] = (u16)j;
        // from xx to the next survivor that begins where a block ends (k: the first list entry not passed yet): survivors inside what is
        // walked over are struck, blocks the scan did not see are entered -- run by run.  false: no way through
        auto walk = [&](u32 xx, u32& k) -> bool
        {
          for (;;)
          {
            while (k < n && (u32)s_list[k] < xx)
            {
              if (nFalse >= kScanFalseCap) return false;
              S.falseIdx[nFalse++] = (u16)k;
              k++;
            }
            if (k < n ? xx == (u32)s_list[k] : (lastPiece ? xx == blobRel : xx >= pieceEndRel)) { if (RAG && k >= n && lane == 0) S.mendExit = xx; return true; }
            if (xx >= endTarget || nRuns >= kScanRunCap) return false;
            u32 lx = parseBlock(xx, false, 0u);
            if (RAG)
            {
              // a raw block: whole (what parseBlock says) or an edge block's few pixels -- the first length behind which a known block
              // begins (the next entry, the blob's end) or a block parses
              const u32 bq = (s_in[xx >> 2] >> (8u * (xx & 3u))) & 0xFFu;
              if ((bq & 3u) == 0u && (bq >> 6) == 0u && !(v5 && (bq & 4u)))
              {
                const u32 cand[4] = { 64u, 8u * ragWl, 8u * ragHl, ragWl * ragHl };
                u32 pick = 0u;
                for (u32 ci = 0; ci < 4u && pick == 0u; ci++)
                {
                  if (cand[ci] == 0u) continue;
                  const u32 e = xx + 1u + cand[ci] * G::TB;
                  if (e > blobRel || e + 16u > G::kBytes) continue;
                  const bool known = (k < n && e == (u32)s_list[k]) || (lastPiece && e == blobRel);
                  if (known || (e < blobRel && parseBlock(e, false, 0u) != 0u)) pick = e - xx;
                }
                lx = pick;
              }
            }
#ifdef HIPSIM
            if (lx == 0u && lane == 0 && getenv("LERC_SIM_SCAN_DIAG"))
            {
              printf("piece %u: the walk stops at %u (blob offset %u; end target %u, blob end %u): no block there; bytes:", wg, xx, pieceStart + xx - PRE, endTarget, blobRel);
              for (u32 j = 0; j < 24u; j++) printf(" %02x", (s_in[(xx + j) >> 2] >> (8u * ((xx + j) & 3u))) & 0xFFu);
              printf("\n");
            }
#endif
            if (lx == 0u) return false;
            const u32 b0 = (s_in[xx >> 2] >> (8u * (xx & 3u))) & 0xFFu;
            u32 cnt = 1u;
            if ((b0 & 2u) != 0u && lx <= 9u)    // (all zero, or one value: the same block may follow, again and again)
            {
              const u32 pos = xx + (u32)lane * lx;
              bool go = false;
              if (pos + lx <= G::kBytes && pos < endTarget && pos + lx <= blobRel)
              {
                const u32 bi = (s_in[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu;
                const bool same = ((bi ^ b0) & ~(pattern << 2) & 0xFFu) == 0u;
                const bool known = lane != 0 && ((s_sb[pos >> 5] >> (pos & 31u)) & 1u) != 0u;    // (a survivor begins here: the run's end, mended or not)
                go = same && !known;
              }
              const u64 stop = ~__builtin_amdgcn_ballot_w64(go);
              cnt = stop ? (u32)__ffsll((long long)stop) - 1u : 64u;
              if (cnt == 0u) return false;    // (cannot be: lane 0's block has just parsed)
            }
            if (lane == 0) { S.runPos[nRuns] = (u16)xx; S.runLen[nRuns] = (u16)lx; S.runCnt[nRuns] = (u16)cnt; }
            nRuns++;
            xx += cnt * lx;
          }
        };
        u32 cur = start;
        if (!fail && (start >= n || (u32)s_list[start] > t0))    // blocks in front of the first survivor
        {
          u32 k = start;
          if (!walk(t0, k)) fail = true;
          else if (k < n) cur = k;
          else good = true;
        }
        while (!fail && !good)
        {
          u32 a = 0xFFFFu;    // the first entry from cur on whose link to the next one is broken
          for (u32 j = 0; j < nBad; j++) { const u32 f = (u32)S.badIdx[j]; if (f >= cur && f < a) a = f; }
          if (a == 0xFFFFu) { good = true; break; }
          const u32 pa = (u32)s_list[a], la = parseBlock(pa, false, 0u);
          if (la == 0u) { fail = true; break; }
          u32 k = a + 1u;
          if (!walk(pa + la, k)) { fail = true; break; }
          if (k < n) cur = k; else good = true;
        }
        good = good && !fail;
        __builtin_amdgcn_wave_barrier();
        if (good)
        {
          for (u32 j = (u32)lane; j < nFalse; j += 64u) { const u32 pos = (u32)s_list[S.falseIdx[j]]; atomicAnd(&s_sb[pos >> 5], ~(1u << (pos & 31u))); }
          __builtin_amdgcn_wave_barrier();
          for (u32 r = 0; r < nRuns; r++)
          {
            const u32 rp = S.runPos[r], rl = S.runLen[r], rc = S.runCnt[r];
            if ((u32)lane < rc) { const u32 pos = rp + (u32)lane * rl; atomicOr(&s_sb[pos >> 5], 1u << (pos & 31u)); }
          }
        }
#ifdef HIPSIM
        if (lane == 0 && getenv("LERC_SIM_SCAN_DIAG"))
          printf("piece %u: mending by the first wave: %s; t0 %u (check's: %u) entries %u broken links %u -> runs %u struck %u\n", wg, good ? "good" : "FAILED", t0, S.t0, n, nBad, nRuns, nFalse);
#endif
        if (lane == 0)
        {
          S.nIns = nRuns; S.nFalse = nFalse;
          if (good) S.t0 = t0;
          else S.bad = 1u;
          S.mended = good ? 1u : 0u;
        }
      }
    }
    else if (threadIdx.x == 0)
    {
      {
        const u32 n = S.nEnt, nBad = S.nBad[cp];
        bool good = false;
        // where the piece's first block begins, if the bytes in front of it say so: where the anchor ends
        const u32 t0 = S.t0;
        u32 start0 = 0u;
        while (start0 < n && (u32)s_list[start0] < t0) start0++;
        // (the stream's first block is what it is, and so is what the anchor points at; else the piece's first survivor may be a
        // false one -- and the second)
        const bool sure = t0 != 0u || (dataRel >= PRE && dataRel < pieceEndRel);
        const u32 tries = (S.over || nBad > kScanBadCap) ? 0u : (sure ? 1u : 3u);
        const u32 endTarget = lastPiece ? blobRel : pieceEndRel;
        // (blocks entered: END's bitmap is done with in a masked band -- room for the run behind a raw block, which no flood reaches)
        u16* const insWide = reinterpret_cast<u16*>(&s_end[0]);
        constexpr u32 kInsWide = 8u * G::kMapVecs < 2048u ? 8u * G::kMapVecs : 2048u;    // (16-bit entries in the bitmap's bytes)
        // the length of the block at xx (k: the first list entry behind it), seen by the scan or not
        auto lenAt = [&](u32 xx, u32 k) -> u32
        {
          const u32 b0 = (s_in[xx >> 2] >> (8u * (xx & 3u))) & 0xFFu;
          // (a block of pixels that are all zero, or all invalid: one byte -- a masked band has runs of them)
          u32 lx = 0u;
          if ((b0 & 3u) == 2u && !(v5 && (b0 & 4u))) lx = 1u;
          else if (OFFS && (b0 & 3u) == 0u && (b0 >> 6) == 0u && !(v5 && (b0 & 4u)))    // (a raw block as the reference writes it: nothing in bits 6 - 7, Lerc2.cpp:1973)
          {
            // A RAW block of a masked band: the flag byte and its valid pixels' values -- how many, the stream does not say (the
            // mask does, to who knows the block's place).  Few, where raw beats bit-stuffing (one or two pixels at a mask's edge):
            // the first count behind which a known block begins, or blocks that parse, one behind the other, up to a known one
            // (the decode kernel checks every block's length against the mask; a wrong guess here sends the band the long way)
            // (a known block begins at p, or the piece / the stream ends there)
            auto known = [&](u32 p) -> bool
            {
              u32 lo = k, hi = n;
              while (lo < hi) { const u32 mid = (lo + hi) >> 1; if ((u32)s_list[mid] < p) lo = mid + 1u; else hi = mid; }
              return (lo < n && (u32)s_list[lo] == p) || (lastPiece ? p == blobRel : p >= pieceEndRel);
            };
            // blocks that parse from p2 on, the signature going on, all the way to a known one (a run of one-byte blocks may lie
            // between, and one more raw block: a mask's edge, the invalid blocks up to the edge in the next block row, its raw block)
            // (The signature: block jt of a block row carries jt & pattern -- from codec 5 on two blocks share a value, before that none do;
            // 0 begins a block row.  Two tilings of the same bytes can agree on everything else: ... raw block of TWO values ... read as a
            // raw block of one value and a constant block costs the run behind it one byte, and the count comes out the same.)
            auto chain = [&](auto& self, u32 p2, u32 sgPrev, u32 same, bool sure, u32 depth) -> bool
            {
              const u32 maxSame = v5 ? 2u : 1u, step = v5 ? 2u : 1u;
              for (u32 stp = 0, parses = 0; stp < 4096u && parses < 96u; stp++)
              {
                if (known(p2)) return true;
                if (p2 + 24u > G::kBytes || p2 >= blobRel) return false;
                const u32 fq = (s_in[p2 >> 2] >> (8u * (p2 & 3u))) & 0xFFu, sg = (fq >> 2) & pattern;
                if (v5 && (fq & 4u)) return false;
                if (sg == 0u) { same = sgPrev == 0u ? same + 1u : 1u; sure = sure || sgPrev != 0u; }
                else if (sg == sgPrev) { if (++same > maxSame) return false; }
                else if (sg == ((sgPrev + step) & pattern) && (!sure || same == maxSame)) { same = 1u; sure = true; }
                else return false;
                if ((fq & 3u) != 1u && (fq & 3u) != 3u && (fq >> 6) != 0u) return false;    // (raw and all-zero blocks carry no offset type)
                if ((fq & 3u) == 0u)
                {
                  if (depth == 0u) return false;
                  for (u32 nv2 = 1u; nv2 <= 8u; nv2++) if (self(self, p2 + 1u + nv2 * G::TB, sg, same, sure, depth - 1u)) return true;
                  return false;
                }
                const u32 lq = (fq & 3u) == 2u ? 1u : parseBlock(p2, false, 0u);
                if ((fq & 3u) != 2u) parses++;
                if (lq == 0u) return false;
                sgPrev = sg; p2 += lq;
              }
              return false;
            };
            for (u32 nv = 1u; nv <= 8u && lx == 0u; nv++)
            {
              const u32 q = xx + 1u + nv * G::TB;
              if (q > endTarget && lastPiece) break;
              if (chain(chain, q, (b0 >> 2) & pattern, 1u, false, 1u)) lx = 1u + nv * G::TB;
  #ifdef HIPSIM
              if (lx && getenv("LERC_SIM_SCAN_DIAG")) printf("raw block at blob offset %u: %u values\n", pieceStart + xx - PRE, nv);
  #endif
            }
          }
          else lx = parseBlock(xx, false, 0u);
          return lx;
        };
        // (a masked band: the first attempt walks from the anchor's end, if that lies in front of the first entry; an anchor that leads
        // nowhere is forgotten -- the piece in front has the last word on where this one's first block begins)
        const bool frontGap = OFFS && t0 != 0u && (start0 >= n || t0 < (u32)s_list[start0]) && t0 < endTarget;
        for (u32 att = frontGap ? 0u : 1u; tries != 0u && att <= tries && !good; att++)
        {
          const u32 tr = att == 0u ? 0u : att - 1u;
          if (att != 0u && start0 + tr >= n) break;
          const u32 start = start0 + tr;
          u32 nFalse = 0u, nIns = 0u, cur = start, mendExit = 0u, firstOwn = 0u;
          bool fail = start > kScanFalseCap;
          for (u32 j = 0; j < start && !fail; j++) S.falseIdx[nFalse++] = (u16)j;
          // (a masked band: blocks between the anchor's end and the first entry are walked like those behind a broken link)
          bool wasFront = false;
          bool front = att == 0u;
          while (!fail)
          {
            u32 xx, k;
            if (front) { xx = t0; k = start; front = false; wasFront = true; }
            else
            {
              u32 a = 0xFFFFu;    // the first entry from cur on whose link to the next one is broken
              for (u32 j = 0; j < nBad; j++) { const u32 f = (u32)S.badIdx[j]; if (f >= cur && f < a) a = f; }
              if (a == 0xFFFFu) { good = true; break; }
              const u32 pa = (u32)s_list[a], la = lenAt(pa, a + 1u);
              if (la == 0u) { fail = true; break; }
              xx = pa + la; k = a + 1u;
            }
            // from this block's end to the next survivor that begins where a block ends: survivors inside what is walked over are
            // struck, blocks the scan did not see -- not bit-stuffed, or bit-stuffed behind one that is not -- are entered
            for (;;)
            {
              while (k < n && (u32)s_list[k] < xx && !fail)
              {
                if (nFalse < kScanFalseCap) S.falseIdx[nFalse++] = (u16)k; else fail = true;
                k++;
              }
              if (fail) break;
              if (OFFS && xx >= PRE && firstOwn == 0u) firstOwn = xx;
              if (k < n ? xx == (u32)s_list[k] : (lastPiece ? xx == blobRel : xx >= pieceEndRel)) break;
              if (xx >= endTarget || nIns >= kInsWide) { fail = true; break; }
              const u32 lx = lenAt(xx, k);
              if (lx == 0u) { fail = true; break; }
              if (xx >= PRE) insWide[nIns++] = (u16)xx;    // (blocks in front of the piece's own bytes are the piece's in front)
              xx += lx;
            }
            if (fail) break;
            if (k < n) cur = k; else { good = true; if (OFFS) mendExit = xx; break; }
          }
          if (good)
          {
            for (u32 j = 0; j < nFalse; j++) { const u32 pos = (u32)s_list[S.falseIdx[j]]; s_sb[pos >> 5] &= ~(1u << (pos & 31u)); }
            for (u32 j = 0; j < nIns; j++) { const u32 pos = (u32)insWide[j]; s_sb[pos >> 5] |= 1u << (pos & 31u); }
            S.nIns = nIns; S.nFalse = nFalse;
            if (OFFS) { S.mendExit = mendExit; if (wasFront) S.t0 = firstOwn; else if (frontGap) S.t0 = 0u; }    // (where the piece's first block begins, now that the walk from the anchor's end has been there)
          }
        }
  #ifdef HIPSIM
        if (OFFS && getenv("LERC_SIM_SCAN_DIAG") && !good)
        {
          printf("piece %u: mending failed: n %u nBad %u t0 %u start0 %u tries %u; bad entries:", wg, n, nBad, t0, start0, tries);
          for (u32 j = 0; j < nBad && j < 8u; j++) { const u32 f = S.badIdx[j]; const u32 pos = s_list[f]; printf(" [%u] at %u len %u flag %02x next %u;", f, pos, parseBlock(pos, false, 0u), (s_in[pos >> 2] >> (8u * (pos & 3u))) & 0xFFu, f + 1u < n ? (u32)s_list[f + 1u] : 0u); }
          printf("\n   survivors in front:");
          for (u32 wd = 0; wd < G::kOwnWord0; wd++) for (u32 bt = 0; bt < 32u; bt++) if (s_sb[wd] >> bt & 1u) printf(" %u(+%u)", 32u * wd + bt, parseBlock(32u * wd + bt, false, 0u));
          printf("\n");
          { const u32 f = S.badIdx[0]; const u32 pos = s_list[f]; const u32 e = nBad ? pos + parseBlock(pos, false, 0u) : t0 - 40u; printf("   bytes from %u:", e); for (u32 j = 0; j < 160u; j++) printf(" %02x", (s_in[(e + j) >> 2] >> (8u * ((e + j) & 3u))) & 0xFFu); printf("\n"); }
        }
  #endif
        if (!good) S.bad = 1u;
        S.mended = good ? 1u : 0u;
      }
    }
    __syncthreads();
    if (S.mended)
    {
      buildList(false);
      tilePass(cp + 1u, true);
      if (S.nBad[cp + 1u] != 0u) S.bad = 1u;    // (mended once; a list that still does not tile goes the long way)
    }
  }

  // ---- count out: blocks of this piece, and where its last block ends (relative to the piece's end)
  const u32 total = S.bad ? 0u : S.nEnt;
  if (threadIdx.x == 0)
  {
    const u32 ex = S.exitRel >= pieceEndRel ? min(S.exitRel - pieceEndRel, 0xFFFDu) : 0xFFFFu;
    const u32 earlyCount = early ? S.earlyCount : 0u;
    if (early && !lastPiece && min(total, 0xFFFFu) != earlyCount) raiseFlag(b, 1);    // (the pieces behind have been told another count; nobody is behind the last piece -- a ragged raster's corner block, a few pixels the scan does not see, is found there by the mending)
    publish64(b.wgCell + wg, tag | ((u64)ex << 16) | (u64)(early ? earlyCount : min(total, 0xFFFFu)));
  }
  // The blocks' places need the cells of the pieces in front -- those of this group, and one per group in front; the piece right in
  // front also says where its last block ends: this piece's first block has to begin there.  The cells are asked for now;
  // while they travel (the pieces in front publish when this one does, and a cell takes a few microseconds to be seen) the
  // PIXELS are taken out of the stream: what a block's pixels ARE does not hang on where the block lies -- only where they go
  // does.  A wave takes BPW blocks of the list at a time, a lane V consecutive pixels of one row of one block, and keeps up to
  // kHeld such 16-byte vectors in registers; they are stored when the places are known.
  // float: a lane takes TWO vectors of its block -- the same four columns of row r and of row r + 4 -- so that what a lane does per block
  // (the block's word, offset and place out of LDS, bit width, mask, payload position, the address) is done once for eight pixels instead
  // of once for four; a wave tile is 8 blocks, a store instruction 4 raster rows of 256 bytes on end.  (The two vectors of one block ROW
  // in a lane were tried first: every store instruction then writes every other 16 bytes, and the kernel took 125 us instead of 100.)
  constexpr int NV = (!RAG && LERC_SCAN_WIDE && DT == DT_Float) ? 2 : 1;       // vectors a lane (32-bit integers: their 64-bit dequantiser leaves no registers for it)
  constexpr int PXL = V * NV, BPWL = BPW * NV;                          // pixels a lane, blocks a wave tile
  constexpr int RSTEP = 8 / NV;                                         // rows between a lane's vectors
  struct alignas(sizeof(T) * V) Vec1 { T e[V]; };
  struct alignas(sizeof(T) * V) Vec { T e[PXL]; };
  constexpr u32 kHeldV = sizeof(T) == 2 ? LERC_SCAN_HELD16 : (DT == DT_Float || DT == DT_Double) ? LERC_SCAN_HELD : LERC_SCAN_HELD32;
  constexpr u32 kHeld = RAG ? (kHeldV >= 2u ? kHeldV / 2u : kHeldV) / 1u : kHeldV / (u32)NV;                               // (as many registers; RAG: only wave tiles of whole, plain blocks are held -- an edge block's shape hangs on its place)
  const int r = lane / (8 * NV), c = lane % (8 * NV), bb = c / LPR, h = c % LPR;    // row (of the lane's first vector), block of the wave tile, vector of the block row
  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  bool bad = false;
  // the lane's V pixels of the block in round slot tSlot (parseBlock's word and offset); a wave-uniform fast path for the common
  // case, all blocks of the wave alike: bit-stuffed without a table, the lane's V values inside 64 bits, no clamp -- three words
  // of the stream, one funnel shift each way, V shifts
  bool tilePlain = false;    // RAG: the wave tile blockRow has just decoded holds whole, plain blocks only
  auto blockRow = [&](u32 tSlot, bool have, bool dimsKnown = true) -> Vec
  {
    const u32 code = have ? s_code[tSlot] : 0u;
    const double offRaw = s_offs[tSlot];
    i64 offBits; memcpy(&offBits, &offRaw, 8);
    const double offset = DT < DT_Float ? (double)offBits : offRaw;    // (integer types keep the integer: see parseBlock)
    const u32 nbC = (code >> 16) & 31u, mode = (code >> 21) & 3u, lut = (code >> 23) & 1u;
    const u32 pbit = 8u * (code & 0xFFFFu);        // payload / first raw value
    int e0 = r * 8 + h * V, vc = V;                // the lane's first element of the block, and how many of its V pixels exist
    Vec o;
#pragma unroll
    for (int k = 0; k < PXL; k++) o.e[k] = T(0);
    const bool plain = ((code >> 24) & 1u) != 0u;
    const bool plainTile = __all(plain || !code);
    if (RAG)
    {
      tilePlain = plainTile;                       // (a plain block is a whole one: the stores need no shapes either)
      if (!plainTile)
      {
        const u32 dims = (have && dimsKnown) ? (u32)s_dims[tSlot] : 0x88u;
        const int bw = (int)(dims & 15u), bh = (int)(dims >> 4);
        e0 = r * bw + h * V;
        vc = r < bh ? max(0, min(V, bw - h * V)) : 0;
      }
    }
    if (plainTile)
    {
      if (code)
      {
        const u32 nb = nbC;
        const u32 mask = nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
        const i64 offI = DT < DT_Float ? offBits : (i64)offset;
#pragma unroll
        for (int hv = 0; hv < NV; hv++)    // (a vector's V values lie inside 64 bits: three words of the stream, a funnel shift each way)
        {
          const u32 bit0 = pbit + (u32)(e0 + hv * 8 * RSTEP) * nb, wi = bit0 >> 5;
          const u32 x0 = s_in[wi], x1 = s_in[wi + 1], x2 = s_in[wi + 2];
          const u64 all = ((u64)__builtin_amdgcn_alignbit(x2, x1, bit0) << 32) | __builtin_amdgcn_alignbit(x1, x0, bit0);
#ifndef HIPSIM
          if constexpr (sizeof(T) == 2 && V == 8 && DT < DT_Float)
          {
            // 16-bit pixels: eight values of at most eight bits -- the first four lie in the low word, the other four in the word that begins
            // 4 nb bits on: a bit-field extract a value (no 64-bit shifts), and without a scale (lossless: 1) an addition
            const u32 lo = (u32)all, mid = (u32)(all >> (4u * nb));
            const u32 off32 = (u32)offI, inv32 = (u32)invI;
            u32 q8[8];
#pragma unroll
            for (int k = 0; k < 4; k++) { q8[k] = __builtin_amdgcn_ubfe(lo, (u32)k * nb, nb); q8[4 + k] = __builtin_amdgcn_ubfe(mid, (u32)k * nb, nb); }
            if (inv32 == 1u)
            {
#pragma unroll
              for (int k = 0; k < 8; k++) o.e[hv * V + k] = (T)(off32 + q8[k]);
            }
            else
            {
#pragma unroll
              for (int k = 0; k < 8; k++) o.e[hv * V + k] = (T)(off32 + q8[k] * inv32);
            }
            continue;
          }
#endif
#pragma unroll
          for (int k = 0; k < V; k++)
          {
            const u32 q = (u32)(all >> ((u32)k * nb)) & mask;
            if (DT >= DT_Float) o.e[hv * V + k] = (T)(offset + (double)q * p.invScale);    // Lerc2.cpp:2159-2160, no contraction
            else if (sizeof(T) <= 4) o.e[hv * V + k] = (T)((u32)offI + q * (u32)invI);    // (the low 32 bits of the sum are the pixel: no 64-bit product)
            else o.e[hv * V + k] = (T)(offI + (i64)q * invI);
          }
        }
      }
    }
    else if (code)
    {
      if (mode == 0)
      {
#pragma unroll
        for (int k = 0; k < PXL; k++)
        {
          const u32 bp = pbit + (u32)(e0 + (k / V) * 8 * RSTEP + k % V) * 8u * (u32)sizeof(T);
          u64 bits = ldsBits(s_in, bp, 32);
          if (sizeof(T) == 8) bits |= (u64)ldsBits(s_in, bp + 32, 32) << 32;
          else if (sizeof(T) < 4) bits &= (1ull << (8 * sizeof(T))) - 1;
          memcpy(&o.e[k], &bits, sizeof(T));
        }
      }
      else if (mode == 3)
      {
#pragma unroll
        for (int k = 0; k < PXL; k++) o.e[k] = (T)offset;
      }
      else if (mode == 1)
      {
        const int nb = (int)nbC;
        const i64 offI = DT < DT_Float ? offBits : (i64)offset;
        if (!lut)
        {
#pragma unroll
          for (int k = 0; k < PXL; k++)
            o.e[k] = dequant<T>(offset, ldsBits(s_in, pbit + (u32)(e0 + (k / V) * 8 * RSTEP + k % V) * (u32)nb, nb), p.invScale, p.zMaxHdr, offI, invI, zMaxI);
        }
        else
        {
          const u32 nLut = (ldsBits(s_in, pbit - 8u, 8) - 1u) & 0xFFu;    // (the byte in front of the table: its size + 1)
          const int nbIdx = bitLen(nLut);
          const u32 idxBit = pbit + 8u * ((nLut * (u32)nb + 7) >> 3);
#pragma unroll
          for (int k = 0; k < PXL; k++)
          {
            u32 ix = ldsBits(s_in, idxBit + (u32)(e0 + (k / V) * 8 * RSTEP + k % V) * (u32)nbIdx, nbIdx);
            if (ix > nLut) { ix = 0; bad = bad || !RAG || (k % V) < vc; }    // the reference would read outside its table here (RAG: pixels that do not exist have no index)
            const u32 q = ix ? ldsBits(s_in, pbit + (ix - 1) * (u32)nb, nb) : 0u;
            o.e[k] = dequant<T>(offset, q, p.invScale, p.zMaxHdr, offI, invI, zMaxI);
          }
        }
      }
    }
    return o;
  };
  const u32 grp = wg / kOneGroup, g0 = grp * kOneGroup, nIn = wg - g0;
  // (tuning, LERC_SCAN_DIRECT=n: the launch's first n pieces -- resident together -- add up the cells of ALL the pieces in front of them instead of
  // their group's and the groups' totals: no chain through the groups' last pieces while everybody starts at once)
  const bool direct = LERC_SCAN_DIRECT != 0 && !OFFS && wg < (u32)LERC_SCAN_DIRECT;
  const u32 nCells = direct ? wg : nIn + grp + ((nIn == 0u && wg != 0u) ? 1u : 0u);
  auto cellOf = [&](u32 i) -> const u64* { return direct ? b.wgCell + i : i < nIn ? b.wgCell + g0 + i : i < nIn + grp ? b.wgGroupCell + (i - nIn) : b.wgCell + (wg - 1u); };
  u64 cell0 = 0;
  if (threadIdx.x < nCells) cell0 = observe64(cellOf(threadIdx.x));
  const u32 nFirst = OFFS ? 0u : min(total, R);     // blocks of the first round: their headers are parsed
  Vec held[kHeld ? kHeld : 1u];
  u32 heldMask = 0u;                               // RAG: wave tiles that are held (all their blocks whole and plain)
#pragma unroll
  for (u32 j = 0; j < kHeld; j++)
  {
    const u32 f = (u32)BPWL * ((u32)w + kWaves * j) + (u32)bb;
    if ((u32)BPWL * ((u32)w + kWaves * j) < nFirst)    // (the same for all lanes of the wave)
    {
      if (RAG)
      {
        const u32 cd = f < nFirst ? s_code[f] : 0u;
        // (every block of the tile parsed, whole and plain: a block without a word yet -- a raw edge block gets its own when its place
        // is known -- must not be held as zeros)
        if (__all(f >= nFirst || ((cd >> 24) & 1u) != 0u)) { heldMask |= 1u << j; held[j] = blockRow(f, f < nFirst, false); }
      }
      else held[j] = blockRow(f, f < nFirst);
    }
  }
  {
    u64 part = 0;
    bool lost = false;
    for (u32 i = threadIdx.x; i < nCells; i += NT)
    {
      const u64* pc = cellOf(i);
      u64 c = i == threadIdx.x ? cell0 : observe64(pc);
      for (u32 spin = 0; (u32)(c >> 32) != epoch && spin < b.spinLimit; spin++)
      {
        __builtin_amdgcn_s_sleep(LERC_SCAN_SLEEP);
        c = observe64(pc);
      }
      if ((u32)(c >> 32) != epoch) { lost = true; c = 0; }
      else if (direct ? i == wg - 1u : (i == nIn - 1u || i == nIn + grp)) S.prevExit = ((u32)c >> 16) & 0xFFFFu;    // (the piece right in front)
      if (direct) part += i >= g0 ? (u64)((u32)c & 0xFFFFu) : ((u64)((u32)c & 0xFFFFu) << 32);
      else if (i < nIn + grp) part += i < nIn ? (u64)((u32)c & 0xFFFFu) : ((u64)(u32)c << 32);    // (a piece's cell: exit (16) | blocks (16))
    }
    if (__any(lost) && lane == 0) S.lost = 1u;
    if (nCells != 0u)
    {
      part = waveSum(part);
      if (lane == 0 && part) atomicAdd((unsigned long long*)&S.part, (unsigned long long)part);
    }
  }
  __syncthreads();
  TRACES(5);
  if (LERC_DEC_EXIT == 5)    // (the held vectors are kept alive)
  {
#pragma unroll
    for (u32 j = 0; j < kHeld; j++)
    {
      u32 w4[sizeof(Vec) / 4];
      memcpy(w4, &held[j], sizeof(Vec));
#pragma unroll
      for (u32 q = 0; q < sizeof(Vec) / 4; q++) asm volatile("" :: "v"(w4[q]));
    }
    return;
  }
  if (S.lost)    // gave up waiting (never seen; the general path takes the band)
  {
    if (threadIdx.x == 0) raiseFlag(b, 3);
    return;
  }
  const u32 inGroup = (u32)S.part, base = (u32)(S.part >> 32) + inGroup;
  if (threadIdx.x == 0)
  {
    if (wg == g0 + kOneGroup - 1u) publish64(b.wgGroupCell + grp, tag | (u64)(inGroup + total));    // this group's total, for the groups behind
    if (S.over) raiseFlag(b, 0);
  }
  // the piece's verdict (one thread).  EARLY: where the piece in front ends may not be known yet (0xFFFE) -- then what the verdict needs is put
  // aside (LDS: nothing of it stays in registers over the pixel loop) and it is given behind the pixels, when that piece has long said
  auto verdict = [&](const ScanVerdictIn& v, u32 prevExit)
  {
    const bool lastP = v.blobRel <= PRE + P;          // (the blob ends in this piece's own bytes)
    bool bad = S.bad != 0u;
    // where this piece's blocks begin: with the stream (the first piece), else where the piece in front says its last block ends
    u32 exitRel = S.exitRel;
    if (v.total == 0u && lastP && wg != 0u && prevExit != 0xFFFFu) exitRel = PRE + prevExit;    // (the stream's last block began in the piece in front)
    const u32 first = v.total ? (u32)s_list[0] : exitRel;
    const u32 firstPiece = v.dataBegin / P;    // (a masked band: the mask's bytes may fill pieces of their own in front of the stream)
    if (wg < firstPiece) bad = bad || v.total != 0u;
    else if (wg == firstPiece) bad = bad || first != v.dataRel;
    else bad = bad || prevExit == 0xFFFFu || first != PRE + prevExit;
    if (v.total == 0u && !lastP && wg >= firstPiece) bad = true;    // (a piece is longer than any block)
    TRACEV(8, v.total); TRACEV(9, S.nBad[0]); TRACEV(10, S.nBad[1] | (S.nBad[2] << 16)); TRACEV(11, (S.over ? 1u : 0u) | (S.bad ? 2u : 0u) | (bad ? 4u : 0u) | (S.mended ? 8u : 0u));
    TRACEV(12, first); TRACEV(13, PRE + prevExit); TRACEV(14, S.nIns); TRACEV(15, S.nFalse);
#ifdef HIPSIM
    if (OFFS && getenv("LERC_SIM_SCAN_DIAG") && (bad || S.over || getenv("LERC_SIM_SCAN_DIAG")[0] == '2'))
      printf("piece %u: total %u first %u expected %u t0 %u nBad %u %u %u frontBad %u over %u S.bad %u mended %u nIns %u nFalse %u exit %u pieceEnd %u blobRel %u dataRel %u\n", wg, v.total, first, PRE + prevExit, S.t0,
             S.nBad[0], S.nBad[1], S.nBad[2], S.frontBad, S.over, S.bad, S.mended, S.nIns, S.nFalse, S.exitRel, pieceEndRel, v.blobRel, v.dataRel);
#endif
    if (bad) raiseFlag(b, 1);
    // the pieces hold all the raster's blocks, or the band goes the long way
    if (lastP && (v.base + v.total != v.nWanted || exitRel != v.blobRel)) raiseFlag(b, 2);
  };
  if (threadIdx.x == 0)
  {
    ScanVerdictIn v;
    v.total = total; v.base = base; v.dataRel = dataRel; v.blobRel = blobRel; v.dataBegin = hl.dataBegin; v.nWanted = OFFS ? job.nPos : hp.nBlocks;
    if (early && S.prevExit == 0xFFFEu) { S.vIn = v; S.vDefer = 1u; }
    else verdict(v, S.prevExit);
  }
  if (OFFS)
  {
    // where block k of the stream begins
    for (u32 f = threadIdx.x; f < total; f += NT)
      if (base + f < job.nPos) job.blockOff[base + f] = pieceStart + (u32)s_list[f] - PRE;
    return;
  }

  // ---- rounds of at most R blocks: the blocks' places (lane = block), then the pixels' way out
  const bool pow2 = (hp.nTH & (hp.nTH - 1u)) == 0u;
  const u32 thShift = 31u - (u32)__clz((int)hp.nTH);
  for (u32 fLo = 0; fLo < total; )
  {
    // (no vectors held: the rounds are cut on multiples of BPW blocks of the RASTER, like the wave tiles below)
    const u32 fHi = kHeld ? min(total, fLo + R) : min(total, ((base + fLo + R) / (u32)BPWL) * (u32)BPWL - base);
    {
      const u32 f = fLo + threadIdx.x;
      if (f < fHi)
      {
        const u32 t = threadIdx.x;
        if (fLo != 0u) (void)parseBlock((u32)s_list[f], true, t);
        u32 code = s_code[t];
        const u32 sigHdr = s_at[t];
        const u32 blk = base + f;          // index of the block in the raster
        const u32 it = pow2 ? (blk >> thShift) : blk / hp.nTH, jt = blk - it * hp.nTH;
        if (sigHdr != (jt & pattern) || blk >= hp.nBlocks) code = 0;    // signature = (j0 >> 3) & pattern, j0 = 8 jt
        if (RAG)
        {
          // the block's size by its place; a bit-stuffed or raw block has to hold as many pixels as that (Lerc2.cpp:1504-1519)
          const u32 nTV = ((u32)nRows + 7u) >> 3;
          const u32 bw = (ragWl != 0u && jt == hp.nTH - 1u) ? ragWl : 8u, bh = (ragHl != 0u && it == nTV - 1u) ? ragHl : 8u;
          {
            // (a raw block says nothing of its size: it reaches to where the next block begins -- the list knows; parseBlock took it for a whole one)
            const u32 posB = (u32)s_list[f], nextB = f + 1u < total ? (u32)s_list[f + 1u] : S.exitRel;
            const u32 bq = (s_in[posB >> 2] >> (8u * (posB & 3u))) & 0xFFu;
            if ((bq & 3u) == 0u && (bq >> 6) == 0u && !(v5 && (bq & 4u)) && nextB > posB && ragRawOk(nextB - posB) && sigHdr == (jt & pattern) && blk < hp.nBlocks)
              code = ((posB + 1u) & 0xFFFFu) | 0x80000000u | ((((nextB - posB - 1u) / G::TB - 1u) & 63u) << 25);
          }
          const u32 modeB = (code >> 21) & 3u;
          if (code && (modeB == 0u || modeB == 1u) && ((code >> 25) & 63u) + 1u != bw * bh) code = 0;
          if (code) { code &= ~(63u << 25); s_code[t] = code; }
          s_dims[t] = (u8)(bw | (bh << 4));
        }
#ifdef HIPSIM
        if (code == 0u && getenv("LERC_SIM_SCAN_DIAG"))
          printf("piece %u: block %u of the piece (raster block %u = row %u column %u of %u) does not belong there: word %08x, signature %u, at %u, next %u\n", wg, f, blk, it, jt, hp.nTH, s_code[t], sigHdr,
                 (u32)s_list[f], f + 1u < total ? (u32)s_list[f + 1u] : S.exitRel);
#endif
        if (code == 0u) { s_code[t] = 0u; bad = true; }
        s_at[t] = code ? (it * 8u) * (u32)p.nCols + jt * 8u : kNoOffset;
      }
    }
    __syncthreads();
    const u32 nRound = fHi - fLo;
    const bool rowsAligned = (((size_t)p.nCols * sizeof(T)) & 15u) == 0u, rowsDword = sizeof(T) * V == 16 && (((size_t)p.nCols * sizeof(T)) & 3u) == 0u;
    auto store = [&](u32 tSlot, const Vec& o, bool wholeTile = false)
    {
      const u32 at0 = s_at[tSlot];
#ifdef LERC_TUNE_WRAP_STORES    // (tuning: the pixels written into the raster's first 4 MB over and over -- what the kernel takes without HBM writes; results invalid)
      if (at0 != kNoOffset)
      {
#pragma unroll
        for (int hv = 0; hv < NV; hv++)
        {
          T* dst = outPix + (((size_t)at0 + (size_t)(r + hv * RSTEP) * (size_t)p.nCols + (size_t)(h * V)) & (size_t)((1u << 20) - 1u));
          Vec1 v1; memcpy(&v1, &o.e[hv * V], sizeof(Vec1)); *reinterpret_cast<Vec1*>(dst) = v1;
        }
      }
#else
      if (RAG && wholeTile)
      {
        // (whole blocks: a vector a lane -- aligned where the rows are, else 16 bytes at dword alignment, else pixel by pixel)
        if (at0 != kNoOffset)
        {
          T* dst = outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V);
          Vec1 v1; memcpy(&v1, &o.e[0], sizeof(Vec1));
          if (rowsAligned) DECODE_STORE(reinterpret_cast<Vec1*>(dst), v1);
          else if (rowsDword) { if constexpr (sizeof(Vec1) == 16) storeStreamingA4(dst, v1); }
          else
          {
#pragma unroll
            for (int k = 0; k < V; k++) dst[k] = o.e[k];
          }
        }
      }
      else if (RAG)
      {
        if (at0 != kNoOffset)
        {
          const u32 dims = (u32)s_dims[tSlot];
          const int bw = (int)(dims & 15u), bh = (int)(dims >> 4);
          const int vc = r < bh ? max(0, min(V, bw - h * V)) : 0;
          T* dst = outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V);
          const size_t pitch = (size_t)p.nCols * sizeof(T);
          Vec1 v1; memcpy(&v1, &o.e[0], sizeof(Vec1));
          if (vc == V && (pitch & 15u) == 0u) DECODE_STORE(reinterpret_cast<Vec1*>(dst), v1);
          else if (vc == V && sizeof(Vec1) == 16 && (pitch & 3u) == 0u) { if constexpr (sizeof(Vec1) == 16) storeStreamingA4(dst, v1); }    // (16 bytes at dword alignment)
          else
          {
#pragma unroll
            for (int k = 0; k < V; k++) if (k < vc) dst[k] = o.e[k];
          }
        }
      }
      else if (at0 != kNoOffset)
      {
        T* dst = outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V);
#pragma unroll
        for (int hv = 0; hv < NV; hv++) { Vec1 v1; memcpy(&v1, &o.e[hv * V], sizeof(Vec1)); DECODE_STORE(reinterpret_cast<Vec1*>(dst + (size_t)(hv * RSTEP) * (size_t)p.nCols), v1); }
      }
#endif
    };
    if (kHeld == 0u)
    {
      // wave tiles on multiples of BPW blocks of the raster: a tile row is a whole 128-byte line of the output
      const u32 blkLo = base + fLo, blkHi = base + fHi;
      const u32 g1 = (blkHi + BPWL - 1) / BPWL;
      for (u32 g = blkLo / BPWL + (u32)w; g < g1; g += kWaves)
      {
        const u32 blk = g * BPWL + (u32)bb;
        const bool have = blk >= blkLo && blk < blkHi;
        const u32 tSlot = have ? blk - blkLo : 0u;     // (the tile's blocks outside the round: lanes that do nothing)
        const Vec o = blockRow(tSlot, have);
        if (have) store(tSlot, o);
      }
      fLo = fHi;
      if (fLo < total) __syncthreads();
      continue;
    }
    u32 jFrom = 0;
    if (fLo == 0u)
    {
#pragma unroll
      for (u32 j = 0; j < kHeld; j++)
      {
        const u32 tSlot = (u32)BPWL * ((u32)w + kWaves * j) + (u32)bb;
        if (RAG && !((heldMask >> j) & 1u) && (u32)BPWL * ((u32)w + kWaves * j) < nRound)
        {
          const Vec o = blockRow(tSlot, tSlot < nRound);    // (not held: an edge block among the tile's, or a block that is not plain)
          if (tSlot < nRound) store(tSlot, o, tilePlain);
        }
        else if (tSlot < nRound) store(tSlot, held[j], RAG);
      }
      jFrom = kHeld;
    }
    for (u32 g = (u32)w + kWaves * jFrom; g * (u32)BPWL < nRound; g += kWaves)
    {
      const u32 tSlot = g * (u32)BPWL + (u32)bb;
      const Vec o = blockRow(tSlot, tSlot < nRound);
      if (tSlot < nRound) store(tSlot, o, RAG && tilePlain);
    }
    fLo = fHi;
    if (fLo < total) __syncthreads();    // (the round's arrays are taken again)
  }
  TRACES(6);
  if (__any(bad) && lane == 0) raiseFlag(b, 3);
  if (early && threadIdx.x == 0 && S.vDefer)
  {
    u64 c = observe64(b.wgCell + (wg - 1u));
    for (u32 spin = 0; ((u32)(c >> 32) != epoch || (((u32)c >> 16) & 0xFFFFu) == 0xFFFEu) && spin < b.spinLimit; spin++)
    {
      __builtin_amdgcn_s_sleep(LERC_SCAN_SLEEP);
      c = observe64(b.wgCell + (wg - 1u));
    }
    const bool seen = (u32)(c >> 32) == epoch && (((u32)c >> 16) & 0xFFFFu) != 0xFFFEu;
    verdict(S.vIn, seen ? (((u32)c >> 16) & 0xFFFFu) : 0xFFFFu);
  }

  // ---- checksum: the launch's last workgroup waits for everybody's terms (they were sent off microseconds after each
  // workgroup started), folds them and clears the accumulators for the next call (Lerc2.cpp:1037-1064)
  if (!lastPiece) return;
  __syncthreads();    // (S.fa / S.fb are free)
  {
    const u32 nGroups = fastOneGroups(nWG);
    u64 A = 0, B = 0;
    bool lostF = false;
    for (u32 i = threadIdx.x; i < nGroups; i += NT)
    {
      const u64 want = (u64)min(kOneGroup, nWG - i * kOneGroup);
      u64 v = observe64(b.wgAcc + i);
      for (u32 spin = 0; (v >> 48) != want && spin < (1u << 22); spin++)
      {
        __builtin_amdgcn_s_sleep(8);
        v = observe64(b.wgAcc + i);
      }
      if ((v >> 48) != want) lostF = true;
      publish64(b.wgAcc + i, 0ull);
      A += v & 0xFFFFFFull; B += (v >> 24) & 0xFFFFFFull;
    }
    A = waveSum(A % 65535u); B = waveSum(B % 65535u);
    if (lane == 0) { S.fa[w] = A; S.fb[w] = B; }
    if (__any(lostF) && lane == 0) raiseFlag(b, 3);
    __syncthreads();
    if (threadIdx.x != 0) return;
    A = 0; B = 0;
    for (u32 i = 0; i < kWaves; i++) { A += S.fa[i]; B += S.fb[i]; }
    A %= 65535u; B %= 65535u;
    const u64 N = ((u64)(blobEnd - 14u) + 1) / 2;
    u64 s1 = A, s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
    if (s1 == 0) s1 = 0xffff;
    if (s2 == 0) s2 = 0xffff;
    const u32 good = ((u32)((s2 << 16) | s1) == S.hp.expectChecksum) ? 1u : 0u;
    publish32(&b.params->checksumOk, good);
    if (b.hostParams) b.hostParams->checksumOk = good;
  }
}

// blockIdx.y = tile of a batch (one raster: a batch of 1); each tile has its own slice of every buffer
#ifdef HIPSIM
#define LERC_SCAN_SGPR_CAP
#else
#define LERC_SCAN_SGPR_CAP __attribute__((amdgpu_num_sgpr(80)))
#endif
template<class T, bool RAG = false>
__global__ void __launch_bounds__(kScanThreads) LERC_SCAN_SGPR_CAP
k_fast_decode_scan(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, u32 specEnd, int nRows, int nCols, T* __restrict__ outPix)
{
  const size_t tile = blockIdx.y;
  b.params += tile; b.fallback += 4 * tile;
  b.wgCell += tile * b.wgStride;
  b.wgGroupCell += tile * b.wgGroupStride;
  b.wgAcc += tile * b.wgGroupStride;
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
  __shared__ ScanShared<T> sm;
  fastScanBody<T, 0, RAG>(sm, b, blob, sizeGiven, specEnd, nRows, nCols, outPix + tile * t.tileElems, blockIdx.x, ScanOffsetsJob());
}

// MODE 1: a masked band's block stream cut into blocks (see ScanOffsetsJob); no pixels, no checksum
template<class T>
__global__ void __launch_bounds__(kScanThreads) LERC_SCAN_SGPR_CAP
k_fast_scan_offsets(FastDecodeBuffers b, ScanOffsetsJob job, const u8* blob, int nRows, int nCols)
{
  __shared__ ScanShared<T> sm;
  fastScanBody<T, 1>(sm, b, blob, job.blobEnd, job.blobEnd, nRows, nCols, (T*)nullptr, blockIdx.x, job);
}

template<class T>
static void launchFastDecodeScanT(int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven, const FastDecodeBuffers& b, void* out,
                                  hipStream_t st)
{
  const dim3 grid(fastScanNumWG(b.scanGridBytes ? min(b.scanGridBytes, sizeGiven) : sizeGiven), t.nTiles), block(kScanThreads);    // (sizeGiven: the largest blob of the batch)
  // (rows / columns no multiples of 8: the instantiation that knows edge blocks)
  if (nRows % 8 != 0 || nCols % 8 != 0) hipLaunchKernelGGL((k_fast_decode_scan<T, true>), grid, block, 0, st, b, t, blob, sizeGiven, b.scanSpecEnd, nRows, nCols, (T*)out);
  else hipLaunchKernelGGL((k_fast_decode_scan<T, false>), grid, block, 0, st, b, t, blob, sizeGiven, b.scanSpecEnd, nRows, nCols, (T*)out);
}

// diagnostic: workgroups of the float kernel a CU holds, by the runtime's count
extern "C" __attribute__((visibility("default"))) int lerc_amd_probe_decode_scan_residency()
{
#ifdef HIPSIM
  return 0;
#else
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fast_decode_scan<float>, (int)kScanThreads, 0) != hipSuccess) return -1;
  return n;
#endif
}

bool fastDecodeScanEligible(int nRows, int nCols)
{
  static const bool ragOn = []() { const char* e = getenv("LERC_AMD_SCAN_RAGGED"); return !e || atoi(e) != 0; }();    // (0: ragged rasters keep to the walking tier)
  return (nRows % 8 == 0 && nCols % 8 == 0) || (ragOn && nRows >= 8 && nCols >= 8);
}

// a masked band's block offsets by the scanning decoder's first half (8 x 8 blocks, one value per pixel, 16-bit and wider types)
void launchFastScanOffsets(int dt, int nRows, int nCols, const u8* band, u32 version, u32 dataBegin, u32 blobEnd, u32* blockOff, u32 nPos,
                           const FastDecodeBuffers& b, hipStream_t st)
{
  ScanOffsetsJob job;
  job.version = version; job.dataBegin = dataBegin; job.blobEnd = blobEnd; job.blockOff = blockOff; job.nPos = nPos;
  const dim3 grid(fastScanNumWG(blobEnd)), block(kScanThreads);
  switch (dt)
  {
    case DT_Short:  hipLaunchKernelGGL((k_fast_scan_offsets<short>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    case DT_UShort: hipLaunchKernelGGL((k_fast_scan_offsets<unsigned short>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    case DT_Int:    hipLaunchKernelGGL((k_fast_scan_offsets<int>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    case DT_UInt:   hipLaunchKernelGGL((k_fast_scan_offsets<unsigned int>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    case DT_Float:  hipLaunchKernelGGL((k_fast_scan_offsets<float>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    case DT_Double: hipLaunchKernelGGL((k_fast_scan_offsets<double>), grid, block, 0, st, b, job, band, nRows, nCols); break;
    default: break;
  }
}

void launchFastDecodeScan(int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                          const FastDecodeBuffers& b, void* out, hipStream_t st)
{
  switch (dt)
  {
    case DT_Short:  launchFastDecodeScanT<short>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UShort: launchFastDecodeScanT<unsigned short>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Int:    launchFastDecodeScanT<int>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UInt:   launchFastDecodeScanT<unsigned int>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Float:  launchFastDecodeScanT<float>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Double: launchFastDecodeScanT<double>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    default: break;
  }
}

}    // namespace lerc
