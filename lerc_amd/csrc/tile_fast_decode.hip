// tile_fast_decode.hip -- streaming decoder kernels for the common case (one band, nDepth == 1, every
// pixel valid, 8 x 8 blocks, rows and columns multiples of 8).  Same results as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends), so decoding starts with a
// discovery pass over 2 KiB chunks of the blob (tile_fast.h).  Everything serial about it runs out of LDS:
//   k_fast_discover    a workgroup = kDiscChunks chunks staged with 16-byte loads (their Fletcher32 terms are summed
//                      on the way, so the decode kernel does not look at the checksum at all).  Every workgroup reads
//                      the band header itself (the host has not seen a byte of the blob when it enqueues the two
//                      kernels).  A bit-stuffed block reads flag byte, offset, 10?nnnnn, count 64: "64 behind
//                      10?nnnnn" is found four positions per lane and step in each chunk's first `window` bytes and is
//                      true for one position in a thousand of anything else.  Of the blocks found, those that are
//                      not the block right behind another one start a walk: one wave walks all chunks' heads in
//                      lockstep (lane = chunk x head), fetching the words of the next block as soon as its start is
//                      known, until each has landed on a header found in the NEXT chunk's window.  Walks that are
//                      not on the path end within a few steps (signature sequence).
//   k_fast_decode      its first blocks resolve, 256 chunks each: whatever ALL live walks of a chunk agree on is true
//                      without knowing which one is real: entry of chunk c = agreed exit of chunk c-1; the walk that
//                      starts exactly there (or passes it with one of its first blocks) is the true path and its
//                      length the chunk's block count; scanned -- inside the block, then over the totals of the blocks
//                      in front -- it is the index of the chunk's first block, left in an epoch-tagged cell per chunk.
//                      Block 0 also folds the checksum terms.
//                      The other workgroups decode the blocks that start in kDecodeChunks chunks: they stage the byte
//                      span in LDS, read their chunks' cells, gather the block starts from the true walks' lists, parse
//                      the block headers once (lane = block; signature and contiguity checks = ReadTile's integrity
//                      checks), then every lane extracts V consecutive pixels of one raster row, dequantises (double
//                      precision in the reference's expression order for float types, exact integer arithmetic for
//                      integer types) and stores one 16-byte vector.
// Streams the scan cannot follow (a chunk whose window holds no bit-stuffed block on the path: long runs of constant
// or raw blocks) and anything else unexpected raise an epoch tagged flag; the host then repeats the band with the
// general kernels.
// Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540; Lerc2.cpp:1037-1064 (checksum).
#include "tile_fast_decode_dev.h"

namespace lerc {

PROBE_DEFINE(fast_decode)
#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time lines (constant-rate counter) of the discovery kernel (slots 0 .. 7 of row blockIdx.x) and of the
// decode workgroups (rows behind 8192), read by tools/trace_decode.py
static __device__ unsigned long long g_traceD[8 * 32768];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace_decode(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceD), sizeof(unsigned long long) * (size_t)n); }
#define TRACED(row, slot) do { if (threadIdx.x == 0 && (row) < 32768u) g_traceD[8 * (row) + (slot)] = wall_clock64(); } while (0)
#else
#define TRACED(row, slot)
#endif


// sizes of a discovery workgroup of NCH chunks
template<int DT, u32 NCH> struct DiscGeom
{
  static constexpr int TBYTES = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  static constexpr u32 W = kFastWindow(TBYTES), CH = kFastChunkBytes, NW = (u32)kDiscWalks;
  static constexpr u32 kUnits = NCH * CH / 16;                    // 16-byte units a workgroup owns
  static constexpr u32 kOverhang = (W + 16 + 15) / 16;            // + the next workgroup's first window (walks end on a block start there)
  static constexpr u32 kStageUnits = kUnits + kOverhang;
  static constexpr u32 kBitWords = (W + 31) / 32;
  static constexpr u32 kFoundCap = 512, kHitCap = 512;            // count bytes / block headers found in the workgroup's windows (a few dozen)
  static constexpr u32 kScanWords = (W + 2 + 8 + 3) / 4 + 1;      // dwords of a window that can hold the count byte of a block starting in it
};
// ... and its LDS (a struct, so that the one-launch decoder can lay the three roles' LDS over each other)
template<int DT, u32 NCH, u32 NT> struct DiscShared
{
  typedef DiscGeom<DT, NCH> G;
  alignas(16) u32 in[G::kStageUnits * 4];
  u32 hits[NCH + 1][G::kBitWords];               // window positions where a bit-stuffed block header stands
  u32 heads[NCH][G::kBitWords];                  // ... that are not the block right behind another one
  u16 found[G::kFoundCap], hit[G::kHitCap];      // window (6) << 10 | position
  u32 nFound, nHit;
  u16 fin[NCH][G::NW];
  u32 nFinal[NCH];
  u32 exit[NCH][G::NW];
  u64 fa[NT / 64], fb[NT / 64];
  u32 over;
  u16 cnt[NCH][G::NW];                           // (ONE: the walks' counts on their way to the records)
};

// ONE: a workgroup of the one-launch decoder (k_fast_decode1): what it leaves is read by other workgroups of the SAME launch, on
// other XCDs -- lists, records and checksum terms leave as write-through stores, and a tagged cell per workgroup says "all out"
template<int DT, bool RAG, u32 NCH, u32 NT, bool ONE>
__device__ __forceinline__ void
fastDiscoverBody(DiscShared<DT, NCH, NT>& S, const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols, const FastDecodeBuffers& b, u32 wg)
{
  const RagCounts rc = ragCounts(nRows, nCols);
  typedef DiscGeom<DT, NCH> G;
  constexpr u32 W = G::W, CH = G::CH, NW = G::NW;
  constexpr u32 kWaves = NT / 64, kHeadsPerWave = 64 / NCH;    // walks: lane = (chunk, head), a wave takes kHeadsPerWave heads of every chunk
  constexpr u32 kUnits = G::kUnits, kStageUnits = G::kStageUnits, kBitWords = G::kBitWords;
  constexpr u32 kFoundCap = G::kFoundCap, kHitCap = G::kHitCap, kScanWords = G::kScanWords;
  static_assert((NCH == 32 || NCH == 16 || NCH == 8) && NW == 8 && W + 16 < 1024 && NT >= NCH * NW && W < 1024 && CH + 2 * W < 65536, "lane layout / 16-bit list entries");
  auto& s_in = S.in; auto& s_hits = S.hits; auto& s_heads = S.heads; auto& s_found = S.found; auto& s_hit = S.hit;
  auto& s_nFound = S.nFound; auto& s_nHit = S.nHit; auto& s_final = S.fin; auto& s_nFinal = S.nFinal; auto& s_exit = S.exit;
  auto& s_fa = S.fa; auto& s_fb = S.fb; auto& s_over = S.over;

  PROBE_BEGIN;
  TRACED(wg, 0);
  const int lane = laneId(), w = waveId();
  const u32 c0 = wg * NCH;                         // first chunk of this workgroup
  const u32 r0 = c0 * CH;                                  // blob offset of LDS byte 0

  // ---- the band header first: a caller that only knows the capacity of the blob's buffer (a decode enqueued behind the
  // encode that writes it) launches workgroups for all of it, and those behind the stream's end must not drag a third of a
  // raster's worth of bytes through the chip before they find out
  HeadLite hl;
  if (wg == 0)
  {
    const FastDecodeParams hp = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    if (threadIdx.x == 0) { storeParams<ONE>(b.params, hp); if (b.hostParams) *b.hostParams = hp; }
    hl.ok = hp.ok; hl.version = hp.version; hl.dataBegin = hp.dataBegin; hl.blobEnd = hp.blobEnd;
  }
  else hl = parseHeadLite<DT>(blob, sizeGiven);
  const u32 nChunks = (hl.blobEnd + CH - 1) / CH;
  if (!hl.ok || c0 >= nChunks) return;    // (the grid is sized for the largest stream the blob could hold)
  const int version = (int)hl.version;
  const bool v5 = version >= 5;
  const u32 dataBegin = hl.dataBegin, blobEnd = hl.blobEnd;
  const u32 pattern = v5 ? 14u : 15u;
  PROBE(12);

  // ---- the workgroup's chunks, all loads in flight at once (clipped to what the caller says is readable)
  constexpr int kRounds = (int)((kStageUnits + NT - 1) / NT);
  uint4 x[kRounds];
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * NT + threadIdx.x;
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

  // ---- stage + Fletcher terms of the units this workgroup owns (bytes 14 ... blobEnd - 1 of the blob are checksummed)
  u32 fA = 0;
  u64 fB = 0;
  PROBE(13);
  PROBE_DRAIN;
  PROBE(15);
  const bool inner = r0 != 0u && (u64)r0 + 16ull * kUnits <= blobEnd;    // no unit of this workgroup needs blanking
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * NT + threadIdx.x;
    if (i < kStageUnits) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
    const u32 a = r0 + 16u * i;                                           // (< 2^32: the blob is)
    if (inner)
    {
      if (i < kUnits) fletcherUnit(x[k], (a - 14u) / 2u, fA, fB);         // unit at blob offset a holds words (a - 14) / 2 ...
    }
    else if (i < kUnits && a < blobEnd)
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
  }
  PROBE(14);
  {
    // (no reduction mod 65535 before the sums: a lane holds 9 units, A < 2^23 and B < 2^54 per lane)
    const u64 A = waveSum(fA), B = waveSum(fB);
    if (lane == 0) { s_fa[w] = A; s_fb[w] = B; }
  }
  if (threadIdx.x == 0) s_over = 0u;
  for (u32 i = threadIdx.x; i < (NCH + 1) * kBitWords; i += NT) (&s_hits[0][0])[i] = 0u;
  if (threadIdx.x == 0) { s_nFound = 0u; s_nHit = 0u; }
  __syncthreads();
  PROBE(16);
  TRACED(wg, 1);
  if (threadIdx.x == 0)
  {
    u64 A = 0, B = 0;
#pragma unroll
    for (u32 k = 0; k < kWaves; k++) { A += s_fa[k]; B += s_fb[k]; }
    if (ONE) { publish64(b.waveFletcher + 2 * (size_t)wg, A % 65535u); publish64(b.waveFletcher + 2 * (size_t)wg + 1, B % 65535u); }
    else { b.waveFletcher[2 * (size_t)wg] = A % 65535u; b.waveFletcher[2 * (size_t)wg + 1] = B % 65535u; }
  }

  // ---- bit-stuffed block headers in the first `window` bytes of every chunk (+ the next workgroup's first one).
  // Such a block reads: flag byte (bits 0-1 == 1, bits 6-7 the type of the offset, bit 2 clear from codec 5 on), the
  // offset in that type, then 10?nnnnn (bits per element n != 0, bit 5: look-up table, bits 6-7: the count field
  // is one byte) and the count 64 (Lerc2.cpp:1961-2021, BitStuffer2.cpp:35-77).  "a byte 64 behind a byte 10?nnnnn" is
  // true for one position in a thousand of anything else, so the scan finds the path's blocks almost alone; whatever
  // else it finds dies within a few steps of its walk.  Four positions per lane and step.
  for (u32 f0 = 0; f0 < (NCH + 1) * kScanWords; f0 += NT)
  {
    const u32 f = f0 + threadIdx.x;
    const u32 win = f / kScanWords, d = f - win * kScanWords;
    const u32 chunk = c0 + win;
    const bool scan = win <= NCH && chunk < nChunks && chunk * CH > dataBegin;    // (the chunk that holds the first block: that block only)
    u32 m = 0;
    if (scan)
    {
      const u32 wAt = win * (CH / 4) + d;
      const u32 cur4 = s_in[wAt], prev4 = d ? s_in[wAt - 1] : 0u;
      const u32 hdr4 = __builtin_amdgcn_alignbit(cur4, prev4, 24);        // the bytes in front of cur4's
      u32 isCount = zeroBytes(cur4 ^ 0x40404040u);
      if (RAG) isCount |= zeroBytes(cur4 ^ (rc.cR * 0x01010101u)) | zeroBytes(cur4 ^ (rc.cB * 0x01010101u)) | zeroBytes(cur4 ^ (rc.cC * 0x01010101u));
      m = isCount & zeroBytes((hdr4 & 0xC0C0C0C0u) ^ 0x80808080u) & ~zeroBytes(hdr4 & 0x1F1F1F1Fu);
    }
    // the few count bytes found (one lane in a hundred has any) go to a queue: window (6) << 10 | position
    while (m)
    {
      const u32 j = (u32)(__ffs((int)m) - 1) >> 3;
      m &= m - 1u;
      const u32 at = atomicAdd(&s_nFound, 1u);
      if (at < kFoundCap) s_found[at] = (u16)((win << 10) | (4u * d + j)); else s_over = 1u;
    }
  }
  TRACED(wg, 5);
  __syncthreads();
  TRACED(wg, 6);
  // a count byte stands 2 + (bytes of the offset) behind the block's flag byte: try each offset type, one lane each
  {
    const u32 nFound = min(s_nFound, kFoundCap);
    for (u32 h = threadIdx.x; h < 4u * nFound; h += NT)
    {
      const u32 e = s_found[h >> 2], tc = h & 3u;
      const u32 win = e >> 10, q = e & 0x3FFu;
      const u32 offB = (offBytesTable<DT>() >> (4u * tc)) & 15u;
      if (offB == 0u || q < 2u + offB) continue;
      const u32 p = q - 2u - offB;
      if (p >= W || (c0 + win) * CH + p >= blobEnd) continue;
      const u32 rel = win * CH + p;
      const u32 flag = (s_in[rel >> 2] >> (8u * (rel & 3u))) & 0xFFu;
      if ((flag & 3u) != 1u || (flag >> 6) != tc || (v5 && (flag & 4u))) continue;
      atomicOr(&s_hits[win][p >> 5], 1u << (p & 31u));
      if (win < NCH) { const u32 at = atomicAdd(&s_nHit, 1u); if (at < kHitCap) s_hit[at] = (u16)((win << 10) | p); else s_over = 1u; }
    }
  }
  if (threadIdx.x == 0 && c0 * CH <= dataBegin)    // the stream's first block, whatever it is
  {
    const u32 p = dataBegin - c0 * CH;
    atomicOr(&s_hits[0][p >> 5], 1u << (p & 31u));
    const u32 at = atomicAdd(&s_nHit, 1u);
    if (at < kHitCap) s_hit[at] = (u16)p; else s_over = 1u;
  }
  __syncthreads();
  PROBE(17);
  TRACED(wg, 2);

  // ---- of the blocks found, those that are not the block right behind another one start a walk (the true path crosses
  // a window in several blocks, each of them found)
  for (u32 i = threadIdx.x; i < NCH * kBitWords; i += NT) (&s_heads[0][0])[i] = (&s_hits[0][0])[i];
  if (threadIdx.x < NCH) s_nFinal[threadIdx.x] = 0u;
  __syncthreads();
  constexpr u32 kMaxRel = NCH * CH + W - 1;                               // last staged byte a block may start at
  const u32 nHit = min(s_nHit, kHitCap);
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 e = s_hit[h];
    const u32 hWin = e >> 10, hPos = e & 0x3FFu;
    u32 sg;
    const u32 cur = (c0 + hWin) * CH + hPos;
    const u32 len = stepLean<DT, false, RAG>(s_in, cur - r0, blobEnd - cur, v5, kNoOffset, pattern, sg, rc);
    // (only where the walk from here would pass the block behind: its signature has to follow this one's -- then both
    // walks are the same from there on, and the earlier one lists the later one's blocks)
    const u32 nx = hPos + len;
    const u32 relNx = cur - r0 + len;
    const u32 sgNx = ((s_in[relNx >> 2] >> (8u * (relNx & 3u))) >> 2) & pattern;
    const bool follows = sigOk(sg, sgNx, pattern);
    if (len != 0u && nx < W && follows) atomicAnd(&s_heads[hWin][nx >> 5], ~(1u << (nx & 31u)));
    // what is no block, or is followed by something that cannot be the next block, would end its walk at once: a byte of
    // a block's offset often looks like a flag byte in front of the same header (one such twin per block)
    if (len == 0u || (!follows && cur + len < blobEnd)) atomicAnd(&s_heads[hWin][hPos >> 5], ~(1u << (hPos & 31u)));
  }
  __syncthreads();
  // (walk slots in the order of the heads' positions: the path's head is nearly always the first one, so that the decode
  // kernel can fetch "walk 0 of the chunk" before it has been told which walk it is)
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 e = s_hit[h];
    const u32 hWin = e >> 10, hPos = e & 0x3FFu;
    if ((s_heads[hWin][hPos >> 5] >> (hPos & 31u)) & 1u)
    {
      u32 at = (u32)__popc(s_heads[hWin][hPos >> 5] & ((1u << (hPos & 31u)) - 1u));
      for (u32 k = 0; k < (hPos >> 5); k++) at += (u32)__popc(s_heads[hWin][k]);
      if (at < NW) s_final[hWin][at] = (u16)hPos; else s_over = 1u;
      atomicAdd(&s_nFinal[hWin], 1u);
    }
  }
  PROBE(19);
  __syncthreads();
  PROBE(20);
  TRACED(wg, 3);

  // ---- walks: lane = (chunk, head); the first wave takes the first heads of every chunk (there are seldom more than two).
  // A walk ends on the first block header of the next chunk's window it lands on (or with the blob).
  // (rotating the walking waves over the workgroup's waves, as the encoder's statistics pass does, changes nothing here)
  if ((u32)w < NW / kHeadsPerWave)
  {
    const u32 wc = (u32)lane / kHeadsPerWave, slot = ((u32)lane % kHeadsPerWave) + kHeadsPerWave * (u32)w;    // chunk inside the workgroup, head
    const u32 wChunk = c0 + wc;
    const u32 wStart = wChunk * CH;
    const bool wLive = wChunk < nChunks;
    const u32 wEnd = wLive ? min(wStart + CH, blobEnd) : wStart;
    const bool walker = wLive && slot < min(s_nFinal[wc], NW);
    u16* __restrict__ list = b.lists + ((size_t)wChunk * NW + slot) * kFastListCap;
    const u32* __restrict__ nextHits = s_hits[wc + 1];
    // Inside the chunk.  A lone wave issues an instruction every seven cycles or so whatever it depends on, and this loop
    // is what the kernel waits for, so it is written for few instructions: positions relative to the staged bytes, a walk
    // that is over is a position no chunk reaches (no flag beside it), the signature of the walk's first block stands in
    // for "the block before" at the first step, and "bit-stuffed with a one-byte count of 64 and 1 .. 31 bits" is one
    // range test.  The words of the next block are fetched as soon as its start is known; what is not on the way from
    // one start to the next -- is this a block, does its signature follow, the list -- fills the wait.
    constexpr u32 kOver = 0xFFFFFFFFu;
    const u32 startRel = wc * CH, endRel = wEnd - r0, blobRel = blobEnd - r0;
    const u32 sigStep = (pattern == 14u) ? 2u : 1u;
    u32 rel = walker ? startRel + (u32)s_final[wc][slot] : kOver;
    u32 count = 0;
    u64 acc = 0ull;    // (ONE: the block starts of the current group of four)
    LeanWords<DT> xw = leanFetch<DT>(s_in, min(rel, kMaxRel));
    u32 sig = (__builtin_amdgcn_alignbit(xw.x1, xw.x0, 8u * rel) >> 2) & pattern;
    bool active = rel < endRel;
    while (__builtin_amdgcn_ballot_w64(active) != 0ull)
    {
      const LeanBlock k = leanLength<DT, true, RAG>(xw, rel, blobRel - min(rel, blobRel), rc);
      const u32 behind = rel + k.len;
      const u32 nxt = min(behind, kMaxRel);                               // (a length that is none stays inside the staged bytes)
      xw = leanFetch<DT>(s_in, nxt);
      const u32 sg = (k.h0 >> 2) & pattern, d = (sg - sig) & pattern;
      const bool stuffedOk = RAG ? (((k.t & 0xC0u) == 0x80u) & ((k.t & 31u) != 0u) & rc.allowed((k.t >> 8) & 0xFFu))
                                 : (((k.t & 0xFFDFu) - 0x4081u) <= 30u);
      const bool valid = (((k.h0 & 3u) != 1u) | stuffedOk) & (behind <= blobRel)
        & ((d == 0u) | (d == sigStep) | (sg == 0u));
      const bool ok = active & valid & (count < (u32)kFastListCap);
#ifndef LERC_WALK_NOSTORE
      if (ONE)
      {
        // (four block starts leave together: two-byte write-through stores would each be a memory transaction of their own)
        if (ok) acc |= (u64)(rel - startRel) << (16u * (count & 3u));
        if (ok && (count & 3u) == 3u) { publish64(reinterpret_cast<u64*>(list) + (count >> 2), acc); acc = 0ull; }
      }
      else if (ok) list[count] = (u16)(rel - startRel);
#endif
      rel = active ? (ok ? nxt : kOver) : rel;
      count += ok ? 1u : 0u;
      sig = ok ? sg : sig;
      active = rel < endRel;
    }
    PROBE(22);
    if (w == 0) TRACED(wg, 7);
    bool alive = rel != kOver;
    u32 cur = r0 + rel;                                                   // (absolute from here on: a few steps at most)
    bool tooMany = count == (u32)kFastListCap;                            // (a walk that filled its list: it may have been cut short)
    // behind the chunk: done on a block header of the next window (or at the end of the blob), lost behind that window
    bool landed = false;
    for (;;)
    {
      const u32 past = cur - wEnd;                                        // (alive lanes have cur >= wEnd now)
      const u32 pb = min(past, W - 1u);
      landed = landed || (alive && (cur == blobEnd || (past < W && ((nextHits[pb >> 5] >> (pb & 31u)) & 1u))));
      alive = alive && (landed || past < W);
      const bool active = alive && !landed;
      if (!__any(active)) break;
      u32 sg;
      const u32 len = stepLean<DT, true, RAG>(s_in, min(cur - r0, kMaxRel), blobEnd - min(cur, blobEnd), v5, sig, pattern, sg, rc);
      const bool room = count < (u32)kFastListCap;
      const bool ok = active && len != 0u && room;
      tooMany = tooMany || (active && len != 0u && !room);
      if (ONE)
      {
        if (ok) acc |= (u64)(cur - wStart) << (16u * (count & 3u));
        if (ok && (count & 3u) == 3u) { publish64(reinterpret_cast<u64*>(list) + (count >> 2), acc); acc = 0ull; }
      }
      else if (ok) list[count] = (u16)(cur - wStart);
      alive = alive && (!active || ok);
      cur += ok ? len : 0u;
      count += ok ? 1u : 0u;
      sig = ok ? sg : sig;
    }
    s_exit[wc][slot] = alive ? cur : kNoOffset;
    if (ONE && (count & 3u) != 0u) publish64(reinterpret_cast<u64*>(list) + (count >> 2), acc);
    if (ONE) S.cnt[wc][slot] = alive ? (u16)count : (u16)0xFFFFu;
    else if (wLive) b.recs[wChunk].count[slot] = alive ? (u16)count : (u16)0xFFFFu;
    if (__any(tooMany) && lane == 0) s_over = 1u;
  }
  PROBE(21);
  __syncthreads();
  TRACED(wg, 4);
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
    const u32 ex = (n != 0u && lo == hi) ? lo : kNoOffset;
    if (ONE)
    {
      static_assert(sizeof(FastChunkRec) == 24 && kDiscWalks == 8, "a record is three 8-byte stores");
      u64* dst = reinterpret_cast<u64*>(rec);
      const u16* cn = S.cnt[threadIdx.x];
      publish64(dst, (u64)ex | ((u64)n << 32));
      publish64(dst + 1, (u64)cn[0] | ((u64)cn[1] << 16) | ((u64)cn[2] << 32) | ((u64)cn[3] << 48));
      publish64(dst + 2, (u64)cn[4] | ((u64)cn[5] << 16) | ((u64)cn[6] << 32) | ((u64)cn[7] << 48));
    }
    else { rec->exit = ex; rec->nLive = n; }
  }
  if (threadIdx.x == 0 && s_over) raiseFlag(b, 0);
  if (ONE)
  {
    // everything this workgroup leaves is on its way: wait until it has arrived, then say so
    drainVmem();
    __syncthreads();
    if (threadIdx.x == 0) publish64(b.discCell + wg, ((u64)b.publishEpoch << 32) | 1u);
  }
}

// ------------------------------------------------------------------------------------------------
// resolve
// ------------------------------------------------------------------------------------------------
// One thread per chunk.  Only the exits need agreement (they break the chunk-to-chunk dependency): once the entry of
// a chunk is known, the walk that starts there IS the true path.  The counts are scanned inside the workgroup; the
// gather step adds the sums of the workgroups before its own.  Workgroup 0 also folds the checksum terms.
struct ResolveShared
{
  u32 w[kResolveWG / 64];
  u64 a[kResolveWG / 64], b[kResolveWG / 64];
  u32 base[kResolveWG / 64];
};
// (ONE: a resolving block of the one-launch decoder.  It reads the band header itself, waits for the discovery workgroups of
// its chunks -- and of the chunk in front of them -- to say "all out", and reads what they left past the L2; the LAST
// resolving block folds the checksum terms: it is the one that has, by way of the totals in front of it, waited for everybody)
template<int DT, bool ONE>
__device__ __forceinline__ void fastResolveBody(ResolveShared& S, const FastDecodeBuffers& b, u32 nWavesBound, u32 discChunks, u32 group,
                                                const u8* __restrict__ blob = nullptr, u32 sizeGiven = 0, int nRows = 0, int nCols = 0)
{
  auto& s_w = S.w; auto& s_a = S.a; auto& s_b = S.b; auto& s_base = S.base;
  constexpr u32 kDiscPer = (u32)kOneDiscChunks;
  const FastDecodeParams hp = ONE ? parseBandHeader<DT>(blob, sizeGiven, nRows, nCols) : *b.params;
  const u32 blobEnd = hp.blobEnd;
  const u32 c = group * kResolveChunks + threadIdx.x;
  const u32 cPrev = c ? c - 1u : 0u;
  u32 prevExit = 0;
  if (!ONE) prevExit = b.recs[cPrev].exit;    // (the records first: their addresses do not hang on the header)
  if (!hp.ok || group * kResolveChunks >= hp.nChunks) return;    // (the grid is sized for the largest stream the blob could hold)
  const int lane = laneId(), w = waveId();
  if (ONE)
  {
    // the discovery workgroups of chunks [first - 1, last]: one thread each
    const u32 first = group * kResolveChunks, last = min(first + kResolveChunks, hp.nChunks) - 1u;
    const u32 d0 = (first ? first - 1u : 0u) / kDiscPer, d1 = last / kDiscPer;
    static_assert(kResolveChunks / kOneDiscChunks + 2 <= kResolveWG, "one thread per discovery workgroup");
    const u32 d = d0 + threadIdx.x;
    const bool mineD = d <= d1;
    u64 cell = mineD ? observe64(b.discCell + d) : 0ull;
    for (u32 spin = 0; mineD && (u32)(cell >> 32) != b.epoch && spin < b.spinLimit; spin++)
    {
      __builtin_amdgcn_s_sleep(8);
      cell = observe64(b.discCell + d);
    }
    if (mineD && (u32)(cell >> 32) != b.epoch) raiseFlag(b, 1);    // (gave up waiting: never seen; the general path takes the band)
    __syncthreads();
    prevExit = (u32)observe64(reinterpret_cast<const u64*>(b.recs + cPrev));
  }
  u32 count = 0, laneOfPath = kNoOffset;
  bool bad = false;
  const bool mine = threadIdx.x < kResolveChunks && c < hp.nChunks;
  if (mine)
  {
    const u32 chunkStart = c * kFastChunkBytes, chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
    const u32 e = (chunkStart <= hp.dataBegin) ? hp.dataBegin : prevExit;
    if (e == kNoOffset || e < chunkStart) bad = true;
    else if (e >= chunkEnd) bad = (e != blobEnd);    // the last block may begin before the last chunk and end with it
    else
    {
      FastChunkRec rec;
      if (ONE)
      {
        const u64* src = reinterpret_cast<const u64*>(b.recs + c);
        const u64 r0 = observe64(src), r1 = observe64(src + 1), r2 = observe64(src + 2);
        rec.exit = (u32)r0; rec.nLive = (u32)(r0 >> 32);
#pragma unroll
        for (int l = 0; l < 4; l++) { rec.count[l] = (u16)(r1 >> (16 * l)); rec.count[4 + l] = (u16)(r2 >> (16 * l)); }
      }
      else rec = b.recs[c];
      const u32 rel = e - chunkStart;
      // the walk the entry lies on: normally a walk starts there; else it is one of the first blocks of a walk that
      // began a little earlier (something in front of the entry that looks like a block ending right there)
#pragma unroll
      for (int l = 0; l < kDiscWalks; l++)
      {
        if (rec.count[l] == 0xFFFFu) continue;
        const u16* lp = b.lists + ((size_t)c * kDiscWalks + l) * kFastListCap;    // its first four block starts
        uint2 pre;
        if (ONE) { const u64 v = observe64(reinterpret_cast<const u64*>(lp)); pre = make_uint2((u32)v, (u32)(v >> 32)); }
        else pre = *reinterpret_cast<const uint2*>(lp);
        const u32 st[4] = { pre.x & 0xFFFFu, pre.x >> 16, pre.y & 0xFFFFu, pre.y >> 16 };
#pragma unroll
        for (u32 k = 0; k < 4; k++)
          if (st[k] == rel && k < (u32)rec.count[l] && laneOfPath == kNoOffset) { laneOfPath = (u32)l | (k << 8); count = rec.count[l] - k; }
      }
      if (laneOfPath == kNoOffset || rec.exit == kNoOffset) bad = true;    // (no agreement on the exit: the next chunk says so too)
    }
    if (bad) { count = 0; laneOfPath = kNoOffset; }
    // which walk and how many blocks: out at once (the decode workgroups gather the block starts with it while this block
    // still waits for the totals in front of it); epoch (32) | walk (16, 0xFFFF: none) | count (16)
    publish64(b.chunkCell + 2 * (size_t)c + 1, ((u64)b.publishEpoch << 32) | ((u64)(laneOfPath & 0xFFFFu) << 16) | (count & 0xFFFFu));
  }
  if (__any(bad) && lane == 0) raiseFlag(b, 1);
  // exclusive scan of the counts inside the workgroup
  u32 inc = count;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  u32 before = 0;
  for (int i = 0; i < w; i++) before += s_w[i];
  // this block's total goes out at once; the blocks in front of it -- dispatched earlier, waiting for nobody before they
  // publish theirs -- are read in one go (a cell is read by the few hundred resolving blocks behind it, no more)
  const u32 epoch = b.epoch;
  const u32 nGroups = (hp.nChunks + kResolveChunks - 1u) / kResolveChunks;
  if (threadIdx.x == kResolveWG - 1) publish64(b.groupCell + group, ((u64)b.publishEpoch << 32) | (before + inc));
  u32 base = 0;
  for (u32 g0 = 0; g0 < group; g0 += kResolveWG)
  {
    const u32 g = g0 + threadIdx.x;
    u64 cell = g < group ? observe64(b.groupCell + g) : 0ull;
    for (u32 spin = 0; __any(g < group && (u32)(cell >> 32) != epoch) && spin < b.spinLimit; spin++)    // (never that long)
    {
      __builtin_amdgcn_s_sleep(4);
      if (g < group && (u32)(cell >> 32) != epoch) cell = observe64(b.groupCell + g);
    }
    if (g < group && (u32)(cell >> 32) != epoch) raiseFlag(b, 1);    // (gave up waiting: never seen; the general path takes the band)
    base += g < group ? (u32)cell : 0u;
  }
  base = waveSum(base);
  __syncthreads();    // (s_w has been read)
  if (lane == 0) s_base[w] = base;
  __syncthreads();
  base = 0;
  for (u32 i = 0; i < kResolveWG / 64; i++) base += s_base[i];
  if (mine) publish64(b.chunkCell + 2 * (size_t)c, ((u64)b.publishEpoch << 32) | (base + before + inc - count));
  // the chunks hold all the raster's blocks, or the band goes the long way
  if (group == nGroups - 1u && threadIdx.x == kResolveWG - 1 && base + before + inc != hp.nBlocks) raiseFlag(b, 2);

  if (group != (ONE ? nGroups - 1u : 0u)) return;
  // checksum: Fletcher32 over blob[14 ..) from the discovery waves' partial sums (Lerc2.cpp:1037-1064)
  const u32 perWave = ONE ? (u32)kOneDiscChunks : discChunks;
  const u32 nWaves = min((hp.nChunks + perWave - 1u) / perWave, nWavesBound);
  u64 A = 0, B = 0;
  for (u32 i = threadIdx.x; i < nWaves; i += kResolveWG)    // each < 65535
  {
    if (ONE) { A += observe64(b.waveFletcher + 2 * (size_t)i); B += observe64(b.waveFletcher + 2 * (size_t)i + 1); }
    else { A += b.waveFletcher[2 * (size_t)i]; B += b.waveFletcher[2 * (size_t)i + 1]; }
  }
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
  const u32 good = ((u32)((s2 << 16) | s1) == hp.expectChecksum) ? 1u : 0u;
  if (ONE) publish32(&b.params->checksumOk, good); else b.params->checksumOk = good;
  if (b.hostParams) b.hostParams->checksumOk = good;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------

// A workgroup decodes the blocks that start in kDecodeChunks consecutive chunks: their bytes lie in a range known
// beforehand (the chunks, plus the blocks a walk passed behind its chunk before it landed), so the bytes, the chunks'
// counts and -- as soon as it is known which walk of a chunk is the path -- the walks' lists travel together; block i
// of chunk c is block base(c) + i of the raster (base: the scan the resolve step left in pieces).
template<class T, bool RAG> struct DecodeShared
{
  static constexpr u32 CH = kFastChunkBytes, CPD = kDecodeChunks, CAP = (u32)kFastListCap;
  static constexpr u32 TB = (u32)sizeof(T), RAW = 1 + 64 * TB, W = kFastWindow((int)sizeof(T));
  static constexpr u32 kStageUnits = (CPD * CH + W + RAW + 16 + 15) / 16;
  static constexpr u32 kMaxBlocks = CPD * CAP;
  alignas(16) u32 in[kStageUnits * 4];
  alignas(16) u16 spec[CPD][CAP];      // the list of each chunk's walk 0, fetched before anybody knows which walk is the path
  double offs[kMaxBlocks];
  u32 code[kMaxBlocks];                // parseCode of the block, 0 = bad
  u32 at[kMaxBlocks];                  // raster offset (pixels) of the block's first pixel, ~0: no such block
  u32 n[CPD + 1], first[CPD], lane[CPD], bad;
  u16 pos[kMaxBlocks + 1];             // block starts relative to the workgroup's first byte
  u8 dims[RAG ? kMaxBlocks : 1];       // RAG: width | height << 4 of each block (8 x 8 but for the raster's last block column / row)
};

// (ONE: a decoding workgroup of the one-launch decoder: it reads the band header itself and, once its chunks' cells are
// there, what the discovery workgroups left past the L2 -- not before: a line read too early would stay in this XCD's L2)
template<class T, bool RAG, bool ONE>
__device__ __forceinline__ void
fastDecodeBody(DecodeShared<T, RAG>& S, const FastDecodeBuffers& b, const u8* __restrict__ blob, T* __restrict__ outPix, u32 wgIndex,
               u32 sizeGiven = 0, int nRows = 0, int nCols = 0)
{
  constexpr int DT = DtOf<T>::v;
  const FastDecodeParams hp = ONE ? parseBandHeader<DT>(blob, sizeGiven, nRows, nCols) : *b.params;
  const u32 raised0 = ONE ? ~b.epoch : b.fallback[0];    // (read together with the parameters: one round trip, not two)
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  typedef DecodeShared<T, RAG> G;
  constexpr u32 CH = G::CH, CPD = G::CPD, CAP = G::CAP, kStageUnits = G::kStageUnits;
  auto& s_in = S.in; auto& s_pos = S.pos; auto& s_code = S.code; auto& s_at = S.at; auto& s_offs = S.offs;
  auto& s_n = S.n; auto& s_first = S.first; auto& s_lane = S.lane; auto& s_bad = S.bad; auto& s_dims = S.dims; auto& s_spec = S.spec;
  const u32 blobEnd = hp.blobEnd, epoch = b.epoch;
  const struct { int nCols, version; double invScale, zMaxHdr; } p = { (int)hp.nCols, (int)hp.version, hp.invScale, hp.zMaxHdr };
  PROBE_BEGIN;
  TRACED(8192u + wgIndex, 0);
  const int w = waveId(), lane = laneId();
  const u32 c0 = wgIndex * CPD;
  // not ours, given up by an earlier kernel, or behind the stream's end (the grid is sized for the largest stream the blob could hold)
  if (!hp.ok || raised0 == epoch || c0 >= hp.nChunks) return;
  const u32 r0 = c0 * CH;

  // ---- everything whose address is known goes out at once: the bytes ...
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
      if (a + 16 <= blobEnd) x[k] = *reinterpret_cast<const uint4*>(blob + a);
      else if (a < blobEnd)    // never read past the blob
      {
        u32 t4[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (u32 q = 0; q < 16; q++) if (a + q < blobEnd) t4[q >> 2] |= (u32)blob[a + q] << (8 * (q & 3));
        x[k] = make_uint4(t4[0], t4[1], t4[2], t4[3]);
      }
    }
  }
  // ... walk 0's lists ...
  static_assert(CPD * CAP * 2 / 16 <= 256, "one 16-byte load per thread");
  // (ONE: past the L2, so that a list that is not there yet leaves nothing behind in it.  Such a list is only used by a workgroup
  // that found its cells at the first look -- they are written microseconds after the lists have arrived -- and whatever is
  // used is checked block by block against the stream: first start == the chunk's entry, every block ends where the next begins)
  if (threadIdx.x < CPD * CAP / 8)
  {
    const u32 q = threadIdx.x / (CAP / 8), part = threadIdx.x % (CAP / 8);
    const u16* src = b.lists + ((size_t)(c0 + q) * kDiscWalks) * kFastListCap + 8u * part;    // (chunks behind the last one: the buffer's slack)
    uint4 l;
    if (ONE)
    {
      const u64 lo = observe64(reinterpret_cast<const u64*>(src)), hi = observe64(reinterpret_cast<const u64*>(src) + 1);
      l = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32));
    }
    else l = *reinterpret_cast<const uint4*>(src);
    *reinterpret_cast<uint4*>(&s_spec[q][8u * part]) = l;
  }
  // ... and the chunks' cells: which walk is the path, how many blocks, the index of the first one (the resolving blocks of
  // this launch leave them; only the workgroups of the launch's first round ever wait)
  u64 firstCell = 0;    // (threads 0 .. CPD - 1: the chunk's other cell, asked for in the same breath and looked at further down)
  if (threadIdx.x < CPD)
  {
    const u32 c = c0 + threadIdx.x;
    u32 n = 0, ln = 0;
    if (c < hp.nChunks)
    {
      u64 cell = observe64(b.chunkCell + 2 * (size_t)c + 1);
      firstCell = observe64(b.chunkCell + 2 * (size_t)c);
      const bool waited = (u32)(cell >> 32) != epoch;
      for (u32 spin = 0; (u32)(cell >> 32) != epoch && spin < b.spinLimit; spin++)    // (never that long: the resolving blocks were dispatched first)
      {
        __builtin_amdgcn_s_sleep(ONE ? 16 : 4);
        cell = observe64(b.chunkCell + 2 * (size_t)c + 1);
      }
      if ((u32)(cell >> 32) != epoch) { raiseFlag(b, 3); cell = 0xFFFFull << 16; }    // (gave up waiting: never seen)
      n = (u32)cell & 0xFFFFu; ln = ((u32)cell >> 16) & 0xFFFFu;
      if (ln == 0xFFFFu) { n = 0; ln = 0; }    // (no path through this chunk: the resolving block has raised the flag)
      if (ONE && waited) ln |= 0x4000u;         // (the list fetched at the start may have been fetched too early)
    }
    s_n[threadIdx.x] = n; s_lane[threadIdx.x] = ln;
  }
  if (threadIdx.x == 0) s_bad = 0u;
  __syncthreads();
  const u32 before = 0u;    // (the cells hold raster indices)
  PROBE(8);
  TRACED(8192u + wgIndex, 1);
  // ---- the lists of the walks that are the path: flat index f = blocks of chunk 0, then of chunk 1, ...
  u32 nAll = 0, cum[CPD + 1];
#pragma unroll
  for (u32 q = 0; q < CPD; q++) { cum[q] = nAll; nAll += min(s_n[q], CAP); }
  cum[CPD] = nAll;
  for (u32 f = threadIdx.x; f < nAll; f += 256u)
  {
    u32 q = 0;
#pragma unroll
    for (u32 k = 1; k < CPD; k++) q += (f >= cum[k]) ? 1u : 0u;
    const u32 i = f - cum[q], ls = s_lane[q];      // walk | index of the chunk's first block in its list << 8
    const u32 src = min(((ls >> 8) & 3u) + i, CAP - 1u);
    const u16* lp = b.lists + ((size_t)(c0 + q) * kDiscWalks + (ls & 7u)) * kFastListCap + src;
    u32 v;
    if ((ls & 0x4007u) == 0u) v = (u32)s_spec[q][src];
    else if (ONE) v = (u32)(observe64(reinterpret_cast<const u64*>(lp - (src & 3u))) >> (16u * (src & 3u))) & 0xFFFFu;    // (written in groups of four)
    else v = (u32)*lp;
    s_pos[f] = (u16)(q * CH + v);
  }
  // (the end of the last block: the next chunk's entry, which is the exit its walks agreed on)
  if (threadIdx.x == 0 && nAll)
  {
    u32 last = CPD - 1;
    while (last > 0 && s_n[last] == 0u) last--;    // (a last chunk in which no block starts has no walks and no exit)
    const u32 ex = ONE ? (u32)observe64(reinterpret_cast<const u64*>(b.recs + c0 + last)) : b.recs[c0 + last].exit;
    if (ONE)    // the first block starts at its chunk's entry: the exit of the chunk in front (the stream's first block: behind the header)
    {
      u32 qf = 0;
      while (qf < CPD - 1 && s_n[qf] == 0u) qf++;
      const u32 entry = (c0 + qf == 0u) ? hp.dataBegin : (u32)observe64(reinterpret_cast<const u64*>(b.recs + c0 + qf - 1u));
      if (entry != r0 + (u32)s_pos[0]) s_bad = 1u;
    }
    s_pos[nAll] = (u16)min(ex - min(ex, r0), 0xFFFFu);
  }
  // ---- stage the bytes
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * 256u + threadIdx.x;
    if (i < kStageUnits) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
  }
  // ... and, by now, where the chunks' blocks lie in the raster
  if (threadIdx.x < CPD)
  {
    const u32 c = c0 + threadIdx.x;
    u32 first = 0;
    if (c < hp.nChunks)
    {
      u64 cell = firstCell;    // (asked for at the start; only the workgroups of the launch's first round find it missing)
      for (u32 spin = 0; (u32)(cell >> 32) != epoch && spin < b.spinLimit; spin++)
      {
        __builtin_amdgcn_s_sleep(4);
        cell = observe64(b.chunkCell + 2 * (size_t)c);
      }
      if ((u32)(cell >> 32) != epoch) raiseFlag(b, 3);    // (gave up waiting: never seen)
      first = (u32)cell;
    }
    s_first[threadIdx.x] = first;
  }
  __syncthreads();
  PROBE(9);
  TRACED(8192u + wgIndex, 2);

  // ---- parse the block headers once: lane = block
  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  constexpr u32 kMaxRel = kStageUnits * 16 - 16;
  const bool pow2 = (hp.nTH & (hp.nTH - 1u)) == 0u;
  const u32 thShift = 31u - (u32)__clz((int)hp.nTH);
  for (u32 f = threadIdx.x; f < nAll; f += 256u)
  {
    u32 q = 0;
#pragma unroll
    for (u32 k = 1; k < CPD; k++) q += (f >= cum[k]) ? 1u : 0u;
    const u32 blk = before + s_first[q] + (f - cum[q]);          // index of the block in the raster
    const u32 off = min((u32)s_pos[f], kMaxRel);
    u32 h0, h1, h2;
    ldsHeader<DT>(s_in, off, h0, h1, h2);
    const u32 it = pow2 ? (blk >> thShift) : blk / hp.nTH, jt = blk - it * hp.nTH;
    u32 bw = 8u, bh = 8u;
    if (RAG)
    {
      bw = min(8u, hp.nCols - 8u * min(jt, hp.nTH - 1u)); bh = min(8u, hp.nRows - 8u * min(it, (hp.nRows + 7u) / 8u - 1u));
      s_dims[f] = (u8)(bw | (bh << 4));
    }
    u32 code = parseCode<DT>(h0, h1, h2, p.version, bw * bh);
    if ((u32)s_pos[f] + codeLen(code) != (u32)s_pos[f + 1]) code = 0;      // the blocks tile the stream
    if (((h0 >> 2) & pattern) != (jt & pattern)) code = 0;                // signature = (j0 >> 3) & pattern, j0 = 8 jt
    if (blk >= hp.nBlocks) code = 0;
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
    // bit 30: not even the largest value nb bits can hold reaches the header's zMax, so the pixels need no clamp
    // (the dequantiser is monotone in q)
    if (code && mode == 1)
    {
      const u32 qTop = codeBits(code) >= 32u ? 0xFFFFFFFFu : ((1u << codeBits(code)) - 1u);
      const bool below = (DT >= DT_Float) ? (offset + (double)qTop * p.invScale < p.zMaxHdr)
                                          : ((i64)offset + (i64)qTop * (i64)p.invScale < (i64)p.zMaxHdr);
      if (below) code |= 1u << 30;
    }
    s_offs[f] = offset;
    s_code[f] = code;
    s_at[f] = code ? (it * 8u) * (u32)p.nCols + jt * 8u : kNoOffset;
    if (code == 0u) s_bad = 1u;
  }
  __syncthreads();
  PROBE(10);
  TRACED(8192u + wgIndex, 3);

  // ---- pixels: a wave takes BPW blocks at a time, a lane V consecutive pixels of one raster row of one block
  const int r = lane >> 3, c = lane & 7, bb = c / LPR, h = c % LPR;
  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  bool bad = false;
  // (wave tiles lie on multiples of BPW blocks of the RASTER, not of this workgroup's first block: a tile row is then a
  // whole 128-byte line of the output, written by one instruction; only the first and last tile may be partial)
  const u32 blk0 = before + s_first[0];            // the workgroup's blocks are blk0 ... blk0 + nAll - 1
  const u32 g1 = (blk0 + nAll + BPW - 1) / BPW;
  for (u32 g = blk0 / BPW + (u32)w; g < g1; g += 4u)
  {
    const u32 blk = g * BPW + (u32)bb;
    const u32 f = blk - blk0;                      // (wraps for the blocks in front of the workgroup's first one)
    const bool have = blk >= blk0 && f < nAll;
    const u32 code = have ? s_code[f] : 0u;
    const double offset = have ? s_offs[f] : 0.0;
    const u32 at0 = have ? s_at[f] : kNoOffset;
    const u32 mode = codeMode(code), lut = codeLut(code), offB = codeOffBytes(code);
    const u32 pbit = 8u * ((have ? (u32)s_pos[f] : 0u) + ((mode == 1u) ? 3u + offB + lut : 1u));    // payload / first raw value
    // RAG: the block is bw x bh pixels, this lane holds the first vc of its V (or none), element = row-major index among them
    int bw = 8, vc = V;
    if (RAG)
    {
      const u32 dims = have ? (u32)s_dims[f] : 0x88u;
      bw = (int)(dims & 15u);
      vc = r < (int)(dims >> 4) ? max(0, min(V, bw - h * V)) : 0;
    }
    const bool rowsAligned = !RAG || (((size_t)p.nCols * sizeof(T)) & 15u) == 0u;    // (else: no 16-byte stores)
    const int e0 = r * bw + h * V;
    T v[V];
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = T(0);
    // the common case, all blocks of the wave alike: bit-stuffed without a table, the lane's V values inside 64 bits, no
    // clamp -- three words of the stream, one funnel shift each way, V shifts
    const bool plain = mode == 1u && !lut && (code >> 30) != 0u && (u32)V * codeBits(code) <= 64u && (!RAG || (vc == V && bw == 8 && rowsAligned));
    if (__all(plain || !code))
    {
      if (code)
      {
        const u32 nb = codeBits(code);
        const u32 bit0 = pbit + (u32)e0 * nb, wi = bit0 >> 5;
        const u32 x0 = s_in[wi], x1 = s_in[wi + 1], x2 = s_in[wi + 2];
        const u64 all = ((u64)__builtin_amdgcn_alignbit(x2, x1, bit0) << 32) | __builtin_amdgcn_alignbit(x1, x0, bit0);
        const u32 mask = nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
        const i64 offI = (i64)offset;
#pragma unroll
        for (int k = 0; k < V; k++)
        {
          const u32 q = (u32)(all >> ((u32)k * nb)) & mask;
          if (DT >= DT_Float) v[k] = (T)(offset + (double)q * p.invScale);    // Lerc2.cpp:2159-2160, no contraction
          else if (sizeof(T) <= 4) v[k] = (T)((u32)offI + q * (u32)invI);    // (the low 32 bits of the sum are the pixel: no 64-bit product)
            else v[k] = (T)(offI + (i64)q * invI);
        }
        struct alignas(sizeof(T) * V) Vec { T e[V]; };
        Vec o;
#pragma unroll
        for (int k = 0; k < V; k++) o.e[k] = v[k];
        DECODE_STORE(reinterpret_cast<Vec*>(outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V)), o);
      }
    }
    else if (code)
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
            if (ix > nLut) { ix = 0; bad = bad || !RAG || k < vc; }    // the reference would read outside its table here (RAG: pixels that do not exist have no index)
            const u32 q = ix ? ldsBits(s_in, pbit + (ix - 1) * (u32)nb, nb) : 0u;
            v[k] = dequant<T>(offset, q, p.invScale, p.zMaxHdr, offI, invI, zMaxI);
          }
        }
      }
      if (RAG && !(vc == V && rowsAligned))
      {
        T* dst = outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V);
#pragma unroll
        for (int k = 0; k < V; k++) if (k < vc) dst[k] = v[k];
      }
      else
      {
        struct alignas(sizeof(T) * V) Vec { T e[V]; };
        Vec o;
#pragma unroll
        for (int k = 0; k < V; k++) o.e[k] = v[k];
        DECODE_STORE(reinterpret_cast<Vec*>(outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V)), o);
      }
    }
  }
  PROBE(11);
  TRACED(8192u + wgIndex, 4);
  if ((__any(bad) && lane == 0) || (threadIdx.x == 0 && s_bad)) raiseFlag(b, 3);
}

// ------------------------------------------------------------------------------------------------
u32 fastTestGiveUp()
{
  static const u32 bits = []() -> u32 { const char* e = getenv("LERC_AMD_TEST_GIVEUP"); return e ? (u32)strtoul(e, nullptr, 0) : 0u; }();
  return bits;
}

bool fastDecodeEligible(int dt, int version, int mb, int nRows, int nCols, int nDepth, bool allValid)
{
  if (!allValid || nDepth != 1 || mb != 8 || version < 3) return false;
  if (dt == DT_Char || dt == DT_Byte) return false;
  if (!fastDimsOkRagged(nRows, nCols)) return false;    // (rows / columns need not be multiples of 8)
  return true;
}

FastWalkPlan makeFastWalkPlan(int nRows, int nCols, u32 sizeGiven, u32 nTiles)
{
  // upper bounds from what the caller knows without reading the blob
  FastWalkPlan wp;
  wp.nChunks = ((sizeGiven ? sizeGiven : 1u) + kFastChunkBytes - 1) / kFastChunkBytes;
  wp.nBlocks = (u32)((nRows + 7) / 8) * (u32)((nCols + 7) / 8);
  // a batch of small blobs, a grid of its own per tile: where workgroups of half as many chunks leave fewer chunk slots empty in
  // every tile's last workgroup they win (8 100 tiles of 51 chunks: 4 x 16 against 7 x 8 slots, discovery 460 -> 410 us); one large
  // raster is 8 % faster with the full ones
  const u32 full = (u32)kDiscChunks, half = full / 2u;
  const u32 slotsFull = (wp.nChunks + full - 1) / full * full, slotsHalf = (wp.nChunks + half - 1) / half * half;
  static const int force = []() { const char* e = getenv("LERC_AMD_DISC_CHUNKS"); return e ? atoi(e) : 0; }();    // tuning knob: 8 or 16
  wp.discChunks = (nTiles > 1 && slotsHalf < slotsFull) ? half : full;
  if (force == (int)half || force == (int)full) wp.discChunks = (u32)force;
  wp.nWaves = (wp.nChunks + wp.discChunks - 1) / wp.discChunks;
  return wp;
}

// ------------------------------------------------------------------------------------------------
// kernels: blockIdx.y = tile of a batch (one raster: a batch of 1).  Each tile has its own slice of every buffer.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tileSlice(FastDecodeBuffers& b, const FastDecodeBatch& t, const u8*& blob, u32& sizeGiven, u32 tileIndex)
{
  const size_t tile = tileIndex;
  const size_t sChunk = fastChunkStride(t.nChunks);
  b.params += tile; b.fallback += 4 * tile;
  b.recs += tile * t.nChunks; b.lists += tile * t.nChunks * (size_t)(kDiscWalks * kFastListCap);
  b.chunkCell += tile * 2 * sChunk;
  b.groupCell += tile * fastGroupStride(t.nChunks);
  b.waveFletcher += tile * 2 * (size_t)t.nWaves;
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
}

template<int DT, bool RAG, int NCH>
__global__ void __launch_bounds__(NCH * kDiscThreads / kDiscChunks)
k_fast_discover(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, int nRows, int nCols)
{
  constexpr u32 NT = (u32)(NCH * kDiscThreads / kDiscChunks);
  tileSlice(b, t, blob, sizeGiven, blockIdx.y);
  __shared__ DiscShared<DT, (u32)NCH, NT> sm;
  fastDiscoverBody<DT, RAG, (u32)NCH, NT, false>(sm, blob, sizeGiven, nRows, nCols, b, blockIdx.x);
}
// The first blocks of the launch resolve (kResolveChunks chunks each; all tiles' resolving blocks first, so that a batch's decode
// workgroups find the cells of their tile ready like those of a single raster do), the others decode (kDecodeChunks chunks each).
template<class T, bool RAG>
__global__ void __launch_bounds__(256)
k_fast_decode(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, T* __restrict__ outPix)
{
  static_assert(kResolveWG == 256, "a resolving block is a block of this launch");
  const u32 nResolve = (t.nChunks + kResolveChunks - 1u) / kResolveChunks, nDecode = (t.nChunks + kDecodeChunks - 1u) / kDecodeChunks;
  const bool resolving = blockIdx.x < t.nTiles * nResolve;
  const u32 rest = resolving ? blockIdx.x : blockIdx.x - t.nTiles * nResolve, per = resolving ? nResolve : nDecode;
  const u32 tile = rest / per, index = rest - tile * per;
  u32 sizeGiven = 0;
  tileSlice(b, t, blob, sizeGiven, tile);
  __shared__ union Sm { ResolveShared r; DecodeShared<T, RAG> x; } sm;
  if (resolving) fastResolveBody<DtOf<T>::v, false>(sm.r, b, t.nWaves, t.discChunks, index);
  else fastDecodeBody<T, RAG, false>(sm.x, b, blob, outPix + (size_t)tile * t.tileElems, index);
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
    {
      const bool rag = nRows % 8 != 0 || nCols % 8 != 0, half = t.discChunks != (u32)kDiscChunks;
      const dim3 grid(t.nWaves, nT), block(half ? kDiscThreads / 2 : kDiscThreads);
      if (rag && half) hipLaunchKernelGGL((k_fast_discover<DT, true, kDiscChunks / 2>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols);
      else if (rag) hipLaunchKernelGGL((k_fast_discover<DT, true, kDiscChunks>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols);
      else if (half) hipLaunchKernelGGL((k_fast_discover<DT, false, kDiscChunks / 2>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols);
      else hipLaunchKernelGGL((k_fast_discover<DT, false, kDiscChunks>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols);
      break;
    }
    default:
      if (nRows % 8 != 0 || nCols % 8 != 0)
        hipLaunchKernelGGL((k_fast_decode<T, true>), dim3(nT * ((t.nChunks + kResolveChunks - 1) / kResolveChunks + (t.nChunks + kDecodeChunks - 1) / kDecodeChunks)),
                           dim3(256), 0, st, b, t, blob, (T*)out);
      else
        hipLaunchKernelGGL((k_fast_decode<T, false>), dim3(nT * ((t.nChunks + kResolveChunks - 1) / kResolveChunks + (t.nChunks + kDecodeChunks - 1) / kDecodeChunks)),
                           dim3(256), 0, st, b, t, blob, (T*)out);
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
