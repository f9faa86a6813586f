// tile_fast_decode_one.hip -- the streaming decoder in ONE launch: every workgroup finds the block starts of its own piece of
// the blob AND decodes them.  Same results as tile_fast_decode.hip's two launches and as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends).  The two-launch form (tile_fast_decode.hip) reads
// the blob twice -- once to find the block starts (k_fast_discover: lists of starts per chunk go to memory), once to decode
// (k_fast_decode, behind a resolving step that turns per-chunk counts into block indices).  Here a workgroup of 512 threads
//   1. stages 32 KiB of the blob (+ the first window behind them) in LDS as 32 sub-chunks of 1 KiB, summing the Fletcher32
//      terms of its own sub-chunks on the way -- all but the first: that one is the last sub-chunk of the workgroup in front,
//      staged and walked again here, so that the entry of this workgroup's first own sub-chunk (the exit all live walks of
//      the sub-chunk in front agree on) is known without asking anybody; 1/32 of the discovery work done twice buys a launch,
//      a pass over the blob, and the lists' way through memory;
//   2. finds the bit-stuffed block headers in every sub-chunk's first window by their byte pattern ("a byte 64 behind a byte
//      10?nnnnn": a lane takes a 16-byte unit of a window) and lets the heads among them walk, lane = sub-chunk x head --
//      k_fast_discover's steps on chunks half as long, so a walk is ten steps, and walk 0's list of block starts stays in LDS
//      (sub-chunks of 512 bytes: walks of 2 us instead of 3.5, but twice the windows to scan: 149 against 119 us; of 2 KiB: 122);
//   3. settles, per own sub-chunk, which walk is the path and how many blocks belong to it, publishes the workgroup's block
//      count in an epoch-tagged cell, and closes the lists up into one flat list of block starts;
//   4. adds up the cells of the workgroups in front of it (those of its group of 64, and one cell per group in front, left by
//      each group's last workgroup): the raster index of its first block -- the only thing it ever waits for;
//   5. parses the block headers (lane = block: signature, contiguity, count -- ReadTile's integrity checks), then every lane
//      extracts V consecutive pixels of one raster row, dequantises and stores one 16-byte vector, 512 blocks per round.
// The launch's last workgroup waits for all checksum terms (fire-and-forget atomics, one accumulator per 64 workgroups), folds
// them and compares with the header's checksum.
// Streams the walks cannot follow raise an epoch tagged flag, as in the two-launch form; the host then repeats the band with
// the general kernels.
// Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540; Lerc2.cpp:1037-1064 (checksum).
#include "tile_fast_decode_dev.h"
#include <cstddef>

namespace lerc {

#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time lines (constant-rate counter), read by tools/trace_decode_one.py
static __device__ unsigned long long g_traceO[16 * 8192];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace_decode_one(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceO), sizeof(unsigned long long) * (size_t)n); }
#define TRACEO(slot) do { if (threadIdx.x == 0 && wg < 8192u) g_traceO[16 * wg + (slot)] = wall_clock64(); } while (0)
#else
#define TRACEO(slot)
#endif

template<class T> struct OneGeom
{
  static constexpr int DT = DtOf<T>::v;
  static constexpr u32 TB = (u32)sizeof(T), W = kFastWindow((int)sizeof(T)), CH = fastOneSub((int)sizeof(T));
  static constexpr u32 NCH = kOneStage / CH, NW = (u32)kDiscWalks, NT = kOneThreads, CAP = CH / 8u < 128u ? CH / 8u : 128u;    // (list cap: blocks of 8 bytes on average; the last block row of a 257 x 257 tile has 14-byte blocks)
  static constexpr u32 kUnits = kOneStage / 16;                   // 16-byte units of the staged sub-chunks
  static constexpr u32 kOverhang = (W + 16 + 15) / 16;            // + the next sub-chunk's first window (walks end on a block start there)
  static constexpr u32 kStageUnits = kUnits + kOverhang;
  static constexpr u32 kBitWords = (W + 31) / 32;
  static constexpr u32 kFoundCap = 768, kHitCap = 768;            // count bytes / block headers found in the windows (a few hundred)
  static constexpr u32 R = NT;                                    // blocks per decode round: one header per thread
  static constexpr u32 kMaxRel = NCH * CH + W - 1;                // last staged byte a block may start at
  static constexpr u32 kWalkWaves = (NCH * NW + 63u) / 64u;          // walks: thread = sub-chunk + NCH * head
  static constexpr u32 kChShift = CH == 512u ? 9u : CH == 1024u ? 10u : 11u;
};

template<class T, bool RAG> struct OneShared
{
  typedef OneGeom<T> G;
  alignas(16) u32 in[G::kStageUnits * 4];
  alignas(16) u16 list[G::NCH * G::CAP + 8];       // walk 0's block starts per sub-chunk (relative to the staged bytes); later: the starts of all own blocks, flat
  union U
  {
    struct D                                       // steps 2 - 3
    {
      u32 hits[G::NCH + 1][G::kBitWords];          // window positions where a bit-stuffed block header stands
      alignas(16) u16 fin[G::NCH][G::NW];          // window position of each walk's head
      union P
      {
        struct A                                   // ... until the walks start
        {
          u32 removed[G::NCH][G::kBitWords];       // headers that are the block right behind another one (or no block at all): no heads
          u32 strong[G::NCH][G::kBitWords];        // heads that are followed, exactly where they end, by another header found
          u16 found[G::kFoundCap], hit[G::kHitCap];    // count bytes / block headers found: position in the staged bytes
        } a;
        struct B                                   // ... from the walks on
        {
          alignas(16) u16 mini[G::NCH][G::NW][4];  // the first four block starts of every walk
          alignas(16) u16 exit[G::NCH][G::NW];     // where a walk landed (relative to the staged bytes), 0xFFFF: it did not
          alignas(8) u8 cnt[G::NCH][G::NW];        // blocks a walk passed; 0xFF: no such walk / it ran into something that is no block
          u32 key[G::NCH];                         // the path: walk * 4 + index of the entry among its first four blocks (the least wins)
        } b;
      } p;
    } d;
    struct X                                       // step 5, one round
    {
      double offs[G::R];
      u32 code[G::R];                              // parseCode of the block, 0 = bad
      u32 at[G::R];                                // raster offset (pixels) of the block's first pixel
      u8 dims[RAG ? G::R : 1];                     // RAG: width | height << 4
    } x;
  } u;
  u32 nFinal[G::NCH];
  u32 agreed[G::NCH];                              // the exit all live walks of a sub-chunk agree on (blob offset), or ~0
  u32 count[G::NCH];                               // blocks of the path that belong to own sub-chunk q
  u32 path[G::NCH];                                // walk | index of the sub-chunk's first block in its list << 8, or ~0
  u32 cum[G::NCH + 1];
  u64 fa[G::NT / 64], fb[G::NT / 64];
  u64 part;                                        // sum of the cells: this group's in the low half, the groups' in front in the high half
  u32 nFound, nHit, over, bad, rewalk, lost, maxCount;
  FastDecodeParams hp;                             // the band header, parsed in full by the first wave
};

// the exit all live walks of a sub-chunk agree on (relative to the staged bytes), or 0xFFFF; e: the eight walks' exits
__device__ __forceinline__ u32 agreedExit(const uint4& e)
{
  const u32 wds[4] = { e.x, e.y, e.z, e.w };
  u32 lo = 0xFFFFu, hi = 0u, n = 0u;
#pragma unroll
  for (int k = 0; k < 4; k++)
  {
    const u32 a = wds[k] & 0xFFFFu, b = wds[k] >> 16;
    if (a != 0xFFFFu) { lo = min(lo, a); hi = max(hi, a); n++; }
    if (b != 0xFFFFu) { lo = min(lo, b); hi = max(hi, b); n++; }
  }
  return (n != 0u && lo == hi) ? lo : 0xFFFFu;
}

template<class T, bool RAG>
__device__ __forceinline__ void
fastOneBody(OneShared<T, RAG>& S, const FastDecodeBuffers& b, const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols,
            T* __restrict__ outPix, u32 wg)
{
  typedef OneGeom<T> G;
  constexpr int DT = G::DT;
  constexpr u32 W = G::W, CH = G::CH, NCH = G::NCH, NW = G::NW, NT = G::NT, CAP = G::CAP, SH = G::kChShift;
  constexpr u32 kWaves = NT / 64;
  constexpr u32 kUnits = G::kUnits, kStageUnits = G::kStageUnits, kBitWords = G::kBitWords;
  constexpr u32 kFoundCap = G::kFoundCap, kHitCap = G::kHitCap, kMaxRel = G::kMaxRel;
  static_assert(NCH <= 64 && NCH * NW <= NT && NT % 64 == 0 && NT % NCH == 0 && CAP % (NT / NCH) == 0 && NW == 8 && W <= CH && NCH * CH + 2 * W < 65535 && CAP <= 254, "lane layout / 16-bit positions / 8-bit counts");
  static_assert(NCH * NW <= NT, "a thread per walk when the path is picked");
  auto& s_in = S.in; auto& s_hits = S.u.d.hits; auto& s_removed = S.u.d.p.a.removed; auto& s_strong = S.u.d.p.a.strong;
  auto& s_found = S.u.d.p.a.found; auto& s_hit = S.u.d.p.a.hit; auto& s_fin = S.u.d.fin;
  auto& s_mini = S.u.d.p.b.mini; auto& s_exit = S.u.d.p.b.exit; auto& s_cnt = S.u.d.p.b.cnt; auto& s_key = S.u.d.p.b.key;

  const RagCounts rc = ragCounts(nRows, nCols);
  const int lane = laneId(), w = waveId();
  // ---- the band header: every wave reads what the front part needs of it (three loads of the same address in all lanes);
  // the first wave reads it in full -- Lerc2::ReadHeader's checks -- while the staged bytes are on their way, and leaves the
  // result in LDS (workgroup 0: also where the host wants to see it)
  const HeadLite hl = parseHeadLite<DT>(blob, sizeGiven);
  const u32 blobEnd = hl.blobEnd;
  const u32 nSub = (blobEnd + CH - 1u) >> SH;      // sub-chunk c = blob bytes [c * CH, (c + 1) * CH)
  const u32 cs = wg * (NCH - 1u);                  // first staged sub-chunk; own: cs + 1 ... cs + NCH - 1 (workgroup 0: cs too)
  const bool ours = hl.ok && headLiteEligible<DT>(blob, hl.version, nRows, nCols);
  if (wg == 0u && !ours && threadIdx.x == 0)      // (not a band of ours: say so)
  {
    const FastDecodeParams hp0 = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    storeParams<true>(b.params, hp0); if (b.hostParams) *b.hostParams = hp0;
  }
  if (!ours || (wg != 0u && cs + 1u >= nSub)) return;    // (the grid is sized for the largest stream the blob could hold)
  TRACEO(0);
  const u32 nWG = fastOneNumWG(blobEnd, (int)sizeof(T));
  const u32 r0 = cs << SH;                         // blob offset of LDS byte 0
  const u32 qOwn0 = wg ? 1u : 0u;
  const int version = (int)hl.version;
  const bool v5 = version >= 5;
  const u32 dataBegin = hl.dataBegin;
  const u32 pattern = v5 ? 14u : 15u;
  const u32 epoch = b.epoch;
  const u64 tag = (u64)b.publishEpoch << 32;

  // ---- the staged bytes, all loads in flight at once (clipped to what the caller says is readable)
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
  if (w == 0)
  {
    const FastDecodeParams hpFull = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
    if (lane == 0)
    {
      S.hp = hpFull;
      if (wg == 0u)
      {
        // (the verdict on the checksum comes from the launch's last workgroup, microseconds later for a small blob: one writer
        // per byte -- the host's copy, which travels over PCIe, gets everything BUT that word here (the host has zeroed it);
        // the device's copy is complete before this thread sends its checksum terms off, which the last workgroup waits for)
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
  for (u32 i = threadIdx.x; i < (NCH + 1) * kBitWords; i += NT) (&s_hits[0][0])[i] = 0u;
  for (u32 i = threadIdx.x; i < NCH * kBitWords; i += NT) { (&s_removed[0][0])[i] = 0u; (&s_strong[0][0])[i] = 0u; }
  if (threadIdx.x < NCH) S.nFinal[threadIdx.x] = 0u;
  if (threadIdx.x == 0) { S.nFound = 0u; S.nHit = 0u; S.over = 0u; S.bad = 0u; S.rewalk = 0u; S.lost = 0u; S.part = 0ull; S.maxCount = 0u; }
  __syncthreads();
  if (!S.hp.ok) return;    // (not a band the streaming kernels take: the header says so in full only)

  // ---- stage; Fletcher terms of the units of the own sub-chunks (bytes 14 ... blobEnd - 1 of the blob are checksummed)
  u32 fA = 0;
  u64 fB = 0;
  const u32 ownUnit0 = qOwn0 * (CH / 16u);
  const bool inner = r0 != 0u && (u64)r0 + 16ull * kUnits <= blobEnd;    // no unit of this workgroup needs blanking
#pragma unroll
  for (int k = 0; k < kRounds; k++)
  {
    const u32 i = (u32)k * NT + threadIdx.x;
    if (i < kStageUnits) *reinterpret_cast<uint4*>(&s_in[i * 4]) = x[k];
    const u32 a = r0 + 16u * i;                                           // (< 2^32: the blob is)
    if (inner)
    {
      if (i >= ownUnit0 && i < kUnits) fletcherUnit(x[k], (a - 14u) / 2u, fA, fB);    // unit at blob offset a holds words (a - 14) / 2 ...
    }
    else if (i >= ownUnit0 && i < kUnits && a < blobEnd)
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
  {
    // (no reduction mod 65535 before the sums: a lane holds 5 units, A < 2^23 and B < 2^54 per lane)
    const u64 A = waveSum(fA), B = waveSum(fB);
    if (lane == 0) { S.fa[w] = A; S.fb[w] = B; }
  }
  __syncthreads();
  TRACEO(1);
  if (threadIdx.x == 0)
  {
    // this workgroup's checksum terms: one atomic nobody waits for (the launch's last workgroup folds the accumulators)
    u64 A = 0, B = 0;
#pragma unroll
    for (u32 k = 0; k < kWaves; k++) { A += S.fa[k]; B += S.fb[k]; }
    A %= 65535u; B %= 65535u;
    if (wg == 0u) drainVmem();    // (the band's parameters have arrived)
    __hip_atomic_fetch_add(b.wgAcc + wg / kOneGroup, A | (B << 24) | (1ull << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ---- bit-stuffed block headers in every sub-chunk's first window.  Such a block reads: flag byte (bits 0-1 == 1, bits 6-7
  // the type of the offset, bit 2 clear from codec 5 on), the offset in that type, then 10?nnnnn (bits per element n != 0,
  // bit 5: look-up table, bits 6-7: the count field is one byte) and the count 64 (Lerc2.cpp:1961-2021, BitStuffer2.cpp:35-77).
  // "a byte 64 behind a byte 10?nnnnn" is true for one position in a thousand of anything else.  A lane takes a 16-byte unit of
  // a window (only the windows' units: every lane works); the count byte stands 2 + (bytes of the offset) behind the flag
  // byte, so each one found is tried with every offset type.
  constexpr u32 UPW = (W + 12 + 15) / 16;    // units of a window that can hold the count byte of a block starting in it
  static_assert(UPW <= G::kOverhang && UPW * 16u <= CH, "the window behind the staged sub-chunks is staged as far as it is scanned");
  for (u32 u0 = 0; u0 < (NCH + 1) * UPW; u0 += NT)
  {
    const u32 u = u0 + threadIdx.x;
    const u32 win = u / UPW, k = u - win * UPW, chunk = cs + win;
    if (win > NCH || chunk >= nSub || (chunk << SH) <= dataBegin) continue;    // (the sub-chunk that holds the first block: that block only)
    const u32 i = win * (CH / 16u) + k;
    const uint4 cu = *reinterpret_cast<const uint4*>(&s_in[4u * i]);
    u32 pv = i ? s_in[4u * i - 1u] : 0u;
    const u32 wd[4] = { cu.x, cu.y, cu.z, cu.w };
#pragma unroll
    for (u32 j = 0; j < 4; j++)
    {
      const u32 cur4 = wd[j];
      const u32 hdr4 = __builtin_amdgcn_alignbit(cur4, pv, 24);        // the bytes in front of cur4's
      pv = cur4;
      u32 isCount = zeroBytes(cur4 ^ 0x40404040u);
      if (RAG) isCount |= zeroBytes(cur4 ^ (rc.cR * 0x01010101u)) | zeroBytes(cur4 ^ (rc.cB * 0x01010101u)) | zeroBytes(cur4 ^ (rc.cC * 0x01010101u));
      // 0x80 in every byte of hdr4 that reads 10?nnnnn, n != 0: bit 7 set and bit 6 clear; the low five bits + 31 reach bit 5
      const u32 m10 = hdr4 & ~(hdr4 << 1), mN = ((hdr4 & 0x1F1F1F1Fu) + 0x1F1F1F1Fu) << 2;
      u32 m = isCount & m10 & mN;
      // the few count bytes found (one lane in six has any) go to a queue
      while (m)
      {
        const u32 q = 16u * i + 4u * j + ((u32)(__ffs((int)m) - 1) >> 3);
        m &= m - 1u;
        const u32 at = atomicAdd(&S.nFound, 1u);
        if (at < kFoundCap) s_found[at] = (u16)q; else S.over = 1u;
      }
    }
  }
  __syncthreads();
  // a count byte stands 2 + (bytes of the offset) behind the block's flag byte: try each offset type, one lane each
  {
    const u32 nFound = min(S.nFound, kFoundCap);
    for (u32 h = threadIdx.x; h < 4u * nFound; h += NT)
    {
      const u32 q = s_found[h >> 2], tc = h & 3u;
      const u32 offB = (offBytesTable<DT>() >> (4u * tc)) & 15u;
      if (offB == 0u || q < 2u + offB) continue;
      const u32 p = q - 2u - offB;                                       // the flag byte (relative to the staged bytes)
      const u32 win = p >> SH, pw = p & (CH - 1u);
      if (pw >= W || r0 + p >= blobEnd || ((cs + win) << SH) <= dataBegin) continue;
      const u32 flag = (s_in[p >> 2] >> (8u * (p & 3u))) & 0xFFu;
      if ((flag & 3u) != 1u || (flag >> 6) != tc || (v5 && (flag & 4u))) continue;
      atomicOr(&s_hits[win][pw >> 5], 1u << (pw & 31u));
      if (win < NCH) { const u32 at = atomicAdd(&S.nHit, 1u); if (at < kHitCap) s_hit[at] = (u16)p; else S.over = 1u; }
    }
  }
  if (threadIdx.x == 0 && r0 <= dataBegin)    // the stream's first block, whatever it is
  {
    const u32 p = dataBegin - r0;
    atomicOr(&s_hits[0][p >> 5], 1u << (p & 31u));
    const u32 at = atomicAdd(&S.nHit, 1u);
    if (at < kHitCap) s_hit[at] = (u16)p; else S.over = 1u;
  }
  __syncthreads();
  TRACEO(2);

  // ---- of the blocks found, those that are not the block right behind another one start a walk (the true path crosses a
  // window in several blocks, each of them found)
  const u32 nHit = min(S.nHit, kHitCap);
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 p = s_hit[h];
    const u32 hWin = p >> SH, hPos = p & (CH - 1u);
    u32 sg;
    const u32 cur = r0 + p;
    const u32 len = stepLean<DT, false, RAG>(s_in, p, blobEnd - cur, v5, kNoOffset, pattern, sg, rc);
    // (only where the walk from here would pass the block behind: its signature has to follow this one's -- then both
    // walks are the same from there on, and the earlier one lists the later one's blocks)
    const u32 nx = hPos + len;
    const u32 relNx = min(p + len, kMaxRel);
    const u32 sgNx = ((s_in[relNx >> 2] >> (8u * (relNx & 3u))) >> 2) & pattern;
    const bool follows = sigOk(sg, sgNx, pattern);
    if (len != 0u && nx < W && follows)
    {
      atomicOr(&s_removed[hWin][nx >> 5], 1u << (nx & 31u));
      // (a header found exactly where this block ends: this one is on the path, as good as certainly -- such heads get the
      // first walk slots of their sub-chunk, and walk 0 is the one whose list is kept)
      if ((s_hits[hWin][nx >> 5] >> (nx & 31u)) & 1u) atomicOr(&s_strong[hWin][hPos >> 5], 1u << (hPos & 31u));
    }
    // what is no block, or is followed by something that cannot be the next block, would end its walk at once: a byte of
    // a block's offset often looks like a flag byte in front of the same header (one such twin per block)
    if (len == 0u || (!follows && cur + len < blobEnd)) atomicOr(&s_removed[hWin][hPos >> 5], 1u << (hPos & 31u));
  }
  __syncthreads();
  // walk slots: the heads that are followed by a header first, in the order of their positions, then the others
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 p = s_hit[h];
    const u32 hWin = p >> SH, hPos = p & (CH - 1u);
    const u32 bit = 1u << (hPos & 31u), wd = hPos >> 5;
    if (s_hits[hWin][wd] & ~s_removed[hWin][wd] & bit)
    {
      const bool strong = (s_strong[hWin][wd] & bit) != 0u;
      u32 before = 0, nStrong = 0;
      for (u32 k = 0; k < kBitWords; k++)
      {
        const u32 hd = s_hits[hWin][k] & ~s_removed[hWin][k], st = hd & s_strong[hWin][k], mine = strong ? st : (hd & ~st);
        nStrong += (u32)__popc(st);
        before += (k < wd) ? (u32)__popc(mine) : (k == wd) ? (u32)__popc(mine & (bit - 1u)) : 0u;
      }
      const u32 at = strong ? before : nStrong + before;
      if (at < NW) s_fin[hWin][at] = (u16)hPos; else S.over = 1u;
      atomicAdd(&S.nFinal[hWin], 1u);
    }
  }
  __syncthreads();
  TRACEO(3);

  // ---- walks: lane = sub-chunk, a wave takes one head of every sub-chunk (there are seldom more than two, so the later
  // waves are through at once).  A walk ends on the first block header of the next window it lands on (or with the blob).
  // Written for few instructions, see k_fast_discover.
  if (threadIdx.x < NCH) s_key[threadIdx.x] = 0xFFFFFFFFu;
  if ((u32)w < G::kWalkWaves)
  {
    static_assert((NCH * NW) % 64u == 0u, "whole waves of walks");
    const u32 wc = threadIdx.x % NCH, slot = threadIdx.x / NCH;
    const u32 wChunk = cs + wc;
    const u32 wStart = wChunk << SH;
    const bool wLive = wChunk < nSub;
    const u32 wEnd = wLive ? min(wStart + CH, blobEnd) : wStart;
    const bool walker = wLive && slot < min(S.nFinal[wc], NW);
    // (walk 0 keeps all its block starts, the others their first four: enough to tell which walk the entry lies on)
    if (__any(walker))
    {
    u16* __restrict__ lst = slot == 0u ? &S.list[wc * CAP] : &s_mini[wc][slot][0];
    const u32 keep = slot == 0u ? CAP : 4u;
    const u32* __restrict__ nextHits = s_hits[wc + 1];
    constexpr u32 kOver = 0xFFFFFFFFu;
    const u32 startRel = wc << SH, endRel = wEnd - r0, blobRel = blobEnd - r0;
    const u32 sigStep = (pattern == 14u) ? 2u : 1u;
    u32 rel = walker ? startRel + (u32)s_fin[wc][slot] : kOver;
    u32 count = 0;
    LeanWords<DT> xw = leanFetch<DT>(s_in, min(rel, kMaxRel));
    u32 sig = (__builtin_amdgcn_alignbit(xw.x1, xw.x0, 8u * rel) >> 2) & pattern;
    bool active = rel < endRel;
#if !defined(HIPSIM) && defined(LERC_ONE_WALK_PRIO)
    __builtin_amdgcn_s_setprio(3);    // (the walks are what the workgroup waits for; a walking wave issues an instruction every few cycles)
#endif
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
      const bool ok = active & valid & (count < CAP);
      if (ok && count < keep) lst[count] = (u16)rel;
      rel = active ? (ok ? nxt : kOver) : rel;
      count += ok ? 1u : 0u;
      sig = ok ? sg : sig;
      active = rel < endRel;
    }
    bool alive = rel != kOver;
    u32 cur = r0 + rel;                                                   // (absolute from here on: a few steps at most)
    bool tooMany = count == CAP;                                          // (a walk that filled its list: it may have been cut short)
    // behind the sub-chunk: done on a block header of the next window (or at the end of the blob), lost behind that window
    bool landed = false;
    for (;;)
    {
      const u32 past = cur - wEnd;                                        // (alive lanes have cur >= wEnd now)
      const u32 pb = min(past, W - 1u);
      landed = landed || (alive && (cur == blobEnd || (past < W && ((nextHits[pb >> 5] >> (pb & 31u)) & 1u))));
      alive = alive && (landed || past < W);
      const bool act = alive && !landed;
      if (!__any(act)) break;
      u32 sg;
      const u32 len = stepLean<DT, true, RAG>(s_in, min(cur - r0, kMaxRel), blobEnd - min(cur, blobEnd), v5, sig, pattern, sg, rc);
      const bool room = count < CAP;
      const bool ok = act && len != 0u && room;
      tooMany = tooMany || (act && len != 0u && !room);
      if (ok && count < keep) lst[count] = (u16)(cur - r0);
      alive = alive && (!act || ok);
      cur += ok ? len : 0u;
      count += ok ? 1u : 0u;
      sig = ok ? sg : sig;
    }
#if !defined(HIPSIM) && defined(LERC_ONE_WALK_PRIO)
    __builtin_amdgcn_s_setprio(0);
#endif
    s_exit[wc][slot] = alive ? (u16)(cur - r0) : (u16)0xFFFFu;
    s_cnt[wc][slot] = alive ? (u8)count : (u8)0xFFu;
    if (__any(tooMany) && lane == 0) S.over = 1u;
    }
    else { s_exit[wc][slot] = (u16)0xFFFFu; s_cnt[wc][slot] = (u8)0xFFu; }    // (no sub-chunk has that many heads)
  }
  __syncthreads();
  TRACEO(4);

  // ---- what all live walks of a sub-chunk agree on is true without knowing which one is real: the entry of sub-chunk q is
  // the agreed exit of q - 1; the walk that starts there (or passes it with one of its first four blocks: something in
  // front of the entry that looks like a block ending right there) is the path.  A thread per walk looks; the least wins.
  // (Where the walks of q - 1 do NOT agree -- a walk of half a dozen steps is short enough for a stray one to survive now
  // and then -- the entry of q is where the PATH of q - 1 ends, known once that is: such sub-chunks wait for the serial
  // step below.  Only the first staged sub-chunk of a workgroup behind the first, whose path nobody here knows, needs the
  // agreement.)
  if (threadIdx.x < NCH * NW)
  {
    const u32 q = threadIdx.x / NW, l = threadIdx.x % NW, chunk = cs + q;
    if (q >= qOwn0 && chunk < nSub)
    {
      const u32 chunkStart = chunk << SH, chunkEnd = min(chunkStart + CH, blobEnd);
      u32 e = dataBegin;
      if (chunkStart > dataBegin)
      {
        const u32 ex = agreedExit(*reinterpret_cast<const uint4*>(&s_exit[q ? q - 1u : 0u][0]));
        e = ex == 0xFFFFu ? kNoOffset : r0 + ex;
      }
      const u32 c = s_cnt[q][l];
      if (e != kNoOffset && e >= chunkStart && e < chunkEnd && c != 0xFFu)
      {
        const u32 rel = e - r0;
        const uint2 st2 = *reinterpret_cast<const uint2*>(l == 0u ? &S.list[q * CAP] : &s_mini[q][l][0]);
        const u32 st[4] = { st2.x & 0xFFFFu, st2.x >> 16, st2.y & 0xFFFFu, st2.y >> 16 };
        u32 k0 = 4u;
#pragma unroll
        for (u32 k = 4; k-- > 0; ) if (k < c && st[k] == rel) k0 = k;
        if (k0 < 4u) atomicMin(&s_key[q], l * 4u + k0);
      }
    }
  }
  __syncthreads();
  // the walk of sub-chunk q that starts at e or passes it with one of its first four blocks (one thread looking at all walks)
  auto findKey = [&](u32 q, u32 e) -> u32
  {
    u32 key = 0xFFFFFFFFu;
    if (e == kNoOffset || e < r0) return key;
    const u32 rel = e - r0;
    for (u32 l = NW; l-- > 0; )
    {
      const u32 c = s_cnt[q][l];
      if (c == 0xFFu) continue;
      const u16* st = l == 0u ? &S.list[q * CAP] : &s_mini[q][l][0];
      for (u32 k = 4; k-- > 0; ) if (k < c && (u32)st[k] == rel) key = l * 4u + k;
    }
    return key;
  };
  // per own sub-chunk: entry e (a blob offset, or none) -> the path, its blocks, where it ends; false: the band goes the long way
  auto settle = [&](u32 q, u32 e, u32 key, u32& count, u32& path, u32& pathExit) -> bool
  {
    const u32 chunk = cs + q;
    const u32 chunkStart = chunk << SH, chunkEnd = min(chunkStart + CH, blobEnd);
    count = 0; path = kNoOffset; pathExit = kNoOffset;
    if (e == kNoOffset || e < chunkStart) return false;
    if (e >= chunkEnd) { pathExit = e; return e == blobEnd; }    // the last block may begin before the last sub-chunk and end with it
    if (key == 0xFFFFFFFFu) return false;
    const u32 ex = (u32)s_exit[q][key >> 2];
    if (ex == 0xFFFFu) return false;
    path = (key >> 2) | ((key & 3u) << 8); count = (u32)s_cnt[q][key >> 2] - (key & 3u); pathExit = r0 + ex;
    return true;
  };
  if (threadIdx.x < NCH)
  {
    const u32 q = threadIdx.x, chunk = cs + q;
    u32 count = 0, path = kNoOffset, pathExit = kNoOffset;
    bool bad = false;
    if (q >= qOwn0 && chunk < nSub)
    {
      u32 e = dataBegin, keyOwn = 0xFFFFFFFFu;
      bool pending = false;
      if ((chunk << SH) > dataBegin)
      {
        // the walks of the sub-chunk in front: agreed, or several exits (then: where its path ends, if it has one to be known)
        const uint4 pe = *reinterpret_cast<const uint4*>(&s_exit[q ? q - 1u : 0u][0]);
        const u32 ex = agreedExit(pe);
        e = ex == 0xFFFFu ? kNoOffset : r0 + ex;
        const bool any = (pe.x & pe.y & pe.z & pe.w) != 0xFFFFFFFFu;
        pending = ex == 0xFFFFu && any && q > qOwn0;
        if (ex == 0xFFFFu && any && q == qOwn0)
        {
          // the walks of the sub-chunk in front of this workgroup's own ones do not agree: the workgroup in front, whose own
          // sub-chunk it is, knows where its path ends and says so in its cell (one workgroup in a few hundred gets here)
          const u64* p = b.wgCell + (wg - 1u);
          u64 c = observe64(p);
          for (u32 spin = 0; (u32)(c >> 32) != epoch && spin < b.spinLimit; spin++)
          {
            __builtin_amdgcn_s_sleep(8);
            c = observe64(p);
          }
          const u32 exn = ((u32)c >> 16) & 0xFFFFu;
          if ((u32)(c >> 32) == epoch && exn != 0xFFFFu) { e = r0 + exn; keyOwn = findKey(q, e); }
        }
      }
      if (pending) { path = 0xFFFFFFFEu; S.rewalk = 1u; }    // (settled by the serial step)
      else bad = !settle(q, e, keyOwn != 0xFFFFFFFFu ? keyOwn : s_key[q], count, path, pathExit);
      if (bad) { count = 0; path = kNoOffset; }
    }
    else if (chunk < nSub)    // the first staged sub-chunk of a workgroup behind the first: its walks have to agree
    {
      const u32 own = agreedExit(*reinterpret_cast<const uint4*>(&s_exit[q][0]));
      pathExit = own != 0xFFFFu ? r0 + own : kNoOffset;
    }
    S.agreed[q] = pathExit;
    S.count[q] = count; S.path[q] = path;
    if (path < 0xFFFFFFFEu && ((path & 0xFFu) != 0u || b.testRewalk)) S.rewalk = 1u;    // (test knob: every path is walked again)
    if (bad) S.bad = 1u;
  }
  __syncthreads();
  if (S.rewalk)
  {
    // the serial step: sub-chunks behind walks that did not agree, front to back
    if (threadIdx.x == 0)
    {
      for (u32 q = qOwn0 + 1u; q < NCH; q++)
      {
        if (S.path[q] != 0xFFFFFFFEu) continue;
        const u32 e = S.agreed[q - 1u];    // where the path of the sub-chunk in front ends (none: it has none)
        const u32 key = findKey(q, e);
        u32 count, path, pathExit;
        if (!settle(q, e, key, count, path, pathExit)) { S.bad = 1u; count = 0; path = kNoOffset; pathExit = kNoOffset; }
        S.agreed[q] = pathExit; S.count[q] = count; S.path[q] = path;
      }
    }
    __syncthreads();
  }
  // ---- a path that is not walk 0 (a stray header in front of the path's first block of the window that happened to be
  // followed by a header itself: one sub-chunk in thousands) is walked again from the entry, with all checks and with its list
  if (S.rewalk)
  {
    if (threadIdx.x < NCH)
    {
      const u32 q = threadIdx.x, path = S.path[q];
      if (path < 0xFFFFFFFEu && ((path & 0xFFu) != 0u || b.testRewalk))
      {
        const u32 chunk = cs + q;
        const u32 e = ((chunk << SH) <= dataBegin) ? dataBegin : S.agreed[q ? q - 1u : 0u];
        const u32 endRel = S.agreed[q] - r0;
        u32 cur = e - r0, sig = kNoOffset, n = 0;
        bool good = true;
        while (cur < endRel && n < CAP)
        {
          u32 sg;
          const u32 len = stepLean<DT, false, RAG>(s_in, min(cur, kMaxRel), blobEnd - min(r0 + cur, blobEnd), v5, sig, pattern, sg, rc);
          if (len == 0u) { good = false; break; }
          S.list[q * CAP + n] = (u16)cur;
          n++; cur += len; sig = sg;
        }
        if (!good || cur != endRel || n != S.count[q]) { S.bad = 1u; n = 0; }
        S.count[q] = n; S.path[q] = 0u;
      }
    }
    __syncthreads();
  }
  // ---- this workgroup's blocks: how many (out at once), and their starts closed up into one list
  if (threadIdx.x < 64u)
  {
    const u32 c = (u32)lane < NCH ? S.count[lane] : 0u;
    u32 inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
    if ((u32)lane < NCH) S.cum[lane] = inc - c;
    const u32 cMax = waveMax(c);
    if ((u32)lane == NCH - 1u)
    {
      S.cum[NCH] = inc;
      S.maxCount = cMax;
      // (with where the path of the last sub-chunk ends, relative to the first staged byte of the workgroup behind: that one
      // asks for it if the walks it runs over the same sub-chunk do not agree)
      const u32 ex = S.agreed[NCH - 1u], rNext = r0 + (NCH - 1u) * CH;
      const u32 exNext = (ex != kNoOffset && ex >= rNext && ex - rNext < 0xFFFFu) ? ex - rNext : 0xFFFFu;
      publish64(b.wgCell + wg, tag | ((u64)exNext << 16) | (u64)inc);
    }
  }
  // the cells of the workgroups in front -- those of this group, and one per group in front -- are asked for now and looked at
  // when the lists are closed up: they publish when this one does, or did long ago
  const u32 grp = wg / kOneGroup, g0 = grp * kOneGroup, nIn = wg - g0;
  const u64* cellAt = threadIdx.x < nIn ? b.wgCell + g0 + threadIdx.x : b.wgGroupCell + (threadIdx.x - nIn);
  u64 cell0 = 0;
  if (threadIdx.x < nIn + grp) cell0 = observe64(cellAt);
  // closing up: a sub-chunk's list is moved by NT / NCH threads, entries j, j + NT / NCH, ... each (sub-chunks hold half a
  // dozen blocks, so one turn as a rule); everything is read before anything is written -- the lists move in place
  constexpr u32 kLanesPer = NT / NCH, kTurns = CAP / kLanesPer;
  const u32 cq = threadIdx.x / kLanesPer, cj = threadIdx.x % kLanesPer;
  const u32 cCount = S.count[cq], cFrom = cq * CAP + ((S.path[cq] >> 8) & 3u);
  u16 keep[kTurns];
  __syncthreads();    // (S.maxCount)
  const u32 turns = (S.maxCount + kLanesPer - 1u) / kLanesPer;
#pragma unroll
  for (u32 e = 0; e < kTurns; e++)
  {
    keep[e] = 0;
    if (e < turns)
    {
      const u32 i = cj + kLanesPer * e;
      if (i < cCount) keep[e] = S.list[min(cFrom + i, NCH * CAP - 1u)];
    }
  }
  __syncthreads();
  const u32 total = S.cum[NCH];
  {
    const u32 cTo = S.cum[cq];
#pragma unroll
    for (u32 e = 0; e < kTurns; e++)
      if (e < turns)
      {
        const u32 i = cj + kLanesPer * e;
        if (i < cCount) S.list[cTo + i] = keep[e];
      }
  }
  if (threadIdx.x == 0 && total)
  {
    // (the end of the last block: the exit the walks of the last sub-chunk a block belongs to agreed on)
    u32 last = NCH - 1u;
    while (last > 0u && S.count[last] == 0u) last--;
    const u32 ex = S.agreed[last];
    S.list[total] = (u16)min(ex - min(ex, r0), 0xFFFFu);
  }
  TRACEO(5);

  // ---- rounds of at most R blocks (cut on multiples of BPW blocks of the RASTER, like the wave tiles below)
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  const FastDecodeParams hp = S.hp;
  const struct { int nCols, version; double invScale, zMaxHdr; } p = { (int)hp.nCols, (int)hp.version, hp.invScale, hp.zMaxHdr };
  const bool pow2 = (hp.nTH & (hp.nTH - 1u)) == 0u;
  const u32 thShift = 31u - (u32)__clz((int)hp.nTH);
  const int r = lane >> 3, c = lane & 7, bb = c / LPR, h = c % LPR;
  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  auto& s_offs = S.u.x.offs; auto& s_code = S.u.x.code; auto& s_at = S.u.x.at; auto& s_dims = S.u.x.dims;
  bool bad = false;
  // a block's header, lane = block.  What does not hang on where the block lies in the raster (PRE: length, mode, bits, offset,
  // "the blocks tile the stream") and what does (POST: the signature, the block's place; RAG: its size, and so everything)
  auto parseHeader = [&](u32 f, u32 t, u32 base, bool pre, bool post)
  {
    const u32 pos = (u32)S.list[f], off = min(pos, kMaxRel);
    u32 code = 0, sigHdr = 0;
    if (pre)
    {
      u32 h0, h1, h2;
      ldsHeader<DT>(s_in, off, h0, h1, h2);
      u32 n = 64u;
      if (RAG)
      {
        const u32 blk = base + f;
        const u32 it = pow2 ? (blk >> thShift) : blk / hp.nTH, jt = blk - it * hp.nTH;
        const u32 bw = min(8u, hp.nCols - 8u * min(jt, hp.nTH - 1u)), bh = min(8u, hp.nRows - 8u * min(it, (hp.nRows + 7u) / 8u - 1u));
        s_dims[t] = (u8)(bw | (bh << 4));
        n = bw * bh;
      }
      code = parseCode<DT>(h0, h1, h2, p.version, n);
      if (pos + codeLen(code) != (u32)S.list[f + 1] || pos > kMaxRel) code = 0;    // the blocks tile the stream
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
      // what the pixel loop wants to know of the block, in one word: where its payload begins (16: byte among the staged ones;
      // the first raw value of a raw block), bits per value (5) << 16, mode (2) << 21, look-up table << 23, "plain" << 24 --
      // bit-stuffed without a table, a lane's V values inside 64 bits, and not even the largest value nb bits can hold
      // reaches the header's zMax, so the pixels need no clamp -- and bit 31 (0: no such block)
      if (code)
      {
        const u32 nb = codeBits(code), lutB = codeLut(code);
        bool plainB = false;
        if (mode == 1)
        {
          const u32 qTop = nb >= 32u ? 0xFFFFFFFFu : ((1u << nb) - 1u);
          const bool below = (DT >= DT_Float) ? (offset + (double)qTop * p.invScale < p.zMaxHdr)
                                              : ((i64)offset + (i64)qTop * (i64)p.invScale < (i64)p.zMaxHdr);
          plainB = below && !lutB && (u32)V * nb <= 64u;
        }
        const u32 pay = pos + ((mode == 1u) ? 3u + codeOffBytes(code) + lutB : 1u);
        code = (pay & 0xFFFFu) | (nb << 16) | (mode << 21) | (lutB << 23) | ((plainB ? 1u : 0u) << 24) | 0x80000000u;
      }
      sigHdr = (h0 >> 2) & pattern;
      s_offs[t] = offset;
      s_code[t] = code;
      s_at[t] = sigHdr;                  // (until the block's place is known)
    }
    if (post)
    {
      if (!pre) { code = s_code[t]; sigHdr = s_at[t]; }
      const u32 blk = base + f;          // index of the block in the raster
      const u32 it = pow2 ? (blk >> thShift) : blk / hp.nTH, jt = blk - it * hp.nTH;
      if (sigHdr != (jt & pattern) || blk >= hp.nBlocks) code = 0;    // signature = (j0 >> 3) & pattern, j0 = 8 jt
      if (!pre || code == 0u) s_code[t] = code;
      s_at[t] = code ? (it * 8u) * (u32)p.nCols + jt * 8u : kNoOffset;
      if (code == 0u) bad = true;
    }
  };
  // the first round's headers while the cells asked for above are on their way
  const bool early = !RAG;
  __syncthreads();    // (the flat list is complete)
  if (early && threadIdx.x < min(total, G::R)) parseHeader(threadIdx.x, threadIdx.x, 0u, true, false);

  // ---- the raster index of this workgroup's first block (only the launch's first round ever waits here)
  {
    u64 part = 0;
    bool lost = false;
    for (u32 i = threadIdx.x; i < nIn + grp; i += NT)
    {
      const u64* p = i < nIn ? b.wgCell + g0 + i : b.wgGroupCell + (i - nIn);
      u64 c = i == threadIdx.x ? cell0 : observe64(p);
      for (u32 spin = 0; (u32)(c >> 32) != epoch && spin < b.spinLimit; spin++)
      {
        __builtin_amdgcn_s_sleep(4);
        c = observe64(p);
      }
      if ((u32)(c >> 32) != epoch) { lost = true; c = 0; }
      part += i < nIn ? (u64)((u32)c & 0xFFFFu) : ((u64)(u32)c << 32);    // (a workgroup's cell: exit (16) | blocks (16))
    }
    if (__any(lost) && lane == 0) S.lost = 1u;
    if (nIn + grp != 0u)
    {
      part = waveSum(part);
      if (lane == 0 && part) atomicAdd((unsigned long long*)&S.part, (unsigned long long)part);
    }
  }
  __syncthreads();
  TRACEO(6);
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
    if (S.bad) raiseFlag(b, 1);
    // the sub-chunks hold all the raster's blocks, or the band goes the long way
    if (wg == nWG - 1u && base + total != S.hp.nBlocks) raiseFlag(b, 2);
  }

  for (u32 fLo = 0; fLo < total; )
  {
    const u32 fHi = min(total, ((base + fLo + G::R) / (u32)BPW) * (u32)BPW - base);
    // ---- parse the block headers once: lane = block
    {
      const u32 f = fLo + threadIdx.x;
      if (f < fHi) parseHeader(f, threadIdx.x, base, !(early && fLo == 0u), true);
    }
    __syncthreads();
    // ---- pixels: a wave takes BPW blocks at a time, a lane V consecutive pixels of one raster row of one block.  Wave tiles
    // lie on multiples of BPW blocks of the raster: a tile row is then a whole 128-byte line of the output
    const u32 blkLo = base + fLo, blkHi = base + fHi;
    const u32 g1 = (blkHi + BPW - 1) / BPW;
    for (u32 g = blkLo / BPW + (u32)w; g < g1; g += kWaves)
    {
      const u32 blk = g * BPW + (u32)bb;
      const bool have = blk >= blkLo && blk < blkHi;
      const u32 t = have ? blk - blkLo : 0u;         // (the tile's blocks outside the round: lanes that do nothing)
      const u32 code = have ? s_code[t] : 0u;        // (parseHeader's word)
      const double offset = s_offs[t];
      const u32 at0 = s_at[t];
      const u32 nbC = (code >> 16) & 31u, mode = (code >> 21) & 3u, lut = (code >> 23) & 1u;
      const u32 pbit = 8u * (code & 0xFFFFu);        // payload / first raw value
      int bw = 8, vc = V;
      if (RAG)
      {
        const u32 dims = have ? (u32)s_dims[t] : 0x88u;
        bw = (int)(dims & 15u);
        vc = r < (int)(dims >> 4) ? max(0, min(V, bw - h * V)) : 0;
      }
      const bool rowsAligned = !RAG || (((size_t)p.nCols * sizeof(T)) & 15u) == 0u;    // (else: no 16-byte ALIGNED stores)
      const bool rowsDword = sizeof(T) * V == 16 && (((size_t)p.nCols * sizeof(T)) & 3u) == 0u;    // (... but dwordx4 stores at dword alignment)
      const int e0 = r * bw + h * V;
      T v[V];
#pragma unroll
      for (int k = 0; k < V; k++) v[k] = T(0);
      // the common case, all blocks of the wave alike: bit-stuffed without a table, the lane's V values inside 64 bits, no
      // clamp -- three words of the stream, one funnel shift each way, V shifts
      const bool plain = ((code >> 24) & 1u) != 0u && (!RAG || (vc == V && bw == 8 && (rowsAligned || rowsDword)));
      if (__all(plain || !code))
      {
        if (code)
        {
          const u32 nb = nbC;
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
          if (RAG && !rowsAligned)
          {
            if constexpr (sizeof(Vec) == 16) storeStreamingA4(outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V), o);
          }
          else DECODE_STORE(reinterpret_cast<Vec*>(outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V)), o);
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
          const int nb = (int)nbC;
          const i64 offI = (i64)offset;
          if (!lut)
          {
#pragma unroll
            for (int k = 0; k < V; k++)
              v[k] = dequant<T>(offset, ldsBits(s_in, pbit + (u32)(e0 + k) * (u32)nb, nb), p.invScale, p.zMaxHdr, offI, invI, zMaxI);
          }
          else
          {
            const u32 nLut = (ldsBits(s_in, pbit - 8u, 8) - 1u) & 0xFFu;    // (the byte in front of the table: its size + 1)
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
        if (RAG && vc == V && !rowsAligned && rowsDword)
        {
          struct Vec16 { T e[V]; };
          Vec16 o;
#pragma unroll
          for (int k = 0; k < V; k++) o.e[k] = v[k];
          if constexpr (sizeof(Vec16) == 16) storeStreamingA4(outPix + (size_t)at0 + (size_t)r * (size_t)p.nCols + (size_t)(h * V), o);
        }
        else if (RAG && !(vc == V && rowsAligned))
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
    fLo = fHi;
    if (fLo < total) __syncthreads();    // (the round's arrays are taken again)
  }
  TRACEO(7);
  if (__any(bad) && lane == 0) raiseFlag(b, 3);

  // ---- checksum: the launch's last workgroup waits for everybody's terms (they were sent off microseconds after each
  // workgroup started), folds them and clears the accumulators for the next call (Lerc2.cpp:1037-1064)
  if (wg != nWG - 1u) return;
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
// (80 scalar registers: the waves of four workgroups fit a CU only if a SIMD can hold eight -- 800 scalar registers per SIMD, 16
// more than a wave asks for go with each; at 90 it is seven, and a workgroup of six waves puts two on two of the SIMDs)
#ifdef HIPSIM
#define LERC_ONE_SGPR_CAP
#else
#define LERC_ONE_SGPR_CAP __attribute__((amdgpu_num_sgpr(80)))
#endif
template<class T, bool RAG>
__global__ void __launch_bounds__(kOneThreads) LERC_ONE_SGPR_CAP
k_fast_decode_one(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, int nRows, int nCols, T* __restrict__ outPix)
{
  const size_t tile = blockIdx.y;
  b.params += tile; b.fallback += 4 * tile;
  b.wgCell += tile * b.wgStride;
  b.wgGroupCell += tile * b.wgGroupStride;
  b.wgAcc += tile * b.wgGroupStride;
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
  __shared__ OneShared<T, RAG> sm;
  fastOneBody<T, RAG>(sm, b, blob, sizeGiven, nRows, nCols, outPix + tile * t.tileElems, blockIdx.x);
}

template<class T>
static void launchFastDecodeOneT(int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven, const FastDecodeBuffers& b, void* out,
                                 hipStream_t st)
{
  const dim3 grid(fastOneNumWG(sizeGiven, (int)sizeof(T)), t.nTiles), block(kOneThreads);    // (sizeGiven: the largest blob of the batch)
  if (nRows % 8 != 0 || nCols % 8 != 0)
    hipLaunchKernelGGL((k_fast_decode_one<T, true>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols, (T*)out);
  else
    hipLaunchKernelGGL((k_fast_decode_one<T, false>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols, (T*)out);
}

// diagnostic: workgroups of the float kernel a CU holds, by the runtime's count
extern "C" __attribute__((visibility("default"))) int lerc_amd_probe_decode_one_residency()
{
#ifdef HIPSIM
  return 0;
#else
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_fast_decode_one<float, false>, (int)kOneThreads, 0) != hipSuccess) return -1;
  return n;
#endif
}

void launchFastDecodeOne(int dt, int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven,
                         const FastDecodeBuffers& b, void* out, hipStream_t st)
{
  switch (dt)
  {
    case DT_Short:  launchFastDecodeOneT<short>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UShort: launchFastDecodeOneT<unsigned short>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Int:    launchFastDecodeOneT<int>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_UInt:   launchFastDecodeOneT<unsigned int>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Float:  launchFastDecodeOneT<float>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    case DT_Double: launchFastDecodeOneT<double>(nRows, nCols, t, blob, sizeGiven, b, out, st); break;
    default: break;
  }
}

}    // namespace lerc
