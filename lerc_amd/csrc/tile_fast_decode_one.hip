// tile_fast_decode_one.hip -- the streaming decoder in ONE launch: every workgroup finds the block starts of its own piece of
// the blob AND decodes them.  Same results as tile_fast_decode.hip's two launches and as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends).  The two-launch form (tile_fast_decode.hip) reads
// the blob twice -- once to find the block starts (k_fast_discover: lists of starts per chunk go to memory), once to decode
// (k_fast_decode, behind a resolving step that turns per-chunk counts into block indices).  Here a workgroup of 512 threads
//   1. stages NCH = 16 consecutive chunks of 2 KiB (+ the next chunk's first window) in LDS, summing the Fletcher32 terms of
//      its own chunks on the way (all but the first: that one is the last chunk of the workgroup in front, staged and walked
//      again here, so that the entry of this workgroup's first own chunk -- the exit all live walks of the chunk in front agree
//      on -- is known without asking anybody; 1/15 of the discovery work done twice buys a launch, a pass over the blob, and
//      the lists' way through memory);
//   2. finds the bit-stuffed block headers in every chunk's first window by their byte pattern and lets the heads among them
//      walk, one wave, lane = (chunk, head) -- k_fast_discover's steps, with walk 0's list of block starts kept in LDS;
//   3. settles, per own chunk, which walk is the path and how many blocks start in the chunk, publishes the workgroup's block
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

namespace lerc {

#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time lines (constant-rate counter), read by tools/trace_decode.py
static __device__ unsigned long long g_traceO[16 * 8192];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace_decode_one(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceO), sizeof(unsigned long long) * (size_t)n); }
#define TRACEO(slot) do { if (threadIdx.x == 0 && wg < 8192u) g_traceO[16 * wg + (slot)] = wall_clock64(); } while (0)
#else
#define TRACEO(slot)
#endif

template<class T, u32 NCH_> struct OneGeom
{
  static constexpr int DT = DtOf<T>::v;
  static constexpr u32 TB = (u32)sizeof(T), W = kFastWindow((int)sizeof(T)), CH = kFastChunkBytes;
  static constexpr u32 NCH = NCH_, NW = (u32)kDiscWalks, NT = 32u * NCH_, CAP = (u32)kFastListCap;
  static constexpr u32 kUnits = NCH * CH / 16;                    // 16-byte units of the staged chunks
  static constexpr u32 kOverhang = (W + 16 + 15) / 16;            // + the next chunk's first window (walks end on a block start there)
  static constexpr u32 kStageUnits = kUnits + kOverhang;
  static constexpr u32 kBitWords = (W + 31) / 32;
  static constexpr u32 kFoundCap = 512, kHitCap = 512;
  static constexpr u32 kScanWords = (W + 2 + 8 + 3) / 4 + 1;
  static constexpr u32 R = NT;                                    // blocks per decode round: one header per thread
  static constexpr u32 kMaxRel = NCH * CH + W - 1;                // last staged byte a block may start at
};

template<class T, bool RAG, u32 NCH> struct OneShared
{
  typedef OneGeom<T, NCH> G;
  alignas(16) u32 in[G::kStageUnits * 4];
  alignas(16) u16 list[G::NCH * G::CAP + 8];       // walk 0's block starts per chunk (relative to the staged bytes); later: the starts of all own blocks, flat
  union U
  {
    struct D                                       // steps 2 - 3
    {
      u32 hits[G::NCH + 1][G::kBitWords];          // window positions where a bit-stuffed block header stands
      u32 heads[G::NCH][G::kBitWords];             // ... that are not the block right behind another one
      u32 strong[G::NCH][G::kBitWords];            // ... and are followed, exactly where they end, by another header found
      u16 found[G::kFoundCap], hit[G::kHitCap];    // window (6) << 10 | position
    } d;
    struct X                                       // step 5, one round
    {
      double offs[G::R];
      u32 code[G::R];                              // parseCode of the block, 0 = bad
      u32 at[G::R];                                // raster offset (pixels) of the block's first pixel
      u8 dims[RAG ? G::R : 1];                     // RAG: width | height << 4
    } x;
  } u;
  u16 mini[G::NCH][G::NW][4];                      // the first four block starts of every walk
  u32 exit[G::NCH][G::NW];
  u16 fin[G::NCH][G::NW];
  u16 cnt[G::NCH][G::NW];                          // blocks a walk passed; 0xFFFF: no such walk / it ran into something that is no block
  u32 nFinal[G::NCH];
  u32 agreed[G::NCH];                              // the exit all live walks of a chunk agree on, or ~0
  u32 count[G::NCH];                               // blocks of the path that belong to own chunk q
  u32 path[G::NCH];                                // walk | index of the chunk's first block in its list << 8, or ~0
  u32 cum[G::NCH + 1];
  u64 fa[G::NT / 64], fb[G::NT / 64];
  u64 part;                                        // sum of the cells: this group's in the low half, the groups' in front in the high half
  u32 nFound, nHit, over, bad, rewalk, lost;
};

template<class T, bool RAG, u32 NCH>
__device__ __forceinline__ void
fastOneBody(OneShared<T, RAG, NCH>& S, const FastDecodeBuffers& b, const u8* __restrict__ blob, u32 sizeGiven, int nRows, int nCols,
            T* __restrict__ outPix, u32 wg)
{
  typedef OneGeom<T, NCH> G;
  constexpr int DT = G::DT;
  constexpr u32 W = G::W, CH = G::CH, NW = G::NW, NT = G::NT, CAP = G::CAP;
  constexpr u32 kWaves = NT / 64, kHeadsPerWave = 64 / NCH;
  constexpr u32 kUnits = G::kUnits, kStageUnits = G::kStageUnits, kBitWords = G::kBitWords;
  constexpr u32 kFoundCap = G::kFoundCap, kHitCap = G::kHitCap, kScanWords = G::kScanWords, kMaxRel = G::kMaxRel;
  static_assert((NCH == 32 || NCH == 16 || NCH == 8) && NW == 8 && W + 16 < 1024 && NCH * CH + 2 * W < 65536, "lane layout / 16-bit list entries");
  static_assert(NCH * CAP == 4 * NT, "four list entries per thread when the lists are closed up");
  auto& s_in = S.in; auto& s_hits = S.u.d.hits; auto& s_heads = S.u.d.heads; auto& s_strong = S.u.d.strong;
  auto& s_found = S.u.d.found; auto& s_hit = S.u.d.hit;

  const RagCounts rc = ragCounts(nRows, nCols);
  const int lane = laneId(), w = waveId();
  // ---- the band header: every workgroup reads it for itself (all lanes the same 128 bytes); workgroup 0 leaves what the
  // host wants to see
  const FastDecodeParams hp = parseBandHeader<DT>(blob, sizeGiven, nRows, nCols);
  if (wg == 0 && threadIdx.x == 0) { storeParams<true>(b.params, hp); if (b.hostParams) *b.hostParams = hp; }
  const u32 nChunks = hp.nChunks;
  const u32 cs = wg * (NCH - 1u);                  // first staged chunk; own chunks: cs + 1 ... cs + NCH - 1 (workgroup 0: cs too)
  if (!hp.ok || (wg != 0u && cs + 1u >= nChunks)) return;    // (the grid is sized for the largest stream the blob could hold)
  TRACEO(0);
  const u32 nWG = fastOneNumWG(nChunks);
  const u32 r0 = cs * CH;                          // blob offset of LDS byte 0
  const u32 qOwn0 = wg ? 1u : 0u;
  const int version = (int)hp.version;
  const bool v5 = version >= 5;
  const u32 dataBegin = hp.dataBegin, blobEnd = hp.blobEnd;
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

  // ---- stage + Fletcher terms of the units of the own chunks (bytes 14 ... blobEnd - 1 of the blob are checksummed)
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
  for (u32 i = threadIdx.x; i < (NCH + 1) * kBitWords; i += NT) (&s_hits[0][0])[i] = 0u;
  for (u32 i = threadIdx.x; i < NCH * kBitWords; i += NT) (&s_strong[0][0])[i] = 0u;
  if (threadIdx.x == 0) { S.nFound = 0u; S.nHit = 0u; S.over = 0u; S.bad = 0u; S.rewalk = 0u; S.lost = 0u; S.part = 0ull; }
  __syncthreads();
  TRACEO(1);
  if (threadIdx.x == 0)
  {
    // this workgroup's checksum terms: one atomic nobody waits for (the launch's last workgroup folds the accumulators)
    u64 A = 0, B = 0;
#pragma unroll
    for (u32 k = 0; k < kWaves; k++) { A += S.fa[k]; B += S.fb[k]; }
    A %= 65535u; B %= 65535u;
    __hip_atomic_fetch_add(b.wgAcc + wg / kOneGroup, A | (B << 24) | (1ull << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }

  // ---- bit-stuffed block headers in the first `window` bytes of every chunk (+ the next workgroup's second one): see
  // k_fast_discover.  "a byte 64 behind a byte 10?nnnnn", four positions per lane and step.
  for (u32 f0 = 0; f0 < (NCH + 1) * kScanWords; f0 += NT)
  {
    const u32 f = f0 + threadIdx.x;
    const u32 win = f / kScanWords, d = f - win * kScanWords;
    const u32 chunk = cs + win;
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
    while (m)
    {
      const u32 j = (u32)(__ffs((int)m) - 1) >> 3;
      m &= m - 1u;
      const u32 at = atomicAdd(&S.nFound, 1u);
      if (at < kFoundCap) s_found[at] = (u16)((win << 10) | (4u * d + j)); else S.over = 1u;
    }
  }
  __syncthreads();
  // a count byte stands 2 + (bytes of the offset) behind the block's flag byte: try each offset type, one lane each
  {
    const u32 nFound = min(S.nFound, kFoundCap);
    for (u32 h = threadIdx.x; h < 4u * nFound; h += NT)
    {
      const u32 e = s_found[h >> 2], tc = h & 3u;
      const u32 win = e >> 10, q = e & 0x3FFu;
      const u32 offB = (offBytesTable<DT>() >> (4u * tc)) & 15u;
      if (offB == 0u || q < 2u + offB) continue;
      const u32 p = q - 2u - offB;
      if (p >= W || (cs + win) * CH + p >= blobEnd) continue;
      const u32 rel = win * CH + p;
      const u32 flag = (s_in[rel >> 2] >> (8u * (rel & 3u))) & 0xFFu;
      if ((flag & 3u) != 1u || (flag >> 6) != tc || (v5 && (flag & 4u))) continue;
      atomicOr(&s_hits[win][p >> 5], 1u << (p & 31u));
      if (win < NCH) { const u32 at = atomicAdd(&S.nHit, 1u); if (at < kHitCap) s_hit[at] = (u16)((win << 10) | p); else S.over = 1u; }
    }
  }
  if (threadIdx.x == 0 && cs * CH <= dataBegin)    // the stream's first block, whatever it is
  {
    const u32 p = dataBegin - cs * CH;
    atomicOr(&s_hits[0][p >> 5], 1u << (p & 31u));
    const u32 at = atomicAdd(&S.nHit, 1u);
    if (at < kHitCap) s_hit[at] = (u16)p; else S.over = 1u;
  }
  __syncthreads();
  TRACEO(2);

  // ---- of the blocks found, those that are not the block right behind another one start a walk
  for (u32 i = threadIdx.x; i < NCH * kBitWords; i += NT) (&s_heads[0][0])[i] = (&s_hits[0][0])[i];
  if (threadIdx.x < NCH) S.nFinal[threadIdx.x] = 0u;
  __syncthreads();
  const u32 nHit = min(S.nHit, kHitCap);
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 e = s_hit[h];
    const u32 hWin = e >> 10, hPos = e & 0x3FFu;
    u32 sg;
    const u32 cur = (cs + hWin) * CH + hPos;
    const u32 len = stepLean<DT, false, RAG>(s_in, cur - r0, blobEnd - cur, v5, kNoOffset, pattern, sg, rc);
    const u32 nx = hPos + len;
    const u32 relNx = cur - r0 + len;
    const u32 sgNx = ((s_in[relNx >> 2] >> (8u * (relNx & 3u))) >> 2) & pattern;
    const bool follows = sigOk(sg, sgNx, pattern);
    if (len != 0u && nx < W && follows)
    {
      atomicAnd(&s_heads[hWin][nx >> 5], ~(1u << (nx & 31u)));
      // (a header found exactly where this block ends: this one is on the path, as good as certainly -- it gets the first
      // walk slot of its chunk, the only one whose list is kept)
      if ((s_hits[hWin][nx >> 5] >> (nx & 31u)) & 1u) atomicOr(&s_strong[hWin][hPos >> 5], 1u << (hPos & 31u));
    }
    if (len == 0u || (!follows && cur + len < blobEnd)) atomicAnd(&s_heads[hWin][hPos >> 5], ~(1u << (hPos & 31u)));
  }
  __syncthreads();
  // walk slots: the heads that are followed by a header first, in the order of their positions, then the others
  for (u32 h = threadIdx.x; h < nHit; h += NT)
  {
    const u32 e = s_hit[h];
    const u32 hWin = e >> 10, hPos = e & 0x3FFu;
    const u32 bit = 1u << (hPos & 31u), wd = hPos >> 5;
    if (s_heads[hWin][wd] & bit)
    {
      const bool strong = (s_strong[hWin][wd] & bit) != 0u;
      u32 before = 0, nStrong = 0;
      for (u32 k = 0; k < kBitWords; k++)
      {
        const u32 hd = s_heads[hWin][k], st = hd & s_strong[hWin][k], mine = strong ? st : (hd & ~st);
        nStrong += (u32)__popc(st);
        before += (k < wd) ? (u32)__popc(mine) : (k == wd) ? (u32)__popc(mine & (bit - 1u)) : 0u;
      }
      const u32 at = strong ? before : nStrong + before;
      if (at < NW) S.fin[hWin][at] = (u16)hPos; else S.over = 1u;
      atomicAdd(&S.nFinal[hWin], 1u);
    }
  }
  __syncthreads();
  TRACEO(3);

  // ---- walks: lane = (chunk, head); the first wave takes the first heads of every chunk.  A walk ends on the first block
  // header of the next chunk's window it lands on (or with the blob).  Written for few instructions: see k_fast_discover.
  if ((u32)w < NW / kHeadsPerWave)
  {
    const u32 wc = (u32)lane / kHeadsPerWave, slot = ((u32)lane % kHeadsPerWave) + kHeadsPerWave * (u32)w;
    const u32 wChunk = cs + wc;
    const u32 wStart = wChunk * CH;
    const bool wLive = wChunk < nChunks;
    const u32 wEnd = wLive ? min(wStart + CH, blobEnd) : wStart;
    const bool walker = wLive && slot < min(S.nFinal[wc], NW);
    // (walk 0 keeps all its block starts, the others their first four: enough to tell which walk the chunk's entry lies on)
    u16* __restrict__ lst = slot == 0u ? &S.list[wc * CAP] : &S.mini[wc][slot][0];
    const u32 keep = slot == 0u ? CAP : 4u;
    const u32* __restrict__ nextHits = s_hits[wc + 1];
    constexpr u32 kOver = 0xFFFFFFFFu;
    const u32 startRel = wc * CH, endRel = wEnd - r0, blobRel = blobEnd - r0;
    const u32 sigStep = (pattern == 14u) ? 2u : 1u;
    u32 rel = walker ? startRel + (u32)S.fin[wc][slot] : kOver;
    u32 count = 0;
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
    // behind the chunk: done on a block header of the next window (or at the end of the blob), lost behind that window
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
    S.exit[wc][slot] = alive ? cur : kNoOffset;
    S.cnt[wc][slot] = alive ? (u16)count : (u16)0xFFFFu;
    if (__any(tooMany) && lane == 0) S.over = 1u;
  }
  __syncthreads();
  TRACEO(4);

  // ---- what all live walks of a chunk agree on is true without knowing which one is real
  if (threadIdx.x < NCH)
  {
    u32 lo = kNoOffset, hi = 0u, n = 0u;
#pragma unroll
    for (u32 k = 0; k < NW; k++)
    {
      const u32 e = S.exit[threadIdx.x][k];
      if (e != kNoOffset) { lo = min(lo, e); hi = max(hi, e); n++; }
    }
    S.agreed[threadIdx.x] = (cs + threadIdx.x < nChunks && n != 0u && lo == hi) ? lo : kNoOffset;
  }
  __syncthreads();
  // ... so the entry of chunk q is the agreed exit of chunk q - 1, the walk that starts there (or passes it with one of its
  // first four blocks: something in front of the entry that looks like a block ending right there) is the path, and what it
  // counted from there on are the blocks that belong to the chunk
  if (threadIdx.x < NCH)
  {
    const u32 q = threadIdx.x, chunk = cs + q;
    u32 count = 0, path = kNoOffset;
    bool bad = false;
    if (q >= qOwn0 && chunk < nChunks)
    {
      const u32 chunkStart = chunk * CH, chunkEnd = min(chunkStart + CH, blobEnd);
      const u32 e = (chunkStart <= dataBegin) ? dataBegin : S.agreed[q ? q - 1u : 0u];
      if (e == kNoOffset || e < chunkStart) bad = true;
      else if (e >= chunkEnd) bad = (e != blobEnd);    // the last block may begin before the last chunk and end with it
      else
      {
        const u32 rel = e - r0;
#pragma unroll
        for (u32 l = 0; l < NW; l++)
        {
          const u32 c = S.cnt[q][l];
          if (c == 0xFFFFu) continue;
          const u16* st = l == 0u ? &S.list[q * CAP] : &S.mini[q][l][0];
#pragma unroll
          for (u32 k = 0; k < 4; k++)
            if (k < c && (u32)st[k] == rel && path == kNoOffset) { path = l | (k << 8); count = c - k; }
        }
        if (path == kNoOffset || S.agreed[q] == kNoOffset) bad = true;    // (no agreement on the exit: the next chunk says so too)
      }
      if (bad) { count = 0; path = kNoOffset; }
    }
    S.count[q] = count; S.path[q] = path;
    if (path != kNoOffset && ((path & 0xFFu) != 0u || b.testRewalk)) S.rewalk = 1u;    // (test knob: every path is walked again)
    if (bad) S.bad = 1u;
  }
  __syncthreads();
  // ---- a path that is not walk 0 (a stray header in front of the path's first block of the window that was not followed by
  // a header itself: one chunk in a few hundred) is walked again from the chunk's entry, now with all checks and with its list
  if (S.rewalk)
  {
    if (threadIdx.x < NCH)
    {
      const u32 q = threadIdx.x, path = S.path[q];
      if (path != kNoOffset && ((path & 0xFFu) != 0u || b.testRewalk))
      {
        const u32 chunk = cs + q;
        const u32 e = (chunk * CH <= dataBegin) ? dataBegin : S.agreed[q ? q - 1u : 0u];
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
    for (int d = 1; d < (int)NCH; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
    if ((u32)lane < NCH) S.cum[lane] = inc - c;
    if ((u32)lane == NCH - 1u)
    {
      S.cum[NCH] = inc;
      publish64(b.wgCell + wg, tag | (u64)inc);
    }
  }
  u16 keepStart[4];
  u32 keepTo[4];
#pragma unroll
  for (u32 e = 0; e < 4; e++)
  {
    const u32 idx = threadIdx.x + NT * e, q = idx / CAP, i = idx % CAP;
    const bool have = i < S.count[q];
    keepStart[e] = have ? S.list[q * CAP + min(((S.path[q] >> 8) & 3u) + i, CAP - 1u)] : (u16)0;
    keepTo[e] = have ? i : kNoOffset;
  }
  __syncthreads();
  const u32 total = S.cum[NCH];
#pragma unroll
  for (u32 e = 0; e < 4; e++)
  {
    const u32 idx = threadIdx.x + NT * e, q = idx / CAP;
    if (keepTo[e] != kNoOffset) S.list[S.cum[q] + keepTo[e]] = keepStart[e];
  }
  if (threadIdx.x == 0 && total)
  {
    // (the end of the last block: the exit the walks of the last chunk in which a block starts agreed on)
    u32 last = NCH - 1u;
    while (last > 0u && S.count[last] == 0u) last--;
    const u32 ex = S.agreed[last];
    S.list[total] = (u16)min(ex - min(ex, r0), 0xFFFFu);
  }
  TRACEO(5);

  // ---- the raster index of this workgroup's first block: the cells of the workgroups in front of it -- those of its group,
  // and one per group in front.  (They publish when this one does, or did long ago: only the launch's first round waits.)
  const u32 grp = wg / kOneGroup, g0 = grp * kOneGroup, nIn = wg - g0;
  {
    u64 part = 0;
    bool lost = false;
    for (u32 i = threadIdx.x; i < nIn + grp; i += NT)
    {
      const u64* p = i < nIn ? b.wgCell + g0 + i : b.wgGroupCell + (i - nIn);
      u64 c = observe64(p);
      for (u32 spin = 0; (u32)(c >> 32) != epoch && spin < b.spinLimit; spin++)
      {
        __builtin_amdgcn_s_sleep(4);
        c = observe64(p);
      }
      if ((u32)(c >> 32) != epoch) { lost = true; c = 0; }
      part += i < nIn ? (u64)(u32)c : ((u64)(u32)c << 32);
    }
    if (__any(lost) && lane == 0) S.lost = 1u;
    if (nIn + grp != 0u)
    {
      part = waveSum(part);
      if (lane == 0 && part) atomicAdd((unsigned long long*)&S.part, (unsigned long long)part);
    }
  }
  __syncthreads();    // (also: the flat list is complete)
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
    // the chunks hold all the raster's blocks, or the band goes the long way
    if (wg == nWG - 1u && base + total != hp.nBlocks) raiseFlag(b, 2);
  }

  // ---- rounds of at most R blocks (cut on multiples of BPW blocks of the RASTER, like the wave tiles below)
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  const struct { int nCols, version; double invScale, zMaxHdr; } p = { (int)hp.nCols, (int)hp.version, hp.invScale, hp.zMaxHdr };
  const bool pow2 = (hp.nTH & (hp.nTH - 1u)) == 0u;
  const u32 thShift = 31u - (u32)__clz((int)hp.nTH);
  const int r = lane >> 3, c = lane & 7, bb = c / LPR, h = c % LPR;
  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  auto& s_offs = S.u.x.offs; auto& s_code = S.u.x.code; auto& s_at = S.u.x.at; auto& s_dims = S.u.x.dims;
  bool bad = false;
  for (u32 fLo = 0; fLo < total; )
  {
    const u32 fHi = min(total, ((base + fLo + G::R) / (u32)BPW) * (u32)BPW - base);
    // ---- parse the block headers once: lane = block
    {
      const u32 f = fLo + threadIdx.x;
      if (f < fHi)
      {
        const u32 blk = base + f;                                  // index of the block in the raster
        const u32 pos = (u32)S.list[f], off = min(pos, kMaxRel);
        u32 h0, h1, h2;
        ldsHeader<DT>(s_in, off, h0, h1, h2);
        const u32 it = pow2 ? (blk >> thShift) : blk / hp.nTH, jt = blk - it * hp.nTH;
        u32 bw = 8u, bh = 8u;
        if (RAG)
        {
          bw = min(8u, hp.nCols - 8u * min(jt, hp.nTH - 1u)); bh = min(8u, hp.nRows - 8u * min(it, (hp.nRows + 7u) / 8u - 1u));
          s_dims[threadIdx.x] = (u8)(bw | (bh << 4));
        }
        u32 code = parseCode<DT>(h0, h1, h2, p.version, bw * bh);
        if (pos + codeLen(code) != (u32)S.list[f + 1]) code = 0;              // the blocks tile the stream
        if (((h0 >> 2) & pattern) != (jt & pattern)) code = 0;                // signature = (j0 >> 3) & pattern, j0 = 8 jt
        if (blk >= hp.nBlocks || pos > kMaxRel) code = 0;
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
        if (code && mode == 1)
        {
          const u32 qTop = codeBits(code) >= 32u ? 0xFFFFFFFFu : ((1u << codeBits(code)) - 1u);
          const bool below = (DT >= DT_Float) ? (offset + (double)qTop * p.invScale < p.zMaxHdr)
                                              : ((i64)offset + (i64)qTop * (i64)p.invScale < (i64)p.zMaxHdr);
          if (below) code |= 1u << 30;
        }
        s_offs[threadIdx.x] = offset;
        s_code[threadIdx.x] = code;
        s_at[threadIdx.x] = code ? (it * 8u) * (u32)p.nCols + jt * 8u : kNoOffset;
        if (code == 0u) bad = true;
      }
    }
    __syncthreads();
    // ---- pixels: a wave takes BPW blocks at a time, a lane V consecutive pixels of one raster row of one block.  Wave tiles
    // lie on multiples of BPW blocks of the raster: a tile row is then a whole 128-byte line of the output
    const u32 blkLo = base + fLo, blkHi = base + fHi;
    const u32 g1 = (blkHi + BPW - 1) / BPW;
    for (u32 g = blkLo / BPW + (u32)w; g < g1; g += kWaves)
    {
      const u32 blk = g * BPW + (u32)bb;
      const u32 t = blk - blkLo;                     // (wraps for the blocks in front of the round's first one)
      const bool have = blk >= blkLo && blk < blkHi;
      const u32 code = have ? s_code[t] : 0u;
      const double offset = have ? s_offs[t] : 0.0;
      const u32 at0 = have ? s_at[t] : kNoOffset;
      const u32 mode = codeMode(code), lut = codeLut(code), offB = codeOffBytes(code);
      const u32 pbit = 8u * ((have ? (u32)S.list[fLo + t] : 0u) + ((mode == 1u) ? 3u + offB + lut : 1u));    // payload / first raw value
      int bw = 8, vc = V;
      if (RAG)
      {
        const u32 dims = have ? (u32)s_dims[t] : 0x88u;
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
    const u32 good = ((u32)((s2 << 16) | s1) == hp.expectChecksum) ? 1u : 0u;
    publish32(&b.params->checksumOk, good);
    if (b.hostParams) b.hostParams->checksumOk = good;
  }
}

// blockIdx.y = tile of a batch (one raster: a batch of 1); each tile has its own slice of every buffer
template<class T, bool RAG, u32 NCH>
__global__ void __launch_bounds__(32 * NCH)
k_fast_decode_one(FastDecodeBuffers b, FastDecodeBatch t, const u8* blob, u32 sizeGiven, int nRows, int nCols, T* __restrict__ outPix)
{
  const size_t tile = blockIdx.y;
  b.params += tile; b.fallback += 4 * tile;
  b.wgCell += tile * fastOneWgStride(t.nChunks);
  b.wgGroupCell += tile * fastOneGroupStride(t.nChunks);
  b.wgAcc += tile * fastOneGroupStride(t.nChunks);
  if (t.tileOffset) { blob += t.tileOffset[tile]; sizeGiven = t.tileSize[tile]; }
  __shared__ OneShared<T, RAG, NCH> sm;
  fastOneBody<T, RAG, NCH>(sm, b, blob, sizeGiven, nRows, nCols, outPix + tile * t.tileElems, blockIdx.x);
}

template<class T>
static void launchFastDecodeOneT(int nRows, int nCols, const FastDecodeBatch& t, const u8* blob, u32 sizeGiven, const FastDecodeBuffers& b, void* out,
                                 hipStream_t st)
{
  const dim3 grid(fastOneNumWG(t.nChunks), t.nTiles), block(kOneThreads);
  if (nRows % 8 != 0 || nCols % 8 != 0)
    hipLaunchKernelGGL((k_fast_decode_one<T, true, kOneChunks>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols, (T*)out);
  else
    hipLaunchKernelGGL((k_fast_decode_one<T, false, kOneChunks>), grid, block, 0, st, b, t, blob, sizeGiven, nRows, nCols, (T*)out);
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
