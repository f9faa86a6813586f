// tile_fast_decode.hip -- streaming decoder kernels for the common case (one band, nDepth == 1, every
// pixel valid, 8 x 8 blocks, nRows % 8 == 0, nCols % 512 == 0).  Same results as tile_decode.hip.
//
// The block stream stores no offsets (block k+1 starts where block k ends), so decoding starts with a
// discovery pass:
//   k_fast_walk     LDS staged.  A workgroup stages a few 4 KiB chunks, tries every position of each chunk's
//                   first `window` bytes as a block start (a few steps filter out almost all of them) and
//                   walks the survivors to the chunk end.  The true first block of the chunk is always among
//                   the survivors, so whatever ALL survivors agree on is true without knowing which survivor
//                   is the real one: the chunk's exit (= entry of the next chunk) and the first block start
//                   behind every 512-byte sub-chunk boundary.  Per surviving start it also records the
//                   number of blocks up to the chunk end.
//   k_fast_resolve  entry of chunk c = agreed exit of chunk c-1; #blocks of chunk c = count recorded for the
//                   survivor that starts exactly there; an exclusive scan turns counts into block indices.
//   k_fast_decode   a workgroup owns 2 chunks: it stages their bytes (accumulating the Fletcher32 sums of the
//                   bytes it owns, word-wise), re-walks them from the resolved entry with one lane per
//                   512-byte sub-chunk, parses every block header once (thread = block), then every lane
//                   extracts V consecutive pixels of one raster row, dequantises (double precision in the
//                   reference's expression order for float types, exact integer arithmetic for integer
//                   types) and stores one 16-byte vector; a wave covers 8 raster rows x 128 bytes.
// Whenever a precondition fails (a block longer than its raw size, disagreeing survivors, more than
// kMaxBlocksPerWG blocks in two chunks, ...) the kernels raise `fallback`, and the host repeats the band
// with the general kernels.  Reference: Lerc2.cpp:1672-1713, :2025-2230; BitStuffer2.cpp:159-258, :476-540.
#include "tile_fast.h"
#include "kernels.h"
#include "wave_utils.h"

namespace lerc {

static const u32 kNoOffset = 0xFFFFFFFFu;

struct BlkLite
{
  u32 len, payload;
  u32 nLut;
  u8 flag, mode, offBytes, nb, lut, dtUsed;
};

// Header parser for all-valid 8 x 8 blocks (64 elements) of data type DT from an LDS word image whose
// byte 0 is blob byte `a0`.  All header bytes come out of one group of aligned word reads (a single LDS
// round trip per block).  Returns false if no valid block starts at pos.  Mirrors Lerc2::ReadTile
// (Lerc2.cpp:2025-2110) and BitStuffer2::Decode (BitStuffer2.cpp:159-258).
template<int DT>
__device__ __forceinline__ bool parseLds(const u32* words, u32 a0, u32 pos, u32 end, int version, BlkLite& b)
{
  constexpr int TBYTES = (DT <= DT_Byte) ? 1 : (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  if (pos >= end) return false;
  const u32 rel = pos - a0, w = rel >> 2, sh = 8u * (rel & 3u);
  const u32 x0 = words[w], x1 = words[w + 1], x2 = words[w + 2], x3 = (DT == DT_Double) ? words[w + 3] : 0u;
  const u32 h0 = sh ? ((x0 >> sh) | (x1 << (32 - sh))) : x0;
  const u32 h1 = sh ? ((x1 >> sh) | (x2 << (32 - sh))) : x1;
  const u32 h2 = sh ? ((x2 >> sh) | (x3 << (32 - sh))) : x2;
  const u64 lo = ((u64)h1 << 32) | h0;    // header bytes 0..7; h2 = bytes 8..11
  const u32 flag = h0 & 0xFFu;
  b.flag = (u8)flag;
  if (version >= 5 && (flag & 4u)) return false;    // slice difference needs nDepth > 1
  b.mode = (u8)(flag & 3u);
  const int tc = (int)(flag >> 6);
  b.offBytes = 0; b.nb = 0; b.lut = 0; b.nLut = 0; b.payload = 1; b.dtUsed = (u8)DT;
  u32 len = 1;
  if (b.mode == 2) { b.len = 1; return true; }
  if (b.mode == 0) len = 1 + 64 * TBYTES;
  else
  {
    const int dtU = typeUsed(DT, tc);
    if (dtU == DT_Undefined) return false;
    b.dtUsed = (u8)dtU;
    b.offBytes = (u8)dtSize(dtU);
    len = 1 + b.offBytes;
    if (b.mode == 1)
    {
      // bytes len, len + 1, len + 2 of the header: numBits byte, count, (LUT size)
      const u32 t3 = (len < 8) ? (u32)(lo >> (8 * len)) | ((len > 4) ? (h2 << (8 * (8 - len))) : 0u) : (h2 >> (8 * (len - 8)));
      if (pos + len + 2 > end) return false;
      const u32 b0 = t3 & 0xFFu;
      if ((b0 >> 6) != 2u) return false;            // 64 elements -> one-byte count field
      if (((t3 >> 8) & 0xFFu) != 64u) return false;
      b.lut = (b0 & 32u) ? 1 : 0;
      b.nb = (u8)(b0 & 31u);
      if (b.nb == 0) return false;
      len += 2;
      if (!b.lut) { b.payload = len; len += 8u * b.nb; }
      else
      {
        if (pos + len >= end) return false;
        const int nLut = (int)((t3 >> 16) & 0xFFu) - 1;
        if (nLut < 1) return false;
        b.nLut = (u32)nLut;
        len += 1;
        b.payload = len;
        len += ((u32)nLut * b.nb + 7) >> 3;
        len += (64u * (u32)bitLen((u32)nLut) + 7) >> 3;
      }
    }
  }
  if (pos + len > end) return false;
  b.len = len;
  return true;
}

__device__ __forceinline__ bool sigOk(u32 prev, u32 cur, u32 pattern)
{
  const u32 step = (pattern == 14u) ? 2u : 1u;    // 8 x 8 blocks: signature = (j0 >> 3) & pattern
  return cur == prev || cur == ((prev + step) & pattern) || cur == 0;
}

static const int kWalkChunksPerWG = 4;
static const int kFilterSteps = 4;
static const int kMaxSurvivors = 448;

template<int DT>
__global__ void __launch_bounds__(256)
k_fast_walk(int version, FastWalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
            u32* __restrict__ chunkExit, u16* __restrict__ countAt, u32* __restrict__ subEntry, u32* __restrict__ fallback)
{
  constexpr int TBYTES = (DT <= DT_UShort) ? 2 : (DT <= DT_Float) ? 4 : 8;
  constexpr u32 W = kFastWindow(TBYTES);
  constexpr u32 kStage = kWalkChunksPerWG * kFastChunkBytes + W + 48;
  constexpr int NS = kFastSubPerChunk;
  __shared__ __align__(16) u32 s_in[kStage / 4 + 8];
  __shared__ u32 s_svStart[kMaxSurvivors];
  __shared__ u32 s_nSv, s_over;
  __shared__ u32 s_min[kWalkChunksPerWG], s_max[kWalkChunksPerWG], s_n[kWalkChunksPerWG];
  __shared__ u32 s_subMin[kWalkChunksPerWG][NS], s_subMax[kWalkChunksPerWG][NS];

  const u32 c0 = blockIdx.x * kWalkChunksPerWG;
  const u32 groupStart = dataBegin + c0 * kFastChunkBytes;
  const u32 a0 = groupStart & ~15u;
  const u32 stageEnd = min(a0 + kStage, blobEnd);
  for (u32 i = threadIdx.x * 16u; a0 + i < stageEnd; i += 256u * 16u)
  {
    if (a0 + i + 16 <= stageEnd)
      *reinterpret_cast<uint4*>(reinterpret_cast<u8*>(s_in) + i) = *reinterpret_cast<const uint4*>(blob + a0 + i);
    else
    {
      u32 t4[4] = { 0, 0, 0, 0 };
      for (u32 k = 0; a0 + i + k < stageEnd; k++) t4[k >> 2] |= (u32)blob[a0 + i + k] << (8 * (k & 3));    // never read past the blob
      *reinterpret_cast<uint4*>(reinterpret_cast<u8*>(s_in) + i) = make_uint4(t4[0], t4[1], t4[2], t4[3]);
    }
  }
  if (threadIdx.x == 0) { s_nSv = 0; s_over = 0; }
  if (threadIdx.x < kWalkChunksPerWG) { s_min[threadIdx.x] = kNoOffset; s_max[threadIdx.x] = 0; s_n[threadIdx.x] = 0; }
  if (threadIdx.x < kWalkChunksPerWG * NS) { (&s_subMin[0][0])[threadIdx.x] = kNoOffset; (&s_subMax[0][0])[threadIdx.x] = 0; }
  __syncthreads();

  const u32 pattern = (version >= 5) ? 14u : 15u;
  const u32 nChunksHere = min((u32)kWalkChunksPerWG, wp.nChunks - c0);

  // ---- phase 1: every window position, a few steps
  for (u32 f = threadIdx.x; f < nChunksHere * W; f += 256)
  {
    const u32 g = f / W, o = f - g * W;
    const u32 chunkStart = groupStart + g * kFastChunkBytes;
    const u32 chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
    if (c0 + g == 0 && o != 0) continue;    // the very first block of the stream is known
    u32 cur = chunkStart + o;
    if (cur >= chunkEnd) continue;
    u32 sig = kNoOffset;
    bool alive = true;
    for (int s = 0; s < kFilterSteps && cur < chunkEnd; s++)
    {
      BlkLite b;
      if (!parseLds<DT>(s_in, a0, cur, stageEnd, version, b)) { alive = false; break; }
      const u32 sg = ((u32)b.flag >> 2) & pattern;
      if (sig != kNoOffset && !sigOk(sig, sg, pattern)) { alive = false; break; }
      sig = sg; cur += b.len;
    }
    if (!alive) continue;
    const u32 slot = atomicAdd(&s_nSv, 1u);
    if (slot >= (u32)kMaxSurvivors) { s_over = 1; continue; }
    s_svStart[slot] = (chunkStart + o) | 0u;
  }
  __syncthreads();

  // ---- phase 2: survivors walk (again from their start) to the end of their chunk, noting where they
  // pass every sub-chunk boundary
  const u32 nSv = min(s_nSv, (u32)kMaxSurvivors);
  for (u32 s = threadIdx.x; s < nSv; s += 256)
  {
    const u32 start = s_svStart[s];
    const u32 g = (start - groupStart) / kFastChunkBytes;
    const u32 chunkStart = groupStart + g * kFastChunkBytes;
    const u32 chunkEnd = min(chunkStart + kFastChunkBytes, blobEnd);
    u32 cur = start, count = 0, sig = kNoOffset;
    u32 nextSub = 1;    // sub-chunk boundaries passed so far + 1
    bool alive = true;
    while (cur < chunkEnd)
    {
      while (nextSub < (u32)NS && cur >= chunkStart + nextSub * kFastSubBytes)
      {
        atomicMin(&s_subMin[g][nextSub], cur);
        atomicMax(&s_subMax[g][nextSub], cur);
        nextSub++;
      }
      BlkLite b;
      if (!parseLds<DT>(s_in, a0, cur, stageEnd, version, b)) { alive = false; break; }
      const u32 sg = ((u32)b.flag >> 2) & pattern;
      if (sig != kNoOffset && !sigOk(sig, sg, pattern)) { alive = false; break; }
      sig = sg; cur += b.len; count++;
    }
    if (!alive)
    {
      // a chain that died after leaving marks: its marks must not count as agreement
      for (u32 j = 1; j < nextSub; j++) { atomicMin(&s_subMin[g][j], 0u); atomicMax(&s_subMax[g][j], kNoOffset - 1); }
      continue;
    }
    for (; nextSub < (u32)NS; nextSub++) { atomicMin(&s_subMin[g][nextSub], cur); atomicMax(&s_subMax[g][nextSub], cur); }
    atomicMin(&s_min[g], cur);
    atomicMax(&s_max[g], cur);
    atomicAdd(&s_n[g], 1u);
    countAt[(size_t)(c0 + g) * W + (start - chunkStart)] = (u16)count;    // #blocks from this start to the chunk end
  }
  __syncthreads();
  if (threadIdx.x < nChunksHere)
  {
    const u32 g = threadIdx.x;
    const bool ok = !s_over && s_n[g] > 0 && s_min[g] == s_max[g];
    chunkExit[c0 + g] = ok ? s_min[g] : kNoOffset;
    if (!ok && c0 + g + 1 < wp.nChunks) atomicOr(fallback, 1u);    // the exit of the last chunk is not needed
  }
  if (threadIdx.x < nChunksHere * NS)
  {
    const u32 g = threadIdx.x / NS, j = threadIdx.x % NS;
    const bool ok = j > 0 && s_subMin[g][j] == s_subMax[g][j] && s_subMin[g][j] != kNoOffset;
    subEntry[(size_t)(c0 + g) * NS + j] = ok ? s_subMin[g][j] : kNoOffset;
  }
}

__global__ void __launch_bounds__(256)
k_fast_resolve(FastWalkPlan wp, u32 window, u32 dataBegin, u32 blobEnd, const u32* __restrict__ chunkExit, const u16* __restrict__ countAt,
               u32* __restrict__ chunkEntry, u32* __restrict__ chunkCount, u32* __restrict__ fallback)
{
  const u32 c = blockIdx.x * 256u + threadIdx.x;
  if (c > wp.nChunks) return;
  if (c == wp.nChunks) { chunkEntry[c] = blobEnd; return; }
  const u32 e = (c == 0) ? dataBegin : chunkExit[c - 1];
  chunkEntry[c] = e;
  const u32 chunkStart = dataBegin + c * kFastChunkBytes;
  u32 n = 0xFFFFu;
  if (e != kNoOffset && e >= chunkStart && e - chunkStart < window) n = countAt[(size_t)c * window + (e - chunkStart)];
  if (n == 0xFFFFu) { atomicOr(fallback, 2u); n = 0; }
  chunkCount[c] = n;
}

// ------------------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------------------
template<class T> struct DCfg
{
  static constexpr int V = (sizeof(T) >= 4) ? 16 / (int)sizeof(T) : 8;
  static constexpr int LPR = 8 / V;
  static constexpr int BPW = 8 / LPR;    // blocks per wave tile
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

static const int kDecodeChunksPerWG = 2;
static const int kMaxBlocksPerWG = 1024;

template<class T>
__global__ void __launch_bounds__(256)
k_fast_decode(BandParams p, FastWalkPlan wp, const u8* __restrict__ blob, u32 dataBegin, u32 blobEnd,
              const u32* __restrict__ chunkEntry, const u32* __restrict__ chunkBase, const u32* __restrict__ subEntry,
              T* __restrict__ outPix, u64* __restrict__ slotFletcher, u32* __restrict__ fallback, DeviceStatus* st)
{
  typedef DCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  constexpr int DT = DtOf<T>::v;
  constexpr int NS = kFastSubPerChunk, NL = kDecodeChunksPerWG * NS;
  constexpr u32 kStage = kDecodeChunksPerWG * kFastChunkBytes + kFastWindow((int)sizeof(T)) + 48;
  __shared__ __align__(16) u32 s_in[kStage / 4 + 8];
  __shared__ u16 s_boff[kMaxBlocksPerWG];      // block start, relative to a0
  __shared__ u32 s_pbit[kMaxBlocksPerWG];      // LDS bit position of the payload (bit stuffed) / first raw value
  __shared__ u32 s_meta[kMaxBlocksPerWG];      // mode | lut << 2 | numBits << 3 | nLut << 8 | ok << 31
  __shared__ double s_offs[kMaxBlocksPerWG];
  __shared__ u64 s_fa[4], s_fb[4];
  __shared__ u32 s_nB, s_bad, s_skip;
  // the earlier kernels gave up?  (read once per workgroup: other workgroups of this launch may raise it too)
  if (threadIdx.x == 0) s_skip = *fallback;
  __syncthreads();
  if (s_skip) return;

  const int w = waveId(), lane = laneId();
  const u32 c0 = blockIdx.x * kDecodeChunksPerWG;
  const u32 c1 = min(c0 + (u32)kDecodeChunksPerWG, wp.nChunks);
  const u32 g0 = chunkEntry[c0], g1 = chunkEntry[c1];    // chunkEntry[nChunks] = blobEnd
  if (g0 == kNoOffset || g1 < g0 || g1 - g0 > kStage - 48 || g1 > blobEnd)
  {
    if (threadIdx.x == 0) atomicOr(fallback, 16u);
    return;
  }
  // ---- stage the span (16-byte loads from the aligned-down start) + Fletcher sums of the owned bytes
  const u32 a0 = g0 & ~15u;
  const u32 shift = g0 - a0, spanLen = g1 - g0;
  const u32 nChunks16 = (shift + spanLen + 15) >> 4;
  u64 A = 0, B = 0;
  for (u32 ch = threadIdx.x; ch < nChunks16; ch += 256)
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
  if (threadIdx.x == 0) { s_nB = 0; s_bad = 0; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const u32 slot = blockIdx.x & (kFastSlots - 1);
    atomicAdd(&slotFletcher[2 * slot], (s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3]) % 65535u);
    atomicAdd(&slotFletcher[2 * slot + 1], (s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]) % 65535u);
  }

  // ---- block starts: one lane per sub-chunk walks from its (agreed) entry to the next known entry
  const u32 pattern = (p.version >= 5) ? 14u : 15u;
  if (w == 0)
  {
    u32 start = kNoOffset;
    if (lane < NL)
    {
      const u32 g = (u32)lane / NS, j = (u32)lane % NS;
      if (c0 + g < c1) start = (j == 0) ? chunkEntry[c0 + g] : subEntry[(size_t)(c0 + g) * NS + j];
      // an entry that is not behind the previous known one cannot be right (never happens for agreed values)
      if (start != kNoOffset && (start < g0 || start >= g1)) start = kNoOffset;    // nothing left to walk from there
    }
    const u64 known = __ballot(start != kNoOffset);
    // limit = start of the next lane with a known start, or the end of the span
    const u64 later = (lane < 63) ? (known >> (lane + 1)) : 0ull;
    const int nextLane = later ? lane + 1 + (__ffsll((long long)later) - 1) : -1;
    const u32 nextStart = __shfl(start, nextLane < 0 ? lane : nextLane);
    const u32 limit = (nextLane < 0) ? g1 : nextStart;
    // a later lane may name a start inside an earlier lane's range only if both are on the same chain;
    // duplicates (two boundaries passed by one block) are dropped: a lane whose start equals the previous
    // known start would emit the same blocks twice
    u32 n = 0;
    bool bad = false;
    if (start != kNoOffset && start < limit)
    {
      u32 cur = start;
      while (cur < limit)
      {
        BlkLite b;
        if (!parseLds<DT>(s_in, a0, cur, g1, p.version, b)) { bad = true; break; }
        cur += b.len; n++;
      }
      if (cur != limit) bad = true;
    }
    u32 inc = n;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
    const u32 total = __shfl(inc, 63);
    u32 at = inc - n;
    if (total > (u32)kMaxBlocksPerWG) bad = true;
    if (!__any(bad) && start != kNoOffset && start < limit)
    {
      u32 cur = start;
      while (cur < limit)
      {
        BlkLite b;
        parseLds<DT>(s_in, a0, cur, g1, p.version, b);
        s_boff[at++] = (u16)(cur - a0);
        cur += b.len;
      }
    }
    if (lane == 0) { s_nB = total; s_bad = __any(bad) ? 1u : 0u; }
    else (void)__any(bad);
  }
  __syncthreads();
  const u32 nB = s_nB;
  if (s_bad || nB != chunkBase[c1] - chunkBase[c0])
  {
    if (threadIdx.x == 0) atomicOr(fallback, 32u);
    return;
  }

  // ---- parse every block header once: thread = block
  const u32 B0 = chunkBase[c0];
  bool anyBad = false;
  for (u32 i = threadIdx.x; i < nB; i += 256)
  {
    const u32 off = a0 + s_boff[i];
    const u32 blkIdx = B0 + i;
    const int jt = (int)(blkIdx % (u32)p.nTH);
    const int j0 = jt * 8;
    BlkLite bl;
    bool ok = parseLds<DT>(s_in, a0, off, g1, p.version, bl);
    ok = ok && ((((u32)bl.flag >> 2) & pattern) == (((u32)j0 >> 3) & pattern));
    double offset = 0;
    if (ok && (bl.mode == 1 || bl.mode == 3))
      offset = typedFromBits(getBytes(reinterpret_cast<const u8*>(s_in) + (off - a0) + 1, bl.offBytes), bl.dtUsed);
    s_offs[i] = offset;
    s_pbit[i] = 8u * (off - a0 + bl.payload);
    s_meta[i] = ok ? ((u32)bl.mode | ((u32)bl.lut << 2) | ((u32)bl.nb << 3) | (bl.nLut << 8) | 0x80000000u) : 0u;
    if (!ok) anyBad = true;
  }
  if (anyBad) raiseError(st, kFailed, blockIdx.x);
  __syncthreads();

  // ---- pixels: wave tiles of BPW adjacent blocks (global block index / BPW), 8 rows x 128 bytes per wave
  const int r = lane >> 3, c = lane & 7, b = c / LPR, h = c % LPR;
  const i64 invI = (i64)p.invScale, zMaxI = (i64)p.zMaxHdr;
  const u32 firstTile = B0 / BPW, lastTile = (B0 + nB - 1) / BPW;
  bool bad = false;
  for (u32 tile = firstTile + (u32)w; tile <= lastTile && nB > 0; tile += 4)
  {
    const u32 blkIdx = tile * BPW + (u32)b;
    if (blkIdx < B0 || blkIdx >= B0 + nB) continue;    // that block belongs to a neighbouring workgroup
    const u32 i = blkIdx - B0;
    const u32 meta = s_meta[i];
    const u32 pbit = s_pbit[i];
    const double offset = s_offs[i];
    const int mode = (int)(meta & 3u);
    const int e0 = r * 8 + h * V;
    T v[V];
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = T(0);
    if (meta >> 31)
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
        const int nb = (int)((meta >> 3) & 31u);
        const i64 offI = (i64)offset;
        if (!((meta >> 2) & 1u))
        {
#pragma unroll
          for (int k = 0; k < V; k++)
            v[k] = dequant<T>(offset, ldsBits(s_in, pbit + (u32)(e0 + k) * (u32)nb, nb), p.invScale, p.zMaxHdr, offI, invI, zMaxI);
        }
        else
        {
          const u32 nLut = (meta >> 8) & 0xFFu;
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
    const u32 it = blkIdx / (u32)p.nTH, jt = blkIdx - it * (u32)p.nTH;
    struct alignas(sizeof(T) * V) Vec { T e[V]; };
    Vec o;
#pragma unroll
    for (int k = 0; k < V; k++) o.e[k] = v[k];
    *reinterpret_cast<Vec*>(outPix + (i64)(it * 8 + (u32)r) * p.nCols + (i64)jt * 8 + h * V) = o;
  }
  if (bad) raiseError(st, kFailed, blockIdx.x);
  // the last workgroup proves that the stream holds exactly the expected number of blocks
  if (c1 == wp.nChunks && threadIdx.x == 0 && B0 + nB != wp.nBlocks) atomicOr(fallback, 64u);
}

__global__ void __launch_bounds__(64) k_fast_fletcher_sum(u64* __restrict__ slotFletcher, u64* __restrict__ out2)
{
  const int lane = laneId();
  const u64 A = waveSum(slotFletcher[2 * lane] % 65535u), B = waveSum(slotFletcher[2 * lane + 1] % 65535u);
  if (lane == 0) { out2[0] = A % 65535u; out2[1] = B % 65535u; }
}

// ------------------------------------------------------------------------------------------------
bool fastDecodeEligible(int dt, int version, int mb, int nRows, int nCols, int nDepth, bool allValid)
{
  if (!allValid || nDepth != 1 || mb != 8 || version < 3) return false;
  if (nRows % 8 != 0 || nCols % (kFastBlocksPerWG * 8) != 0) return false;
  if (dt == DT_Char || dt == DT_Byte) return false;
  return true;
}

FastWalkPlan makeFastWalkPlan(int nRows, int nCols, u32 dataBegin, u32 blobEnd)
{
  FastWalkPlan wp;
  const u32 span = blobEnd > dataBegin ? blobEnd - dataBegin : 0;
  wp.nChunks = span ? (span + kFastChunkBytes - 1) / kFastChunkBytes : 1;
  wp.nBlocks = (u32)(nRows / 8) * (u32)(nCols / 8);
  return wp;
}

template<class T>
static void launchFastDecodeT(int stage, const BandParams& p, const FastWalkPlan& wp, const u8* blob, u32 dataBegin, u32 blobEnd,
                              const FastDecodeBuffers& b, void* out, DeviceStatus* status, hipStream_t st)
{
  constexpr int DT = DtOf<T>::v;
  constexpr int TBYTES = (int)sizeof(T);
  if (stage == 0)
  {
    const u32 nWG = (wp.nChunks + kWalkChunksPerWG - 1) / kWalkChunksPerWG;
    hipMemsetAsync(b.countAt, 0xFF, (size_t)wp.nChunks * kFastWindow(TBYTES) * 2, st);
    hipLaunchKernelGGL(k_fast_walk<DT>, dim3(nWG), dim3(256), 0, st, p.version, wp, blob, dataBegin, blobEnd, b.chunkExit, b.countAt,
                       b.subEntry, b.fallback);
  }
  else if (stage == 1)
  {
    hipLaunchKernelGGL(k_fast_resolve, dim3((wp.nChunks + 256) / 256), dim3(256), 0, st, wp, (u32)kFastWindow(TBYTES), dataBegin, blobEnd,
                       (const u32*)b.chunkExit, (const u16*)b.countAt, b.chunkEntry, b.chunkCount, b.fallback);
    launchExclusiveScan(b.chunkCount, b.chunkBase, wp.nChunks, b.scanScratch, st);
  }
  else
  {
    const u32 nWG = (wp.nChunks + kDecodeChunksPerWG - 1) / kDecodeChunksPerWG;
    hipMemsetAsync(b.slotFletcher, 0, 2 * kFastSlots * 8, st);
    hipLaunchKernelGGL(k_fast_decode<T>, dim3(nWG), dim3(256), 0, st, p, wp, blob, dataBegin, blobEnd, (const u32*)b.chunkEntry,
                       (const u32*)b.chunkBase, (const u32*)b.subEntry, (T*)out, b.slotFletcher, b.fallback, status);
    hipLaunchKernelGGL(k_fast_fletcher_sum, dim3(1), dim3(64), 0, st, b.slotFletcher, b.fletcherOut);
  }
}

void launchFastDecode(int stage, const BandParams& p, const FastWalkPlan& wp, const u8* blob, u32 dataBegin, u32 blobEnd,
                      const FastDecodeBuffers& b, void* out, DeviceStatus* status, hipStream_t st)
{
  switch (p.dt)
  {
    case DT_Short:  launchFastDecodeT<short>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    case DT_UShort: launchFastDecodeT<unsigned short>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    case DT_Int:    launchFastDecodeT<int>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    case DT_UInt:   launchFastDecodeT<unsigned int>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    case DT_Float:  launchFastDecodeT<float>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    case DT_Double: launchFastDecodeT<double>(stage, p, wp, blob, dataBegin, blobEnd, b, out, status, st); break;
    default: break;
  }
}

}    // namespace lerc
