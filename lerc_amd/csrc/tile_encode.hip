// tile_encode.hip -- Lerc2 "tiling" encoder on CDNA4: one wave64 per micro-block position.
//
// Replaces the reference's CPU block loop Lerc2::WriteTiles (Lerc2.cpp:1474-1668) and everything it
// calls per block: GetValidDataAndStats (:1717-1799), NeedToQuantize / Quantize (Lerc2.h:345-376),
// NumBytesTile (Lerc2.h:416-453), ComputeDiffSliceInt (Lerc2.cpp:1803-1874), WriteTile (:1949-2021),
// BitStuffer2::EncodeSimple / EncodeLut / BitStuff (BitStuffer2.cpp:35-153, :432-472).
//
// Mapping (this is the full-coverage kernel: every dtype, masks, partial edge blocks, nDepth > 1 with
// slice-difference encoding, 8x8 and 16x16 blocks; the specialised streaming kernels for the
// all-valid nDepth == 1 case live in tile_fast.hip):
//   * workgroup = 4 waves, wave w handles block position 4 * blockIdx.x + w, all depth slices in turn
//   * lane l holds elements l, l + 64, ... of the block in row-major order; valid elements are
//     compacted with __ballot / __popcll so that "rank" equals the reference's dataBuf index
//   * min / max / same-as-previous counts are wave reductions; the bit width comes from the clz of
//     the quantised maximum; packing ORs each element's bits into an LDS image of the block
//     (ds_or_b32, at most two per element), which is then copied to the blob with coalesced stores
//   * all control flow around cross-lane operations is wave-uniform
//
// FP parity: the quantiser is evaluated in double precision in the reference's
// expression order; this file must be compiled with -ffp-contract=off.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "wave_utils.h"
#include "block_plan.h"

namespace lerc {

// element i of a bit-stuffed field of n elements, nb bits each, that starts at bit `at` of the block image
// (BitStuffer2::BitStuff, BitStuffer2.cpp:432-472; codec 2: BitStuff_Before_Lerc2v3, :292-351)
__device__ __forceinline__ void stuffElement(u32* obuf, u32 at, u32 i, u32 v, int nb, u32 n, int version)
{
  if (version >= 3) { orBits(obuf, at + i * (u32)nb, v, nb); return; }
  const OldBitLayout o = oldBitLayout(i, nb, n);
  orBits(obuf, at + o.pos0, v >> o.n1, (int)o.n0);
  if (o.n1) orBits(obuf, at + o.pos1, v & ((1u << o.n1) - 1u), (int)o.n1);
}

// Distinct values of the block in increasing order (what the reference gets from SortQuantArray,
// Lerc2.cpp:2255-2266): repeated wave-min extraction.  Returns the number of distinct values, stores
// them to lutOut (if not null) and the per-element index into idx.
template<int E>
__device__ __forceinline__ u32 extractDistinct(const u32 (&q)[E], const int (&rank)[E], u32* lutOut, u32 (&idx)[E])
{
  u32 count = 0, last = 0;
  for (;;)
  {
    u32 m = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < E; k++)
      if (rank[k] >= 0 && (count == 0 || q[k] > last) && q[k] < m) m = q[k];
    m = waveMin(m);
    if (m == 0xFFFFFFFFu) break;
#pragma unroll
    for (int k = 0; k < E; k++)
      if (rank[k] >= 0 && q[k] == m) idx[k] = count;
    if (lutOut && laneId() == 0) lutOut[count] = m;
    last = m;
    count++;
  }
  return count;
}


// Builds the byte image of one block in LDS (obuf, zeroed here) -- Lerc2::WriteTile.
template<class Z, int E>
__device__ __forceinline__ void composeBlock(u32* obuf, u32* lutBuf, const BandParams& p, const Plan& pl, int n, int j0,
                                             bool diff, Z zMin, const Z (&val)[E], const u32 (&q)[E],
                                             const int (&rank)[E], u32 qMax)
{
  const int lane = laneId();
  const int nWords = (pl.nBytes + 3) / 4 + 2;
  for (int i = lane; i < nWords; i += 64) obuf[i] = 0;
  waveSync();

  u32 flag = (u32)(((j0 >> 3) & 15) << 2);
  if (p.version >= 5) flag = diff ? (flag | 4u) : (flag & 0x38u);

  if (pl.kind == 0)
  {
    if (lane == 0) orBits(obuf, 0, flag | 2u, 8);
  }
  else if (pl.kind == 1)
  {
    if (lane == 0) orBits(obuf, 0, flag, 8);
#pragma unroll
    for (int k = 0; k < E; k++)
      if (rank[k] >= 0)
      {
        u64 bits = 0;
        Z tmp = val[k];
        memcpy(&bits, &tmp, sizeof(Z));
        const u32 bp = 8u * (1u + (u32)rank[k] * (u32)sizeof(Z));
        if (sizeof(Z) <= 4) orBits(obuf, bp, (u32)bits, 8 * (int)sizeof(Z));
        else { orBits(obuf, bp, (u32)bits, 32); orBits(obuf, bp + 32, (u32)(bits >> 32), 32); }
      }
  }
  else
  {
    flag |= (pl.kind == 2) ? 3u : 1u;
    flag |= (u32)pl.tc << 6;
    const int offBytes = dtSize(pl.dtRed);
    if (lane == 0)
    {
      orBits(obuf, 0, flag, 8);
      const u64 ob = typedBits((double)zMin, pl.dtRed);
      if (offBytes <= 4) orBits(obuf, 8, (u32)ob, 8 * offBytes);
      else { orBits(obuf, 8, (u32)ob, 32); orBits(obuf, 40, (u32)(ob >> 32), 32); }
    }
    if (pl.kind >= 3)
    {
      const int cb = countFieldBytes((u32)n);
      const u32 code = (cb == 4) ? 0u : (u32)(3 - cb);
      const int nb = bitLen(qMax);
      u32 at = 8u * (1u + (u32)offBytes);    // bit cursor
      if (pl.kind == 3)
      {
        if (lane == 0) { orBits(obuf, at, (u32)nb | (code << 6), 8); orBits(obuf, at + 8, (u32)n, 8 * cb); }
        at += 8u * (1u + (u32)cb);
#pragma unroll
        for (int k = 0; k < E; k++)
          if (rank[k] >= 0) stuffElement(obuf, at, (u32)rank[k], q[k], nb, (u32)n, p.version);
      }
      else
      {
        u32 idx[E];
#pragma unroll
        for (int k = 0; k < E; k++) idx[k] = 0;
        const u32 nDistinct = extractDistinct<E>(q, rank, lutBuf, idx);
        waveSync();
        const u32 nLut = nDistinct - 1;
        const int nbIdx = bitLen(nLut);
        if (lane == 0)
        {
          orBits(obuf, at, (u32)nb | (code << 6) | 32u, 8);
          orBits(obuf, at + 8, (u32)n, 8 * cb);
          orBits(obuf, at + 8u * (1u + (u32)cb), nLut + 1, 8);
        }
        at += 8u * (2u + (u32)cb);
        for (u32 i = (u32)lane; i < nLut; i += 64) stuffElement(obuf, at, i, lutBuf[i + 1], nb, nLut, p.version);
        at += 8u * ((nLut * (u32)nb + 7) >> 3);
#pragma unroll
        for (int k = 0; k < E; k++)
          if (rank[k] >= 0) stuffElement(obuf, at, (u32)rank[k], idx[k], nbIdx, (u32)n, p.version);
      }
    }
  }
  waveSync();
}

template<class T, int MB, bool WRITE>
__global__ void __launch_bounds__(256)
k_encode_tiles(const T* __restrict__ data, const u8* __restrict__ maskBits, BandParams p, u32* __restrict__ sizes,
               const u32* __restrict__ offsets, u8* __restrict__ out, DeviceStatus* st)
{
  constexpr int E = MB * MB / 64, NMAX = MB * MB;
  constexpr int OBW = (1 + NMAX * (int)sizeof(T) + 3) / 4 + 4;
  __shared__ T s_val[2][4][NMAX];
  __shared__ int s_diff[4][NMAX];
  __shared__ u32 s_obuf[4][WRITE ? OBW : 1];
  __shared__ u32 s_lut[4][WRITE ? NMAX : 1];

  const int w = waveId(), lane = laneId();
  const int pos = (int)blockIdx.x * 4 + w;
  if (pos >= p.nTV * p.nTH) return;    // whole wave leaves together
  const int it = pos / p.nTH, jt = pos - it * p.nTH;
  const int i0 = it * MB, j0 = jt * MB;
  const int tileH = min(MB, p.nRows - i0), tileW = min(MB, p.nCols - j0);
  const int nElem = tileH * tileW;
  const int nD = p.nDepth;
  const u64 lt = laneMaskLt();

  // --- which elements are valid, and their rank in the reference's row-major gather order
  int rank[E];
  i64 pix[E];
  int n = 0;
#pragma unroll
  for (int k = 0; k < E; k++)
  {
    const int e = k * 64 + lane;
    const bool inb = e < nElem;
    const int r = inb ? e / tileW : 0, c = inb ? e - r * tileW : 0;
    pix[k] = (i64)(i0 + r) * p.nCols + (j0 + c);
    const bool valid = inb && (p.allValid || maskBit(maskBits, pix[k]));
    const u64 bal = __ballot(valid);
    rank[k] = valid ? n + __popcll(bal & lt) : -1;
    n += __popcll(bal);
  }

  u32 total = 0;
  u32 outOff = 0;
  if (WRITE) outOff = offsets[pos];
  u32* obuf = s_obuf[w];

  for (int iD = 0; iD < nD; iD++)
  {
    const int cur = iD & 1;
    T* valBuf = s_val[cur][w];
    const T* prevBuf = s_val[cur ^ 1][w];

    if (n == 0)    // empty position: one "all zero" byte per slice (Lerc2.cpp:1534-1538, :1960-1966)
    {
      if (WRITE && lane == 0)
      {
        u32 flag = (u32)(((j0 >> 3) & 15) << 2);
        if (p.version >= 5) flag &= 0x38u;
        out[outOff] = (u8)(flag | 2u);
      }
      outOff += 1; total += 1;
      continue;
    }

    // --- gather this slice
    T v[E];
#pragma unroll
    for (int k = 0; k < E; k++)
    {
      v[k] = T(0);
      if (rank[k] >= 0) { v[k] = data[pix[k] * nD + iD]; valBuf[rank[k]] = v[k]; }
    }
    waveSync();

    // --- statistics (GetValidDataAndStats)
    T mn = valBuf[0], mx = valBuf[0];
#pragma unroll
    for (int k = 0; k < E; k++)
      if (rank[k] >= 0) { mn = (v[k] < mn) ? v[k] : mn; mx = (v[k] > mx) ? v[k] : mx; }
    mn = waveMinT(mn);
    mx = waveMaxT(mx);
    int same = 0;
#pragma unroll
    for (int k = 0; k < E; k++)
    {
      bool s = false;
      if (rank[k] > 0) s = (v[k] == valBuf[rank[k] - 1]);
      else if (rank[k] == 0) s = p.allValid ? (v[k] == T(0)) : false;    // prevVal starts at 0 (all-valid branch only)
      same += __popcll(__ballot(s));
    }
    const bool tryLut = (n > 4) && ((double)mx > (double)mn + 3 * p.maxZErr) && (2 * same > n);

    double mv = 0;
    bool quantOk = false;
    if (p.maxZErr > 0)
    {
      mv = ((double)mx - (double)mn) * p.scale;
      quantOk = !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
    }
    u32 q[E];
    u32 qMax = 0;
#pragma unroll
    for (int k = 0; k < E; k++) q[k] = 0;
    if (quantOk)
    {
      const double z0 = (double)mn;
#pragma unroll
      for (int k = 0; k < E; k++)
        if (rank[k] >= 0)
        {
          q[k] = p.intLossless ? quantLossless<T>(v[k], mn) : (u32)(((double)v[k] - z0) * p.scale + 0.5);
          qMax = q[k] > qMax ? q[k] : qMax;
        }
      qMax = waveMax(qMax);
    }
    u32 nDistinct = 0;
    if (tryLut && quantOk)
    {
      u32 idxTmp[E];
      nDistinct = extractDistinct<E>(q, rank, nullptr, idxTmp);
    }
    const Plan plan = planBlock<T>(p, n, mn, mx, p.dt, tryLut, mv, qMax, nDistinct);

    // --- optional: difference to the previous depth slice (integer lossless only)
    bool useDiff = false;
    Plan planD = plan;
    int d[E], mnD = 0;
    u32 qD[E], qMaxD = 0;
#pragma unroll
    for (int k = 0; k < E; k++) { d[k] = 0; qD[k] = 0; }
    if (p.tryDiff && iD > 0)
    {
      bool ovf = false;
#pragma unroll
      for (int k = 0; k < E; k++)
        if (rank[k] >= 0)
        {
          const T pv = prevBuf[rank[k]];
          if (!p.checkOverflow) d[k] = (int)((i64)(int)v[k] - (i64)(int)pv);
          else
          {
            const double z = (double)v[k] - (double)pv;
            if (z < -2147483648.0 || z > 2147483647.0) ovf = true; else d[k] = (int)z;
          }
          s_diff[w][rank[k]] = d[k];
        }
      waveSync();
      if (!__any(ovf))
      {
        int lo = s_diff[w][0], hi = lo;
#pragma unroll
        for (int k = 0; k < E; k++)
          if (rank[k] >= 0) { lo = d[k] < lo ? d[k] : lo; hi = d[k] > hi ? d[k] : hi; }
        lo = waveMin(lo); hi = waveMax(hi);
        mnD = lo;
        int sameD = 0;
#pragma unroll
        for (int k = 0; k < E; k++)
        {
          bool s = false;
          if (rank[k] > 0) s = (d[k] == s_diff[w][rank[k] - 1]);
          else if (rank[k] == 0) s = (d[k] == 0);    // ComputeDiffSliceInt: prevVal(0) also for masked blocks
          sameD += __popcll(__ballot(s));
        }
        const bool tryLutD = (n > 4) && ((double)hi > (double)lo + 3 * p.maxZErr) && (2 * sameD > n);
        const double mvD = ((double)hi - (double)lo) * p.scale;
        const bool quantOkD = !(mvD > (double)p.maxQ || (u32)(mvD + 0.5) == 0);
        if (quantOkD)
        {
#pragma unroll
          for (int k = 0; k < E; k++)
            if (rank[k] >= 0) { qD[k] = (u32)((i64)d[k] - (i64)lo); qMaxD = qD[k] > qMaxD ? qD[k] : qMaxD; }
          qMaxD = waveMax(qMaxD);
        }
        u32 nDistinctD = 0;
        if (tryLutD && quantOkD)
        {
          u32 idxTmp[E];
          nDistinctD = extractDistinct<E>(qD, rank, nullptr, idxTmp);
        }
        planD = planBlock<int>(p, n, lo, hi, DT_Int, tryLutD, mvD, qMaxD, nDistinctD);
        useDiff = plan.nBytes > planD.nBytes;
      }
      waveSync();
    }

    const int nBytes = useDiff ? planD.nBytes : plan.nBytes;

    if (WRITE)
    {
      if (!useDiff) composeBlock<T, E>(obuf, s_lut[w], p, plan, n, j0, false, mn, v, q, rank, qMax);
      else composeBlock<int, E>(obuf, s_lut[w], p, planD, n, j0, true, mnD, d, qD, rank, qMaxD);
      const u8* ob8 = reinterpret_cast<const u8*>(obuf);
      for (int b = lane; b < nBytes; b += 64) out[outOff + b] = ob8[b];
      waveSync();
    }
    outOff += (u32)nBytes;
    total += (u32)nBytes;
  }
  if (!WRITE && lane == 0) sizes[pos] = total;
  (void)st;
}

// ---- sizes only, 8-bit values, every pixel valid, lossless, whole 8 x 8 blocks, a few values per pixel -----------------
// One LANE per block position instead of one wave: the block's 8 rows of 8 * D bytes sit in the lane's registers (8-byte
// loads; the lanes of a wave read neighbouring blocks, so whole cache lines are used), and what k_encode_tiles gets from wave
// reductions -- range, "same as the previous element", distinct values, the same for the difference to the previous depth
// slice -- is plain serial arithmetic over 64 bytes.  This is the dry run behind the choice between tiling and Huffman
// coding for 8-bit imagery (Lerc2.cpp:317-373), where the general kernel's ~9 000 cycles per position were the largest
// single item of the encode.  Same decisions, same sums: tests compare the two kernels' totals.
struct DistinctBits
{
  u64 bm[8];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int i = 0; i < 8; i++) bm[i] = 0ull; }
  __device__ __forceinline__ void set(u32 q)    // q < 512
  {
    const u32 hi = q >> 6;
    const u64 bit = 1ull << (q & 63u);
#pragma unroll
    for (int i = 0; i < 8; i++) if ((u32)i == hi) bm[i] |= bit;
  }
  __device__ __forceinline__ u32 count() const
  {
    u32 n = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) n += (u32)__popcll(bm[i]);
    return n;
  }
};

template<class T, int D>
__global__ void __launch_bounds__(256)
k_tile_sizes_bytes(const T* __restrict__ data, BandParams p, u32* __restrict__ sizes)
{
  const int pos = (int)(blockIdx.x * 256u + threadIdx.x);
  if (pos >= p.nTV * p.nTH) return;
  const int it = pos / p.nTH, jt = pos - it * p.nTH;
  const i64 rowBytes = (i64)p.nCols * D;
  const u8* base = reinterpret_cast<const u8*>(data) + ((i64)it * 8 * p.nCols + (i64)jt * 8) * D;
#define LERC_PX(c, d) ((int)(T)(u8)(w[((c) * D + (d)) >> 3] >> (8 * (((c) * D + (d)) & 7))))
  // ---- first sweep over the rows: ranges and "same as the previous element" counts of every slice and of its
  // difference to the slice in front (GetValidDataAndStats, all-valid branch: prevVal starts at 0; ComputeDiffSliceInt,
  // Lerc2.cpp:1803-1874)
  int mn[D], mx[D], same[D], prev[D], lo[D], hi[D], sameD[D], prevD[D];
#pragma unroll
  for (int d = 0; d < D; d++) { mn[d] = 0x7FFFFFFF; mx[d] = -0x7FFFFFFF; same[d] = 0; prev[d] = 0; lo[d] = 0x7FFFFFFF; hi[d] = -0x7FFFFFFF; sameD[d] = 0; prevD[d] = 0; }
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    u64 w[D];
#pragma unroll
    for (int x = 0; x < D; x++) w[x] = *reinterpret_cast<const u64*>(base + r * rowBytes + 8 * x);
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
      for (int d = 0; d < D; d++)
      {
        const int v = LERC_PX(c, d);
        mn[d] = v < mn[d] ? v : mn[d]; mx[d] = v > mx[d] ? v : mx[d];
        same[d] += (v == prev[d]) ? 1 : 0;
        prev[d] = v;
        if (d > 0)
        {
          const int dv = v - LERC_PX(c, d > 0 ? d - 1 : 0);
          lo[d] = dv < lo[d] ? dv : lo[d]; hi[d] = dv > hi[d] ? dv : hi[d];
          sameD[d] += (dv == prevD[d]) ? 1 : 0;
          prevD[d] = dv;
        }
      }
  }
  u32 total = 0;
#pragma unroll
  for (int d = 0; d < D; d++)
  {
    const bool tryLut = ((double)mx[d] > (double)mn[d] + 3 * p.maxZErr) && (2 * same[d] > 64);
    double mv = 0;
    bool quantOk = false;
    if (p.maxZErr > 0)
    {
      mv = ((double)mx[d] - (double)mn[d]) * p.scale;
      quantOk = !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
    }
    const u32 qMax = quantOk ? (u32)(mx[d] - mn[d]) : 0u;    // lossless: the quantised value is v - mn
    const bool diff = d > 0 && p.tryDiff;
    const bool tryLutD = diff && ((double)hi[d] > (double)lo[d] + 3 * p.maxZErr) && (2 * sameD[d] > 64);
    const double mvD = diff ? ((double)hi[d] - (double)lo[d]) * p.scale : 0;
    const bool quantOkD = diff && !(mvD > (double)p.maxQ || (u32)(mvD + 0.5) == 0);
    const u32 qMaxD = quantOkD ? (u32)(hi[d] - lo[d]) : 0u;
    // ---- second sweep where a look-up table is on the cards: the number of distinct values
    u32 nDistinct = 0, nDistinctD = 0;
    if ((tryLut && quantOk) || (tryLutD && quantOkD))
    {
      DistinctBits plain, delta;
      plain.clear(); delta.clear();
      for (int r = 0; r < 8; r++)
      {
        u64 w[D];
#pragma unroll
        for (int x = 0; x < D; x++) w[x] = *reinterpret_cast<const u64*>(base + r * rowBytes + 8 * x);
#pragma unroll
        for (int c = 0; c < 8; c++)
        {
          const int v = LERC_PX(c, d);
          plain.set((u32)(v - mn[d]));
          if (d > 0) delta.set((u32)(v - LERC_PX(c, d > 0 ? d - 1 : 0) - lo[d]));
        }
      }
      nDistinct = plain.count(); nDistinctD = delta.count();
    }
    const Plan plan = planBlock<T>(p, 64, (T)mn[d], (T)mx[d], p.dt, tryLut, mv, qMax, nDistinct);
    int nBytes = plan.nBytes;
    if (diff)
    {
      const Plan planD = planBlock<int>(p, 64, lo[d], hi[d], DT_Int, tryLutD, mvD, qMaxD, nDistinctD);
      if (plan.nBytes > planD.nBytes) nBytes = planD.nBytes;
    }
    total += (u32)nBytes;
  }
#undef LERC_PX
  sizes[pos] = total;
}

template<class T>
static bool launchSizesBytes(int mb, const void* data, const u8* maskBits, const BandParams& p, u32* sizes, hipStream_t stream)
{
  if (sizeof(T) != 1 || mb != 8 || maskBits || !p.allValid || !p.intLossless || p.checkOverflow || p.nDepth > 4
      || (p.nRows & 7) || (p.nCols & 7) || ((uintptr_t)data & 7)) return false;
  const int nPos = p.nTV * p.nTH;
  const dim3 grid((nPos + 255) / 256), block(256);
  switch (p.nDepth)
  {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_sizes_bytes<T, 1>), grid, block, 0, stream, (const T*)data, p, sizes); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_sizes_bytes<T, 2>), grid, block, 0, stream, (const T*)data, p, sizes); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_sizes_bytes<T, 3>), grid, block, 0, stream, (const T*)data, p, sizes); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_tile_sizes_bytes<T, 4>), grid, block, 0, stream, (const T*)data, p, sizes); break;
  }
  return true;
}

template<class T>
static void launchSizesT(int mb, const void* data, const u8* maskBits, const BandParams& p, u32* sizes, DeviceStatus* st,
                         hipStream_t stream)
{
  const int nPos = p.nTV * p.nTH;
  const dim3 grid((nPos + 3) / 4), block(256);
  if (mb == 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_encode_tiles<T, 8, false>), grid, block, 0, stream, (const T*)data, maskBits, p, sizes,
                       (const u32*)nullptr, (u8*)nullptr, st);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_encode_tiles<T, 16, false>), grid, block, 0, stream, (const T*)data, maskBits, p, sizes,
                       (const u32*)nullptr, (u8*)nullptr, st);
}

template<class T>
static void launchWriteT(int mb, const void* data, const u8* maskBits, const BandParams& p, const u32* offsets, u8* out,
                         DeviceStatus* st, hipStream_t stream)
{
  const int nPos = p.nTV * p.nTH;
  const dim3 grid((nPos + 3) / 4), block(256);
  if (mb == 8)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_encode_tiles<T, 8, true>), grid, block, 0, stream, (const T*)data, maskBits, p,
                       (u32*)nullptr, offsets, out, st);
  else
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_encode_tiles<T, 16, true>), grid, block, 0, stream, (const T*)data, maskBits, p,
                       (u32*)nullptr, offsets, out, st);
}

#define LERC_DT_SWITCH(dt, CALL)                                      \
  switch (dt) {                                                       \
    case DT_Char:   { typedef signed char TT; CALL; break; }          \
    case DT_Byte:   { typedef unsigned char TT; CALL; break; }        \
    case DT_Short:  { typedef short TT; CALL; break; }                \
    case DT_UShort: { typedef unsigned short TT; CALL; break; }       \
    case DT_Int:    { typedef int TT; CALL; break; }                  \
    case DT_UInt:   { typedef unsigned int TT; CALL; break; }         \
    case DT_Float:  { typedef float TT; CALL; break; }                \
    case DT_Double: { typedef double TT; CALL; break; }               \
    default: break;                                                   \
  }

// Float types at maxZErr 0 (the lossless float mode's rival, Lerc2.cpp:1418-1452): a block is one byte (no valid pixel, or all
// of them zero), raw if two of its values differ, else a constant (Lerc2::NumBytesTile, Lerc2.h:416-431) -- all a block's size
// takes is its valid count and its range.  A LANE per block position and nothing else, instead of k_encode_tiles' wave per
// block with its quantiser, LUT candidates and LDS staging (8192^2 float32: 337 -> R us).
template<class T>
__global__ void __launch_bounds__(256) k_tile_sizes_lossless_flt(const T* __restrict__ data, const u8* __restrict__ maskBits, BandParams p, u32* __restrict__ sizes)
{
  const int pos = (int)(blockIdx.x * 256 + threadIdx.x);
  if (pos >= p.nTV * p.nTH) return;
  const int it = pos / p.nTH, jt = pos - it * p.nTH;
  const int i0 = it * p.mb, j0 = jt * p.mb;
  const int i1 = min(p.nRows, i0 + p.mb), j1 = min(p.nCols, j0 + p.mb);
  int n = 0;
  T mn = T(0), mx = T(0);
  // whole blocks of 8 x 8 values, every pixel valid, rows of whole 16-byte units: a row of the block in 16-byte loads
  constexpr int PV = 16 / (int)sizeof(T);
  if (p.mb == 8 && p.allValid && i1 - i0 == 8 && j1 - j0 == 8 && (p.nCols % PV) == 0 && ((uintptr_t)data & 15) == 0)
  {
    struct alignas(16) Vec { T v[PV]; };
    bool first = true;
#pragma unroll
    for (int r = 0; r < 8; r++)
    {
      const Vec* src = reinterpret_cast<const Vec*>(data + (i64)(i0 + r) * p.nCols + j0);
#pragma unroll
      for (int u = 0; u < 8 / PV; u++)
      {
        const Vec x = src[u];
#pragma unroll
        for (int q = 0; q < PV; q++)
        {
          const T v = x.v[q];
          if (first) { mn = v; mx = v; first = false; }
          else { mn = (v < mn) ? v : mn; mx = (v > mx) ? v : mx; }
        }
      }
    }
    n = 64;
  }
  else
  for (int i = i0; i < i1; i++)
  {
    const i64 row = (i64)i * p.nCols;
    for (int j = j0; j < j1; j++)
    {
      if (!p.allValid && !maskBit(maskBits, row + j)) continue;
      const T v = data[row + j];
      if (n == 0) { mn = v; mx = v; }
      else { mn = (v < mn) ? v : mn; mx = (v > mx) ? v : mx; }
      n++;
    }
  }
  const Plan pl = planBlock<T>(p, n, mn, mx, p.dt, false, 0.0, 0u, 0u);
  sizes[pos] = (u32)pl.nBytes;
}

void launchTileSizes(int dt, int mb, const void* data, const u8* maskBits, const BandParams& p, u32* sizes,
                     DeviceStatus* st, hipStream_t stream)
{
  if (dt == DT_Char && launchSizesBytes<signed char>(mb, data, maskBits, p, sizes, stream)) return;
  if (dt == DT_Byte && launchSizesBytes<unsigned char>(mb, data, maskBits, p, sizes, stream)) return;
  if ((dt == DT_Float || dt == DT_Double) && p.maxZErr == 0 && p.nDepth == 1 && !p.tryDiff && p.mb == mb)
  {
    const int nPos = p.nTV * p.nTH;
    const dim3 grid((nPos + 255) / 256), block(256);
    if (dt == DT_Float) hipLaunchKernelGGL(k_tile_sizes_lossless_flt<float>, grid, block, 0, stream, (const float*)data, maskBits, p, sizes);
    else hipLaunchKernelGGL(k_tile_sizes_lossless_flt<double>, grid, block, 0, stream, (const double*)data, maskBits, p, sizes);
#ifdef HIPSIM
    {   // (emulator builds: the general kernel's sizes, block by block)
      hipStreamSynchronize(stream);
      std::vector<u32> want((size_t)nPos);
      u32* tmp = nullptr;
      if (hipMalloc(&tmp, (size_t)nPos * 4) == hipSuccess)
      {
        LERC_DT_SWITCH(dt, launchSizesT<TT>(mb, data, maskBits, p, tmp, st, stream))
        hipStreamSynchronize(stream);
        for (int i = 0; i < nPos; i++)
          if (tmp[i] != sizes[i]) { fprintf(stderr, "k_tile_sizes_lossless_flt: block %d: %u bytes, the general kernel says %u\n", i, sizes[i], tmp[i]); abort(); }
        hipFree(tmp);
      }
    }
#endif
    return;
  }
  LERC_DT_SWITCH(dt, launchSizesT<TT>(mb, data, maskBits, p, sizes, st, stream))
}

void launchTileWrite(int dt, int mb, const void* data, const u8* maskBits, const BandParams& p, const u32* offsets,
                     u8* out, DeviceStatus* st, hipStream_t stream)
{
  LERC_DT_SWITCH(dt, launchWriteT<TT>(mb, data, maskBits, p, offsets, out, st, stream))
}

}    // namespace lerc
