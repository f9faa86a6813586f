// tile_fast.hip -- streaming encoder kernels for the common case (see tile_fast.h for the conditions).
//
// Same arithmetic and the same bytes as the general wave-per-block kernel in tile_encode.hip, arranged
// for HBM bandwidth instead of generality (measured: the first version of these kernels was VALU-issue
// bound at ~420 instructions per 4-block tile, profiles/r01_pmc_fast_v1.txt; this layout cuts that ~3x):
//   * a lane owns V consecutive pixels of one raster row (one 16-byte global load); 8 / V lanes form a
//     block row, 64 / V lanes an 8 x 8 block, so a wave64 covers 8 raster rows x 128 bytes -- whole
//     cache lines per row -- with the pixels in registers (no LDS staging of the input)
//   * per block only min / max / "same as previous" are reduced across lanes (butterflies over lane
//     bits 0-1 and 3-5); everything the reference decides per block (NeedToQuantize, NumBytesTile,
//     ReduceDataType) is evaluated ONCE per block with lane = block over the workgroup's 64 blocks
//   * pass 1 (k_fast_stats) = global statistics of Lerc::FilterNoDataAndNaN / ComputeMinMaxRanges + the
//     size-only dry run of Lerc2::WriteTiles; it leaves a 16-byte descriptor per block
//   * pass 2 (k_fast_pack) reads pixels + descriptors, assembles the workgroup's contiguous output span
//     in LDS with ds_or_b32 and flushes it with 16-byte stores, accumulating the Fletcher32 sums of the
//     bytes it stores (word-wise)
//   * the decisions the reference takes between its sweeps are taken on the device (k_fast_decide), so
//     an encode costs one host synchronisation
// Reference: Lerc2.cpp:1474-1668, :1717-1799, :1949-2021; Lerc2.h:337-453; BitStuffer2.cpp:35-153.
#include "tile_fast.h"
#include "kernels.h"
#include "wave_utils.h"
#include "block_plan.h"

namespace lerc {

PROBE_DEFINE(fast_encode)

enum FastRedo : u32
{
  kRedoNaN = 1, kRedoAllInt = 2, kRedoRaise = 4, kRedoConst = 8, kRedoMb16 = 16, kRedoOneSweep = 32, kRedoCapacity = 64
};

template<class T> struct FastCfg
{
  static constexpr int V = (sizeof(T) >= 4) ? 16 / (int)sizeof(T) : 8;    // pixels per lane: f32 4, f64 2, 16-bit 8
  static constexpr int LPR = 8 / V;                                      // lanes per block row
  static constexpr int BPW = 8 / LPR;                                    // blocks per wave tile
  static constexpr int TILE_COLS = 8 * V;                                // raster columns per wave tile
  static constexpr int IT = kFastBlocksPerWG / (4 * BPW);                // wave tiles per wave
};

// butterfly reductions over the lanes of one block
template<int LPR, class X> __device__ __forceinline__ X groupMin(X v)
{
  if (LPR > 1) { X o = __shfl_xor(v, 1); v = o < v ? o : v; }
  if (LPR > 2) { X o = __shfl_xor(v, 2); v = o < v ? o : v; }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) { X o = __shfl_xor(v, m); v = o < v ? o : v; }
  return v;
}
template<int LPR, class X> __device__ __forceinline__ X groupMax(X v)
{
  if (LPR > 1) { X o = __shfl_xor(v, 1); v = o > v ? o : v; }
  if (LPR > 2) { X o = __shfl_xor(v, 2); v = o > v ? o : v; }
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) { X o = __shfl_xor(v, m); v = o > v ? o : v; }
  return v;
}
template<int LPR> __device__ __forceinline__ int groupSum(int v)
{
  if (LPR > 1) v += __shfl_xor(v, 1);
  if (LPR > 2) v += __shfl_xor(v, 2);
#pragma unroll
  for (int m = 8; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

template<class T> __device__ __forceinline__ T shflT(T v, int src) { return (T)__shfl((typename ShflT<T>::type)v, src); }

template<class T> __device__ __forceinline__ bool notIntegral(T) { return false; }
template<> __device__ __forceinline__ bool notIntegral<float>(float v) { return !(v == truncf(v)); }    // == !IsInt (Lerc.h:271) for finite v
template<> __device__ __forceinline__ bool notIntegral<double>(double v) { return !(v == trunc(v)); }
template<class T> __device__ __forceinline__ bool isNaNv(T) { return false; }
template<> __device__ __forceinline__ bool isNaNv<float>(float v) { return v != v; }
template<> __device__ __forceinline__ bool isNaNv<double>(double v) { return v != v; }

// loads the V pixels of this lane (one aligned vector load)
template<class T, int V>
__device__ __forceinline__ void loadLane(const T* __restrict__ p, T (&v)[V])
{
  struct alignas(sizeof(T) * V) Vec { T e[V]; };
  const Vec x = *reinterpret_cast<const Vec*>(p);
#pragma unroll
  for (int k = 0; k < V; k++) v[k] = x.e[k];
}

template<class T, int V>
__device__ __forceinline__ void quantizeLane(int intLossless, double scale, const T (&v)[V], T mn, u32 (&q)[V])
{
  const double z0 = (double)mn;
#pragma unroll
  for (int k = 0; k < V; k++)
    q[k] = (DtOf<T>::v < DT_Float && intLossless) ? quantLossless<T>(v[k], mn) : (u32)(((double)v[k] - z0) * scale + 0.5);
}

// number of distinct quantised values in each block of the wave (only meaningful where `need`)
template<int LB, int V>
__device__ __forceinline__ u32 groupDistinct(const u32 (&q)[V], bool need)
{
  u32 count = 0, last = 0;
  bool active = need;
  for (;;)
  {
    u32 m = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < V; k++)
      if ((count == 0 || q[k] > last) && q[k] < m) m = q[k];
    m = groupReduce<LB>(m, OpMin());
    if (m == 0xFFFFFFFFu) active = false;
    if (!__any(active)) break;
    if (active) { last = m; count++; }
  }
  return count;
}

template<class T> struct FKey;
template<> struct FKey<float> { static __device__ u64 enc(float v) { u32 b; memcpy(&b, &v, 4); b = (b & 0x80000000u) ? ~b : (b | 0x80000000u); return b; } };
template<> struct FKey<double> { static __device__ u64 enc(double v) { u64 b; memcpy(&b, &v, 8); return (b >> 63) ? ~b : (b | (1ull << 63)); } };
template<> struct FKey<int> { static __device__ u64 enc(int v) { return (u64)((i64)v + (1ll << 62)); } };
template<> struct FKey<unsigned int> { static __device__ u64 enc(unsigned int v) { return (u64)v + (1ull << 62); } };
template<> struct FKey<short> { static __device__ u64 enc(short v) { return (u64)((i64)v + (1ll << 62)); } };
template<> struct FKey<unsigned short> { static __device__ u64 enc(unsigned short v) { return (u64)v + (1ull << 62); } };

// Raster offset of the first pixel a lane owns in wave tile `tl` (0 .. 15) of its workgroup.  WIDE: the 64 blocks of
// the workgroup lie in one block row (nTH % 64 == 0), the tiles are a constant stride apart and the compiler folds the
// stride into the load / store instruction; otherwise the workgroup spans several block rows (fastSpanOf).
template<bool WIDE, int BPW, int V>
__device__ __forceinline__ i64 laneOrigin(const FastSpan& span, int tl, int r, int c, int nCols)
{
  if (WIDE) return (i64)(span.it0 * 8u + (u32)r) * nCols + (i64)span.jt0 * 8 + tl * (BPW * 8) + c * V;
  constexpr int LPR = 8 / V;    // lanes per block row; c = block of the wave tile * LPR + lane of the row
  const u32 j = (u32)tl * BPW + (u32)(c / LPR);
  return (i64)(fastSpanRow(span, j) * 8u + (u32)r) * nCols + (i64)fastSpanCol(span, j) * 8 + (c % LPR) * V;
}

template<class T> __device__ __forceinline__ u64 rawBits(T v) { u64 b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template<class T> __device__ __forceinline__ T fromRawBits(u64 b) { T v; memcpy(&v, &b, sizeof(T)); return v; }

// descriptor word 1: nBytes (16) | kind (3) << 16 | tc (2) << 19 | dtRed (3) << 21 | numBits (5) << 24
__device__ __forceinline__ u32 packDesc(const Plan& pl, int nb) { return (u32)pl.nBytes | ((u32)pl.kind << 16) | ((u32)pl.tc << 19) | ((u32)pl.dtRed << 21) | ((u32)nb << 24); }

// ------------------------------------------------------------------------------------------------
// pass 1: global statistics + block decisions + sizes
// ------------------------------------------------------------------------------------------------
template<class T, bool WIDE>
__global__ void __launch_bounds__(256)
k_fast_stats(const T* __restrict__ data, BandParams p, FastBlockDesc* __restrict__ desc, u32* __restrict__ wgSize,
             u64* __restrict__ wgMinKey, u64* __restrict__ wgMaxKey, u32* __restrict__ wgFlags, FastBatch batch)
{
  {
    const size_t tile = blockIdx.y;    // this tile's slice of every array
    data += tile * batch.tileElems; desc += tile * batch.nWG * kFastBlocksPerWG; wgSize += tile * fastWgStride(batch.nWG);
    wgMinKey += tile * batch.nWG; wgMaxKey += tile * batch.nWG; wgFlags += tile * batch.nWG;
  }
  typedef FastCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW;
  typedef typename ShflT<T>::type ST;
  __shared__ T s_mn[kFastBlocksPerWG], s_mx[kFastBlocksPerWG];
  __shared__ u32 s_same[kFastBlocksPerWG], s_nd[kFastBlocksPerWG];
  __shared__ u32 s_fl[4];
  PROBE_BEGIN;
  // a block is LB consecutive lanes (a DPP row for 32-bit types), so that its reductions never leave the VALU:
  // lane = b * LB + r * LPR + h  (block of the wave tile, raster row of the block, lane of that row)
  constexpr int LB = 8 * LPR;
  const int w = waveId(), lane = laneId();
  const int b = lane / LB, r = (lane % LB) / LPR, h = lane % LPR, c = b * LPR + h;
  const FastSpan span = fastSpanOf(blockIdx.x, (u32)p.nTH, (u32)p.nTV);
  const bool leader = (lane % LB == 0);

  // all loads of the wave in flight before the first use (the data-dependent LUT branch below keeps the compiler
  // from hoisting them itself)
  T vAll[C::IT][V];
#pragma unroll
  for (int t = 0; t < C::IT; t++) loadLane<T, V>(data + laneOrigin<WIDE, BPW, V>(span, t * 4 + w, r, c, p.nCols), vAll[t]);

  u32 flags = 0;
#pragma unroll
  for (int t = 0; t < C::IT; t++)
  {
    const int tile = t * 4 + w;
    T (&v)[V] = vAll[t];
    if (DtOf<T>::v >= DT_Float)
    {
#pragma unroll
      for (int k = 0; k < V; k++) if (isNaNv(v[k])) flags |= 1u;
      if (!(flags & 2u))    // one fractional value settles "not all integers" for good (skipped once every lane has one)
      {
#pragma unroll
        for (int k = 0; k < V; k++) if (notIntegral(v[k])) flags |= 2u;
      }
    }
    T mn = v[0], mx = v[0];
#pragma unroll
    for (int k = 1; k < V; k++) { mn = v[k] < mn ? v[k] : mn; mx = v[k] > mx ? v[k] : mx; }
    mn = (T)groupReduce<LB>((ST)mn, OpMin());
    mx = (T)groupReduce<LB>((ST)mx, OpMax());
    // "same as previous" in row-major block order, prevVal starts at 0 (Lerc2.cpp:1729-1758): the previous
    // pixel vector of the block lives in the previous lane
    T prev = (T)dppMovT<kDppWaveShr1>((ST)v[V - 1]);
    if (leader) prev = T(0);
    int same = (v[0] == prev) ? 1 : 0;
#pragma unroll
    for (int k = 1; k < V; k++) same += (v[k] == v[k - 1]) ? 1 : 0;
    // only "more than half of the 64" matters (tryLut); that needs a lane above the average of 32 / LB
    if (__any(same > 32 / LB)) same = groupReduce<LB>(same, OpSum());
    else same = 0;
    // LUT candidates need the number of distinct quantised values, which only the pixel owners can count
    u32 nd = 0;
    const bool tryLut = (2 * same > 64) && ((double)mx > (double)mn + 3 * p.maxZErr);
    if (__any(tryLut))
    {
      const double mv = ((double)mx - (double)mn) * p.scale;
      const bool need = tryLut && !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
      u32 q[V];
      quantizeLane<T, V>(p.intLossless, p.scale, v, mn, q);
      nd = groupDistinct<LB, V>(q, need);
    }
    if (leader)
    {
      const int blk = tile * BPW + b;
      s_mn[blk] = mn; s_mx[blk] = mx; s_same[blk] = (u32)same; s_nd[blk] = nd;
    }
  }
  const bool f1 = __any(flags & 1u), f2 = __any(flags & 2u);
  if (lane == 0) s_fl[w] = (f1 ? 1u : 0u) | (f2 ? 2u : 0u);
  __syncthreads();
  PROBE(0);
  // the serial per-block phase rotates over the waves (= SIMDs) from workgroup to workgroup, or one SIMD of the CU
  // would carry it for every resident workgroup
  if (w != (int)((blockIdx.x * 2654435761u) >> 30)) return;

  // ---- lane = block: the per-block decisions of Lerc2::NumBytesTile, once
  const T mn = s_mn[lane], mx = s_mx[lane];
  const int same = (int)s_same[lane];
  const bool tryLut = (2 * same > 64) && ((double)mx > (double)mn + 3 * p.maxZErr);
  double mv = 0;
  bool quantOk = false;
  if (p.maxZErr > 0)
  {
    mv = ((double)mx - (double)mn) * p.scale;
    quantOk = !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
  }
  const u32 qMax = quantOk ? (u32)(mv + 0.5) : 0u;    // == largest quantised element (same expression as Quantize)
  Plan pl = planBlock<T>(p, 64, mn, mx, p.dt, tryLut, mv, qMax, s_nd[lane]);
  if (!fastSpanHas(span, (u32)lane)) { pl.nBytes = 0; pl.kind = 7; }    // behind the raster's last block: nothing to write
  FastBlockDesc d;
  d.mnBits = rawBits<T>(mn);
  d.w1 = packDesc(pl, bitLen(qMax));
  d.pad = 0;
  desc[(size_t)blockIdx.x * kFastBlocksPerWG + lane] = d;
  const u32 total = waveSum((u32)pl.nBytes);
  const u64 kMin = waveMin(FKey<T>::enc(mn)), kMax = waveMax(FKey<T>::enc(mx));
  if (lane == 0)
  {
    // per-workgroup partial results, folded by k_fast_scan_decide (thousands of workgroups hammering a few
    // addresses with atomics cost more than the whole pass)
    wgSize[blockIdx.x] = total;
    wgMinKey[blockIdx.x] = kMin;
    wgMaxKey[blockIdx.x] = kMax;
    wgFlags[blockIdx.x] = s_fl[0] | s_fl[1] | s_fl[2] | s_fl[3];
  }
  PROBE(1);
}

// ------------------------------------------------------------------------------------------------
// decisions between the passes (Lerc.cpp:1486-1502, Lerc2.cpp:205-373) + header (Lerc2.cpp:724-786)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double keyToDouble(int dt, u64 key, u64& raw)
{
  if (dt == DT_Float) { u32 b = (u32)key; b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b; raw = b; float f; memcpy(&f, &b, 4); return (double)f; }
  if (dt == DT_Double) { const u64 b = (key >> 63) ? (key & ~(1ull << 63)) : ~key; raw = b; double d; memcpy(&d, &b, 8); return d; }
  const i64 v = (i64)key - (1ll << 62);
  raw = (u64)v;
  return (double)v;
}

__device__ __forceinline__ void
fastDecide(const BandParams& p, double requestedMaxZErr, u32 raiseCandidates, u32 nWG, const u32* __restrict__ wgBase,
           const u64* __restrict__ slotMinKey, const u64* __restrict__ slotMaxKey, const u32* __restrict__ slotFlags,
           const double* __restrict__ row0RaiseErr, u32 nRaiseSets, u32 nBlobsMore, u8* __restrict__ out, u64 outCapacity, FastEncodeResult* res)
{
  const int lane = laneId();
  const u64 a = waveMin(slotMinKey[lane]), b = waveMax(slotMaxKey[lane]);
  u32 f = slotFlags[lane];
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) f |= __shfl_xor(f, m);
  if (lane != 0) return;

  u64 rawMin = 0, rawMax = 0;
  const double zMin = keyToDouble(p.dt, a, rawMin), zMax = keyToDouble(p.dt, b, rawMax);
  const u32 nBytesTiling = wgBase[nWG];
  const int tb = dtSize(p.dt);
  const u64 nPix = (u64)p.nRows * (u64)p.nCols;
  u32 redo = 0;
  const bool isFlt = p.dt >= DT_Float;
  if (isFlt)
  {
    if (f & 1u) redo |= kRedoNaN;
    const double lim = (p.dt == DT_Float) ? 8388608.0 : 9007199254740992.0;
    const bool allInt = !(f & 2u) && zMin >= -lim && zMin <= lim && zMax >= -lim && zMax <= lim;
    if (allInt) redo |= kRedoAllInt;    // maxZErr becomes max(0.5, floor(.)) and the header says bIsInt
    if (raiseCandidates && row0RaiseErr)
    {
      const int fac[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
      for (int c = 0; c < 9; c++)
      {
        double e = 0;
#pragma unroll
        for (u32 w = 0; w < 16u; w++)    // nRaiseSets <= 16; a fixed trip count lets the loads go out together
        {
          const double x = (w < nRaiseSets) ? row0RaiseErr[w * 9 + c] : 0.0;
          e = x > e ? x : e;
        }
        if (((raiseCandidates >> c) & 1u) && !(e / fac[c] > requestedMaxZErr / 2)) redo |= kRedoRaise;
      }
    }
  }
  if (zMin == zMax) redo |= kRedoConst;
  const u64 nBytesOneSweep = (u64)tb * nPix;
  if (((double)((u64)nBytesTiling * 8) < (double)nPix * 1.5) && ((u64)nBytesTiling < 4 * nBytesOneSweep)) redo |= kRedoMb16;
  if (nBytesOneSweep <= (u64)nBytesTiling) redo |= kRedoOneSweep;
  const u32 prefixLen = 90u + 4u + 2u * (u32)tb + 1u;
  const u64 blobSize = (u64)prefixLen + nBytesTiling;
  if (blobSize > outCapacity || blobSize > 0x7FFFFFFFull) redo |= kRedoCapacity;

  res->redoReason = redo;
  res->redo = redo ? 1u : 0u;
  res->blobSize = (u32)blobSize;
  res->nBytesTiling = nBytesTiling;
  res->zMin = zMin; res->zMax = zMax;
  res->minKey = a; res->maxKey = b;
  res->prefixLen = prefixLen;
  res->checksum = 0;
  if (redo) return;

  // header + "no mask" + ranges + "not one sweep" (Lerc2.cpp:396-430)
  u8* o = out;
  const char key[6] = { 'L', 'e', 'r', 'c', '2', ' ' };
  for (int i = 0; i < 6; i++) o[i] = (u8)key[i];
  putBytes(o + 6, (u64)(u32)kCodecVersion, 4);
  putBytes(o + 10, 0, 4);
  const int ints[8] = { p.nRows, p.nCols, 1, (int)nPix, 8, (int)blobSize, p.dt, (int)nBlobsMore };
  for (int i = 0; i < 8; i++) putBytes(o + 14 + 4 * i, (u64)(u32)ints[i], 4);
  putBytes(o + 46, 0, 4);
  const double dbl[5] = { p.maxZErr, zMin, zMax, 0.0, 0.0 };
  for (int i = 0; i < 5; i++) { u64 bits; memcpy(&bits, &dbl[i], 8); putBytes(o + 50 + 8 * i, bits, 8); }
  putBytes(o + 90, 0, 4);
  putBytes(o + 94, rawMin, tb);
  putBytes(o + 94 + tb, rawMax, tb);
  o[94 + 2 * tb] = 0;
}

// the scan of the workgroup sizes, the fold of the per-workgroup statistics and the decisions in one launch
// (1024 threads; the first wave decides)
__global__ void __launch_bounds__(1024)
k_fast_scan_decide(BandParams p, double requestedMaxZErr, u32 raiseCandidates, u32 nWG, const u32* __restrict__ wgSize,
                   u32* __restrict__ wgBase, const u64* __restrict__ wgMinKey, const u64* __restrict__ wgMaxKey,
                   const u32* __restrict__ wgFlags, const double* __restrict__ row0RaiseErr, u8* __restrict__ prefixStage, u64 outCapacity,
                   FastEncodeResult* res, FastBatch batch)
{
  __shared__ u64 s_min[64], s_max[64];
  __shared__ u32 s_fl[64];
  {
    const size_t tile = blockIdx.y;
    wgSize += tile * fastWgStride(batch.nWG); wgBase += tile * fastWgStride(batch.nWG);
    wgMinKey += tile * batch.nWG; wgMaxKey += tile * batch.nWG; wgFlags += tile * batch.nWG;
    if (row0RaiseErr) row0RaiseErr += tile * batch.nRaiseSets * 9;
    prefixStage += tile * kFastPrefixStage; res += tile;
  }
  u64 kMin = ~0ull, kMax = 0ull;
  u32 fl = 0;
  for (u32 i = threadIdx.x; i < nWG; i += 1024u)
  {
    const u64 a = wgMinKey[i], b = wgMaxKey[i];
    kMin = a < kMin ? a : kMin; kMax = b > kMax ? b : kMax;
    fl |= wgFlags[i];
  }
  kMin = waveMin(kMin); kMax = waveMax(kMax);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) fl |= __shfl_xor(fl, m);
  if (threadIdx.x < 64) { s_min[threadIdx.x] = ~0ull; s_max[threadIdx.x] = 0ull; s_fl[threadIdx.x] = 0u; }
  __syncthreads();
  if (laneId() == 0) { s_min[waveId()] = kMin; s_max[waveId()] = kMax; s_fl[waveId()] = fl; }
  scanSingleWorkgroup(wgSize, wgBase, nWG);
  __syncthreads();
  if (waveId() == 0)
    fastDecide(p, requestedMaxZErr, raiseCandidates, nWG, wgBase, s_min, s_max, s_fl, row0RaiseErr, batch.nRaiseSets, batch.nBlobsMore, prefixStage, outCapacity, res);
}

// ------------------------------------------------------------------------------------------------
// pass 2: pack + checksum
// ------------------------------------------------------------------------------------------------
// OR up to 64 bits into the LDS bit stream
__device__ __forceinline__ void orBits64(u32* words, u32 bitPos, u64 value, int nbits)
{
  const u32 w = bitPos >> 5, sh = bitPos & 31;
  const u32 lo = (u32)value, hi = (u32)(value >> 32);
  atomicOr(&words[w], lo << sh);
  const u32 mid = (sh ? (lo >> (32 - sh)) : 0u) | (hi << sh);
  if (sh + (u32)nbits > 32) atomicOr(&words[w + 1], mid);
  if (sh + (u32)nbits > 64) atomicOr(&words[w + 2], sh ? (hi >> (32 - sh)) : 0u);
}

// Fletcher terms of one little-endian 32-bit word whose first byte sits at an EVEN position `pos` of
// the checksummed range: two big-endian 16-bit words w0 = b0 b1, w1 = b2 b3 with indices pos/2, pos/2 + 1
__device__ __forceinline__ void fletcherWord(u32 x, u32 pos, u64& A, u64& B)
{
  const u32 w0 = ((x & 0xFFu) << 8) | ((x >> 8) & 0xFFu), w1 = ((x >> 8) & 0xFF00u) | (x >> 24);
  const u32 k = pos >> 1;
  A += w0 + w1;
  B += (u64)k * w0 + (u64)(k + 1) * w1;
}

template<class T, bool WIDE>
__global__ void __launch_bounds__(256)
k_fast_pack(const T* __restrict__ data, BandParams p, const FastBlockDesc* __restrict__ desc, const u32* __restrict__ wgBase,
            u8* __restrict__ out, u64* __restrict__ wgFletcher, const FastEncodeResult* __restrict__ res,
            const u8* __restrict__ prefixStage, const u64* __restrict__ tileOffset, FastBatch batch)
{
  {
    const size_t tile = blockIdx.y;
    data += tile * batch.tileElems; desc += tile * batch.nWG * kFastBlocksPerWG; wgBase += tile * fastWgStride(batch.nWG);
    wgFletcher += tile * batch.nWG * 2; res += tile; prefixStage += tile * kFastPrefixStage;
    if (tileOffset) out += tileOffset[tile];
  }
  typedef FastCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW, IT = C::IT;
  constexpr int kMaxBlockBytes = 1 + 64 * (int)sizeof(T);
  constexpr int kSpanWords = (kFastBlocksPerWG * kMaxBlockBytes + 16) / 4 + 8;
  __shared__ __align__(16) u32 s_out[kSpanWords];
  __shared__ u64 s_mn[kFastBlocksPerWG];
  __shared__ u32 s_w1[kFastBlocksPerWG];
  __shared__ u32 s_bit[kFastBlocksPerWG];    // bit position of each block inside s_out
  __shared__ u64 s_fa[4], s_fb[4];
  if (res->redo) return;
  // the bytes in front of the first block (header, mask count, ranges, mode byte) come from the decide step
  if (blockIdx.x == 0 && threadIdx.x < res->prefixLen) out[threadIdx.x] = prefixStage[threadIdx.x];

  PROBE_BEGIN;
  const int w = waveId(), lane = laneId();
  const int r = lane >> 3, c = lane & 7, b = c / LPR, h = c % LPR;
  const FastSpan span = fastSpanOf(blockIdx.x, (u32)p.nTH, (u32)p.nTV);
  const u32 g0 = res->prefixLen + wgBase[blockIdx.x];        // absolute offset of this workgroup's span
  const u32 spanLen = wgBase[blockIdx.x + 1] - wgBase[blockIdx.x];
  const u32 ldsShift = g0 & 15u;                            // LDS byte i <-> blob byte (g0 & ~15) + i

  // pixels first (long latency), descriptors by wave 0, zero the span image meanwhile
  T v[IT][V];
#pragma unroll
  for (int t = 0; t < IT; t++) loadLane<T, V>(data + laneOrigin<WIDE, BPW, V>(span, t * 4 + w, r, c, p.nCols), v[t]);
  FastBlockDesc d;
  const int wPlan = (int)((blockIdx.x * 2654435761u) >> 30);    // the wave that does the per-block work rotates (see k_fast_stats)
  if (w == wPlan) d = desc[(size_t)blockIdx.x * kFastBlocksPerWG + lane];
  for (int i = threadIdx.x; i < kSpanWords; i += 256) s_out[i] = 0;
  if (w == wPlan)
  {
    const u32 sz = d.w1 & 0xFFFFu;
    u32 inc = sz;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(inc, (unsigned)dd); if (lane >= dd) inc += o; }
    s_mn[lane] = d.mnBits; s_w1[lane] = d.w1;
    s_bit[lane] = 8u * (ldsShift + inc - sz);
  }
  __syncthreads();
  PROBE(4);

  // ---- block headers: lane = block (Lerc2::WriteTile, BitStuffer2 stream header)
  if (w == wPlan)
  {
    const u32 w1 = d.w1;
    const int kind = (int)((w1 >> 16) & 7u), tc = (int)((w1 >> 19) & 3u), dtRed = (int)((w1 >> 21) & 7u), nb = (int)(w1 >> 24);
    const int j0 = (int)fastSpanCol(span, (u32)lane) * 8;
    u32 flag = (u32)(((j0 >> 3) & 15) << 2) & 0x38u;    // version 6, no slice difference
    const u32 at0 = 8u * (ldsShift) + (s_bit[lane] - 8u * ldsShift);
    if (kind == 7) { }    // no such block (last workgroup of a raster whose block count is no multiple of 64)
    else if (kind == 0) orBits(s_out, at0, flag | 2u, 8);
    else if (kind == 1) orBits(s_out, at0, flag, 8);
    else
    {
      flag |= (kind == 2) ? 3u : 1u;
      flag |= (u32)tc << 6;
      const int offBytes = dtSize(dtRed);
      orBits(s_out, at0, flag, 8);
      orBits64(s_out, at0 + 8, typedBits((double)fromRawBits<T>(d.mnBits), dtRed), 8 * offBytes);
      if (kind == 3) orBits(s_out, at0 + 8u * (1u + (u32)offBytes), (u32)nb | (2u << 6) | (64u << 8), 16);    // numBits byte, count 64
    }
  }

  PROBE(5);
  // ---- payloads
#pragma unroll
  for (int t = 0; t < IT; t++)
  {
    const int tile = t * 4 + w;
    const int blk = tile * BPW + b;
    const u32 w1 = s_w1[blk];
    const int kind = (int)((w1 >> 16) & 7u);
    const u32 at0 = s_bit[blk];
    const int e0 = r * 8 + h * V;
    if (kind == 3)
    {
      const int nb = (int)(w1 >> 24);
      const int offBytes = dtSize((int)((w1 >> 21) & 7u));
      const T mn = fromRawBits<T>(s_mn[blk]);
      u32 q[V];
      quantizeLane<T, V>(p.intLossless, p.scale, v[t], mn, q);
      const u32 at = at0 + 8u * (3u + (u32)offBytes);
      if (V * nb <= 64)
      {
        u64 s = 0;
#pragma unroll
        for (int k = 0; k < V; k++) s |= (u64)q[k] << (k * nb);
        orBits64(s_out, at + (u32)e0 * (u32)nb, s, V * nb);
      }
      else
      {
#pragma unroll
        for (int k = 0; k < V; k += 2)
          orBits64(s_out, at + (u32)(e0 + k) * (u32)nb, (u64)q[k] | ((u64)q[k + 1] << nb), 2 * nb);
      }
    }
    else if (kind == 1)
    {
#pragma unroll
      for (int k = 0; k < V; k++)
        orBits64(s_out, at0 + 8u + (u32)(e0 + k) * 8u * (u32)sizeof(T), rawBits<T>(v[t][k]), 8 * (int)sizeof(T));
    }
    // LUT blocks (kind 4): all blocks of the wave take part in the group reductions
    if (__any(kind == 4))
    {
      const bool mine = (kind == 4);
      const bool leader = (r == 0 && h == 0);
      const T mn = fromRawBits<T>(s_mn[blk]);
      u32 q[V], idx[V];
      quantizeLane<T, V>(p.intLossless, p.scale, v[t], mn, q);
#pragma unroll
      for (int k = 0; k < V; k++) idx[k] = 0;
      const int nb = (int)(w1 >> 24);
      const int offBytes = dtSize((int)((w1 >> 21) & 7u));
      const u32 hdr = at0 + 8u * (1u + (u32)offBytes);
      const u32 lutAt = hdr + 24;    // numBits byte, count byte, nLut + 1 byte
      u32 count = 0, last = 0;
      bool active = mine;
      for (;;)
      {
        u32 m = 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < V; k++)
          if ((count == 0 || q[k] > last) && q[k] < m) m = q[k];
        m = groupMin<LPR>(m);
        if (m == 0xFFFFFFFFu) active = false;
        if (!__any(active)) break;
        if (active)
        {
#pragma unroll
          for (int k = 0; k < V; k++) if (q[k] == m) idx[k] = count;
          if (leader && count > 0) orBits(s_out, lutAt + (count - 1) * (u32)nb, m, nb);
          last = m; count++;
        }
      }
      if (mine)
      {
        const u32 nLut = count - 1;
        const int nbIdx = bitLen(nLut);
        if (leader) orBits(s_out, hdr, (u32)nb | (2u << 6) | 32u | (64u << 8) | ((nLut + 1) << 16), 24);
        const u32 idxAt = lutAt + 8u * ((nLut * (u32)nb + 7) >> 3);
        u64 s = 0;
#pragma unroll
        for (int k = 0; k < V; k++) s |= (u64)idx[k] << (k * nbIdx);    // nbIdx <= 6
        orBits64(s_out, idxAt + (u32)e0 * (u32)nbIdx, s, V * nbIdx);
      }
    }
  }
  __syncthreads();
  PROBE(6);

  // ---- flush: 16-byte chunks, byte granular at the two ends; Fletcher sums of the bytes we own.
  // Absolute blob offsets of chunk starts are multiples of 16, so positions inside blob[14 ..) are even.
  u64 A = 0, B = 0;
  const u32 gAligned = g0 & ~15u;
  const u32 nChunks = (ldsShift + spanLen + 15) >> 4;
  for (u32 ch = threadIdx.x; ch < nChunks; ch += 256)
  {
    const u32 lo = ch * 16, hi = lo + 16;                                  // LDS byte range of this chunk
    const u32 first = lo < ldsShift ? ldsShift : lo;
    const u32 last = hi > ldsShift + spanLen ? ldsShift + spanLen : hi;    // owned bytes: [first, last)
    uint4 x = *reinterpret_cast<const uint4*>(&s_out[ch * 4]);
    if (first == lo && last == hi) *reinterpret_cast<uint4*>(out + gAligned + lo) = x;
    else
    {
      // partial chunk: store byte-wise and blank the bytes we do not own before summing
      u32 wd[4] = { x.x, x.y, x.z, x.w };
      for (u32 i = lo; i < hi; i++)
      {
        const u32 byte = (wd[(i - lo) >> 2] >> (8 * ((i - lo) & 3))) & 0xFFu;
        if (i >= first && i < last) out[gAligned + i] = (u8)byte;
        else wd[(i - lo) >> 2] &= ~(0xFFu << (8 * ((i - lo) & 3)));
      }
      x = make_uint4(wd[0], wd[1], wd[2], wd[3]);
    }
    const u32 pos = gAligned + lo - 14 + 0;    // position of the chunk's first byte; may be "negative" only for lo < 14 of the first span
    // gAligned + lo >= 16 > 14 always holds here because spans start behind the >= 95-byte prefix
    fletcherWord(x.x, pos, A, B);
    fletcherWord(x.y, pos + 4, A, B);
    fletcherWord(x.z, pos + 8, A, B);
    fletcherWord(x.w, pos + 12, A, B);
  }
  A %= 65535u; B %= 65535u;
  A = waveSum(A); B = waveSum(B);
  if (lane == 0) { s_fa[w] = A; s_fb[w] = B; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    wgFletcher[2 * (size_t)blockIdx.x] = (s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3]) % 65535u;    // folded by k_fast_checksum
    wgFletcher[2 * (size_t)blockIdx.x + 1] = (s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]) % 65535u;
  }
  PROBE(7);
}

// checksum = Fletcher32 over blob[14 ..): the prefix bytes written by the decide step + the workgroups' partial sums
__global__ void __launch_bounds__(1024)
k_fast_checksum(const u64* __restrict__ wgFletcher, u32 nWG, u8* __restrict__ out, FastEncodeResult* res,
                const u8* __restrict__ prefixStage, const u64* __restrict__ tileOffset, FastBatch batch)
{
  __shared__ u64 s_a[16], s_b[16];
  {
    const size_t tile = blockIdx.y;
    wgFletcher += tile * batch.nWG * 2; res += tile; prefixStage += tile * kFastPrefixStage;
    if (tileOffset) out += tileOffset[tile];
  }
  if (res->redo) return;
  const int lane = laneId(), w = waveId();
  u64 A = 0, B = 0;
  if (threadIdx.x < 16) { s_a[threadIdx.x] = 0; s_b[threadIdx.x] = 0; }
  __syncthreads();
  for (u32 i = threadIdx.x; i < nWG; i += blockDim.x) { A += wgFletcher[2 * (size_t)i]; B += wgFletcher[2 * (size_t)i + 1]; }    // each < 65535
  for (u32 pos = threadIdx.x; pos + 14 < res->prefixLen; pos += blockDim.x)
  {
    const u32 cw = (u32)prefixStage[14 + pos] << ((pos & 1u) ? 0 : 8);
    A += cw; B += (u64)(pos >> 1) * cw;
  }
  A = waveSum(A % 65535u); B = waveSum(B % 65535u);
  if (lane == 0) { s_a[w] = A; s_b[w] = B; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  A = 0; B = 0;
  for (int i = 0; i < 16; i++) { A += s_a[i]; B += s_b[i]; }
  const u32 len = res->blobSize - 14;
  const u64 N = ((u64)len + 1) / 2;
  A %= 65535u; B %= 65535u;
  u64 s1 = A, s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
  if (s1 == 0) s1 = 0xffff;
  if (s2 == 0) s2 = 0xffff;
  const u32 cs = (u32)((s2 << 16) | s1);
  putBytes(out + 10, cs, 4);
  res->checksum = cs;
}

// Batches: where each tile's blob goes in the arena.  One workgroup of 1024 threads; a tile that the general path has
// to redo takes no room here (the host appends it behind the batch).  Starts are 16-byte aligned.
__global__ void __launch_bounds__(1024)
k_fast_tile_offsets(FastEncodeResult* __restrict__ res, u32 nTiles, u64 arenaBase, u64 arenaCapacity, u64* __restrict__ tileOffset)
{
  __shared__ u64 s_w[16];
  const u32 per = (nTiles + 1023u) / 1024u;
  const u32 begin = min(threadIdx.x * per, nTiles), end = min(begin + per, nTiles);
  u64 sum = 0;
  for (u32 t = begin; t < end; t++) sum += res[t].redo ? 0ull : (((u64)res[t].blobSize + 15ull) & ~15ull);
  u64 inc = sum;
  const int lane = laneId();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u64 o = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += o; }
  if (lane == 63) s_w[waveId()] = inc;
  __syncthreads();
  u64 run = arenaBase + inc - sum;
  for (int i = 0; i < waveId(); i++) run += s_w[i];
  for (u32 t = begin; t < end; t++)
  {
    tileOffset[t] = run;
    if (!res[t].redo)
    {
      const u64 sz = ((u64)res[t].blobSize + 15ull) & ~15ull;
      if (run + sz > arenaCapacity) { res[t].redo = 1u; res[t].redoReason = kRedoCapacity; }    // does not fit: nothing is written
      run += sz;
    }
  }
  if (threadIdx.x == 1023) tileOffset[nTiles] = run;
}

// Before the passes, for float types: look at the first raster row the way
// Lerc2::TryRaiseMaxZError does (Lerc2.cpp:1245-1290): per candidate factor the largest rounding error, one partial
// result per workgroup (k_fast_decide folds them).
template<class T>
__global__ void __launch_bounds__(256)
k_fast_prepare(const T* __restrict__ data, int nCols, u32 raiseCand, double* __restrict__ row0Partial, FastBatch batch)
{
  data += (size_t)blockIdx.y * batch.tileElems;
  row0Partial += (size_t)blockIdx.y * batch.nRaiseSets * 9;
  __shared__ u64 s_r[4][9];
  const int lane = laneId(), w = waveId();
  double rerr[9];
#pragma unroll
  for (int c = 0; c < 9; c++) rerr[c] = 0;
  const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
  for (int i = (int)(blockIdx.x * 256u + threadIdx.x); i < nCols; i += (int)(gridDim.x * 256u))
  {
    const double x = (double)data[i];
    if (x != x) continue;    // a NaN sends the band to the general path anyway
#pragma unroll
    for (int c = 0; c < 9; c++)    // candidates in increasing factor order, stop at the first exact hit
    {
      if (!((raiseCand >> c) & 1u)) continue;
      const double z = x * facCand[c];
      if (z == (double)(int)z) break;
      const double dlt = fabs(floor(z + 0.5) - z);
      rerr[c] = dlt > rerr[c] ? dlt : rerr[c];
    }
  }
#pragma unroll
  for (int c = 0; c < 9; c++)
  {
    u64 bits; const double e = rerr[c]; memcpy(&bits, &e, 8);    // non-negative doubles order like their bit patterns
    bits = waveMax(bits);
    if (lane == 0) s_r[w][c] = bits;
  }
  __syncthreads();
  if (threadIdx.x < 9)
  {
    u64 m = s_r[0][threadIdx.x];
    for (int i = 1; i < 4; i++) m = s_r[i][threadIdx.x] > m ? s_r[i][threadIdx.x] : m;
    double e; memcpy(&e, &m, 8);
    row0Partial[blockIdx.x * 9 + threadIdx.x] = e;
  }
}

// ------------------------------------------------------------------------------------------------
bool fastEncodeEligible(int dt, int nRows, int nCols, int nDepth, bool hasMask, double maxZErr)
{
  if (hasMask || nDepth != 1) return false;
  if (dt == DT_Char || dt == DT_Byte) return false;    // 8-bit: the Huffman decision needs the general path
  if (!fastDimsOk(dt, nRows, nCols)) return false;
  if (dt >= DT_Float && maxZErr == 0) return false;    // lossless float is out of scope altogether
  return true;
}

u32 fastEncodeNumWG(int nRows, int nCols) { return fastNumWG(nRows, nCols); }

template<class T>
static void launchFastEncodeT(int stage, const BandParams& p, double requested, u32 raiseCand, const void* data, u8* out, u64 cap,
                              u64 arenaBase, const FastEncodeBuffers& b, const FastBatch& batch, hipStream_t st)
{
  const u32 nWG = batch.nWG, nT = batch.nTiles;
  if (stage == -1)
  {
    if (b.row0RaiseErr)
      hipLaunchKernelGGL(k_fast_prepare<T>, dim3(batch.nRaiseSets, nT), dim3(256), 0, st, (const T*)data, p.nCols, raiseCand, b.row0RaiseErr, batch);
  }
  else if (stage == 0)
  {
    if (p.nTH % 64 == 0)
      hipLaunchKernelGGL((k_fast_stats<T, true>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, b.desc, b.wgSize, b.wgMinKey, b.wgMaxKey,
                         b.wgFlags, batch);
    else
      hipLaunchKernelGGL((k_fast_stats<T, false>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, b.desc, b.wgSize, b.wgMinKey, b.wgMaxKey,
                         b.wgFlags, batch);
  }
  else if (stage == 1)
  {
    // a tile of a batch may be as large as it likes here; whether the arena holds it is decided by the placement
    hipLaunchKernelGGL(k_fast_scan_decide, dim3(1, nT), dim3(1024), 0, st, p, requested, raiseCand, nWG, (const u32*)b.wgSize, b.wgBase,
                       (const u64*)b.wgMinKey, (const u64*)b.wgMaxKey, (const u32*)b.wgFlags, (const double*)b.row0RaiseErr, b.prefixStage,
                       b.tileOffset ? ~0ull : cap, b.result, batch);
    if (b.tileOffset)
      hipLaunchKernelGGL(k_fast_tile_offsets, dim3(1), dim3(1024), 0, st, b.result, nT, arenaBase, cap, b.tileOffset);
  }
  else if (stage == 2)
  {
    if (p.nTH % 64 == 0)
      hipLaunchKernelGGL((k_fast_pack<T, true>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, (const FastBlockDesc*)b.desc,
                         (const u32*)b.wgBase, out, b.wgFletcher, (const FastEncodeResult*)b.result, (const u8*)b.prefixStage,
                         (const u64*)b.tileOffset, batch);
    else
      hipLaunchKernelGGL((k_fast_pack<T, false>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, (const FastBlockDesc*)b.desc,
                         (const u32*)b.wgBase, out, b.wgFletcher, (const FastEncodeResult*)b.result, (const u8*)b.prefixStage,
                         (const u64*)b.tileOffset, batch);
  }
  else
    hipLaunchKernelGGL(k_fast_checksum, dim3(1, nT), dim3(nT > 1 ? 256 : 1024), 0, st, (const u64*)b.wgFletcher, nWG, out, b.result, (const u8*)b.prefixStage,
                       (const u64*)b.tileOffset, batch);
}

void launchFastEncode(int stage, const BandParams& assumed, double requestedMaxZErr, u32 raiseCandidates, const void* data, u8* out,
                      u64 outCapacity, u64 arenaBase, const FastEncodeBuffers& b, const FastBatch& batch, hipStream_t st)
{
  switch (assumed.dt)
  {
    case DT_Short:  launchFastEncodeT<short>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    case DT_UShort: launchFastEncodeT<unsigned short>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    case DT_Int:    launchFastEncodeT<int>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    case DT_UInt:   launchFastEncodeT<unsigned int>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    case DT_Float:  launchFastEncodeT<float>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    case DT_Double: launchFastEncodeT<double>(stage, assumed, requestedMaxZErr, raiseCandidates, data, out, outCapacity, arenaBase, b, batch, st); break;
    default: break;
  }
}

}    // namespace lerc
