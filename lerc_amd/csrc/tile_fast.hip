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
//   * the scan of the sizes and the decisions the reference takes between its sweeps are taken on the device: for one
//     raster by the first blocks of the pack launch (k_fast_pack<SOLO>, soloScanDecide), for tile batches by a launch of
//     their own (k_fast_scan_decide); an encode costs one host synchronisation, a queued one none
// Reference: Lerc2.cpp:1474-1668, :1717-1799, :1949-2021; Lerc2.h:337-453; BitStuffer2.cpp:35-153.
#include "tile_fast.h"
#include "kernels.h"
#include "wave_utils.h"
#include "block_plan.h"

#include <limits>

namespace lerc {

// (a workgroup of 256 threads is admitted eight times per CU only if its kernel uses at most 80 scalar registers)
#ifndef LERC_FUSED_SGPR80
#define LERC_FUSED_SGPR80 1
#endif
#if defined(HIPSIM) || !LERC_FUSED_SGPR80
#define LERC_SGPR_CAP
#else
#define LERC_SGPR_CAP __attribute__((amdgpu_num_sgpr(80)))
#endif
PROBE_DEFINE(fast_encode)
#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time line of the one-launch encoder (constant-rate counter), read by tools/trace_encode1.py
static __device__ unsigned long long g_trace[8 * 32768];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * (size_t)n); }
#define TRACE(slot) do { if (threadIdx.x == 0 && wg < 32768u) g_trace[8 * wg + (slot)] = wall_clock64(); } while (0)
#define TRACE_ID() do { if (threadIdx.x == 0 && wg < 32768u) { unsigned xcc, hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); g_trace[8 * wg + 7] = ((unsigned long long)xcc << 32) | hw; } } while (0)
#else
#define TRACE(slot)
#define TRACE_ID()
#endif

#ifndef LERC_ENC_WIDE
#define LERC_ENC_WIDE 0    // (tuning: two vectors a lane in the one-launch encoder -- 8 % fewer vector instructions, but the float kernel no longer fits 64 registers: a pixel vector goes to scratch right behind its load, 90 -> 108 us)
#endif
// tuning: LERC_ENC_EXIT=n builds an encoder that leaves at mark n (profiles/r06_notes.md: instruction counts per phase); results are invalid
#ifndef LERC_ENC_EXIT
#define LERC_ENC_EXIT 99
#endif
#define ENC_EXIT(n) do { if (LERC_ENC_EXIT == (n)) return; } while (0)

enum FastRedo : u32
{
  kRedoNaN = 1, kRedoAllInt = 2, kRedoRaise = 4, kRedoConst = 8, kRedoMb16 = 16, kRedoOneSweep = 32, kRedoCapacity = 64,
  kRedoArena = 128    // (with kRedoCapacity: the ARENA of a batch is full, not a tile's own room)
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
__device__ __forceinline__ void loadLane(const T* __restrict__ p, T (&v)[V], bool streaming = false)
{
  struct alignas(sizeof(T) * V) Vec { T e[V]; };
  const Vec x = streaming ? loadStreaming(reinterpret_cast<const Vec*>(p)) : *reinterpret_cast<const Vec*>(p);
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

// order-preserving 32-bit image of a value of a type of at most 32 bits (for wave reductions in one register)
template<class T> struct FOrd
{
  static __device__ __forceinline__ u32 enc(T v)
  {
    if (std::is_same<T, float>::value) { u32 b; memcpy(&b, &v, 4); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
    if (std::is_signed<T>::value) return (u32)((i64)v + 0x80000000ll);
    return (u32)v;
  }
  static __device__ __forceinline__ T dec(u32 k)
  {
    if (std::is_same<T, float>::value) { const u32 b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; T v; memcpy(&v, &b, 4); return v; }
    if (std::is_signed<T>::value) return (T)((i64)k - 0x80000000ll);
    return (T)k;
  }
};
template<> struct FOrd<double> { static __device__ __forceinline__ u32 enc(double) { return 0u; } static __device__ __forceinline__ double dec(u32) { return 0; } };

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
             u64* __restrict__ wgMinKey, u64* __restrict__ wgMaxKey, u32* __restrict__ wgFlags, u32* __restrict__ tickets, FastBatch batch)
{
  {
    const size_t tile = blockIdx.y;    // this tile's slice of every array
    data += tile * batch.tileElems; desc += tile * batch.nWG * kFastBlocksPerWG; wgSize += tile * fastWgStride(batch.nWG);
    wgMinKey += tile * batch.nWG; wgMaxKey += tile * batch.nWG; wgFlags += tile * batch.nWG;
    tickets += tile * fastTicketStride(batch.nWG);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) tickets[0] = 0u;    // the scan step counts its workgroups here
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
  // (the raster is read with the non-temporal hint here and in k_fast_pack: 268 MB do not fit the 256 MiB Infinity Cache,
  // so nothing of the first pass survives for the second one anyway, and ordinary loads only displace what does get re-used)
#pragma unroll
  for (int t = 0; t < C::IT; t++) loadLane<T, V>(data + laneOrigin<WIDE, BPW, V>(span, t * 4 + w, r, c, p.nCols), vAll[t], true);

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
    for (int k = 1; k < V; k++) { mn = OpMin()(mn, v[k]); mx = OpMax()(mx, v[k]); }
    if constexpr (std::is_same<T, float>::value) rowMinMax(mn, mx);    // (LB == 16)
    else
    {
      mn = (T)groupReduce<LB>((ST)mn, OpMin());
      mx = (T)groupReduce<LB>((ST)mx, OpMax());
    }
    // "same as previous" in row-major block order, prevVal starts at 0 (Lerc2.cpp:1729-1758): the previous
    // pixel vector of the block lives in the previous lane
    T prev = (T)dppMovT<kDppWaveShr1>((ST)v[V - 1]);
    if (leader) prev = T(0);
    int same = (v[0] == prev) ? 1 : 0;
#pragma unroll
    for (int k = 1; k < V; k++) same += (v[k] == v[k - 1]) ? 1 : 0;
    // only "more than half of the 64" matters (tryLut); that needs a lane above the average of 32 / LB.  LUT candidates
    // need the number of distinct quantised values, which only the pixel owners can count
    u32 nd = 0;
    if (__any(same > 32 / LB))
    {
      same = groupReduce<LB>(same, OpSum());
      const bool tryLut = (2 * same > 64) && ((double)mx > (double)mn + 3 * p.maxZErr);
      if (__any(tryLut))
      {
        const double mv = ((double)mx - (double)mn) * p.scale;
        const bool need = tryLut && !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
        u32 q[V];
        quantizeLane<T, V>(p.intLossless, p.scale, v, mn, q);
        nd = groupDistinct<LB, V>(q, need);
      }
    }
    else same = 0;
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
    // per-workgroup partial results, folded by k_fast_scan_decide resp. the first blocks of k_fast_pack<SOLO> (thousands of
    // workgroups hammering a few addresses with atomics cost more than the whole pass)
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

// lane 0 of a wave decides; `prefix` (LDS) receives the bytes in front of the first block
__device__ __forceinline__ u32
fastDecide(const BandParams& p, double requestedMaxZErr, u32 raiseCandidates, u32 nBytesTiling, u64 minKey, u64 maxKey, u32 flags,
           const u64* raiseBits /* [9] largest first-row rounding error per candidate, as bit patterns */, u32 nBlobsMore, u8* prefix,
           u64 outCapacity, FastEncodeResult* res, bool resetStuck = true)
{
  const u64 a = minKey, b = maxKey;
  const u32 f = flags;
  u64 rawMin = 0, rawMax = 0;
  const double zMin = keyToDouble(p.dt, a, rawMin), zMax = keyToDouble(p.dt, b, rawMax);
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
    if (raiseCandidates && raiseBits)
    {
      const int fac[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
      for (int c = 0; c < 9; c++)
      {
        double e; const u64 bits = raiseBits[c]; memcpy(&e, &bits, 8);
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
  // (written through: a workgroup that gives up much later says so the same way.  The one-launch encoder decides LAST, when
  // every workgroup has had its chance to give up: there the host clears the word before the launch)
  if (resetStuck) __hip_atomic_store(&res->stuck, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (redo) return redo;

  // header + "no mask" + ranges + "not one sweep" (Lerc2.cpp:396-430)
  u8* o = prefix;
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
  return 0u;
}

// The scan of the workgroup sizes, the fold of the per-workgroup statistics and the decisions in one launch: every
// workgroup (1024 threads) scans kFastScanGroup sizes and publishes what it found; the one that arrives last scans the
// groups' totals, decides, writes the bytes in front of the first block and resets the pack step's arrival counters.
template<class T>
__global__ void __launch_bounds__(1024)
k_fast_scan_decide(const T* __restrict__ data, BandParams p, double requestedMaxZErr, u32 raiseCandidates, u32 nWG, const u32* __restrict__ wgSize,
                   u32* __restrict__ wgBase, const u64* __restrict__ wgMinKey, const u64* __restrict__ wgMaxKey,
                   const u32* __restrict__ wgFlags, u8* __restrict__ prefixStage, u64 outCapacity,
                   FastEncodeResult* res, u32* __restrict__ groupBase, u64* __restrict__ scanPart, u64* __restrict__ packPart,
                   u32* __restrict__ tickets, FastBatch batch)
{
  __shared__ u64 s_min[16], s_max[16], s_raise[9], s_rw[16][9];
  __shared__ u32 s_fl[16], s_w[16];
  __shared__ u32 s_last;
  __shared__ __align__(16) u8 s_prefix[kFastPrefixStage];
  const u32 nGroups = fastScanGroups(nWG);
  {
    const size_t tile = blockIdx.y;
    wgSize += tile * fastWgStride(batch.nWG); wgBase += tile * fastWgStride(batch.nWG);
    wgMinKey += tile * batch.nWG; wgMaxKey += tile * batch.nWG; wgFlags += tile * batch.nWG;
    data += tile * batch.tileElems;
    prefixStage += tile * kFastPrefixStage; res += tile;
    groupBase += tile * (nGroups + 1); scanPart += tile * kScanPartWords * nGroups; tickets += tile * fastTicketStride(batch.nWG);
    packPart += tile * fastPackGroups(batch.nWG);
  }
  const int lane = laneId(), w = waveId();
  const int nWaves = (int)(blockDim.x >> 6);    // 16, or 4 where a raster has at most 1024 workgroups (mosaic tiles)
  const u32 nThreads = blockDim.x;
  const u32 g = blockIdx.x;
  // ---- float types: this group's share of the first raster row the way Lerc2::TryRaiseMaxZError looks at it
  // (Lerc2.cpp:1245-1290): per candidate factor the largest rounding error (the last workgroup folds the groups' results)
  const bool doRaise = DtOf<T>::v >= DT_Float && raiseCandidates != 0u;
  double rerr[9];
#pragma unroll
  for (int cnd = 0; cnd < 9; cnd++) rerr[cnd] = 0;
  if (doRaise)
  {
    const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
    const u32 per = ((u32)p.nCols + nGroups - 1u) / nGroups;
    for (u32 col = g * per + threadIdx.x; col < min((g + 1u) * per, (u32)p.nCols); col += nThreads)
    {
      const double x = (double)data[col];
      if (x != x) continue;    // a NaN sends the band to the general path anyway
      bool exact = false;
#pragma unroll
      for (int cnd = 0; cnd < 9; cnd++)    // candidates in increasing factor order, stop at the first exact hit
      {
        if (exact || !((raiseCandidates >> cnd) & 1u)) continue;
        const double z = x * facCand[cnd];
        if (z == (double)(int)z) { exact = true; continue; }
        const double dlt = fabs(floor(z + 0.5) - z);
        rerr[cnd] = dlt > rerr[cnd] ? dlt : rerr[cnd];
      }
    }
  }
  // ---- this group's kFastScanGroup entries: four per thread, all loads in flight first
  const u32 i0 = g * kFastScanGroup + 4u * threadIdx.x;
  const u32 lastVec = (nWG - 1u) & ~3u;                      // clamped loads (the arrays have slack up to a multiple of 4)
  const uint4 sz = *reinterpret_cast<const uint4*>(wgSize + min(i0, lastVec));
  u64 kMin = ~0ull, kMax = 0ull;
  u32 fl = 0;
  u32 e[4] = { sz.x, sz.y, sz.z, sz.w };
#pragma unroll
  for (u32 k = 0; k < 4; k++)
  {
    const u32 i = min(i0 + k, nWG - 1u);
    const u64 a = wgMinKey[i], b = wgMaxKey[i];
    const u32 f = wgFlags[i];
    const bool have = i0 + k < nWG && 4u * threadIdx.x + k < kFastScanGroup;
    kMin = (have && a < kMin) ? a : kMin; kMax = (have && b > kMax) ? b : kMax;
    fl |= have ? f : 0u;
    e[k] = have ? e[k] : 0u;
  }
  const u32 sum = e[0] + e[1] + e[2] + e[3];
  const u32 inc = waveInclusiveScan(sum);
  kMin = waveMin(kMin); kMax = waveMax(kMax);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) fl |= __shfl_xor(fl, m);
  if (lane == 63) s_w[w] = inc;
  if (lane == 0) { s_min[w] = kMin; s_max[w] = kMax; s_fl[w] = fl; }
  if (doRaise)
  {
#pragma unroll
    for (int cnd = 0; cnd < 9; cnd++)
    {
      u64 bits; const double e = rerr[cnd]; memcpy(&bits, &e, 8);    // non-negative doubles order like their bit patterns
      bits = waveMax(bits);
      if (lane == 0) s_rw[w][cnd] = bits;
    }
  }
  __syncthreads();
  u32 run = inc - sum;
  for (int i = 0; i < w; i++) run += s_w[i];
  {
    uint4 o;
    o.x = run; o.y = run + e[0]; o.z = o.y + e[1]; o.w = o.z + e[2];
    if (4u * threadIdx.x >= kFastScanGroup) { }
    else if (i0 + 3 < nWG) *reinterpret_cast<uint4*>(wgBase + i0) = o;
    else if (i0 < nWG) { wgBase[i0] = o.x; if (i0 + 1 < nWG) wgBase[i0 + 1] = o.y; if (i0 + 2 < nWG) wgBase[i0 + 2] = o.z; }
  }
  if (threadIdx.x == 0)
  {
    u32 total = 0, f = 0;
    u64 a = ~0ull, b = 0ull;
    for (int i = 0; i < nWaves; i++) { total += s_w[i]; f |= s_fl[i]; a = s_min[i] < a ? s_min[i] : a; b = s_max[i] > b ? s_max[i] : b; }
    publish64(scanPart + kScanPartWords * (size_t)g, (u64)total | ((u64)f << 32));
    publish64(scanPart + kScanPartWords * (size_t)g + 1, a);
    publish64(scanPart + kScanPartWords * (size_t)g + 2, b);
    for (int cnd = 0; cnd < 9; cnd++)
    {
      u64 m = 0;    // (+0.0)
      if (doRaise) for (int i = 0; i < nWaves; i++) m = s_rw[i][cnd] > m ? s_rw[i][cnd] : m;
      publish64(scanPart + kScanPartWords * (size_t)g + 3 + cnd, m);
    }
    drainVmem();
    s_last = (nGroups == 1u || lastArrival(&tickets[0], nGroups)) ? 1u : 0u;
  }
  __syncthreads();
  if (!s_last) return;

  // ---- the last workgroup: the groups' totals, the decisions, the header
  const u32 nPackGroups = fastPackGroups(nWG);
  for (u32 i = threadIdx.x; i < nPackGroups; i += nThreads) packPart[i] = 0ull;    // the pack step adds up its checksum terms here
  if (threadIdx.x < 9)
  {
    u64 m = 0;    // (+0.0)
    for (u32 i = 0; i < nGroups; i++) { const u64 bits = observe64(scanPart + kScanPartWords * (size_t)i + 3 + threadIdx.x); m = bits > m ? bits : m; }
    s_raise[threadIdx.x] = m;
  }
  u64 aMin = ~0ull, aMax = 0ull;
  u32 aFl = 0, carry = 0;
  for (u32 g0 = 0; g0 < nGroups; g0 += nThreads)    // (one round unless the raster has more than 2^28 blocks)
  {
    const u32 gi = g0 + threadIdx.x;
    const bool have = gi < nGroups;
    const u64 tf = have ? observe64(scanPart + kScanPartWords * (size_t)gi) : 0ull;
    const u64 a = have ? observe64(scanPart + kScanPartWords * (size_t)gi + 1) : ~0ull, b = have ? observe64(scanPart + kScanPartWords * (size_t)gi + 2) : 0ull;
    const u32 tot = (u32)tf;
    aFl |= (u32)(tf >> 32); aMin = a < aMin ? a : aMin; aMax = b > aMax ? b : aMax;
    __syncthreads();
    const u32 inc2 = waveInclusiveScan(tot);
    if (lane == 63) s_w[w] = inc2;
    __syncthreads();
    u32 before = carry + inc2 - tot;
    for (int i = 0; i < w; i++) before += s_w[i];
    if (have) groupBase[gi] = before;
    u32 all = 0;
    for (int i = 0; i < nWaves; i++) all += s_w[i];
    carry += all;
  }
  aMin = waveMin(aMin); aMax = waveMax(aMax);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) aFl |= __shfl_xor(aFl, m);
  __syncthreads();
  if (lane == 0) { s_min[w] = aMin; s_max[w] = aMax; s_fl[w] = aFl; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    groupBase[nGroups] = carry;
    u32 f = 0;
    u64 a = ~0ull, b = 0ull;
    for (int i = 0; i < nWaves; i++) { f |= s_fl[i]; a = s_min[i] < a ? s_min[i] : a; b = s_max[i] > b ? s_max[i] : b; }
    fastDecide(p, requestedMaxZErr, raiseCandidates, carry, a, b, f, doRaise ? s_raise : nullptr, batch.nBlobsMore, s_prefix, outCapacity, res);
  }
  __syncthreads();
  if (threadIdx.x < kFastPrefixStage / 4) reinterpret_cast<u32*>(prefixStage)[threadIdx.x] = reinterpret_cast<const u32*>(s_prefix)[threadIdx.x];
}

// ------------------------------------------------------------------------------------------------
// pass 2: pack + checksum
// ------------------------------------------------------------------------------------------------
// OR up to 64 bits into the LDS bit stream
// (three words whatever the length: an OR of nothing costs less than the branches around it; the image has slack behind it)
__device__ __forceinline__ void orBits64(u32* words, u32 bitPos, u64 value, int nbits)
{
  (void)nbits;
  const u32 w = bitPos >> 5, sh = bitPos & 31;
  const u64 lo = value << sh;
  const u32 hi = (u32)((value >> 1) >> (63u - sh));    // bits 64 .. 95 of the shifted value (0 for sh == 0)
  atomicOr(&words[w], (u32)lo);
  atomicOr(&words[w + 1], (u32)(lo >> 32));
  atomicOr(&words[w + 2], hi);
}
// bytes of a data type whose code is known to be 0 .. 7 (dtSize without the comparisons)
__device__ __forceinline__ int dtSize3(u32 dt) { return (int)((0x84442211u >> (4u * dt)) & 15u); }

// Fletcher terms of one little-endian 32-bit word whose first byte sits at an EVEN position `pos` of
// the checksummed range: two big-endian 16-bit words w0 = b0 b1, w1 = b2 b3 with indices pos/2, pos/2 + 1
__device__ __forceinline__ void fletcherWord(u32 x, u32 pos, u64& A, u64& B)
{
  const u32 w0 = ((x & 0xFFu) << 8) | ((x >> 8) & 0xFFu), w1 = ((x >> 8) & 0xFF00u) | (x >> 24);
  const u32 k = pos >> 1;
  A += w0 + w1;
  B += (u64)k * w0 + (u64)(k + 1) * w1;
}

// One raster (k_fast_pack<SOLO>): the first blocks of the launch are no pack workgroups.  Block s places the workgroups
// s * kSoloSlice ... of the stream -- the sum of all sizes in front of its slice, then a scan of the slice; no block needs
// another -- and publishes a cell per workgroup, epoch (32) | first byte behind the header (32), with write-through
// stores; a pack workgroup polls its own cell only (a word that every workgroup polls was tried: memory serves one
// address at about 90 reads per microsecond), and only the workgroups of the first round ever wait, with their pixels
// loaded and packed by then.  Block 0 also takes the decisions the reference takes between its sweeps, out of what the
// statistics step left: the largest first-row rounding errors (Lerc2::TryRaiseMaxZError), the fold of the per-workgroup
// statistics, fastDecide.  The bytes in front of the first block and the verdict go to `prefixStage`, write-through too;
// the last pack workgroup picks them up for the checksum.
template<class T>
__device__ __forceinline__ void
soloScanDecide(u32 s, const T* __restrict__ data, const BandParams& p, double requestedMaxZErr, u32 raiseCandidates, u32 nWG, u32 nBlobsMore,
               const u32* __restrict__ wgSize, const u64* __restrict__ wgMinKey, const u64* __restrict__ wgMaxKey, const u32* __restrict__ wgFlags,
               u64* __restrict__ cells, u32 epoch, u8* __restrict__ prefixStage, u64 outCapacity, FastEncodeResult* __restrict__ res,
               u64* __restrict__ arrived, bool pack)
{
  __shared__ u64 s_min[4], s_max[4], s_rw[4][9], s_raise[9];
  __shared__ u32 s_fl[4], s_sum[4], s_w[4], s_redo;
  __shared__ __align__(16) u8 s_prefix[kFastPrefixStage];
  const int w = waveId(), lane = laneId();
  const u32 lastVec = (nWG - 1u) & ~3u;                      // clamped loads (the array has slack up to a multiple of 4)
  // ---- the sizes in front of the slice (block 0: all of them, that is the total)
  const u32 sliceBegin = s * kSoloSlice, sumEnd = s ? sliceBegin : nWG;
  u32 sum = 0;
#pragma unroll 4
  for (u32 i = 4u * threadIdx.x; i < sumEnd; i += 1024u)
  {
    const uint4 x = *reinterpret_cast<const uint4*>(wgSize + min(i, lastVec));
    sum += x.x + (i + 1u < sumEnd ? x.y : 0u) + (i + 2u < sumEnd ? x.z : 0u) + (i + 3u < sumEnd ? x.w : 0u);
  }
  sum = waveSum(sum);
  if (lane == 0) s_sum[w] = sum;
  // ---- the slice: kPer sizes per thread (16; emulator builds: 1, most threads idle)
  constexpr u32 kPer = (kSoloSlice + 255u) / 256u;
  const u32 sliceEnd = min(sliceBegin + kSoloSlice, nWG);
  u32 e[kPer], mine = 0;
  const u32 i0 = sliceBegin + kPer * threadIdx.x;
  if constexpr (kPer % 4u == 0u)
  {
#pragma unroll
    for (u32 q = 0; q < kPer / 4u; q++)
    {
      const uint4 x = *reinterpret_cast<const uint4*>(wgSize + min(i0 + 4u * q, lastVec));
      e[4 * q] = x.x; e[4 * q + 1] = x.y; e[4 * q + 2] = x.z; e[4 * q + 3] = x.w;
    }
  }
  else
  {
#pragma unroll
    for (u32 k = 0; k < kPer; k++) e[k] = wgSize[min(i0 + k, nWG - 1u)];
  }
#pragma unroll
  for (u32 k = 0; k < kPer; k++) { e[k] = (i0 + k < sliceEnd) ? e[k] : 0u; mine += e[k]; }
  const u32 inc = waveInclusiveScan(mine);
  if (lane == 63) s_w[w] = inc;
  __syncthreads();
  const u32 total = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];    // in front of the slice; block 0: of the whole stream
  u32 run = (s ? total : 0u) + inc - mine;
  for (int i = 0; i < w; i++) run += s_w[i];
#pragma unroll
  for (u32 k = 0; k < kPer; k++)
  {
    if (i0 + k < sliceEnd) publish64(cells + i0 + k, ((u64)epoch << 32) | run);
    run += e[k];
  }
  if (s != 0u) return;

  // ---- block 0: the decisions
  const bool doRaise = DtOf<T>::v >= DT_Float && raiseCandidates != 0u;
  if (doRaise)
  {
    const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
    double rerr[9];
#pragma unroll
    for (int cnd = 0; cnd < 9; cnd++) rerr[cnd] = 0;
#pragma unroll 2
    for (u32 col = threadIdx.x; col < (u32)p.nCols; col += 256u)
    {
      const double x = (double)data[col];
      if (x != x) continue;    // a NaN sends the band to the general path anyway
      bool exact = false;
#pragma unroll
      for (int cnd = 0; cnd < 9; cnd++)    // candidates in increasing factor order, stop at the first exact hit
      {
        if (exact || !((raiseCandidates >> cnd) & 1u)) continue;
        const double z = x * facCand[cnd];
        if (z == (double)(int)z) { exact = true; continue; }
        const double dlt = fabs(floor(z + 0.5) - z);
        rerr[cnd] = dlt > rerr[cnd] ? dlt : rerr[cnd];
      }
    }
#pragma unroll
    for (int cnd = 0; cnd < 9; cnd++)
    {
      u64 bits; const double er = rerr[cnd]; memcpy(&bits, &er, 8);    // non-negative doubles order like their bit patterns
      bits = waveMax(bits);
      if (lane == 0) s_rw[w][cnd] = bits;
    }
  }
  u64 kMin = ~0ull, kMax = 0ull;
  u32 fl = 0;
#pragma unroll 4
  for (u32 i = threadIdx.x; i < nWG; i += 256u)
  {
    const u64 x = wgMinKey[i], y = wgMaxKey[i];
    kMin = x < kMin ? x : kMin; kMax = y > kMax ? y : kMax; fl |= wgFlags[i];
  }
  kMin = waveMin(kMin); kMax = waveMax(kMax);
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) fl |= __shfl_xor(fl, m);
  if (lane == 0) { s_min[w] = kMin; s_max[w] = kMax; s_fl[w] = fl; }
  __syncthreads();
  if (threadIdx.x < 9)
  {
    u64 m = 0;    // (+0.0)
    if (doRaise) for (int i = 0; i < 4; i++) m = s_rw[i][threadIdx.x] > m ? s_rw[i][threadIdx.x] : m;
    s_raise[threadIdx.x] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    u64 a = ~0ull, b = 0ull;
    u32 f = 0;
    for (int i = 0; i < 4; i++) { a = s_min[i] < a ? s_min[i] : a; b = s_max[i] > b ? s_max[i] : b; f |= s_fl[i]; }
    s_redo = fastDecide(p, requestedMaxZErr, raiseCandidates, total, a, b, f, doRaise ? s_raise : nullptr, nBlobsMore, s_prefix, outCapacity, res);
  }
  __syncthreads();
  if (!pack) return;    // (a size query)
  if (threadIdx.x < kFastPrefixStage / 8 - 1) publish64(reinterpret_cast<u64*>(prefixStage) + threadIdx.x, reinterpret_cast<const u64*>(s_prefix)[threadIdx.x]);
  if (threadIdx.x == kFastPrefixStage / 8 - 1) publish64(reinterpret_cast<u64*>(prefixStage) + threadIdx.x, (u64)s_redo);    // (the prefix is at most 111 bytes long)
  drainVmem();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(arrived, 1ull << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SOLO (one raster, no batch; right behind k_fast_stats, no scan step in between): the first blocks scan and decide
// (soloScanDecide), the others pack.  A pack workgroup assembles its span in LDS at byte 16 -- where it goes in the blob
// is only needed when the bytes leave -- and then reads its cell; the last one, once everybody has arrived, adds the
// header to the checksum, writes both and leaves the arrival counters zero for the next call.  If a decision says "not
// this path" (NaN, all-integer floats, ...) the bytes written are garbage and the host redoes the band on the general
// path.  out == nullptr: a size query, launched with block 0 alone.
template<class T, bool WIDE, bool SOLO>
__global__ void __launch_bounds__(256)
k_fast_pack(const T* __restrict__ data, BandParams p, const FastBlockDesc* __restrict__ desc, const u32* __restrict__ wgSize,
            const u32* __restrict__ wgBase, const u32* __restrict__ groupBase, u8* __restrict__ out,
            u64* __restrict__ packPart, FastEncodeResult* __restrict__ res,
            u8* __restrict__ prefixStage, const u64* __restrict__ tileOffset, FastBatch batch,
            FastSolo solo, const u64* __restrict__ wgMinKey, const u64* __restrict__ wgMaxKey, const u32* __restrict__ wgFlags,
            double requestedMaxZErr, u32 raiseCandidates, u64 outCapacity)
{
  if (!SOLO)
  {
    const size_t tile = blockIdx.y;
    data += tile * batch.tileElems; desc += tile * batch.nWG * kFastBlocksPerWG; wgBase += tile * fastWgStride(batch.nWG);
    wgSize += tile * fastWgStride(batch.nWG); groupBase += tile * (fastScanGroups(batch.nWG) + 1);
    packPart += tile * fastPackGroups(batch.nWG);
    res += tile; prefixStage += tile * kFastPrefixStage;
    if (tileOffset) out += tileOffset[tile];
  }
  typedef FastCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR, BPW = C::BPW, IT = C::IT;
  constexpr int kMaxBlockBytes = 1 + 64 * (int)sizeof(T);
  constexpr u32 kLead = 16;                  // SOLO: zero bytes in front of the image (the flush reads up to 15 bytes before it)
  constexpr int kSpanWords = (kFastBlocksPerWG * kMaxBlockBytes + 16) / 4 + 8 + (SOLO ? (int)kLead / 4 : 0);
  __shared__ __align__(16) u32 s_out[kSpanWords];
  __shared__ u64 s_mn[kFastBlocksPerWG];
  __shared__ u32 s_w1[kFastBlocksPerWG];
  __shared__ u32 s_bit[kFastBlocksPerWG];    // bit position of each block inside s_out
  __shared__ u64 s_fa[4], s_fb[4];
  __shared__ u32 s_spanBase;
  PROBE_BEGIN;
  const int w = waveId(), lane = laneId();
  const int r = lane >> 3, c = lane & 7, b = c / LPR, h = c % LPR;
  const u32 nScanBlocks = SOLO ? (batch.nWG + kSoloSlice - 1u) / kSoloSlice : 0u;
  if (SOLO && blockIdx.x < nScanBlocks)
  {
    soloScanDecide<T>(blockIdx.x, data, p, requestedMaxZErr, raiseCandidates, batch.nWG, batch.nBlobsMore, wgSize, wgMinKey, wgMaxKey, wgFlags,
                      solo.cells, solo.epoch, prefixStage, outCapacity, res, packPart + fastPackGroups(batch.nWG), out != nullptr);
    return;
  }
  // (tried: back to front, so that the workgroups dispatched first meet the end of the raster, which the statistics kernel
  // read last and the Infinity Cache might still hold -- no difference, 81.7 against 81.4 us)
  const u32 wg = blockIdx.x - nScanBlocks;
  const FastSpan span = fastSpanOf(wg, (u32)p.nTH, (u32)p.nTV);

  // pixels first (long latency; nothing they need comes out of memory), descriptors by one wave, then -- in one go, not
  // one round trip per question -- what the decide step left: is this band ours at all, where does the span go
  T v[IT][V];
#pragma unroll
  for (int t = 0; t < IT; t++) loadLane<T, V>(data + laneOrigin<WIDE, BPW, V>(span, t * 4 + w, r, c, p.nCols), v[t], true);
  FastBlockDesc d;
  const int wPlan = (int)((wg * 2654435761u) >> 30);    // the wave that does the per-block work rotates (see k_fast_stats)
  if (w == wPlan) d = desc[(size_t)wg * kFastBlocksPerWG + lane];
  u32 prefixLen, spanBase = 0;    // (SOLO: known when the span leaves)
  const u32 spanLen = wgSize[wg];
  // SOLO: where the span goes is in this workgroup's cell -- asked for now, looked at when the span is packed (it has been
  // there long before, except for the workgroups of the launch's first round)
  u64 cell = 0;
  if (SOLO && threadIdx.x == 0) cell = observe64(solo.cells + wg);
  if (SOLO) prefixLen = 90u + 4u + 2u * (u32)sizeof(T) + 1u;    // header, mask byte count, ranges, "not one sweep" (fastDecide)
  else
  {
    const u32 redo = res->redo;
    prefixLen = res->prefixLen;
    spanBase = groupBase[wg / kFastScanGroup] + wgBase[wg];
    if (redo) return;
    // the bytes in front of the first block (header, mask count, ranges, mode byte) come from the decide step
    // (without the checksum, bytes 10 .. 13: the workgroup that arrives last writes it, possibly through another XCD's L2, and
    // a second dirty copy of those bytes here could reach memory after it)
    if (wg == 0 && threadIdx.x < prefixLen && (threadIdx.x < 10 || threadIdx.x >= 14)) out[threadIdx.x] = prefixStage[threadIdx.x];
  }
  // (only what the span can touch: 16-byte stores up to a few words behind its end -- the ORs of three words reach that far)
  {
    const u32 nZero = min((u32)kSpanWords / 4u, ((SOLO ? kLead : 16u) + spanLen + 48u) / 16u + 1u);
    for (u32 i = threadIdx.x; i < nZero; i += 256u) reinterpret_cast<uint4*>(s_out)[i] = make_uint4(0, 0, 0, 0);
  }
  if (w == wPlan)
  {
    const u32 sz = d.w1 & 0xFFFFu;
    u32 inc = sz;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const u32 o = __shfl_up(inc, (unsigned)dd); if (lane >= dd) inc += o; }
    s_mn[lane] = d.mnBits; s_w1[lane] = d.w1;
    s_bit[lane] = 8u * ((SOLO ? kLead : ((prefixLen + spanBase) & 15u)) + inc - sz);
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
    const u32 at0 = s_bit[lane];
    if (kind == 7) { }    // no such block (last workgroup of a raster whose block count is no multiple of 64)
    else if (kind == 0) orBits(s_out, at0, flag | 2u, 8);
    else if (kind == 1) orBits(s_out, at0, flag, 8);
    else
    {
      flag |= (kind == 2) ? 3u : 1u;
      flag |= (u32)tc << 6;
      const int offBytes = dtSize3((u32)dtRed);
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
      const int offBytes = dtSize3((w1 >> 21) & 7u);
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
      const int offBytes = dtSize3((w1 >> 21) & 7u);
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
  if (SOLO && threadIdx.x == 0)
  {
    for (u32 spin = 0; (u32)(cell >> 32) != solo.epoch && spin < (1u << 22); spin++)    // (never that long: the scan blocks were dispatched before this one)
    {
      __builtin_amdgcn_s_sleep(8);
      cell = observe64(solo.cells + wg);
    }
    if ((u32)(cell >> 32) != solo.epoch) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_spanBase = (u32)cell;
  }
  __syncthreads();
  PROBE(6);
  if (SOLO) spanBase = s_spanBase;
  const u32 g0 = prefixLen + spanBase;                      // absolute offset of this workgroup's span
  const bool fits = !SOLO || (u64)g0 + spanLen <= outCapacity;    // (SOLO: the last workgroup reports what does not fit)
  const u32 ldsShift = g0 & 15u;                            // non-SOLO: LDS byte i <-> blob byte (g0 & ~15) + i

  // ---- Fletcher sums of the bytes this workgroup owns (16-byte units of the span image; absolute blob offsets of unit
  // starts are multiples of 16, so positions inside blob[14 ..) are even), before the bytes themselves leave: the sums
  // are handed on below while the stores are still under way
  const u32 gAligned = g0 & ~15u;
  const u32 nChunks = (ldsShift + spanLen + 15) >> 4;
  if constexpr (SOLO)
  {
    // Fletcher sums and flush in one go, in 16-byte units of the BLOB: unit u holds the image bytes from 16 u - (g0 & 15) on,
    // fetched as five words and funnel shifted; what lies outside the span is zero in LDS, adds nothing to the sums and
    // is not stored
    u32 A = 0;
    u64 B = 0;
    for (u32 u = threadIdx.x; u < nChunks; u += 256)
    {
      const u32 at = kLead + 16u * u - ldsShift;            // LDS byte of the unit's first byte (>= 1)
      const u32 wd0 = at >> 2, sh = 8u * (at & 3u);
      const u32 x0 = s_out[wd0], x1 = s_out[wd0 + 1], x2 = s_out[wd0 + 2], x3 = s_out[wd0 + 3], x4 = s_out[wd0 + 4];
      uint4 x;
      x.x = __builtin_amdgcn_alignbit(x1, x0, sh); x.y = __builtin_amdgcn_alignbit(x2, x1, sh);
      x.z = __builtin_amdgcn_alignbit(x3, x2, sh); x.w = __builtin_amdgcn_alignbit(x4, x3, sh);
      // (gAligned + 16 u >= 16 > 14 always holds here because spans start behind the >= 95-byte prefix)
      fletcherUnit(x, (u64)((gAligned + 16u * u - 14u) >> 1), A, B);
      if (!fits) continue;
      const u32 lo = 16u * u, first = lo < ldsShift ? ldsShift : lo, last = min(lo + 16u, ldsShift + spanLen);    // owned bytes of the unit
      if (first == lo && last == lo + 16u) *reinterpret_cast<uint4*>(out + gAligned + lo) = x;    // (k_fast_discover reads them next: no streaming hint)
      else
      {
        const u32 xs[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (u32 i = 0; i < 16; i++)
          if (lo + i >= first && lo + i < last) out[gAligned + lo + i] = (u8)(xs[i >> 2] >> (8 * (i & 3)));
      }
    }
    // (no reduction mod 65535 before the sums: a lane holds at most 5 units, A < 2^22 and B < 2^53 per lane)
    const u64 a = waveSum(A), b2 = waveSum(B);
    if (lane == 0) { s_fa[w] = a; s_fb[w] = b2; }
  }
  else
  {
    u32 A = 0;
    u64 B = 0;
    for (u32 ch = threadIdx.x; ch < nChunks; ch += 256)
    {
      const u32 lo = ch * 16, hi = lo + 16;                                  // LDS byte range of this unit
      uint4 x = *reinterpret_cast<const uint4*>(&s_out[ch * 4]);
      if (lo < ldsShift || hi > ldsShift + spanLen)                          // blank the bytes we do not own
      {
        u32 wd[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
        for (u32 i = 0; i < 16; i++)
          if (lo + i < ldsShift || lo + i >= ldsShift + spanLen) wd[i >> 2] &= ~(0xFFu << (8 * (i & 3)));
        x = make_uint4(wd[0], wd[1], wd[2], wd[3]);
      }
      // (gAligned + lo >= 16 > 14 always holds here because spans start behind the >= 95-byte prefix)
      fletcherUnit(x, (u64)((gAligned + lo - 14u) >> 1), A, B);
    }
    // (no reduction mod 65535 before the sums: a lane holds at most 5 units, A < 2^22 and B < 2^53 per lane)
    const u64 a = waveSum(A), b2 = waveSum(B);
    if (lane == 0) { s_fa[w] = a; s_fb[w] = b2; }
  }
  __syncthreads();
  // ---- checksum = Fletcher32 over blob[14 ..) (Lerc2.cpp:1037-1064).  Every workgroup adds its two sums (each < 65535)
  // and a 1 to its group's accumulator with ONE 64-bit atomic that it does not wait for: A in bits 0-23, B in bits 24-47,
  // arrivals in bits 48-63.  Only the workgroup with the highest index waits: it is dispatched last, so all the others
  // are resident or done, and it polls the accumulators until every group is complete, folds them and patches the header.
  const u32 nWG = batch.nWG, nGroups = fastPackGroups(nWG);
  if (threadIdx.x == 0)
  {
    const u64 a = (s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3]) % 65535u, b2 = (s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]) % 65535u;
    __hip_atomic_fetch_add(packPart + wg / kFastPackGroup, a | (b2 << 24) | (1ull << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if constexpr (!SOLO)
  {
    // ---- flush: 16-byte units, byte granular at the two ends
    for (u32 ch = threadIdx.x; ch < nChunks; ch += 256)
    {
      const u32 lo = ch * 16, hi = lo + 16;
      const u32 first = lo < ldsShift ? ldsShift : lo;
      const u32 last = hi > ldsShift + spanLen ? ldsShift + spanLen : hi;    // owned bytes: [first, last)
      const uint4 x = *reinterpret_cast<const uint4*>(&s_out[ch * 4]);
      if (first == lo && last == hi) *reinterpret_cast<uint4*>(out + gAligned + lo) = x;    // (k_fast_discover reads them next: no streaming hint)
      else
      {
        const u32 wd[4] = { x.x, x.y, x.z, x.w };
  #pragma unroll
        for (u32 i = 0; i < 16; i++)
          if (lo + i >= first && lo + i < last) out[gAligned + lo + i] = (u8)(wd[i >> 2] >> (8 * (i & 3)));
      }
    }
  }
  PROBE(7);
  if (wg != nWG - 1u) return;
  if (!SOLO)
  {
    if (w != 0) return;
    u64 fA = 0, fB = 0;
    for (u32 g0i = 0; g0i < nGroups; g0i += 64u)
    {
      const u32 gi = g0i + (u32)lane;
      const u32 want = gi < nGroups ? min(kFastPackGroup, nWG - gi * kFastPackGroup) : 0u;
      u64 acc = 0;
      for (u32 spin = 0; ; spin++)
      {
        acc = gi < nGroups ? observe64(packPart + gi) : 0ull;
        if (!__any((u32)(acc >> 48) != want)) break;
        if (spin > (1u << 22)) { if (lane == 0) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }    // (never: every other workgroup was dispatched before this one)
        __builtin_amdgcn_s_sleep(8);
      }
      fA += acc & 0xFFFFFFull; fB += (acc >> 24) & 0xFFFFFFull;
    }
    for (u32 pos = (u32)lane; pos + 14 < res->prefixLen; pos += 64u)    // + the bytes in front of the first block
    {
      const u32 cw = (u32)prefixStage[14 + pos] << ((pos & 1u) ? 0 : 8);
      fA += cw; fB += (u64)(pos >> 1) * cw;
    }
    fA = waveSum(fA % 65535u) % 65535u; fB = waveSum(fB % 65535u) % 65535u;
    if (lane != 0) return;
    const u32 len = res->blobSize - 14;
    const u64 N = ((u64)len + 1) / 2;
    u64 s1 = fA, s2 = ((N % 65535u) * fA + 65535u - fB) % 65535u;
    if (s1 == 0) s1 = 0xffff;
    if (s2 == 0) s2 = 0xffff;
    const u32 cs = (u32)((s2 << 16) | s1);
    putBytes(out + 10, cs, 4);
    res->checksum = cs;
    return;
  }

  // ---- SOLO, the last workgroup: wait for everybody (block 0 included), header + checksum
  __shared__ u32 s_redo, s_cs;
  __shared__ __align__(16) u64 s_prefix[kFastPrefixStage / 8];
  const u32 nBytesTiling = spanBase + spanLen;
  if (w == 0)
  {
    u64 fA = 0, fB = 0;
    for (u32 g0i = 0; g0i <= nGroups; g0i += 64u)
    {
      const u32 gi = g0i + (u32)lane;
      const u32 want = gi < nGroups ? min(kFastPackGroup, nWG - gi * kFastPackGroup) : gi == nGroups ? 1u : 0u;    // (+ block 0)
      u64 acc = 0;
      for (u32 spin = 0; ; spin++)
      {
        acc = gi <= nGroups ? observe64(packPart + gi) : 0ull;
        if (!__any((u32)(acc >> 48) != want)) break;
        if (spin > (1u << 22)) { if (lane == 0) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }    // (never: every other workgroup was dispatched before this one)
        __builtin_amdgcn_s_sleep(8);
      }
      fA += acc & 0xFFFFFFull; fB += (acc >> 24) & 0xFFFFFFull;
      if (gi <= nGroups) publish64(packPart + gi, 0ull);    // clean for the next call
    }
    if (lane < kFastPrefixStage / 8) s_prefix[lane] = observe64(reinterpret_cast<const u64*>(prefixStage) + lane);
    __builtin_amdgcn_wave_barrier();
    const u8* pre = reinterpret_cast<const u8*>(s_prefix);
    for (u32 pos = (u32)lane; pos + 14 < prefixLen; pos += 64u)    // + the Fletcher terms of the bytes in front of the first block
    {
      const u32 cw = (u32)pre[14 + pos] << ((pos & 1u) ? 0 : 8);
      fA += cw; fB += (u64)(pos >> 1) * cw;
    }
    fA = waveSum(fA % 65535u) % 65535u; fB = waveSum(fB % 65535u) % 65535u;
    const u32 len = prefixLen + nBytesTiling - 14u;
    const u64 N = ((u64)len + 1) / 2;
    u64 s1 = fA, s2 = ((N % 65535u) * fA + 65535u - fB) % 65535u;
    if (s1 == 0) s1 = 0xffff;
    if (s2 == 0) s2 = 0xffff;
    if (lane == 0) { s_cs = (u32)((s2 << 16) | s1); s_redo = (u32)s_prefix[kFastPrefixStage / 8 - 1]; }
  }
  __syncthreads();
  if (s_redo) return;
  const u32 cs = s_cs;
  if (threadIdx.x < prefixLen && threadIdx.x < outCapacity)
    out[threadIdx.x] = (threadIdx.x >= 10u && threadIdx.x < 14u) ? (u8)(cs >> (8u * (threadIdx.x - 10u))) : reinterpret_cast<const u8*>(s_prefix)[threadIdx.x];
  if (threadIdx.x == 0) res->checksum = cs;
}


// ------------------------------------------------------------------------------------------------
// ONE launch for one raster (k_fast_encode1): statistics, decisions per block, pack and checksum from a single read of the
// raster.  A workgroup keeps its 64 blocks' pixels in registers from the first load to the last bit packed; what used to
// travel through memory between the two passes (16-byte descriptors: 17 MB each way at 8192 x 8192, and the raster a
// second time) stays in LDS.
//
// Where a workgroup's span goes in the blob is the sum of the sizes of all workgroups in front of it.  Every workgroup
// publishes its size in an epoch-tagged cell as soon as its blocks are planned -- long before its span is packed -- and
// adds up, itself, the cells of the workgroups between the start of the GROUP in front of its own (kFusedGroup workgroups
// a group: at most 2 * kFusedGroup - 1 cells, two per thread, asked for before the payload is packed and looked at
// behind it); what lies in front of that comes from one more cell, left by an AGGREGATOR block: block (kFusedGroup + 1) k
// of the grid sums group k - 1 and adds the bytes in front of it (its predecessor's cell).  No chain through the pack
// workgroups, no word that everybody reads (memory serves one address at about 90 reads per microsecond), and everything a
// block waits for was published by blocks in front of it in the grid, which start first.  If that ever fails to hold a
// workgroup gives up after 2^22 polls and says so (`stuck`): the host repeats the band on the general path.
//
// The band's statistics (range, NaN / non-integer flags) ride on fire-and-forget atomics into per-16-workgroup cells (the
// checksum accumulators of k_fast_pack plus two key cells); the workgroup with the highest index waits for all arrivals,
// takes the decisions the reference takes between its sweeps (fastDecide) and writes header and checksum.  A decision
// that says "not this path" makes the bytes written so far garbage nobody reads, exactly as in the two-launch form.
// Reference: Lerc2.cpp:1474-1668 (WriteTiles), :179-381 (ComputeNumBytesNeededToWrite), :1037-1064 (checksum).
// ------------------------------------------------------------------------------------------------
template<class T>
__device__ __forceinline__ void
fusedAggregate(u32 k, u32 nGroups, const T* __restrict__ data, const BandParams& p, u32 raiseCandidates, const FastFused& f, u32 nPackGroups,
               FastEncodeResult* __restrict__ res)
{
  const int lane = laneId(), w = waveId();
  if (k == 0u)
  {
    // ---- aggregator 0 has nothing to add up: it looks at the first raster row the way Lerc2::TryRaiseMaxZError does
    // (Lerc2.cpp:1245-1290) and leaves the largest rounding error per candidate factor for the deciding workgroup
    __shared__ u64 s_rw[4][9];
    const bool doRaise = DtOf<T>::v >= DT_Float && raiseCandidates != 0u;
    if (doRaise)
    {
      const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
      double rerr[9];
#pragma unroll
      for (int cnd = 0; cnd < 9; cnd++) rerr[cnd] = 0;
#pragma unroll 2
      for (u32 col = threadIdx.x; col < (u32)p.nCols; col += 256u)
      {
        const double x = (double)data[col];
        if (x != x) continue;    // a NaN sends the band to the general path anyway
        bool exact = false;
#pragma unroll
        for (int cnd = 0; cnd < 9; cnd++)    // candidates in increasing factor order, stop at the first exact hit
        {
          if (exact || !((raiseCandidates >> cnd) & 1u)) continue;
          const double z = x * facCand[cnd];
          if (z == (double)(int)z) { exact = true; continue; }
          const double dlt = fabs(floor(z + 0.5) - z);
          rerr[cnd] = dlt > rerr[cnd] ? dlt : rerr[cnd];
        }
      }
#pragma unroll
      for (int cnd = 0; cnd < 9; cnd++)
      {
        u64 bits; const double er = rerr[cnd]; memcpy(&bits, &er, 8);    // non-negative doubles order like their bit patterns
        bits = waveMax(bits);
        if (lane == 0) s_rw[w][cnd] = bits;
      }
    }
    __syncthreads();
    if (threadIdx.x < 9)
    {
      u64 m = 0;    // (+0.0)
      if (doRaise) for (int i = 0; i < 4; i++) m = s_rw[i][threadIdx.x] > m ? s_rw[i][threadIdx.x] : m;
      publish64(f.raise + threadIdx.x, m);
    }
    drainVmem();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(f.packPart + nPackGroups, 1ull << 48, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- aggregator k >= 1 adds up group k - 1 and publishes that total at once (it hangs on nothing but the group's own
  // workgroups); the bytes in front of group k are the totals of all groups before it, which the aggregators in front of
  // this one have published -- each for itself, so that no chain runs through the aggregators (a chain of 64 hops of 1.8 us
  // was as long as the whole kernel).  Group k + 1 is the first one to ask for the bytes in front of group k.
  if (k + 1u >= nGroups) return;    // (nobody asks for the bytes in front of the last group, nor for the total of the one before it)
  __shared__ u32 s_sum[4], s_lost;
  if (threadIdx.x == 0) s_lost = 0u;
  __syncthreads();
  u32 size = 0;
  bool lost = false;
  if (threadIdx.x < kFusedGroup)
  {
    const u64* cellAt = f.sizeCell + (size_t)(k - 1u) * kFusedGroup + threadIdx.x;    // (group k - 1 is a whole group)
    u64 cell = observe64(cellAt);
    for (u32 spin = 0; (u32)(cell >> 32) != f.epoch && spin < f.spinLimit; spin++)
    {
      __builtin_amdgcn_s_sleep(2);
      cell = observe64(cellAt);
    }
    lost = (u32)(cell >> 32) != f.epoch;
    size = lost ? 0u : (u32)cell;
  }
  size = waveSum(size);
  bool lostWave = __any(lost);
  if (lane == 0) { s_sum[w] = size; if (lostWave) s_lost = 1u; }
  __syncthreads();
  const u32 total = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
  if (threadIdx.x == 0 && !s_lost) publish64(f.totalCell + (k - 1u), ((u64)f.publishEpoch << 32) | (u64)total);
  // the totals of groups 0 .. k - 2
  u32 before = 0;
  lost = false;
  for (u32 j0 = 0; j0 + 1u < k; j0 += 256u)
  {
    const u32 j = j0 + threadIdx.x;
    if (j + 1u >= k) continue;
    u64 cell = observe64(f.totalCell + j);
    for (u32 spin = 0; (u32)(cell >> 32) != f.epoch && spin < f.spinLimit; spin++)
    {
      __builtin_amdgcn_s_sleep(2);
      cell = observe64(f.totalCell + j);
    }
    lost = lost || (u32)(cell >> 32) != f.epoch;
    before += (u32)cell;
  }
  before = waveSum(before);
  lostWave = __any(lost);
  __syncthreads();    // (s_sum has been read)
  if (lane == 0) { s_sum[w] = before; if (lostWave) s_lost = 1u; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (s_lost) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else publish64(f.baseCell + k, ((u64)f.publishEpoch << 32) | (u64)(total + s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]));
}

// The tail of the one-launch encoders in three pieces.
// (1) Fletcher sums + flush of a span image whose first byte is blob byte g0: 16-byte units of the BLOB -- unit u holds the image
// bytes from 16 u - (g0 & 15) on, fetched as five words and funnel shifted; what lies outside the span is zero in LDS, adds
// nothing to the sums and is not stored.  A and B collect this lane's Fletcher terms (no reduction mod 65535: a lane
// holds a dozen units at most, A < 2^24 and B < 2^55).
__device__ __forceinline__ void
fusedFlush(const u32* s_out, u32 kLead, u32 g0, u32 spanLen, u8* __restrict__ out, u64 outCapacity, u32& A, u64& B)
{
  const bool fits = out != nullptr && (u64)g0 + spanLen <= outCapacity;       // (the last workgroup reports what does not fit)
  const u32 ldsShift = g0 & 15u;
  const u32 gAligned = g0 & ~15u;
  const u32 nUnits = (ldsShift + spanLen + 15) >> 4;
  const u32 spanEnd = ldsShift + spanLen;                   // (relative to gAligned)
  for (u32 u = threadIdx.x; u < nUnits; u += 256)
  {
    const u32 at = kLead + 16u * u - ldsShift;              // LDS byte of the unit's first byte (>= 1)
    const u32 wd0 = at >> 2, sh = 8u * (at & 3u);
    const u32 x0 = s_out[wd0], x1 = s_out[wd0 + 1], x2 = s_out[wd0 + 2], x3 = s_out[wd0 + 3], x4 = s_out[wd0 + 4];
    uint4 x;
    x.x = __builtin_amdgcn_alignbit(x1, x0, sh); x.y = __builtin_amdgcn_alignbit(x2, x1, sh);
    x.z = __builtin_amdgcn_alignbit(x3, x2, sh); x.w = __builtin_amdgcn_alignbit(x4, x3, sh);
    // (gAligned + 16 u >= 16 > 14 always holds here because spans start behind the >= 95-byte prefix)
    fletcherUnit(x, (u64)((gAligned + 16u * u - 14u) >> 1), A, B);
    const u32 lo = 16u * u;
#ifdef LERC_TUNE_WRAP_STORES    // (tuning: the blob written into its first 4 MB over and over -- what the kernel takes without HBM writes; results invalid)
    if (fits && lo >= ldsShift && lo + 16u <= spanEnd) *reinterpret_cast<uint4*>(out + ((gAligned + lo) & ((1u << 22) - 16u))) = x;
#else
    if (fits && lo >= ldsShift && lo + 16u <= spanEnd) *reinterpret_cast<uint4*>(out + gAligned + lo) = x;    // (k_fast_discover reads them next: no streaming hint)
#endif
  }
  // the span's two ragged ends (units shared with the neighbouring spans): lane = byte, sixteen lanes an end, outside the loop --
  // inside it every wave that holds such a unit paid for sixteen predicated byte stores, and two lanes going through their bytes
  // one after the other were a chain of sixteen LDS reads on the first wave's way out
  if (threadIdx.x < 32u && fits && nUnits != 0u)
  {
    const u32 end = threadIdx.x >> 4, i16 = threadIdx.x & 15u;
    const u32 u = end ? nUnits - 1u : 0u;
    const u32 lo = 16u * u, first = lo < ldsShift ? ldsShift : lo, last = min(lo + 16u, spanEnd);    // owned bytes of the unit
    const bool partial = !(first == lo && last == lo + 16u) && !(end == 1u && nUnits == 1u);    // (one unit in all: the first sixteen lanes take it)
    const u32 i = lo + i16;
    if (partial && i >= first && i < last)
    {
      const u8* img = reinterpret_cast<const u8*>(s_out) + kLead - ldsShift;    // image byte of blob byte gAligned
      out[gAligned + i] = img[i];
    }
  }
}

// (2) the arrival: checksum terms (A bits 0-23, B bits 24-47), one arrival (bits 48-52), the two flags as counts (bits 53-57
// NaN seen, bits 58-62 a non-integer value seen) in ONE atomic nobody waits for; by the thread that sent the range
// atomics, behind a wait for those (the deciding workgroup reads the key cells once the arrivals are complete)
__device__ __forceinline__ void
fusedArrive(u32 A, u64 B, u64* s_fa, u64* s_fb, const u32* s_fl, u32 wg, int wPlan, const FastFused& f)
{
  const int w = waveId(), lane = laneId();
  // (the sums are wanted mod 65535: folded to 32 bits first -- a wave's sum of 64 folds stays below 2^24 -- the reduction is DPP adds)
  const u32 a = waveSum(fold65535(A)), b2 = waveSum(fold65535(B));
  if (lane == 0) { s_fa[w] = a; s_fb[w] = b2; }
  __syncthreads();
  if (w == wPlan && lane == 0)
  {
    // (congruent to the sums and below 2^17: sixteen workgroups' terms fit the accumulator's 24-bit fields)
    const u64 a4 = fold65535((u32)(s_fa[0] + s_fa[1] + s_fa[2] + s_fa[3])), b4 = fold65535((u32)(s_fb[0] + s_fb[1] + s_fb[2] + s_fb[3]));
    const u32 fl = s_fl[0] | s_fl[1] | s_fl[2] | s_fl[3];
    drainVmem();
    __hip_atomic_fetch_add(f.packPart + wg / kFastPackGroup, a4 | (b4 << 24) | (1ull << 48) | ((u64)(fl & 1u) << 53) | ((u64)((fl >> 1) & 1u) << 58),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// (3) the workgroup with the highest index: wait for everybody (aggregator 0 included), fold, decide, header + checksum
template<class T>
__device__ __forceinline__ void
fusedFinish(u32 nBytesTiling, u32 prefixLen, u8* __restrict__ out, u64 outCapacity, u32 nWG, u32 nPackGroups, const BandParams& p, const FastFused& f,
            FastEncodeResult* __restrict__ res, double requestedMaxZErr, u32 raiseCandidates, u32 nBlobsMore)
{
  const int w = waveId(), lane = laneId();
  __shared__ u32 s_redo, s_cs;
  __shared__ u64 s_kmax[4], s_kmin[4], s_A[4], s_B[4], s_raise[9];
  __shared__ u32 s_flg[4];
  __shared__ __align__(16) u8 s_prefix[kFastPrefixStage];
  {
    u64 fA = 0, fB = 0, kMaxAll = 0, kMinInv = 0;
    u32 flg = 0;
    for (u32 gq = 0; gq <= nPackGroups; gq += 256u)
    {
      const u32 gi = gq + threadIdx.x;
      const u32 want = gi < nPackGroups ? min(kFastPackGroup, nWG - gi * kFastPackGroup) : gi == nPackGroups ? 1u : 0u;    // (+ aggregator 0)
      u64 acc = 0;
      for (u32 spin = 0; ; spin++)
      {
        acc = gi <= nPackGroups ? observe64(f.packPart + gi) : 0ull;
        if (!__any((u32)((acc >> 48) & 31u) != want)) break;
        if (spin > (1u << 22)) { if (lane == 0) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }    // (never: every other block was dispatched before this one)
        __builtin_amdgcn_s_sleep(8);
      }
      fA += acc & 0xFFFFFFull; fB += (acc >> 24) & 0xFFFFFFull;
      flg |= (((acc >> 53) & 31u) ? 1u : 0u) | (((acc >> 58) & 31u) ? 2u : 0u);
      if (gi <= nPackGroups) publish64(f.packPart + gi, 0ull);    // clean for the next call
      if (gi < nPackGroups)
      {
        const u64 a = observe64(f.keyPart + 2 * (size_t)gi), bInv = observe64(f.keyPart + 2 * (size_t)gi + 1);
        kMaxAll = a > kMaxAll ? a : kMaxAll; kMinInv = bInv > kMinInv ? bInv : kMinInv;
        publish64(f.keyPart + 2 * (size_t)gi, 0ull); publish64(f.keyPart + 2 * (size_t)gi + 1, 0ull);
      }
    }
    fA = waveSum(fA % 65535u) % 65535u; fB = waveSum(fB % 65535u) % 65535u;
    kMaxAll = waveMax(kMaxAll); kMinInv = waveMax(kMinInv);
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) flg |= __shfl_xor(flg, m);
    if (lane == 0) { s_A[w] = fA; s_B[w] = fB; s_kmax[w] = kMaxAll; s_kmin[w] = kMinInv; s_flg[w] = flg; }
  }
  // Aggregator 0's first-row errors may be read once ITS arrival has been seen -- by whichever wave holds the thread that
  // watches its counter (thread nPackGroups % 256), which is this wave only if nPackGroups % 256 < 64: hence the barrier
  // (without it a raster of one residency round -- 2048 x 4096: the last workgroup and aggregator 0 run side by side -- read
  // the cells before they were written a few times in a hundred: zeros in a fresh context, "every candidate fits", and the
  // band went to the general kernels for nothing; in a context that had seen another raster, that raster's errors).
  __syncthreads();
  if (threadIdx.x < 9) s_raise[threadIdx.x] = observe64(f.raise + threadIdx.x);
  __syncthreads();
  if (threadIdx.x == 0)
  {
    res->streamSums = (u32)((s_A[0] + s_A[1] + s_A[2] + s_A[3]) % 65535u) | ((u32)((s_B[0] + s_B[1] + s_B[2] + s_B[3]) % 65535u) << 16);
    u64 a = 0, bInv = 0;
    u32 fl = 0;
    for (int i = 0; i < 4; i++) { a = s_kmax[i] > a ? s_kmax[i] : a; bInv = s_kmin[i] > bInv ? s_kmin[i] : bInv; fl |= s_flg[i]; }
    const bool doRaise = DtOf<T>::v >= DT_Float && raiseCandidates != 0u;
    s_redo = fastDecide(p, requestedMaxZErr, raiseCandidates, nBytesTiling, ~bInv, a, fl, doRaise ? s_raise : nullptr, nBlobsMore, s_prefix, outCapacity, res, false);
    if (f.arenaCursor && outCapacity == 0ull) res->redoReason |= kRedoArena;    // (the ARENA of a batch is full, not a tile's own room)
  }
  __syncthreads();
  if (s_redo) return;
  if (w == 0)
  {
    u64 fA = s_A[0] + s_A[1] + s_A[2] + s_A[3], fB = s_B[0] + s_B[1] + s_B[2] + s_B[3];
    fA = lane == 0 ? fA : 0ull; fB = lane == 0 ? fB : 0ull;
    for (u32 pos = (u32)lane; pos + 14 < prefixLen; pos += 64u)    // + the Fletcher terms of the bytes in front of the first block
    {
      const u32 cw = (u32)s_prefix[14 + pos] << ((pos & 1u) ? 0 : 8);
      fA += cw; fB += (u64)(pos >> 1) * cw;
    }
    fA = waveSum(fA % 65535u) % 65535u; fB = waveSum(fB % 65535u) % 65535u;
    const u32 len = prefixLen + nBytesTiling - 14u;
    const u64 N = ((u64)len + 1) / 2;
    u64 s1 = fA, s2 = ((N % 65535u) * fA + 65535u - fB) % 65535u;
    if (s1 == 0) s1 = 0xffff;
    if (s2 == 0) s2 = 0xffff;
    if (lane == 0) s_cs = (u32)((s2 << 16) | s1);
  }
  __syncthreads();
  const u32 cs = s_cs;
  if (out && threadIdx.x < prefixLen && threadIdx.x < outCapacity)
    out[threadIdx.x] = (threadIdx.x >= 10u && threadIdx.x < 14u) ? (u8)(cs >> (8u * (threadIdx.x - 10u))) : s_prefix[threadIdx.x];
  if (threadIdx.x == 0) res->checksum = cs;
}


// ------------------------------------------------------------------------------------------------
// k_fast_encode1: the one-launch encoder with U consecutive units (64 blocks each) per workgroup.  All U units' pixels are
// asked for at the start; statistics and plans of all of them come first, so the workgroup's size (ONE cell: its units are
// consecutive in the stream, their spans lie back to back) is published long before the first span leaves, and the
// second unit's pixels arrive while the first unit is being packed: the time line of k_fast_encode1 showed a workgroup
// living 10.9 us of which 5.0 went by before its size was out (mostly waiting for its pixels) and 2.8 waiting for the
// cells of the workgroups in front of it, which publish when it does.  Units go through ONE span image, one after the other.
// ------------------------------------------------------------------------------------------------
// (32-bit types: eight workgroups per CU, i.e. at most 64 vector registers -- three values go to scratch and it is still
// 4 us faster at 8192 x 8192 than seven workgroups per CU without)
// PART: rows or columns are no multiples of 8 -- the blocks of the raster's last block row / column are w x h pixels
// (w, h < 8), Lerc2.cpp:1504-1519.  A lane then holds a PREFIX of its V pixels (or none), a block's element count is
// w * h instead of 64, and an element's place in the block's bit stream is its row-major index among the w * h.
// MASKED: the band has a validity mask (f.maskBits, a bit per pixel as BitMask keeps them).  A lane holds ANY subset of its V
// pixels, a block its valid pixels only -- 0 .. 64 of them, in row-major order (Lerc2::GetValidDataAndStats' masked branch,
// Lerc2.cpp:1765-1792: the first valid pixel is not compared with a "value before") -- an element's place in the block's bit
// stream is its rank among them, and a block without a valid pixel is one byte.  The general path runs this form for the block
// stream of masked bands once all decisions about the band are made (codec_encode.cpp); header, mask and checksum are its.
template<class T, bool WIDE, int U, bool PART, bool MASKED = false>
#ifndef LERC_MASKED_WAVES
#define LERC_MASKED_WAVES 4
#endif
#ifndef LERC_PART_WAVES
#define LERC_PART_WAVES 6    // (ragged rasters: at 64 registers the float kernel spills 60 bytes a lane and takes 185 us for 8190^2 instead of 142)
#endif
#ifndef LERC_W32_WAVES
#define LERC_W32_WAVES 7    // (32-bit types: 68 registers and no scratch; at eight waves -- 64 registers -- the float kernel spills three and takes 91 us instead of 85)
#endif
#ifndef LERC_U16_WAVES
#define LERC_U16_WAVES 8
#endif
__global__ void __launch_bounds__(256, (sizeof(T) <= 4 ? (MASKED ? LERC_MASKED_WAVES : PART ? LERC_PART_WAVES : sizeof(T) == 2 ? LERC_U16_WAVES : LERC_W32_WAVES) : 1)) LERC_SGPR_CAP    // (MASKED at 64 registers: 208 bytes of scratch a lane)
k_fast_encode1(const T* __restrict__ data, BandParams p, u8* __restrict__ out, FastEncodeResult* __restrict__ res, u32 nWG, u32 nBlobsMore,
                FastFused f, double requestedMaxZErr, u32 raiseCandidates, u64 outCapacity)
{
  typedef FastCfg<T> C;
  constexpr int V = C::V, LPR = C::LPR;
  // float rasters without a mask whose sides are multiples of 8: a lane holds TWO vectors of its block -- the same columns of row r and of
  // row r + 4 -- so that what is done per lane and block (the reduction steps across the block's lanes, the block's words out of LDS, the
  // place of its payload) is done once for eight pixels; a block is 8 lanes, a wave tile 8 blocks
  constexpr int NV = (LERC_ENC_WIDE && !PART && !MASKED && std::is_same<T, float>::value) ? 2 : 1;
  constexpr int VX = V * NV, RSTEP = 8 / NV;   // pixels of a lane; raster rows between its vectors
  constexpr int LB = 8 * LPR / NV;             // lanes of a block: lane = b * LB + r * LPR + h (block of the wave tile, row of the lane's first vector, lane of the row)
  constexpr int BPW = 64 / LB, IT = kFastBlocksPerWG / (4 * BPW);
  constexpr int DT = DtOf<T>::v;
  typedef typename ShflT<T>::type ST;
  static_assert(U >= 1 && U <= 4, "a plan wave per unit");
  constexpr int kMaxBlockBytes = 1 + 64 * (int)sizeof(T);
  constexpr u32 kLead = 16;                  // zero bytes in front of the span image (the flush reads up to 15 bytes before it)
  // (room for 64 raw blocks of the type -- or for what two units of 16-bit pixels usually come to: 16 KiB, the 32-bit types' size)
#ifndef LERC_IMG_MIN
#define LERC_IMG_MIN 16448
#endif
  constexpr int kImgBytes = kFastBlocksPerWG * kMaxBlockBytes > LERC_IMG_MIN ? kFastBlocksPerWG * kMaxBlockBytes : LERC_IMG_MIN;
  constexpr int kSpanWords = (kImgBytes + 16) / 4 + 8 + (int)kLead / 4;
  __shared__ __align__(16) u32 s_out[kSpanWords];
  __shared__ T s_mnT[U][kFastBlocksPerWG], s_mxT[U][kFastBlocksPerWG];
  __shared__ u32 s_same[U][kFastBlocksPerWG], s_nd[U][kFastBlocksPerWG];
  __shared__ u32 s_w1[U][kFastBlocksPerWG];
  __shared__ u32 s_bit[U][kFastBlocksPerWG]; // bit position of each block inside s_out
  __shared__ u32 s_fl[4];
  __shared__ u64 s_fa[4], s_fb[4], s_kmx[U], s_kmn[U];
  __shared__ u32 s_len[U], s_base, s_retry, s_tile;
  static_assert(!(PART && WIDE), "ragged rasters take the per-block mapping");
  static_assert(!(MASKED && WIDE), "masked bands take the per-block mapping");

  const u32 nGroups = (nWG + kFusedGroup - 1u) / kFusedGroup, nPackGroups = fastPackGroups(nWG);
  if (f.nTiles > 1u)    // a batch: this tile's pixels, cells, counters, result and slot
  {
    const size_t tile = blockIdx.y;
    data += tile * f.tileElems; res += tile;
    if (out && !f.arenaCursor) out += tile * f.outStride;
    if (f.arenaCursor) { f.tileCell += tile; f.tileOffset += tile; }
    f.sizeCell += tile * f.cellStride; f.baseCell += tile * f.cellStride; f.totalCell += tile * f.cellStride; f.raise += tile * f.cellStride;
    f.packPart += tile * f.counterStride; f.keyPart += tile * f.counterStride;
  }
  const u32 grp = blockIdx.x / (kFusedGroup + 1u), inGrp = blockIdx.x - grp * (kFusedGroup + 1u);
  if (inGrp == 0u) { fusedAggregate<T>(grp, nGroups, data, p, raiseCandidates, f, nPackGroups, res); return; }
  const u32 wg = grp * kFusedGroup + inGrp - 1u;    // (the grid has exactly nWG + nGroups blocks)
  TRACE(0); TRACE_ID();
  const int w = waveId(), lane = laneId();
  const int b = lane / LB, r = (lane % LB) / LPR, h = lane % LPR, c = b * LPR + h;
  const bool leader = (lane % LB == 0);
  const u32 nUnits = ((u32)p.nTH * (u32)p.nTV + 63u) / 64u;    // units of the raster (the last workgroup may hold fewer than U)
  FastSpan span[U];
#pragma unroll
  for (int a = 0; a < U; a++) span[a] = fastSpanOf(min(wg * (u32)U + (u32)a, nUnits - 1u), (u32)p.nTH, (u32)p.nTV);

  // ---- the pixels of all units: every load of the wave in flight first (non-temporal: nothing reads them again), the span
  // image is zeroed while they travel
  T v[U][IT][VX];
  // PART: pixels of this lane that exist (a prefix of its V: 0 .. V) and the width of its block, per wave tile
  int vcA[U][IT], bwA[U][IT];
  u32 vmA[U][IT];    // MASKED: bit k = pixel k of the lane is valid
  const u64 blockLanes = (LB == 64) ? ~0ull : (((1ull << (LB & 63)) - 1ull) << (b * LB)), earlierLanes = laneMaskLt() & blockLanes;
#pragma unroll
  for (int a = 0; a < U; a++)
#pragma unroll
    for (int t = 0; t < IT; t++)
    {
      vmA[a][t] = (1u << VX) - 1u;
      if constexpr (!PART)
      {
        vcA[a][t] = VX; bwA[a][t] = 8;
        const i64 at = laneOrigin<WIDE, BPW, V>(span[a], t * 4 + w, r, c, p.nCols);
#pragma unroll
        for (int hv = 0; hv < NV; hv++)
        {
          const i64 atv = at + (i64)(hv * RSTEP) * p.nCols;
#ifdef LERC_TUNE_WRAP_LOADS    // (tuning: every pixel read out of the raster's first 4 MB -- what the kernel takes without HBM reads; results invalid)
          loadLane<T, V>(data + (atv & (i64)((1 << 20) - 1)), reinterpret_cast<T (&)[V]>(v[a][t][hv * V]), true);
#else
          loadLane<T, V>(data + atv, reinterpret_cast<T (&)[V]>(v[a][t][hv * V]), true);
#endif
        }
        if constexpr (MASKED)
        {
          // (columns are multiples of 8 and a lane's first column one of V: its V bits lie in one byte, most significant first)
          const u32 byte = f.maskBits[at >> 3];
          vmA[a][t] = (__brev((byte << ((u32)at & 7u)) & 0xFFu) >> 24) & ((1u << V) - 1u);
        }
      }
      else if constexpr (MASKED)    // ragged AND masked: the pixels that exist, of those the valid ones (their bits may lie in two bytes)
      {
        const u32 j = (u32)(t * 4 + w) * BPW + (u32)b;
        const int bw = min(8, p.nCols - (int)fastSpanCol(span[a], j) * 8), bh = min(8, p.nRows - (int)fastSpanRow(span[a], j) * 8);
        const int vc = r < bh ? max(0, min(V, bw - h * V)) : 0;
        vcA[a][t] = vc; bwA[a][t] = bw;
        const i64 at = laneOrigin<WIDE, BPW, V>(span[a], t * 4 + w, r, c, p.nCols);
        const T* src = data + at;
#pragma unroll
        for (int k = 0; k < V; k++) v[a][t][k] = k < vc ? src[k] : T(0);
        u32 bits = 0;
        if (vc > 0)
        {
          const u32 two = ((u32)f.maskBits[at >> 3] << 8) | (u32)f.maskBits[(at >> 3) + 1];    // (the mask has slack behind its last byte)
          bits = (__brev((two << ((u32)at & 7u)) & 0xFF00u) >> 16) & ((1u << vc) - 1u);
        }
        vmA[a][t] = bits;
      }
      else
      {
        const u32 j = (u32)(t * 4 + w) * BPW + (u32)b;
        const int bw = min(8, p.nCols - (int)fastSpanCol(span[a], j) * 8), bh = min(8, p.nRows - (int)fastSpanRow(span[a], j) * 8);
        const int vc = r < bh ? max(0, min(V, bw - h * V)) : 0;
        vcA[a][t] = vc; bwA[a][t] = bw;
        const T* src = data + laneOrigin<WIDE, BPW, V>(span[a], t * 4 + w, r, c, p.nCols);
        // (whole vectors where the raster's rows start on 16-byte boundaries -- or on 4-byte ones: the same instruction at dword
        // alignment; else, and at the ragged ends, pixel by pixel)
        if (vc == V && (((size_t)p.nCols * sizeof(T)) & 15u) == 0u) loadLane<T, V>(src, v[a][t], true);
        else if (sizeof(T) * V == 16 && vc == V && (((size_t)p.nCols * sizeof(T)) & 3u) == 0u)
        {
          struct Vec16 { T e[16 / sizeof(T)]; };
          const Vec16 x16 = loadStreamingA4<Vec16>(src);
#pragma unroll
          for (int k = 0; k < V; k++) v[a][t][k] = x16.e[k];
        }
        else
        {
#pragma unroll
          for (int k = 0; k < V; k++) v[a][t][k] = k < vc ? src[k] : T(0);
        }
      }
    }
  for (u32 i = threadIdx.x; i < (u32)kSpanWords / 4u; i += 256u) reinterpret_cast<uint4*>(s_out)[i] = make_uint4(0, 0, 0, 0);
  if (threadIdx.x == 0) { s_base = 0u; s_retry = 0u; }
  if (LERC_ENC_EXIT == 1)    // (the loads are kept alive)
  {
#pragma unroll
    for (int a = 0; a < U; a++)
#pragma unroll
      for (int t = 0; t < IT; t++)
#pragma unroll
        for (int k = 0; k < VX; k++) { const u32 bits32 = (u32)rawBits<T>(v[a][t][k]); asm volatile("" :: "v"(bits32)); }
    return;
  }

  // ---- statistics per block (Lerc2::GetValidDataAndStats for an all-valid block + the tryLut count, Lerc2.cpp:1717-1799)
  // (the two flags as per-lane booleans: the compiler keeps them as lane masks in scalar registers, an s_or a tile)
  bool sawNaN = false, sawFrac = false;
#pragma unroll
  for (int a = 0; a < U; a++)
  {
#pragma unroll
    for (int t = 0; t < IT; t++)
    {
      const int tile = t * 4 + w;
      const T (&x)[VX] = v[a][t];
      const int vc = vcA[a][t];    // (V unless PART; pixels that do not exist were loaded as 0: neither NaN nor fractional)
      const u32 vm = vmA[a][t];    // (all ones unless MASKED)
      if (DT >= DT_Float)
      {
#pragma unroll
        for (int k = 0; k < VX; k++) sawNaN = sawNaN | ((!MASKED || ((vm >> k) & 1u)) & isNaNv(x[k]));
        if (!sawFrac)    // one fractional value settles "not all integers" for good
        {
#pragma unroll
          for (int k = 0; k < VX; k++) sawFrac = sawFrac | ((!MASKED || ((vm >> k) & 1u)) & notIntegral(x[k]));
        }
      }
      T mn, mx;
      if constexpr (MASKED)
      {
        mn = std::numeric_limits<T>::has_infinity ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::max();
        mx = std::numeric_limits<T>::has_infinity ? -std::numeric_limits<T>::infinity() : std::numeric_limits<T>::lowest();
#pragma unroll
        for (int k = 0; k < V; k++) if ((vm >> k) & 1u) { mn = OpMin()(mn, x[k]); mx = OpMax()(mx, x[k]); }
      }
      else if constexpr (!PART)
      {
        if constexpr (std::is_same<T, float>::value && NV == 2) laneMinMax8(x, mn, mx);
        else if constexpr (std::is_same<T, float>::value) laneMinMax4(x[0], x[1], x[2], x[3], mn, mx);
        else
        {
          mn = x[0]; mx = x[0];
#pragma unroll
          for (int k = 1; k < V; k++) { mn = OpMin()(mn, x[k]); mx = OpMax()(mx, x[k]); }
        }
      }
      else
      {
        mn = std::numeric_limits<T>::has_infinity ? std::numeric_limits<T>::infinity() : std::numeric_limits<T>::max();
        mx = std::numeric_limits<T>::has_infinity ? -std::numeric_limits<T>::infinity() : std::numeric_limits<T>::lowest();
#pragma unroll
        for (int k = 0; k < V; k++) if (k < vc) { mn = OpMin()(mn, x[k]); mx = OpMax()(mx, x[k]); }
      }
      if constexpr (std::is_same<T, float>::value && LB == 8) halfRowMinMax(mn, mx);
      else if constexpr (std::is_same<T, float>::value) rowMinMax(mn, mx);    // (LB == 16)
      else
      {
        mn = (T)groupReduce<LB>((ST)mn, OpMin());
        mx = (T)groupReduce<LB>((ST)mx, OpMax());
      }
      // "same as previous" in row-major block order, prevVal starts at 0 (Lerc2.cpp:1729-1758)
      int same, nElem = 64;
      bool lutCand = true;    // a block of the wave may have more than half of its pixels equal to their predecessor (PART, MASKED: looked at below)
      if constexpr (MASKED)
      {
        // the valid pixel in front of this lane's first valid one: the last valid pixel of the nearest lane before it in
        // the block that holds any (lanes of a block are in row-major order); the block's first valid pixel has none
        T lastMine = T(0);
#pragma unroll
        for (int k = 0; k < V; k++) if ((vm >> k) & 1u) lastMine = x[k];
        const u64 have = __ballot(vm != 0u) & earlierLanes;
        const int src = have ? 63 - __clzll((long long)have) : lane;
        T pv = shflT<T>(lastMine, src);
        bool seen = have != 0ull;
        same = 0;
#pragma unroll
        for (int k = 0; k < V; k++)
          if ((vm >> k) & 1u) { same += (seen && x[k] == pv) ? 1 : 0; pv = x[k]; seen = true; }
        nElem = 0;
#pragma unroll
        for (int k = 0; k < V; k++) nElem += __popcll(__ballot((vm >> k) & 1u) & blockLanes);
      }
      else if constexpr (!PART)
      {
        // Only "more than half of the 64 equal their predecessor" matters (tryLut), and that takes a lane whose count lies above the
        // average of 32 / LB -- of which all but one are comparisons INSIDE the lane.  Those come first (noise: no lane gets there, and
        // the pixel of the lane in front is never fetched); 32-bit types: "two of the three" as scalar logic on the comparisons' lane masks
        bool maybe;
        if constexpr (V == 4)
        {
          u64 two = 0ull;    // lanes with two of the three comparisons inside a vector equal
#pragma unroll
          for (int hv = 0; hv < NV; hv++)
          {
            const u64 e1 = __ballot(x[hv * V + 1] == x[hv * V]), e2 = __ballot(x[hv * V + 2] == x[hv * V + 1]), e3 = __ballot(x[hv * V + 3] == x[hv * V + 2]);
            two |= (e1 & e2) | (e2 & e3) | (e1 & e3);
          }
          maybe = two != 0ull;    // (two vectors a lane: above the average takes five of eight, of which two look at another lane -- three of six, so two of one vector's three)
          same = 0;
        }
        else
        {
          same = 0;
#pragma unroll
          for (int k = 1; k < V; k++) same += (x[k] == x[k - 1]) ? 1 : 0;
          maybe = __any(same >= 32 / LB);
        }
        lutCand = false;
        if (maybe)
        {
          same = 0;
#pragma unroll
          for (int hv = 0; hv < NV; hv++)
          {
            T prev = (T)dppMovT<kDppWaveShr1>((ST)x[hv * V + V - 1]);    // the previous pixel vector of the block row lives in the previous lane
            if constexpr (NV == 2)
            {
              // (the block's first lane: nothing in front of row 0; in front of row RSTEP the last pixel of row RSTEP - 1, the block's last lane's)
              const T wrap = shflT<T>(x[V - 1], (lane + LB - 1) & 63);
              if (leader) prev = hv == 0 ? T(0) : wrap;
            }
            else if (leader) prev = T(0);
            same += (x[hv * V] == prev) ? 1 : 0;
#pragma unroll
            for (int k = 1; k < V; k++) same += (x[hv * V + k] == x[hv * V + k - 1]) ? 1 : 0;
          }
          lutCand = __any(same > 32 / LB);
        }
      }
      else
      {
        // the pixel in front of this lane's first one: the last existing pixel of the lane to the left, or -- first lane of
        // a row -- of the last lane of the row above that holds pixels
        const int bw = bwA[a][t], hLast = (bw - 1) / V;
        T lastMine = T(0);
#pragma unroll
        for (int k = 0; k < V; k++) if (k == vc - 1) lastMine = x[k];
        const int src = (h > 0) ? lane - 1 : (lane - LPR + hLast);    // (row above: lane - LPR is its first lane)
        T prev = shflT<T>(lastMine, src & 63);
        if (leader) prev = T(0);
        same = (vc > 0 && x[0] == prev) ? 1 : 0;
#pragma unroll
        for (int k = 1; k < V; k++) same += (k < vc && x[k] == x[k - 1]) ? 1 : 0;
        nElem = groupReduce<LB>(vc, OpSum());
      }
      u32 nd = 0;
      if (lutCand)
      {
        same = groupReduce<LB>(same, OpSum());
        const bool tryLut = (nElem > 4) && (2 * same > nElem) && ((double)mx > (double)mn + 3 * p.maxZErr);
        if (__any(tryLut))
        {
          const double mv = ((double)mx - (double)mn) * p.scale;
          const bool need = tryLut && !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
          u32 q[VX];
          quantizeLane<T, VX>(p.intLossless, p.scale, x, mn, q);
          if constexpr (PART)
          {
#pragma unroll
            for (int k = 0; k < V; k++) if (k >= vc) q[k] = 0xFFFFFFFFu;    // (no such pixel: never the smallest value left)
          }
          if constexpr (MASKED)
          {
#pragma unroll
            for (int k = 0; k < V; k++) if (!((vm >> k) & 1u)) q[k] = 0xFFFFFFFFu;
          }
          nd = groupDistinct<LB, VX>(q, need);
        }
      }
      else same = 0;
      if (leader)
      {
        const int blk = tile * BPW + b;
        s_mnT[a][blk] = mn; s_mxT[a][blk] = mx; s_same[a][blk] = (u32)same | ((u32)nElem << 16); s_nd[a][blk] = nd;
      }
    }
  }
  const bool f1 = __any(sawNaN), f2 = __any(sawFrac);
  if (lane == 0) s_fl[w] = (f1 ? 1u : 0u) | (f2 ? 2u : 0u);
  __syncthreads();
  TRACE(5);
  ENC_EXIT(2);

  // ---- lane = block, a wave per unit (they rotate over the SIMDs from workgroup to workgroup): the per-block decisions of
  // Lerc2::NumBytesTile once per block and where each block starts in its unit's span
  const int wPlan = (int)((wg * 2654435761u) >> 30);
  const int myUnit = (w - wPlan) & 3;        // the unit this wave plans (and writes the block headers of), if < U
#pragma unroll
  for (int a = 0; a < U; a++)    // (unrolled: a unit's span stays in registers)
  {
    if (myUnit != a) continue;
    const bool haveUnit = wg * (u32)U + (u32)a < nUnits;
    const T mn = s_mnT[a][lane], mx = s_mxT[a][lane];
    const int same = (int)(s_same[a][lane] & 0xFFFFu), nElem = (int)(s_same[a][lane] >> 16);    // (nElem: 64 unless PART)
    const bool tryLut = (nElem > 4) && (2 * same > nElem) && ((double)mx > (double)mn + 3 * p.maxZErr);
    double mv = 0;
    bool quantOk = false;
    if (p.maxZErr > 0)
    {
      mv = ((double)mx - (double)mn) * p.scale;
      quantOk = !(mv > (double)p.maxQ || (u32)(mv + 0.5) == 0);
    }
    const u32 qMax = quantOk ? (u32)(mv + 0.5) : 0u;    // == largest quantised element (same expression as Quantize)
    Plan pl = planBlock<T>(p, nElem, mn, mx, DT, tryLut, mv, qMax, s_nd[a][lane]);    // (the data type as a constant: the other types' branches fold away)
    if (!haveUnit || !fastSpanHas(span[a], (u32)lane)) { pl.nBytes = 0; pl.kind = 7; }    // behind the raster's last block: nothing to write
    const int nb = bitLen(qMax);
    const u32 sz = (u32)pl.nBytes;
    const u32 inc = waveInclusiveScan(sz);
    // (where the block begins in the span image -- a bit-stuffed block without a table: where its PAYLOAD begins, behind flag byte, offset,
    // bits byte and count: the pixel owners then need nothing of the header's layout)
    s_w1[a][lane] = packDesc(pl, nb); s_bit[a][lane] = 8u * (kLead + inc - sz + (pl.kind == 3 ? 3u + (u32)dtSize3((u32)pl.dtRed) : 0u));
#ifdef HIPSIM
    const u32 total = (u32)__shfl((int)inc, 63);
#else
    const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);    // (the last lane's: into a scalar register, no LDS permute)
#endif
    // (blocks behind the raster's end repeat the last block's range: they change nothing)
    u64 kMin, kMax;
    if (sizeof(T) <= 4)    // (the keys of these types have 34 bits at most: reduce the value, order-preserving in 32 bits, encode once)
    {
      kMin = FKey<T>::enc(FOrd<T>::dec(waveMin(FOrd<T>::enc(mn)))); kMax = FKey<T>::enc(FOrd<T>::dec(waveMax(FOrd<T>::enc(mx))));
    }
    else { kMin = waveMin(FKey<T>::enc(mn)); kMax = waveMax(FKey<T>::enc(mx)); }
    if (lane == 0) { s_len[a] = total; s_kmx[a] = kMax; s_kmn[a] = kMin; }
  }
  __syncthreads();
  u32 len[U], lenAll = 0;
#pragma unroll
  for (int a = 0; a < U; a++) { len[a] = s_len[a]; lenAll += len[a]; }
  if (threadIdx.x == 0)
  {
    publish64(f.sizeCell + wg, ((u64)f.publishEpoch << 32) | (u64)lenAll);
    u64 kMax = s_kmx[0], kMin = s_kmn[0];
#pragma unroll
    for (int a = 1; a < U; a++) { kMax = s_kmx[a] > kMax ? s_kmx[a] : kMax; kMin = s_kmn[a] < kMin ? s_kmn[a] : kMin; }
    // the band's range: fire and forget, both as maxima (the cells are zero between calls)
    __hip_atomic_fetch_max(f.keyPart + 2 * (size_t)(wg / kFastPackGroup), kMax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_max(f.keyPart + 2 * (size_t)(wg / kFastPackGroup) + 1, ~kMin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  TRACE(1);
  ENC_EXIT(3);
  const bool sizeOnly = out == nullptr;    // (a size query of a ragged raster: everything but payloads and stores)

  // ---- where the spans go: the cells of the workgroups from the start of the group in front of this one's up to this
  // one (two per thread), and what lies in front of those.  Asked for now, looked at when the first unit is packed.
  const u32 winBegin = (grp ? grp - 1u : 0u) * kFusedGroup;
  const u32 i0 = winBegin + threadIdx.x, i1 = i0 + 256u;
  static_assert(2u * kFusedGroup <= 512u, "two cells per thread cover the window");
  u64 c0 = 0, c1 = 0, cb = 0;

  const u32 prefixLen = 90u + 4u + 2u * (u32)sizeof(T) + 1u;    // header, mask byte count, ranges, "not one sweep" (fastDecide)
  // All units' spans lie back to back in the blob.  Where they fit the image together (it is sized for 64 raw blocks) they are
  // packed one behind the other and leave in ONE piece: the cells asked for above then have the time of U payloads to
  // arrive -- the workgroups in front publish their sizes when this one does, and a cell takes a microsecond or two to be
  // seen -- and the flush runs once, over full rounds of lanes.  Else (mostly raw blocks) unit after unit through the image.
  const bool together = U > 1 && kLead + lenAll + 64u <= (u32)kSpanWords * 4u;
  // (asked for now where a unit's flush comes right behind its payload; behind the first payload where all units are packed
  // before anything leaves: the workgroups in front publish when this one does, and asking before their cells can be
  // seen only means asking twice)
  if (!together)
  {
    if (i0 < wg) c0 = observe64(f.sizeCell + i0);
    if (i1 < wg) c1 = observe64(f.sizeCell + i1);
    if (threadIdx.x == 0 && grp >= 2u) cb = observe64(f.baseCell + (grp - 1u));
  }
  u32 fA = 0, spanBase = 0, flushed = 0, bitBase = 0;
  u64 fB = 0;
  bool resolved = false;
#pragma unroll
  for (int a = 0; a < U; a++)
  {
    if (a > 0 && !together)
    {
      // the image again, for the next unit (everybody is done reading the last one)
      __syncthreads();
      for (u32 i = threadIdx.x; i < min((u32)kSpanWords / 4u, (kLead + len[a] + 48u) / 16u + 1u); i += 256u) reinterpret_cast<uint4*>(s_out)[i] = make_uint4(0, 0, 0, 0);
      __syncthreads();
    }
    // ---- block headers: lane = block, the wave that planned the unit (Lerc2::WriteTile, BitStuffer2 stream header)
    if (myUnit == a && !sizeOnly)
    {
      const u32 w1 = s_w1[a][lane];
      const int kind = (int)((w1 >> 16) & 7u), tc = (int)((w1 >> 19) & 3u), dtRed = (int)((w1 >> 21) & 7u), nb = (int)(w1 >> 24);
      const int j0 = (int)fastSpanCol(span[a], (u32)lane) * 8;
      u32 flag = (u32)(((j0 >> 3) & 15) << 2) & 0x38u;    // version 6, no slice difference
      const u32 at0 = s_bit[a][lane] + bitBase - (kind == 3 ? 8u * (3u + (u32)dtSize3((u32)dtRed)) : 0u);    // (the block's first byte)
      if (kind == 7) { }
      else if (kind == 0) orBits(s_out, at0, flag | 2u, 8);
      else if (kind == 1) orBits(s_out, at0, flag, 8);
      else
      {
        flag |= (kind == 2) ? 3u : 1u;
        flag |= (u32)tc << 6;
        const int offBytes = dtSize3((u32)dtRed);
        orBits(s_out, at0, flag, 8);
        orBits64(s_out, at0 + 8, typedBits((double)s_mnT[a][lane], dtRed), 8 * offBytes);
        if (kind == 3) orBits(s_out, at0 + 8u * (1u + (u32)offBytes), (u32)nb | (2u << 6) | ((s_same[a][lane] >> 16) << 8), 16);    // numBits byte, count (64)
      }
    }
    // ---- payloads
#pragma unroll
    for (int t = 0; t < IT; t++)
    {
      const int tile = t * 4 + w;
      const int blk = tile * BPW + b;
      const u32 w1 = s_w1[a][blk];
      const int kind = sizeOnly ? 7 : (int)((w1 >> 16) & 7u);
      const u32 at0 = s_bit[a][blk] + bitBase;
      const int vc = vcA[a][t];
      int e0 = PART ? r * bwA[a][t] + h * V : r * 8 + h * V;    // first element of this lane in the block's row-major order
      const u32 vm = vmA[a][t];
      if constexpr (MASKED)    // ... its rank among the block's valid pixels
      {
        e0 = 0;
#pragma unroll
        for (int k = 0; k < V; k++) e0 += __popcll(__ballot((vm >> k) & 1u) & earlierLanes);
      }
      if (kind == 3)
      {
        const int nb = (int)(w1 >> 24);
        const T mn = s_mnT[a][blk];
        u32 q[VX];
        if constexpr (NV == 1) quantizeLane<T, VX>(p.intLossless, p.scale, v[a][t], mn, q);
        if constexpr (PART)
        {
#pragma unroll
          for (int k = 0; k < V; k++) if (k >= vc) q[k] = 0u;    // (no such pixel: no bits)
        }
        const u32 at = at0;    // (the plan wave has left the payload's place)
        if constexpr (MASKED)
        {
          // the lane's valid values, closed up
          int rk = 0;
          if (V * nb <= 64)
          {
            u64 s = 0;
#pragma unroll
            for (int k = 0; k < V; k++) if ((vm >> k) & 1u) { s |= (u64)q[k] << (rk * nb); rk++; }
            if (rk) orBits64(s_out, at + (u32)e0 * (u32)nb, s, rk * nb);
          }
          else
          {
#pragma unroll
            for (int k = 0; k < V; k++) if ((vm >> k) & 1u) { orBits64(s_out, at + (u32)(e0 + rk) * (u32)nb, (u64)q[k], nb); rk++; }
          }
        }
        else if (V * nb <= 64)
        {
          // (nb <= 16 here for V >= 4: pairs fit a 32-bit word, one 64-bit shift in all instead of one per value)
#pragma unroll
          for (int hv = 0; hv < NV; hv++)
          {
            // (two vectors a lane: one after the other -- quantised, packed and gone before the next one's values take registers)
            if constexpr (NV == 2) quantizeLane<T, V>(p.intLossless, p.scale, reinterpret_cast<const T (&)[V]>(v[a][t][hv * V]), mn, reinterpret_cast<u32 (&)[V]>(q[hv * V]));
            const u32* qv = &q[hv * V];
            u64 s = 0;
            if (V == 4) s = (u64)(qv[0] | (qv[1] << nb)) | ((u64)(qv[2] | (qv[3] << nb)) << (2 * nb));
            else if (V == 8)
            {
              const u32 p0 = qv[0] | (qv[1] << nb), p1 = qv[2] | (qv[3] << nb), p2 = qv[4] | (qv[5] << nb), p3 = qv[6] | (qv[7] << nb);    // nb <= 8
              s = (u64)(p0 | (p1 << (2 * nb))) | ((u64)(p2 | (p3 << (2 * nb))) << (4 * nb));
            }
            else
            {
#pragma unroll
              for (int k = 0; k < V; k++) s |= (u64)qv[k] << (k * nb);
            }
            orBits64(s_out, at + (u32)(e0 + hv * 8 * RSTEP) * (u32)nb, s, V * nb);
          }
        }
        else
        {
#pragma unroll
          for (int hv = 0; hv < NV; hv++)
          {
            if constexpr (NV == 2) quantizeLane<T, V>(p.intLossless, p.scale, reinterpret_cast<const T (&)[V]>(v[a][t][hv * V]), mn, reinterpret_cast<u32 (&)[V]>(q[hv * V]));
#pragma unroll
            for (int k = 0; k < V; k += 2)
              orBits64(s_out, at + (u32)(e0 + hv * 8 * RSTEP + k) * (u32)nb, (u64)q[hv * V + k] | ((u64)q[hv * V + k + 1] << nb), 2 * nb);
          }
        }
      }
      else if (kind == 1)
      {
        if constexpr (MASKED)
        {
          int rk = 0;
#pragma unroll
          for (int k = 0; k < V; k++)
            if ((vm >> k) & 1u) { orBits64(s_out, at0 + 8u + (u32)(e0 + rk) * 8u * (u32)sizeof(T), rawBits<T>(v[a][t][k]), 8 * (int)sizeof(T)); rk++; }
        }
        else
        {
#pragma unroll
          for (int k = 0; k < VX; k++)
            if (!PART || k < vc) orBits64(s_out, at0 + 8u + (u32)(e0 + (k / V) * 8 * RSTEP + k % V) * 8u * (u32)sizeof(T), rawBits<T>(v[a][t][k]), 8 * (int)sizeof(T));
        }
      }
      // LUT blocks (kind 4): all blocks of the wave take part in the group reductions
      if (__any(kind == 4))
      {
        const bool mine = (kind == 4);
        const T mn = s_mnT[a][blk];
        u32 q[VX], idx[VX];
        quantizeLane<T, VX>(p.intLossless, p.scale, v[a][t], mn, q);
#pragma unroll
        for (int k = 0; k < VX; k++) idx[k] = 0;
        if constexpr (PART)
        {
#pragma unroll
          for (int k = 0; k < V; k++) if (k >= vc) q[k] = 0xFFFFFFFFu;    // (no such pixel: never the smallest value left, index bits 0)
        }
        if constexpr (MASKED)
        {
#pragma unroll
          for (int k = 0; k < V; k++) if (!((vm >> k) & 1u)) q[k] = 0xFFFFFFFFu;
        }
        const u32 nElemB = s_same[a][blk] >> 16;
        const int nb = (int)(w1 >> 24);
        const int offBytes = dtSize3((w1 >> 21) & 7u);
        const u32 hdr = at0 + 8u * (1u + (u32)offBytes);
        const u32 lutAt = hdr + 24;    // numBits byte, count byte, nLut + 1 byte
        u32 count = 0, last = 0;
        bool active = mine;
        for (;;)
        {
          u32 m = 0xFFFFFFFFu;
#pragma unroll
          for (int k = 0; k < VX; k++)
            if ((count == 0 || q[k] > last) && q[k] < m) m = q[k];
          m = groupReduce<LB>(m, OpMin());
          if (m == 0xFFFFFFFFu) active = false;
          if (!__any(active)) break;
          if (active)
          {
#pragma unroll
            for (int k = 0; k < VX; k++) if (q[k] == m) idx[k] = count;
            if (leader && count > 0) orBits(s_out, lutAt + (count - 1) * (u32)nb, m, nb);
            last = m; count++;
          }
        }
        if (mine)
        {
          const u32 nLut = count - 1;
          const int nbIdx = bitLen(nLut);
          if (leader) orBits(s_out, hdr, (u32)nb | (2u << 6) | 32u | (nElemB << 8) | ((nLut + 1) << 16), 24);
          const u32 idxAt = lutAt + 8u * ((nLut * (u32)nb + 7) >> 3);
          if constexpr (MASKED)
          {
            u64 s = 0;
            int rk = 0;
#pragma unroll
            for (int k = 0; k < V; k++) if ((vm >> k) & 1u) { s |= (u64)idx[k] << (rk * nbIdx); rk++; }
            if (rk) orBits64(s_out, idxAt + (u32)e0 * (u32)nbIdx, s, rk * nbIdx);
          }
          else
          {
#pragma unroll
            for (int hv = 0; hv < NV; hv++)
            {
              u64 s = 0;
#pragma unroll
              for (int k = 0; k < V; k++) s |= (u64)idx[hv * V + k] << (k * nbIdx);    // nbIdx <= 6
              orBits64(s_out, idxAt + (u32)(e0 + hv * 8 * RSTEP) * (u32)nbIdx, s, V * nbIdx);
            }
          }
        }
      }
    }
    if (together) bitBase += 8u * len[a];
    if (together && a == 0)
    {
      if (i0 < wg) c0 = observe64(f.sizeCell + i0);
      if (i1 < wg) c1 = observe64(f.sizeCell + i1);
      if (threadIdx.x == 0 && grp >= 2u) cb = observe64(f.baseCell + (grp - 1u));
    }
    if (together && a < U - 1) continue;    // (more units go into this image)
    if (LERC_ENC_EXIT == 4) { __syncthreads(); return; }
    if (!resolved)
    {
      resolved = true;
      TRACE(2);
      // ---- the cells asked for above: nearly always all there
      {
        const bool need0 = i0 < wg, need1 = i1 < wg, needB = threadIdx.x == 0 && grp >= 2u;
        const bool miss = (need0 && (u32)(c0 >> 32) != f.epoch) || (need1 && (u32)(c1 >> 32) != f.epoch) || (needB && (u32)(cb >> 32) != f.epoch);
        u32 part = (need0 ? (u32)c0 : 0u) + (need1 ? (u32)c1 : 0u) + (needB ? (u32)cb : 0u);
        part = waveSum(part);
        if (lane == 0) atomicAdd(&s_base, part);
        if (__any(miss) && lane == 0) s_retry = 1u;
      }
      __syncthreads();
      if (s_retry)
      {
        // (a workgroup in front of this one was slower than this one: ask again)
        __syncthreads();
        if (threadIdx.x == 0) s_base = 0u;
        __syncthreads();
        const bool need0 = i0 < wg, need1 = i1 < wg, needB = threadIdx.x == 0 && grp >= 2u;
        bool lost = false;
        for (u32 spin = 0; ; spin++)
        {
          if (need0 && (u32)(c0 >> 32) != f.epoch) c0 = observe64(f.sizeCell + i0);
          if (need1 && (u32)(c1 >> 32) != f.epoch) c1 = observe64(f.sizeCell + i1);
          if (needB && (u32)(cb >> 32) != f.epoch) cb = observe64(f.baseCell + (grp - 1u));
          const bool miss = (need0 && (u32)(c0 >> 32) != f.epoch) || (need1 && (u32)(c1 >> 32) != f.epoch) || (needB && (u32)(cb >> 32) != f.epoch);
          if (!miss) break;
          if (spin >= f.spinLimit) { lost = true; break; }
          __builtin_amdgcn_s_sleep(4);
        }
        u32 part = (need0 ? (u32)c0 : 0u) + (need1 ? (u32)c1 : 0u) + (needB ? (u32)cb : 0u);
        part = waveSum(part);
        if (lane == 0) atomicAdd(&s_base, part);
        if (__any(lost) && lane == 0) __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      }
      TRACE(3);
      spanBase = s_base;
      if (f.arenaCursor && !sizeOnly)
      {
        // ---- a batch straight into the arena: where does this TILE go?  Its last workgroup knows the tile's size now (the spans in
        // front of its own are placed), claims that much of the arena and says where; the others have published their sizes long
        // ago and pick the answer up.  (They wait for a workgroup dispatched behind them -- but right behind them: a tile's
        // workgroups are consecutive in dispatch order, and whatever else is resident belongs to tiles in front, which need nobody
        // behind them to finish.  A waiter that gives up says so like everybody else: the host repeats the tile.)
        if (threadIdx.x == 0)
        {
          u32 off16 = 0xFFFFFFFFu;
          if (wg == nWG - 1u)
          {
            const u64 size16 = ((u64)prefixLen + spanBase + lenAll + 15ull) & ~15ull;
            const u64 at = f.arenaBase + __hip_atomic_fetch_add(f.arenaCursor, size16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (at + size16 <= f.arenaCapacity && (at >> 4) < 0xFFFFFFFFull) off16 = (u32)(at >> 4);
            *f.tileOffset = at;
            publish64(f.tileCell, ((u64)f.publishEpoch << 32) | (u64)off16);
          }
          else
          {
            u64 cell = observe64(f.tileCell);
            for (u32 spin = 0; (u32)(cell >> 32) != f.epoch && spin < f.spinLimit; spin++)
            {
              __builtin_amdgcn_s_sleep(4);
              cell = observe64(f.tileCell);
            }
            if ((u32)(cell >> 32) == f.epoch) off16 = (u32)cell;
            else __hip_atomic_store(&res->stuck, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          s_tile = off16;
        }
        __syncthreads();
        const u32 off16 = s_tile;
        if (off16 == 0xFFFFFFFFu) outCapacity = 0ull;    // (no room, or no answer: nothing is stored; the last workgroup's verdict says "capacity")
        else { out += (u64)off16 << 4; outCapacity = f.arenaCapacity - ((u64)off16 << 4); }
      }
    }
    else __syncthreads();    // (the image is complete)
    const u32 segLen = together ? lenAll : len[a];
    fusedFlush(s_out, kLead, (MASKED ? f.payloadAt : prefixLen) + spanBase + flushed, segLen, out, outCapacity, fA, fB);    // (MASKED: behind the band's real sections, which the host writes)
    flushed += segLen;
  }
  if (LERC_ENC_EXIT == 5) { if (fA == 0x12345678u && fB == 77ull) s_base = 1u; return; }
  fusedArrive(fA, fB, s_fa, s_fb, s_fl, wg, wPlan, f);
  TRACE(4);
  if (wg != nWG - 1u) return;
  fusedFinish<T>(spanBase + lenAll, prefixLen, out, outCapacity, nWG, nPackGroups, p, f, res, requestedMaxZErr, raiseCandidates, nBlobsMore);
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
      if (run + sz > arenaCapacity) { res[t].redo = 1u; res[t].redoReason = kRedoCapacity | kRedoArena; }    // does not fit: nothing is written
      run += sz;
    }
  }
  if (threadIdx.x == 1023) tileOffset[nTiles] = run;
}

// Batches through the one-launch encoder: every tile's blob lies in a slot of its own (nobody knows where a tile goes in the
// arena before the tiles in front of it are sized); k_fast_tile_offsets places the tiles, this moves them: 16-byte units,
// both ends aligned.  A tile the general path has to redo is not moved.
__global__ void __launch_bounds__(256)
k_fast_tile_copy(const FastEncodeResult* __restrict__ res, const u64* __restrict__ tileOffset, const u8* __restrict__ slots, u64 slotStride,
                 u8* __restrict__ arena)
{
  const size_t tile = blockIdx.y;
  if (res[tile].redo) return;
  const u32 nUnits = (res[tile].blobSize + 15u) >> 4;
  const uint4* src = reinterpret_cast<const uint4*>(slots + tile * slotStride);
  uint4* dst = reinterpret_cast<uint4*>(arena + tileOffset[tile]);
  for (u32 u = blockIdx.x * 1024u + threadIdx.x; u < min(nUnits, (blockIdx.x + 1u) * 1024u); u += 256u) dst[u] = src[u];
}
void launchFastTileCopy(const FastEncodeResult* res, const u64* tileOffset, const u8* slots, u64 slotStride, u64 slotBytes, u8* arena, u32 nTiles,
                        u64 arenaBase, u64 arenaCapacity, hipStream_t st)
{
  hipLaunchKernelGGL(k_fast_tile_offsets, dim3(1), dim3(1024), 0, st, const_cast<FastEncodeResult*>(res), nTiles, arenaBase, arenaCapacity, const_cast<u64*>(tileOffset));
  hipLaunchKernelGGL(k_fast_tile_copy, dim3((u32)((slotBytes / 16 + 1023) / 1024), nTiles), dim3(256), 0, st, res, tileOffset, slots, slotStride, arena);
}

// ------------------------------------------------------------------------------------------------
bool fastEncodeEligible(int dt, int nRows, int nCols, int nDepth, bool hasMask, double maxZErr, bool ragged)
{
  if (hasMask || nDepth != 1) return false;
  if (dt == DT_Char || dt == DT_Byte) return false;    // 8-bit: the Huffman decision needs the general path
  if (!(ragged ? fastDimsOkRagged(nRows, nCols) : fastDimsOk(dt, nRows, nCols))) return false;
  if (dt >= DT_Float && maxZErr == 0) return false;    // lossless float is out of scope altogether
  return true;
}

u32 fastEncodeNumWG(int nRows, int nCols) { return fastNumWG(nRows, nCols); }

template<class T, bool SOLO>
static void launchFastEncodeT(int stage, const BandParams& p, double requested, u32 raiseCand, const void* data, u8* out, u64 cap,
                              u64 arenaBase, const FastEncodeBuffers& b, const FastBatch& batch, hipStream_t st)
{
  const u32 nWG = batch.nWG, nT = batch.nTiles;
  const bool wide = p.nTH % 64 == 0;
  if (b.fused.sizeCell && (out || p.nRows % 8 != 0 || p.nCols % 8 != 0 || b.fused.maskBits))    // one raster, one launch
  {
    if (stage != 0) return;
    const u32 nW = b.fused.nWG;
    const dim3 grid(nW + fastFusedGroups(nW), b.fused.nTiles > 1u ? b.fused.nTiles : 1u);
#ifndef LERC_U32
#define LERC_U32 2
#endif
    constexpr int U = sizeof(T) == 2 ? 3 : LERC_U32;
    if (b.fused.maskBits && (p.nRows % 8 != 0 || p.nCols % 8 != 0))
      hipLaunchKernelGGL((k_fast_encode1<T, false, U, true, true>), grid, dim3(256), 0, st, (const T*)data, p, out, b.result, nW, batch.nBlobsMore, b.fused, requested, raiseCand, cap);
    else if (b.fused.maskBits)
      hipLaunchKernelGGL((k_fast_encode1<T, false, U, false, true>), grid, dim3(256), 0, st, (const T*)data, p, out, b.result, nW, batch.nBlobsMore, b.fused, requested, raiseCand, cap);
    else if (p.nRows % 8 != 0 || p.nCols % 8 != 0)
      hipLaunchKernelGGL((k_fast_encode1<T, false, U, true>), grid, dim3(256), 0, st, (const T*)data, p, out, b.result, nW, batch.nBlobsMore, b.fused, requested, raiseCand, cap);
    else if (wide)
      hipLaunchKernelGGL((k_fast_encode1<T, true, U, false>), grid, dim3(256), 0, st, (const T*)data, p, out, b.result, nW, batch.nBlobsMore, b.fused, requested, raiseCand, cap);
    else
      hipLaunchKernelGGL((k_fast_encode1<T, false, U, false>), grid, dim3(256), 0, st, (const T*)data, p, out, b.result, nW, batch.nBlobsMore, b.fused, requested, raiseCand, cap);
    return;
  }
  if (stage == 0)
  {
    if (wide)
      hipLaunchKernelGGL((k_fast_stats<T, true>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, b.desc, b.wgSize, b.wgMinKey, b.wgMaxKey,
                         b.wgFlags, b.tickets, batch);
    else
      hipLaunchKernelGGL((k_fast_stats<T, false>), dim3(nWG, nT), dim3(256), 0, st, (const T*)data, p, b.desc, b.wgSize, b.wgMinKey, b.wgMaxKey,
                         b.wgFlags, b.tickets, batch);
  }
  else if (stage == 1)
  {
    if (SOLO) return;    // (k_fast_pack<SOLO> scans and decides itself)
    // a tile of a batch may be as large as it likes here; whether the arena holds it is decided by the placement
    hipLaunchKernelGGL(k_fast_scan_decide<T>, dim3(fastScanGroups(nWG), nT), dim3(nWG <= 1024u ? 256 : 1024), 0, st, (const T*)data, p, requested, raiseCand, nWG, (const u32*)b.wgSize, b.wgBase,
                       (const u64*)b.wgMinKey, (const u64*)b.wgMaxKey, (const u32*)b.wgFlags, b.prefixStage,
                       b.tileOffset ? ~0ull : cap, b.result, b.groupBase, b.scanPart, b.packPart, b.tickets, batch);
    if (b.tileOffset)
      hipLaunchKernelGGL(k_fast_tile_offsets, dim3(1), dim3(1024), 0, st, b.result, nT, arenaBase, cap, b.tileOffset);
  }
  else
  {
    const dim3 grid = SOLO ? dim3(out ? nWG + (nWG + kSoloSlice - 1u) / kSoloSlice : 1u) : dim3(nWG, nT);    // (SOLO: + the scan blocks; a size query needs block 0 only)
    if (wide)
      hipLaunchKernelGGL((k_fast_pack<T, true, SOLO>), grid, dim3(256), 0, st, (const T*)data, p, (const FastBlockDesc*)b.desc,
                         (const u32*)b.wgSize, (const u32*)b.wgBase, (const u32*)b.groupBase, out, b.packPart,
                         b.result, b.prefixStage, (const u64*)b.tileOffset, batch, b.solo, (const u64*)b.wgMinKey, (const u64*)b.wgMaxKey,
                         (const u32*)b.wgFlags, requested, raiseCand, cap);
    else
      hipLaunchKernelGGL((k_fast_pack<T, false, SOLO>), grid, dim3(256), 0, st, (const T*)data, p, (const FastBlockDesc*)b.desc,
                         (const u32*)b.wgSize, (const u32*)b.wgBase, (const u32*)b.groupBase, out, b.packPart,
                         b.result, b.prefixStage, (const u64*)b.tileOffset, batch, b.solo, (const u64*)b.wgMinKey, (const u64*)b.wgMaxKey,
                         (const u32*)b.wgFlags, requested, raiseCand, cap);
  }
}

template<class T>
static void launchFastEncodeT(int stage, const BandParams& p, double requested, u32 raiseCand, const void* data, u8* out, u64 cap,
                              u64 arenaBase, const FastEncodeBuffers& b, const FastBatch& batch, hipStream_t st)
{
  if (b.solo.cells) launchFastEncodeT<T, true>(stage, p, requested, raiseCand, data, out, cap, arenaBase, b, batch, st);
  else launchFastEncodeT<T, false>(stage, p, requested, raiseCand, data, out, cap, arenaBase, b, batch, st);
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
