// fpl_kernels.hip -- device side of the lossless float / double image mode (IEM_DeltaDeltaHuffman).
//
// Reference counterparts: UnitTypes::doFloatTransform / setRowsDerivative / setCrossDerivative / restoreBlockSequence /
// restoreCrossBytes (fpl_UnitTypes.cpp:34-150, :302-357, :436-517, :626-697, :775-849), testBlocksSize and getBestLevel2
// (fpl_Lerc2Ext.cpp:167-322), setDerivative / restoreSequence (:112-165), encodePackBits / decodePackBits /
// getPackBitsSize (fpl_EsriHuffman.cpp:37-236).
//
// The reference walks every array front to back (or back to front); here each of those walks is either elementwise
// (a difference only needs the ORIGINAL neighbours) or a prefix sum (the inverse), so it runs as one thread per element
// resp. as a scan.  The two "which predictor / which level is best" decisions are taken on the host from histograms of
// exactly the bytes the reference samples (every 7th byte of its test blocks / snippets).
#include "fpl_dev.h"
#include "wave_utils.h"

namespace lerc {

namespace {

template<int U> struct UnitOf;
template<> struct UnitOf<4> { typedef u32 T; };
template<> struct UnitOf<8> { typedef u64 T; };

// float bits: sign | exponent | mantissa  ->  exponent | sign | mantissa (fpl_UnitTypes.cpp:39-65); doubles stay as they are
__device__ __forceinline__ u32 fplForward(u32 a) { return (a & 0x007FFFFFu) | (((a >> 23) & 0xFFu) << 24) | ((a >> 31) << 23); }
__device__ __forceinline__ u32 fplBackward(u32 a) { return (a & 0x007FFFFFu) | (((a >> 24) & 0xFFu) << 23) | (((a >> 23) & 1u) << 31); }
__device__ __forceinline__ u64 fplForward(u64 a) { return a; }
__device__ __forceinline__ u64 fplBackward(u64 a) { return a; }

// mantissa and the bits above it are differenced apart (SUB32_BIT_FLT / ADD32_BIT_FLT, SUB64_BIT_DBL / ADD64_BIT_DBL)
__device__ __forceinline__ u32 fplSub(u32 a, u32 b) { return ((a - b) & 0x007FFFFFu) | ((((a >> 23) - (b >> 23)) & 0x1FFu) << 23); }
__device__ __forceinline__ u32 fplAdd(u32 a, u32 b) { return ((a + b) & 0x007FFFFFu) | ((((a >> 23) + (b >> 23)) & 0x1FFu) << 23); }
__device__ __forceinline__ u64 fplSub(u64 a, u64 b)
{
  return ((a - b) & 0x000FFFFFFFFFFFFFull) | ((((a >> 52) - (b >> 52)) & 0xFFFull) << 52);
}
__device__ __forceinline__ u64 fplAdd(u64 a, u64 b)
{
  return ((a + b) & 0x000FFFFFFFFFFFFFull) | ((((a >> 52) + (b >> 52)) & 0xFFFull) << 52);
}

__device__ __forceinline__ bool isNaNBits(u32 a) { return (a & 0x7FFFFFFFu) > 0x7F800000u; }
__device__ __forceinline__ bool isNaNBits(u64 a) { return (a & 0x7FFFFFFFFFFFFFFFull) > 0x7FF0000000000000ull; }

template<int U>
__device__ __forceinline__ typename UnitOf<U>::T loadUnit(const void* __restrict__ data, const u8* __restrict__ byteMask, const FplGeom& g, i64 e)
{
  typedef typename UnitOf<U>::T T;
  T x = ((const T*)data)[e];
  if (g.nanToZero && isNaNBits(x) && (!byteMask || byteMask[e])) x = 0;
  return fplForward(x);
}

// the element at e under the three predictors: none, difference along the row, difference along row and column
template<int U>
__device__ __forceinline__ void predictAll(const void* __restrict__ data, const u8* __restrict__ byteMask, const FplGeom& g, i64 e,
                                           typename UnitOf<U>::T out[3])
{
  typedef typename UnitOf<U>::T T;
  const u32 r32 = (u32)e / (u32)g.cols;    // (elements and columns fit 31 bits: a 32-bit division is a fraction of a 64-bit one)
  const i64 r = r32, c = e - r * g.cols;
  const T u = loadUnit<U>(data, byteMask, g, e);
  const T d1 = (c >= 1) ? fplSub(u, loadUnit<U>(data, byteMask, g, e - 1)) : u;
  T d2 = d1;
  if (r >= 1)
  {
    const T ua = loadUnit<U>(data, byteMask, g, e - g.cols);
    const T d1a = (c >= 1) ? fplSub(ua, loadUnit<U>(data, byteMask, g, e - g.cols - 1)) : ua;
    d2 = fplSub(d1, d1a);
  }
  out[0] = u; out[1] = d1; out[2] = d2;
}

// finite difference of order o (0..5) of a byte sequence, x[j] = the byte j places back
__device__ __forceinline__ u32 byteDifference(const u32 x[6], int o)
{
  switch (o)
  {
    case 0: return x[0] & 255u;
    case 1: return (x[0] - x[1]) & 255u;
    case 2: return (x[0] - 2u * x[1] + x[2]) & 255u;
    case 3: return (x[0] - 3u * x[1] + 3u * x[2] - x[3]) & 255u;
    case 4: return (x[0] - 4u * x[1] + 6u * x[2] - 4u * x[3] + x[4]) & 255u;
    default: return (x[0] - 5u * x[1] + 10u * x[2] - 10u * x[3] + 5u * x[4] - x[5]) & 255u;
  }
}

// ------------------------------------------------------------------------------------------------
// which predictor: histograms over the reference's test blocks (fpl_Lerc2Ext.cpp:57-101, :167-232)
// ------------------------------------------------------------------------------------------------
template<int U>
__global__ void __launch_bounds__(256)
k_fpl_predictor_samples(const void* __restrict__ data, const u8* __restrict__ byteMask, FplGeom g, const FplSpan* __restrict__ blocks,
                        u32* __restrict__ histos)
{
  typedef typename UnitOf<U>::T T;
  __shared__ u32 s_h[3 * 2 * U * 256];
  for (int i = threadIdx.x; i < 3 * 2 * U * 256; i += 256) s_h[i] = 0;
  __syncthreads();
  const FplSpan sp = blocks[blockIdx.x];
  const i64 nSamples = (sp.len + kFplPrime - 1) / kFplPrime;
  for (i64 k = threadIdx.x; k < nSamples; k += 256)
  {
    const i64 i = k * kFplPrime, e = sp.start + i;
    T v[3], w[3] = { 0, 0, 0 };
    predictAll<U>(data, byteMask, g, e, v);
    if (i > 0) predictAll<U>(data, byteMask, g, e - 1, w);
    for (int p = 0; p < 3; p++)
      for (int b = 0; b < U; b++)
      {
        const u32 x = (u32)(v[p] >> (8 * b)) & 255u;
        const u32 y = (i > 0) ? (x - ((u32)(w[p] >> (8 * b)) & 255u)) & 255u : x;    // setDerivativePrime (:81-95)
        atomicAdd(&s_h[((p * 2 + 0) * U + b) * 256 + x], 1u);
        atomicAdd(&s_h[((p * 2 + 1) * U + b) * 256 + y], 1u);
      }
  }
  __syncthreads();
  u32* out = histos + (size_t)blockIdx.x * (3 * 2 * U * 256);
  for (int i = threadIdx.x; i < 3 * 2 * U * 256; i += 256) out[i] = s_h[i];
}

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_predict(const void* __restrict__ data, const u8* __restrict__ byteMask, FplGeom g, int predictor, typename UnitOf<U>::T* __restrict__ units)
{
  typedef typename UnitOf<U>::T T;
  const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
  if (e >= g.nElem) return;
  T v[3];
  predictAll<U>(data, byteMask, g, e, v);
  units[e] = (predictor == 0) ? v[0] : (predictor == 1 ? v[1] : v[2]);    // (no indexed access: that would put v[] in scratch)
}

// four elements of one row per thread, for rasters whose rows are whole quads (cols % 4 == 0, base 16-byte aligned)
template<int U> struct alignas(16) Quad { typename UnitOf<U>::T v[4]; };

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_predict_quads(const void* __restrict__ data, const u8* __restrict__ byteMask, FplGeom g, int predictor,
                    typename UnitOf<U>::T* __restrict__ units)
{
  typedef typename UnitOf<U>::T T;
  const i64 e0 = ((i64)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e0 >= g.nElem) return;
  const u32 r = (u32)e0 / (u32)g.cols, c = (u32)e0 - r * (u32)g.cols;
  T cur[5], out[4];    // cur[k] = element e0 - 1 + k
  {
    const Quad<U> q = *reinterpret_cast<const Quad<U>*>((const T*)data + e0);
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      T x = q.v[k];
      if (g.nanToZero && isNaNBits(x) && (!byteMask || byteMask[e0 + k])) x = 0;
      cur[k + 1] = fplForward(x);
    }
    cur[0] = (c >= 1u) ? loadUnit<U>(data, byteMask, g, e0 - 1) : (T)0;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) out[k] = (predictor == 0 || c + k == 0u) ? cur[k + 1] : fplSub(cur[k + 1], cur[k]);
  if (predictor == 2 && r >= 1u)
  {
    T abv[5];
    const i64 a0 = e0 - g.cols;
    const Quad<U> q = *reinterpret_cast<const Quad<U>*>((const T*)data + a0);
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      T x = q.v[k];
      if (g.nanToZero && isNaNBits(x) && (!byteMask || byteMask[a0 + k])) x = 0;
      abv[k + 1] = fplForward(x);
    }
    abv[0] = (c >= 1u) ? loadUnit<U>(data, byteMask, g, a0 - 1) : (T)0;
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = fplSub(out[k], (c + k == 0u) ? abv[k + 1] : fplSub(abv[k + 1], abv[k]));
  }
  Quad<U> o;
#pragma unroll
  for (int k = 0; k < 4; k++) o.v[k] = out[k];
  *reinterpret_cast<Quad<U>*>(units + e0) = o;
}

// ------------------------------------------------------------------------------------------------
// which extra difference order per byte plane: histograms over the reference's snippets (:237-322)
// ------------------------------------------------------------------------------------------------
template<int U>
__global__ void __launch_bounds__(256)
k_fpl_level_samples(const typename UnitOf<U>::T* __restrict__ units, FplGeom g, const FplSpan* __restrict__ snippets, u32 nSnippets,
                    u32* __restrict__ histos)
{
  __shared__ u32 s_h[kFplLevels * 256];
  for (int i = threadIdx.x; i < kFplLevels * 256; i += 256) s_h[i] = 0;
  __syncthreads();
  const FplSpan sp = snippets[blockIdx.x];
  const int b = (int)blockIdx.y;
  const i64 nSamples = (sp.len + kFplPrime - 1) / kFplPrime;
  for (i64 k = threadIdx.x; k < nSamples; k += 256)
  {
    const i64 off = k * kFplPrime, i = sp.start + off;
    u32 x[6];
    for (int j = 0; j < 6; j++) x[j] = (j <= off) ? (u32)(units[i - j] >> (8 * b)) & 255u : 0u;
    for (int l = 0; l < kFplLevels; l++)
    {
      const int o = (off < l) ? (int)off : l;    // the first bytes of a snippet keep their lower order
      atomicAdd(&s_h[l * 256 + byteDifference(x, o)], 1u);
    }
  }
  __syncthreads();
  u32* out = histos + ((size_t)b * nSnippets + blockIdx.x) * (kFplLevels * 256);
  for (int i = threadIdx.x; i < kFplLevels * 256; i += 256) out[i] = s_h[i];
}

// ------------------------------------------------------------------------------------------------
// entropy estimate of a byte histogram (fpl_Compression.cpp:85-113): sum over the bins, in bin order, of
// log2(total / count) * count in double precision, then (long)((bits + 7) / 8).  The logarithms come out of a table
// the host made with its libm (one per histogram total: the totals are sample counts the host knows), products and
// the running sum are IEEE operations in the reference's order (this file is compiled with -ffp-contract=off), so the
// numbers are the reference's.  One wave per histogram; adding the 0.0 of an empty bin changes nothing.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_fpl_entropy(const u32* __restrict__ histos, u32 nHist, const double* __restrict__ tables, FplEntropyTables tb, i64* __restrict__ out)
{
  __shared__ double s_term[4][256];
  const int w = waveId(), lane = laneId();
  const u32 h = blockIdx.x * 4u + (u32)w;
  const bool live = h < nHist;
  u32 c[4] = { 0, 0, 0, 0 }, total = 0;
  if (live)
    for (int k = 0; k < 4; k++) { c[k] = histos[(size_t)h * 256 + lane + 64 * k]; total += c[k]; }
  for (int d = 1; d < 64; d <<= 1) total += __shfl_xor(total, d);
  i64 at = -1;
  for (u32 t = 0; t < tb.nTables; t++) if (tb.total[t] == total) at = (i64)tb.at[t];
  for (int k = 0; k < 4; k++)
    s_term[w][lane + 64 * k] = (c[k] != 0u && at >= 0) ? tables[at + c[k]] * (double)c[k] : 0.0;
  __syncthreads();
  if (live && lane == 0)
  {
    double bits = 0;
    for (int i = 0; i < 256; i++) bits += s_term[w][i];
    out[h] = (total != 0u && at < 0) ? (i64)-1 : (i64)((bits + 7) / 8);
  }
}

// ------------------------------------------------------------------------------------------------
// the byte planes as they are entropy coded (:557-567 + setDerivative), and their histograms
// ------------------------------------------------------------------------------------------------
static const int kFplSymbolTiles = 16;    // tiles of 1024 elements per workgroup: 16 times fewer histogram flushes

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_symbols(const typename UnitOf<U>::T* __restrict__ units, FplGeom g, FplLevels lv, i64 planeStride, u8* __restrict__ planes,
              u32* __restrict__ histos)
{
  typedef typename UnitOf<U>::T T;
  __shared__ u32 s_h[U * 256 + 8];    // histograms, then per plane: how many bytes equal their successor
  for (int i = threadIdx.x; i < U * 256 + 8; i += 256) s_h[i] = 0;
  __syncthreads();
  const int lane = laneId();
  for (int tile = 0; tile < kFplSymbolTiles; tile++)
  {
    const i64 i0 = (((i64)blockIdx.x * kFplSymbolTiles + tile) * 256 + threadIdx.x) * 4;
    if (i0 - 4 * (i64)threadIdx.x >= g.nElem) break;    // (whole workgroup)
    // (threads behind the last element run along with nothing to count: the ballots below want whole waves)
    T p[10];    // p[5 + k] = element i0 + k  (all loops below unroll: p[] and x[] stay in registers)
#pragma unroll
    for (int k = -5; k < 5; k++) p[5 + k] = (i0 + k >= 0 && i0 + k < g.nElem) ? units[i0 + k] : (T)0;
#pragma unroll
    for (int b = 0; b < U; b++)
    {
      const int L = lv.level[b];
      u32 word = 0, prev = 0, equal = 0;
#pragma unroll
      for (int k = 0; k < 5; k++)    // (the fifth symbol belongs to the next thread: only compared)
      {
        const i64 i = i0 + k;
        const bool in = i < g.nElem;
        u32 x[6];
#pragma unroll
        for (int j = 0; j < 6; j++) x[j] = (u32)(p[5 + k - j] >> (8 * b)) & 255u;
        const u32 s = byteDifference(x, (i < L) ? (int)i : L);
        if (in && k > 0 && s == prev) equal++;
        prev = s;
        if (in && k < 4) word |= s << (8 * k);
        // histogram.  The high planes hold a few values almost everywhere, and 64 lanes adding to one LDS word take 64
        // turns: lanes that agree are counted by ballot and added once, as long as the groups are big
        bool mine = in && k < 4;
        u64 rest = __ballot(mine);
        for (int round = 0; round < 4 && rest; round++)
        {
          const int leader = __ffsll((long long)rest) - 1;
          const u32 sL = __shfl(s, leader);
          const u64 m = __ballot(mine && s == sL);
          if (__popcll(m) < 6) break;
          if (lane == leader) atomicAdd(&s_h[b * 256 + sL], (u32)__popcll(m));
          if (s == sL) mine = false;
          rest &= ~m;
        }
        if (mine) atomicAdd(&s_h[b * 256 + s], 1u);
      }
      equal = waveSum(equal);
      if (lane == 0 && equal) atomicAdd(&s_h[U * 256 + b], equal);
      u8* dst = planes + (size_t)b * planeStride + i0;    // planeStride is a multiple of 16
      if (i0 + 3 < g.nElem) *reinterpret_cast<u32*>(dst) = word;
      else for (int k = 0; k < 4 && i0 + k < g.nElem; k++) dst[k] = (u8)(word >> (8 * k));
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < U * 256 + 8; i += 256) if (s_h[i]) atomicAdd(&histos[i], s_h[i]);
}

// ------------------------------------------------------------------------------------------------
// inclusive scans (identity 0 for both operators): 1024 elements per workgroup, three kernels
// ------------------------------------------------------------------------------------------------
struct ScanMax { template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return a > b ? a : b; } };
struct ScanSum { template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return (X)(a + b); } };

// inclusive scan of one value per thread over the 256 threads of a workgroup; returns the EXCLUSIVE prefix of the thread
template<class X, class Op>
__device__ __forceinline__ X workgroupExclusive(X v, Op op, X* s_wave /* [4] */, X& total)
{
  const int lane = laneId(), w = waveId();
  X inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const X t = __shfl_up(inc, (unsigned)d); if (lane >= d) inc = op(t, inc); }
  X exc = __shfl_up(inc, 1u);
  if (lane == 0) exc = 0;
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  X before = 0;
  for (int i = 0; i < w; i++) before = op(before, s_wave[i]);
  total = op(op(op(s_wave[0], s_wave[1]), s_wave[2]), s_wave[3]);
  __syncthreads();
  return op(before, exc);
}

template<class T, class Op>
__global__ void __launch_bounds__(256) k_gscan_local(T* __restrict__ data, u32 n, T* __restrict__ partial)
{
  __shared__ u32 s_wave[4];
  const Op op;
  const u32 i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
  u32 v[4];
  for (int k = 0; k < 4; k++) v[k] = (i0 + k < n) ? (u32)data[i0 + k] : 0u;
  for (int k = 1; k < 4; k++) v[k] = (u32)(T)op(v[k - 1], v[k]);
  u32 total;
  const u32 before = workgroupExclusive<u32, Op>(v[3], op, s_wave, total);
  for (int k = 0; k < 4; k++) if (i0 + k < n) data[i0 + k] = (T)op(before, v[k]);
  if (threadIdx.x == 0) partial[blockIdx.x] = (T)total;
}

template<class T, class Op>
__global__ void __launch_bounds__(256) k_gscan_partials(T* __restrict__ partial, u32 nPart)    // -> exclusive, one workgroup
{
  __shared__ u32 s_wave[4];
  const Op op;
  u32 carry = 0;
  for (u32 base = 0; base < nPart; base += 4096u)
  {
    const u32 i0 = base + threadIdx.x * 16u;
    u32 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = (i0 + k < nPart) ? (u32)partial[i0 + k] : 0u;
#pragma unroll
    for (int k = 1; k < 16; k++) v[k] = (u32)(T)op(v[k - 1], v[k]);
    u32 total;
    const u32 before = (u32)(T)op(carry, workgroupExclusive<u32, Op>(v[15], op, s_wave, total));
#pragma unroll
    for (int k = 0; k < 16; k++) if (i0 + k < nPart) partial[i0 + k] = (k == 0) ? (T)before : (T)op(before, v[k - 1]);
    carry = (u32)(T)op(carry, total);
  }
}

template<class T, class Op>
__global__ void __launch_bounds__(256) k_gscan_add(T* __restrict__ data, u32 n, const T* __restrict__ partial)
{
  const Op op;
  const u32 i0 = blockIdx.x * 1024u + threadIdx.x * 4u;
  const T before = partial[blockIdx.x];
  for (int k = 0; k < 4; k++) if (i0 + k < n) data[i0 + k] = (T)op(before, data[i0 + k]);
}

template<class T, class Op>
void inclusiveScan(T* data, u32 n, T* scratch, hipStream_t st)
{
  if (n == 0) return;
  const u32 nPart = (n + 1023u) / 1024u;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gscan_local<T, Op>), dim3(nPart), dim3(256), 0, st, data, n, scratch);
  if (nPart == 1) return;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gscan_partials<T, Op>), dim3(1), dim3(256), 0, st, scratch, nPart);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gscan_add<T, Op>), dim3(nPart), dim3(256), 0, st, data, n, (const T*)scratch);
}

// ------------------------------------------------------------------------------------------------
// PackBits.  The reference's greedy scan (fpl_EsriHuffman.cpp:79-236) cuts every run of equal bytes, counted from
// the run's first byte, into tokens of 129 and a remainder: 2 .. 128 bytes make one more run token, a single byte is
// a literal; literals that touch are grouped by 128 behind a count byte.  Everything a position emits therefore
// follows from where its run starts and where its stretch of literals starts -- two max-scans -- and the stream
// offsets are a sum-scan.
// Nothing per position is stored: a workgroup takes 4096 bytes (16 per thread) and recomputes what it needs from the
// bytes and three carries per workgroup -- pass 1 finds where the last run begins, pass 2 (with the run carries) where
// the last stretch of literals begins, pass 3 (with both) the bytes emitted; pass 4 writes the stream.  The plane is
// read four times (67 MB each for 8192^2) instead of 5 GB of u32 tables moved.
// ------------------------------------------------------------------------------------------------
static const u32 kPbChunk = 4096;

template<class Op>
__global__ void __launch_bounds__(256) k_pb_carries(u32* __restrict__ partial, u32 nPart)    // exclusive scan in place, total behind
{
  __shared__ u32 s_wave[4];
  const Op op;
  u32 carry = 0;
  for (u32 base = 0; base < nPart; base += 1024u)
  {
    const u32 i0 = base + threadIdx.x * 4u;
    u32 v[4], inc[4];
    for (int k = 0; k < 4; k++) v[k] = (i0 + k < nPart) ? partial[i0 + k] : 0u;
    inc[0] = v[0];
    for (int k = 1; k < 4; k++) inc[k] = op(inc[k - 1], v[k]);
    u32 total;
    const u32 before = op(carry, workgroupExclusive<u32, Op>(inc[3], op, s_wave, total));
    for (int k = 0; k < 4; k++) if (i0 + k < nPart) partial[i0 + k] = (k == 0) ? before : op(before, inc[k - 1]);
    carry = op(carry, total);
  }
  if (threadIdx.x == 0) partial[nPart] = carry;
}

// PASS 1: carry1[wg] = start of the last run that begins in this chunk (0: none);  PASS 2: carry2[wg] = start of the last
// stretch of literals that begins in it;  PASS 3: carry3[wg] = bytes it emits;  PASS 4: the stream.  Passes > 1 read the
// carries of the passes before, scanned (exclusive) by k_pb_carries.
template<int PASS>
__global__ void __launch_bounds__(256) k_pb_pass(const u8* __restrict__ s, u32 n, u32* __restrict__ carry1, u32* __restrict__ carry2,
                                                 u32* __restrict__ carry3, u8* __restrict__ out)
{
  __shared__ u32 s_wave[4];
  const u32 i0 = blockIdx.x * kPbChunk + threadIdx.x * 16u;
  // b[k] = byte i0 + k, k = 0 .. 17; 256: behind the plane's end (differs from every byte); prev = byte i0 - 1
  u32 b[18], prev = 256u;
  if (i0 + 16u <= n && (((size_t)s) & 15u) == 0u)
  {
    const uint4 q = *reinterpret_cast<const uint4*>(s + i0);
    const u32 w[4] = { q.x, q.y, q.z, q.w };
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = (w[k >> 2] >> (8 * (k & 3))) & 255u;
  }
  else
  {
#pragma unroll
    for (int k = 0; k < 16; k++) b[k] = (i0 + k < n) ? (u32)s[i0 + k] : 256u;
  }
  b[16] = (i0 + 16u < n) ? (u32)s[i0 + 16u] : 256u;
  b[17] = (i0 + 17u < n) ? (u32)s[i0 + 17u] : 256u;
  if (i0 > 0u && i0 < n) prev = s[i0 - 1u];

  // where the run a position lies in starts
  u32 local = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) if (i0 + k < n && b[k] != (k ? b[k - 1] : prev)) local = i0 + k;
  u32 total;
  u32 runBefore = workgroupExclusive<u32, ScanMax>(local, ScanMax(), s_wave, total);    // = run start of position i0 - 1
  if (PASS == 1) { if (threadIdx.x == 0) carry1[blockIdx.x] = total; return; }
  runBefore = max(runBefore, carry1[blockIdx.x]);
  u32 rem[16];    // (length of the run up to and including the position) mod 129
  {
    u32 rs = runBefore;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
      if (b[k] != (k ? b[k - 1] : prev)) rs = i0 + k;
      rem[k] = (i0 + k - rs + 1u) % 129u;
    }
  }
  // literal: a run's last byte that is left over after the tokens of 129;  where the stretch of literals it lies in starts
  u32 litMask = 0;
#pragma unroll
  for (int k = 0; k < 16; k++) if (i0 + k < n && b[k] != b[k + 1] && rem[k] == 1u) litMask |= 1u << k;
  const bool litPrev = i0 > 0u && i0 < n && prev != b[0] && ((i0 - runBefore) % 129u) == 1u;
  local = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
    if (((litMask >> k) & 1u) && i0 + k > 0u && !(k ? ((litMask >> (k - 1)) & 1u) != 0u : litPrev)) local = i0 + k;
  u32 litBefore = workgroupExclusive<u32, ScanMax>(local, ScanMax(), s_wave, total);
  if (PASS == 2) { if (threadIdx.x == 0) carry2[blockIdx.x] = total; return; }
  litBefore = max(litBefore, carry2[blockIdx.x]);
  u32 inGroup[16];    // literals: place in their group of 128
  {
    u32 ls = litBefore;
#pragma unroll
    for (int k = 0; k < 16; k++)
    {
      const bool lit = (litMask >> k) & 1u;
      if (lit && i0 + k > 0u && !(k ? ((litMask >> (k - 1)) & 1u) != 0u : litPrev)) ls = i0 + k;
      inGroup[k] = lit ? (i0 + k - ls) % 128u : 0u;
    }
  }
  // bytes emitted
  u32 cost = 0;
#pragma unroll
  for (int k = 0; k < 16; k++)
  {
    if (i0 + k >= n) continue;
    if (rem[k] == 0u) cost += 2u;
    else if (b[k] != b[k + 1]) cost += (rem[k] >= 2u) ? 2u : (inGroup[k] == 0u ? 2u : 1u);
  }
  u32 at = workgroupExclusive<u32, ScanSum>(cost, ScanSum(), s_wave, total);
  if (PASS == 3) { if (threadIdx.x == 0) carry3[blockIdx.x] = total; return; }
  at += carry3[blockIdx.x];
#pragma unroll
  for (int k = 0; k < 16; k++)
  {
    if (i0 + k >= n) continue;
    const u8 v = (u8)b[k];
    if (rem[k] == 0u) { out[at] = 255; out[at + 1u] = v; at += 2u; continue; }    // 127 + 128 repeats
    if (b[k] == b[k + 1]) continue;
    if (rem[k] >= 2u) { out[at] = (u8)(126u + rem[k]); out[at + 1u] = v; at += 2u; continue; }
    // a literal; the group's count byte sits in front of the group's first literal and is written by its last one
    const u32 pos = at + (inGroup[k] == 0u ? 1u : 0u);
    out[pos] = v;
    at = pos + 1u;
    const bool nextIsLiteral = i0 + k + 1u < n && b[k + 1] != b[k + 2];    // (it begins a run: length 1)
    if (inGroup[k] == 127u || !nextIsLiteral) out[pos - inGroup[k] - 1u] = (u8)inGroup[k];
  }
}

// decode.  Where a token starts depends on every token before it, but a token is at most 129 bytes long: the first
// token start at or behind any stream position p lies in [p, p + 128].  So the stream is cut into segments of 512
// bytes, and for each segment and EACH of the 129 possible entries a lane walks to the segment's end: exit (= entry
// into the next segment) and bytes produced.  Entry -> exit is a function on 129 states; functions compose, so groups
// of 256 segments are folded by 129 lanes, the few groups by one lane, and then every segment knows its true entry.
// States 254 / 255: the stream ended exactly on a token boundary / the walk ran past the stream's end (damaged).
static const u32 kPbSeg = 512, kPbStates = 129, kPbGroup = 256;
static const u32 kPbEnd = 254, kPbBad = 255;

__device__ __forceinline__ u32 pbConsumed(u32 b) { return (b <= 127u) ? b + 2u : 2u; }
__device__ __forceinline__ u32 pbProduced(u32 b) { return (b <= 127u) ? b + 1u : b - 126u; }

__global__ void __launch_bounds__(256) k_pb_seg_tables(const u8* __restrict__ in, u32 n, u8* __restrict__ exitT, u32* __restrict__ outT)
{
  __shared__ u8 s_buf[kPbSeg];
  const u32 seg = blockIdx.x, base = seg * kPbSeg;
  for (u32 i = threadIdx.x; i < kPbSeg; i += 256u) s_buf[i] = (base + i < n) ? in[base + i] : (u8)0;
  __syncthreads();
  const u32 o = threadIdx.x;
  if (o >= kPbStates) return;
  u32 pos = o, out = 0, state = 0;
  if (base + pos > n) state = kPbBad;
  while (state == 0u && pos < kPbSeg && base + pos < n)
  {
    const u32 b = s_buf[pos];
    if (base + pos + pbConsumed(b) > n) { state = kPbBad; break; }
    out += pbProduced(b);
    pos += pbConsumed(b);
  }
  if (state == 0u) state = (base + pos == n) ? kPbEnd : pos - kPbSeg;
  exitT[(size_t)seg * kPbStates + o] = (u8)state;
  outT[(size_t)seg * kPbStates + o] = out;
}

__device__ __forceinline__ u32 pbNext(const u8* table, u32 state) { return state < kPbStates ? (u32)table[state] : state; }

__global__ void __launch_bounds__(256) k_pb_group_compose(const u8* __restrict__ exitT, u32 nSeg, u8* __restrict__ groupExit)
{
  __shared__ u8 s_t[kPbGroup * kPbStates];
  const u32 s0 = blockIdx.x * kPbGroup, cnt = min(kPbGroup, nSeg - s0);
  for (u32 i = threadIdx.x; i < cnt * kPbStates; i += 256u) s_t[i] = exitT[(size_t)s0 * kPbStates + i];
  __syncthreads();
  if (threadIdx.x >= kPbStates) return;
  u32 cur = threadIdx.x;
  for (u32 s = 0; s < cnt; s++) cur = pbNext(s_t + s * kPbStates, cur);
  groupExit[(size_t)blockIdx.x * kPbStates + threadIdx.x] = (u8)cur;
}

__global__ void __launch_bounds__(64) k_pb_group_entries(const u8* __restrict__ groupExit, u32 nGroups, u8* __restrict__ groupEntry)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  u32 cur = 0;
  for (u32 g = 0; g < nGroups; g++) { groupEntry[g] = (u8)cur; cur = pbNext(groupExit + (size_t)g * kPbStates, cur); }
  groupEntry[nGroups] = (u8)cur;    // state behind the last segment: kPbEnd for an intact stream
}

__global__ void __launch_bounds__(256) k_pb_seg_entries(const u8* __restrict__ exitT, const u32* __restrict__ outT, u32 nSeg,
                                                         const u8* __restrict__ groupEntry, u8* __restrict__ entry, u32* __restrict__ segOut)
{
  __shared__ u8 s_t[kPbGroup * kPbStates];
  const u32 s0 = blockIdx.x * kPbGroup, cnt = min(kPbGroup, nSeg - s0);
  for (u32 i = threadIdx.x; i < cnt * kPbStates; i += 256u) s_t[i] = exitT[(size_t)s0 * kPbStates + i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  u32 cur = groupEntry[blockIdx.x];
  for (u32 s = 0; s < cnt; s++)
  {
    entry[s0 + s] = (u8)cur;
    segOut[s0 + s] = (cur < kPbStates) ? outT[(size_t)(s0 + s) * kPbStates + cur] : 0u;
    cur = pbNext(s_t + s * kPbStates, cur);
  }
}

__global__ void __launch_bounds__(64) k_pb_verdict(const u8* __restrict__ groupEntry, u32 nGroups, const u32* __restrict__ segOut, u32 nSeg,
                                                   u32 expected, u32* __restrict__ result)
{
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  result[0] = segOut[nSeg - 1u];    // inclusive sums: bytes the whole stream produces
  result[1] = (groupEntry[nGroups] == kPbEnd && segOut[nSeg - 1u] == expected) ? 1u : 0u;
}

__global__ void __launch_bounds__(256) k_pb_expand(const u8* __restrict__ in, u32 n, const u8* __restrict__ entry, const u32* __restrict__ segOut,
                                                   const u32* __restrict__ result, u8* __restrict__ out)
{
  __shared__ u8 s_buf[kPbSeg + 132];
  __shared__ u32 s_src[kPbSeg / 2 + 2], s_dst[kPbSeg / 2 + 2], s_count;
  const u32 seg = blockIdx.x, base = seg * kPbSeg;
  for (u32 i = threadIdx.x; i < kPbSeg + 132u; i += 256u) s_buf[i] = (base + i < n) ? in[base + i] : (u8)0;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    u32 k = 0;
    const u32 e = entry[seg];
    if (result[1] == 1u && e < kPbStates)
    {
      u32 pos = e, dst = (seg > 0) ? segOut[seg - 1u] : 0u;    // segOut[] holds the inclusive sums
      while (pos < kPbSeg && base + pos < n)
      {
        const u32 b = s_buf[pos];
        s_src[k] = pos; s_dst[k] = dst; k++;
        dst += pbProduced(b);
        pos += pbConsumed(b);
      }
    }
    s_count = k;
  }
  __syncthreads();
  const u32 t = threadIdx.x;
  if (t >= s_count) return;
  const u32 pos = s_src[t], dst = s_dst[t], b = s_buf[pos];
  if (b <= 127u) for (u32 j = 0; j <= b; j++) out[dst + j] = s_buf[pos + 1u + j];
  else { const u8 v = s_buf[pos + 1u]; for (u32 j = 0; j < b - 126u; j++) out[dst + j] = v; }
}

// ------------------------------------------------------------------------------------------------
// decode: planes -> units, predictor undone by prefix sums
// ------------------------------------------------------------------------------------------------
struct FplByteIndex { int idx[8]; };

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_gather(const u8* __restrict__ planes, i64 planeStride, FplByteIndex bi, i64 nElem, int finish, typename UnitOf<U>::T* __restrict__ out)
{
  typedef typename UnitOf<U>::T T;
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i >= nElem) return;
  T x = 0;
  for (int b = 0; b < U; b++) x |= (T)planes[(size_t)b * planeStride + i] << (8 * bi.idx[b]);
  out[i] = finish ? fplBackward(x) : x;
}

// column sums in three steps: totals of row segments, their running totals down the column, the segments themselves
template<int U>
__global__ void __launch_bounds__(256)
k_fpl_col_partial(const typename UnitOf<U>::T* __restrict__ x, i64 cols, i64 rows, u32 segRows, u32 nSeg, typename UnitOf<U>::T* __restrict__ partial)
{
  typedef typename UnitOf<U>::T T;
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)nSeg * cols) return;
  const i64 seg = t / cols, c = t - seg * cols;
  const i64 r0 = seg * segRows, r1 = (r0 + segRows < rows) ? r0 + segRows : rows;
  T sum = 0;
  for (i64 r = r0; r < r1; r++) sum = fplAdd(sum, x[r * cols + c]);
  partial[t] = sum;
}

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_col_scan(typename UnitOf<U>::T* __restrict__ partial, i64 cols, u32 nSeg)
{
  typedef typename UnitOf<U>::T T;
  const i64 c = (i64)blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  T run = 0;
  for (u32 s = 0; s < nSeg; s++) { const T v = partial[(size_t)s * cols + c]; partial[(size_t)s * cols + c] = run; run = fplAdd(run, v); }
}

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_col_apply(typename UnitOf<U>::T* __restrict__ x, i64 cols, i64 rows, u32 segRows, u32 nSeg, const typename UnitOf<U>::T* __restrict__ partial)
{
  typedef typename UnitOf<U>::T T;
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)nSeg * cols) return;
  const i64 seg = t / cols, c = t - seg * cols;
  const i64 r0 = seg * segRows, r1 = (r0 + segRows < rows) ? r0 + segRows : rows;
  T run = partial[t];
  for (i64 r = r0; r < r1; r++) { run = fplAdd(run, x[r * cols + c]); x[r * cols + c] = run; }
}

struct ScanFpl { template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return fplAdd(a, b); } };

// row sums, one workgroup per row; the result goes back to IEEE bit order
template<int U>
__global__ void __launch_bounds__(256)
k_fpl_row_sums_wide(typename UnitOf<U>::T* __restrict__ x, i64 cols)
{
  typedef typename UnitOf<U>::T T;
  __shared__ T s_wave[4];
  const ScanFpl op;
  T* row = x + (size_t)blockIdx.x * cols;
  T carry = 0;
  for (i64 base = 0; base < cols; base += 1024)
  {
    const i64 i0 = base + threadIdx.x * 4;
    T v[4];
    for (int k = 0; k < 4; k++) v[k] = (i0 + k < cols) ? row[i0 + k] : (T)0;
    for (int k = 1; k < 4; k++) v[k] = fplAdd(v[k - 1], v[k]);
    T total;
    const T before = fplAdd(carry, workgroupExclusive<T, ScanFpl>(v[3], op, s_wave, total));
    for (int k = 0; k < 4; k++) if (i0 + k < cols) row[i0 + k] = fplBackward(fplAdd(before, v[k]));
    carry = fplAdd(carry, total);
  }
}

template<int U>
__global__ void __launch_bounds__(256)
k_fpl_row_sums_narrow(typename UnitOf<U>::T* __restrict__ x, i64 cols, i64 rows)
{
  typedef typename UnitOf<U>::T T;
  const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  T* row = x + (size_t)r * cols;
  T run = 0;
  for (i64 c = 0; c < cols; c++) { run = fplAdd(run, row[c]); row[c] = fplBackward(run); }
}

unsigned gridFor(i64 n, int per) { return (unsigned)((n + per - 1) / per); }

}    // namespace

// ------------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------------
void launchFplPredictorSamples(const void* data, const u8* byteMask, const FplGeom& g, const FplSpan* blocks, u32 nBlocks, u32* histos,
                               hipStream_t st)
{
  if (g.unit == 4) hipLaunchKernelGGL(k_fpl_predictor_samples<4>, dim3(nBlocks), dim3(256), 0, st, data, byteMask, g, blocks, histos);
  else hipLaunchKernelGGL(k_fpl_predictor_samples<8>, dim3(nBlocks), dim3(256), 0, st, data, byteMask, g, blocks, histos);
}

void launchFplPredict(const void* data, const u8* byteMask, const FplGeom& g, int predictor, void* units, hipStream_t st)
{
  if (g.cols % 4 == 0 && (((size_t)data | (size_t)units) & 15u) == 0)
  {
    const dim3 grid(gridFor(g.nElem, 1024));
    if (g.unit == 4) hipLaunchKernelGGL(k_fpl_predict_quads<4>, grid, dim3(256), 0, st, data, byteMask, g, predictor, (u32*)units);
    else hipLaunchKernelGGL(k_fpl_predict_quads<8>, grid, dim3(256), 0, st, data, byteMask, g, predictor, (u64*)units);
    return;
  }
  const dim3 grid(gridFor(g.nElem, 256));
  if (g.unit == 4) hipLaunchKernelGGL(k_fpl_predict<4>, grid, dim3(256), 0, st, data, byteMask, g, predictor, (u32*)units);
  else hipLaunchKernelGGL(k_fpl_predict<8>, grid, dim3(256), 0, st, data, byteMask, g, predictor, (u64*)units);
}

void launchFplLevelSamples(const void* units, const FplGeom& g, const FplSpan* snippets, u32 nSnippets, u32* histos, hipStream_t st)
{
  if (nSnippets == 0) return;
  const dim3 grid(nSnippets, (unsigned)g.unit);
  if (g.unit == 4) hipLaunchKernelGGL(k_fpl_level_samples<4>, grid, dim3(256), 0, st, (const u32*)units, g, snippets, nSnippets, histos);
  else hipLaunchKernelGGL(k_fpl_level_samples<8>, grid, dim3(256), 0, st, (const u64*)units, g, snippets, nSnippets, histos);
}

void launchFplSymbols(const void* units, const FplGeom& g, const FplLevels& lv, u8* planes, u32* histos, hipStream_t st)
{
  const dim3 grid(gridFor(g.nElem, 1024 * kFplSymbolTiles));
  const i64 stride = fplPlaneStride(g.nElem);
  if (g.unit == 4) hipLaunchKernelGGL(k_fpl_symbols<4>, grid, dim3(256), 0, st, (const u32*)units, g, lv, stride, planes, histos);
  else hipLaunchKernelGGL(k_fpl_symbols<8>, grid, dim3(256), 0, st, (const u64*)units, g, lv, stride, planes, histos);
}

size_t packBitsCarryCount(u32 n) { return (size_t)(n + kPbChunk - 1u) / kPbChunk + 8; }

void launchPackBitsPlan(const u8* plane, u32 n, const PackBitsBuffers& b, hipStream_t st)
{
  const u32 nWG = (n + kPbChunk - 1u) / kPbChunk;
  const dim3 grid(nWG), block(256);
  hipLaunchKernelGGL(k_pb_pass<1>, grid, block, 0, st, plane, n, b.carry1, b.carry2, b.carry3, (u8*)nullptr);
  hipLaunchKernelGGL(k_pb_carries<ScanMax>, dim3(1), block, 0, st, b.carry1, nWG);
  hipLaunchKernelGGL(k_pb_pass<2>, grid, block, 0, st, plane, n, b.carry1, b.carry2, b.carry3, (u8*)nullptr);
  hipLaunchKernelGGL(k_pb_carries<ScanMax>, dim3(1), block, 0, st, b.carry2, nWG);
  hipLaunchKernelGGL(k_pb_pass<3>, grid, block, 0, st, plane, n, b.carry1, b.carry2, b.carry3, (u8*)nullptr);
  hipLaunchKernelGGL(k_pb_carries<ScanSum>, dim3(1), block, 0, st, b.carry3, nWG);
}

const u32* packBitsSize(const PackBitsBuffers& b, u32 n) { return b.carry3 + (n + kPbChunk - 1u) / kPbChunk; }

void launchPackBitsEmit(const u8* plane, u32 n, const PackBitsBuffers& b, u8* out, hipStream_t st)
{
  hipLaunchKernelGGL(k_pb_pass<4>, dim3((n + kPbChunk - 1u) / kPbChunk), dim3(256), 0, st, plane, n, b.carry1, b.carry2, b.carry3, out);
}

void launchFplEntropy(const u32* histos, u32 nHist, const double* log2Tables, const FplEntropyTables& tb, i64* out, hipStream_t st)
{
  if (nHist) hipLaunchKernelGGL(k_fpl_entropy, dim3((nHist + 3u) / 4u), dim3(256), 0, st, histos, nHist, log2Tables, tb, out);
}

size_t packBitsDecodeScratchBytes(u32 n)
{
  const size_t nSeg = ((size_t)n + kPbSeg - 1) / kPbSeg, nGroups = (nSeg + kPbGroup - 1) / kPbGroup;
  return nSeg * kPbStates * 5 + nSeg * 5 + nGroups * (kPbStates + 1) + (nSeg / 1024 + 16) * 4 + 8192;
}

// result[0] = bytes the stream produces, result[1] = 1 if the stream is intact and produces exactly `expected` bytes
// (only then is anything written to out)
bool launchPackBitsDecode(const u8* in, u32 n, u32 expected, u8* scratch, u32* result, u8* out, hipStream_t st)
{
  if (n == 0) return false;
  const u32 nSeg = (n + kPbSeg - 1u) / kPbSeg, nGroups = (nSeg + kPbGroup - 1u) / kPbGroup;
  auto take = [&](size_t bytes) { u8* p = scratch; scratch += (bytes + 255) & ~(size_t)255; return p; };
  u32* outT = (u32*)take((size_t)nSeg * kPbStates * 4);
  u32* segOut = (u32*)take((size_t)nSeg * 4 + 16);
  u32* scanScratch = (u32*)take(((size_t)nSeg / 1024 + 16) * 4);
  u8* exitT = take((size_t)nSeg * kPbStates);
  u8* entry = take(nSeg);
  u8* groupExit = take((size_t)nGroups * kPbStates);
  u8* groupEntry = take(nGroups + 1);
  hipLaunchKernelGGL(k_pb_seg_tables, dim3(nSeg), dim3(256), 0, st, in, n, exitT, outT);
  hipLaunchKernelGGL(k_pb_group_compose, dim3(nGroups), dim3(256), 0, st, (const u8*)exitT, nSeg, groupExit);
  hipLaunchKernelGGL(k_pb_group_entries, dim3(1), dim3(64), 0, st, (const u8*)groupExit, nGroups, groupEntry);
  hipLaunchKernelGGL(k_pb_seg_entries, dim3(nGroups), dim3(256), 0, st, (const u8*)exitT, (const u32*)outT, nSeg, (const u8*)groupEntry, entry, segOut);
  inclusiveScan<u32, ScanSum>(segOut, nSeg, scanScratch, st);
  hipLaunchKernelGGL(k_pb_verdict, dim3(1), dim3(64), 0, st, (const u8*)groupEntry, nGroups, (const u32*)segOut, nSeg, expected, result);
  hipLaunchKernelGGL(k_pb_expand, dim3(nSeg), dim3(256), 0, st, in, n, (const u8*)entry, (const u32*)segOut, (const u32*)result, out);
  return true;
}

// Running sums of bytes (restoreSequence's inner loop, fpl_Lerc2Ext.cpp:128-165) over a plane of 67 MB: the general scan
// above takes four BYTES a thread (65 000 workgroups of 1 KiB, byte loads and stores: 205 us a pass on an 8192^2 band).  Here a
// thread takes the 16 bytes of one aligned unit -- the plane starts anywhere, so the first and the last unit are partly
// someone else's and go byte by byte -- and the second pass adds the carry to four bytes at a time.
static const u32 kByteSumChunk = 4096;

__device__ __forceinline__ u32 addBytes(u32 x, u32 c4)    // four byte-wise sums mod 256
{
  return ((x & 0x7F7F7F7Fu) + (c4 & 0x7F7F7F7Fu)) ^ ((x ^ c4) & 0x80808080u);
}

// unit u (16 bytes at base + 16 u, base 16-byte aligned) holds plane bytes 16 u - head ... ; valid: those in [0, n)
__global__ void __launch_bounds__(256) k_bytesum_local(u8* __restrict__ base, u32 head, u32 n, u8* __restrict__ partial)
{
  __shared__ u32 s_wave[4];
  const ScanSum op;
  const u32 u = blockIdx.x * 256u + threadIdx.x;
  const i64 first = (i64)16 * u - head;    // plane index of the unit's first byte
  const bool any = first < (i64)n && first + 16 > 0;
  const bool whole = first >= 0 && first + 16 <= (i64)n;
  u32 w[4] = { 0, 0, 0, 0 };
  if (whole) { const uint4 x = reinterpret_cast<const uint4*>(base)[u]; w[0] = x.x; w[1] = x.y; w[2] = x.z; w[3] = x.w; }
  else if (any)
    for (int k = 0; k < 16; k++) if (first + k >= 0 && first + k < (i64)n) w[k >> 2] |= (u32)base[16ull * u + k] << (8 * (k & 3));
  // running sums inside the unit: byte k += byte k - 1, ..., as three doubling steps per word and the words' totals
  u32 run = 0;
#pragma unroll
  for (int q = 0; q < 4; q++)
  {
    u32 x = w[q];
    x = addBytes(x, x << 8);
    x = addBytes(x, x << 16);
    x = addBytes(x, run * 0x01010101u);
    w[q] = x;
    run = x >> 24;
  }
  u32 total;
  const u32 before = workgroupExclusive<u32, ScanSum>(run, op, s_wave, total) & 0xFFu;
  const u32 c4 = before * 0x01010101u;
#pragma unroll
  for (int q = 0; q < 4; q++) w[q] = addBytes(w[q], c4);
  if (whole) reinterpret_cast<uint4*>(base)[u] = make_uint4(w[0], w[1], w[2], w[3]);
  else if (any)
    for (int k = 0; k < 16; k++) if (first + k >= 0 && first + k < (i64)n) base[16ull * u + k] = (u8)(w[k >> 2] >> (8 * (k & 3)));
  if (threadIdx.x == 0) partial[blockIdx.x] = (u8)total;
}

__global__ void __launch_bounds__(256) k_bytesum_add(u8* __restrict__ base, u32 head, u32 n, const u8* __restrict__ partial)
{
  const u32 u = blockIdx.x * 256u + threadIdx.x;
  const u32 c = partial[blockIdx.x];
  if (c == 0u) return;    // (the same for the whole workgroup)
  const i64 first = (i64)16 * u - head;
  if (first >= 0 && first + 16 <= (i64)n)
  {
    const u32 c4 = c * 0x01010101u;
    uint4 x = reinterpret_cast<const uint4*>(base)[u];
    x.x = addBytes(x.x, c4); x.y = addBytes(x.y, c4); x.z = addBytes(x.z, c4); x.w = addBytes(x.w, c4);
    reinterpret_cast<uint4*>(base)[u] = x;
  }
  else if (first < (i64)n && first + 16 > 0)
    for (int k = 0; k < 16; k++) if (first + k >= 0 && first + k < (i64)n) base[16ull * u + k] = (u8)(base[16ull * u + k] + c);
}

void launchBytePrefixSum(u8* p, u32 n, u32* scratch, hipStream_t st)
{
  if (n == 0) return;
  const u32 head = (u32)((uintptr_t)p & 15u);
  u8* base = p - head;
  const u32 nUnits = (head + n + 15u) / 16u;
  const u32 nPart = (nUnits + 255u) / 256u;
  hipLaunchKernelGGL(k_bytesum_local, dim3(nPart), dim3(256), 0, st, base, head, n, (u8*)scratch);
  if (nPart == 1) return;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_gscan_partials<u8, ScanSum>), dim3(1), dim3(256), 0, st, (u8*)scratch, nPart);
  hipLaunchKernelGGL(k_bytesum_add, dim3(nPart), dim3(256), 0, st, base, head, n, (const u8*)scratch);
}

void launchFplGather(const u8* planes, const int* byteIndex, const FplGeom& g, bool finish, void* out, hipStream_t st)
{
  FplByteIndex bi;
  for (int b = 0; b < 8; b++) bi.idx[b] = (b < g.unit) ? byteIndex[b] : 0;
  const dim3 grid(gridFor(g.nElem, 256));
  const i64 stride = fplPlaneStride(g.nElem);
  if (g.unit == 4) hipLaunchKernelGGL(k_fpl_gather<4>, grid, dim3(256), 0, st, planes, stride, bi, g.nElem, finish ? 1 : 0, (u32*)out);
  else hipLaunchKernelGGL(k_fpl_gather<8>, grid, dim3(256), 0, st, planes, stride, bi, g.nElem, finish ? 1 : 0, (u64*)out);
}

u32 fplColumnSegments(i64 rows)
{
  const i64 segRows = fplColumnSegmentRows(rows);
  return (u32)((rows + segRows - 1) / segRows);
}

void launchFplColumnSums(void* units, const FplGeom& g, void* partial, u32 nSeg, hipStream_t st)
{
  const u32 segRows = (u32)fplColumnSegmentRows(g.rows);
  const dim3 grid(gridFor((i64)nSeg * g.cols, 256)), gridC(gridFor(g.cols, 256));
  if (g.unit == 4)
  {
    hipLaunchKernelGGL(k_fpl_col_partial<4>, grid, dim3(256), 0, st, (const u32*)units, g.cols, g.rows, segRows, nSeg, (u32*)partial);
    hipLaunchKernelGGL(k_fpl_col_scan<4>, gridC, dim3(256), 0, st, (u32*)partial, g.cols, nSeg);
    hipLaunchKernelGGL(k_fpl_col_apply<4>, grid, dim3(256), 0, st, (u32*)units, g.cols, g.rows, segRows, nSeg, (const u32*)partial);
  }
  else
  {
    hipLaunchKernelGGL(k_fpl_col_partial<8>, grid, dim3(256), 0, st, (const u64*)units, g.cols, g.rows, segRows, nSeg, (u64*)partial);
    hipLaunchKernelGGL(k_fpl_col_scan<8>, gridC, dim3(256), 0, st, (u64*)partial, g.cols, nSeg);
    hipLaunchKernelGGL(k_fpl_col_apply<8>, grid, dim3(256), 0, st, (u64*)units, g.cols, g.rows, segRows, nSeg, (const u64*)partial);
  }
}

void launchFplRowSums(void* units, const FplGeom& g, hipStream_t st)
{
  if (g.cols > 32)
  {
    if (g.unit == 4) hipLaunchKernelGGL(k_fpl_row_sums_wide<4>, dim3((unsigned)g.rows), dim3(256), 0, st, (u32*)units, g.cols);
    else hipLaunchKernelGGL(k_fpl_row_sums_wide<8>, dim3((unsigned)g.rows), dim3(256), 0, st, (u64*)units, g.cols);
  }
  else
  {
    const dim3 grid(gridFor(g.rows, 256));
    if (g.unit == 4) hipLaunchKernelGGL(k_fpl_row_sums_narrow<4>, grid, dim3(256), 0, st, (u32*)units, g.cols, g.rows);
    else hipLaunchKernelGGL(k_fpl_row_sums_narrow<8>, grid, dim3(256), 0, st, (u64*)units, g.cols, g.rows);
  }
}

}    // namespace lerc
