// huffman_kernels.hip -- device side of the 8-bit Huffman image mode.
//
// Reference counterparts: Lerc2::ComputeHistoForHuffman (Lerc2.cpp:2311-2380), EncodeHuffman
// (:2384-2468) with Huffman::PushValue (Huffman.h:218-255), DecodeHuffman (:2472-2606) with
// Huffman::DecodeOneValue (Huffman.h:144-214).
//
// The pixel stream is one long bit string without sync markers.  Encode: every element of the
// (virtual, mask-agnostic) stream order gets its code length, an exclusive scan over runs of elements
// gives bit offsets, and each thread ORs its run's codes into the stream.  Decode: the stream is cut
// into sub-sequences decoded speculatively in parallel; a sub-sequence's start is corrected to its
// predecessor's exit until nothing changes (Huffman codes self-synchronise after a few symbols), then
// symbols are written by rank and the delta predictor is undone with wave-level segmented scans.
#include "huffman_dev.h"
#include "wave_utils.h"

namespace lerc {

// index of the last valid pixel before k, or -1
__device__ __forceinline__ i64 prevValidPixel(const u8* __restrict__ bits, i64 k)
{
  i64 j = k - 1;
  while (j >= 0)
  {
    if ((j & 7) == 7 && bits[j >> 3] == 0) { j -= 8; continue; }
    if (maskBit(bits, j)) return j;
    j--;
  }
  return -1;
}

// Symbol of virtual stream element v (delta mode: plane major; plain mode: pixel major), or -1 when
// the pixel is invalid.  T is signed char or unsigned char.
template<class T>
__device__ __forceinline__ int huffSymbol(const T* __restrict__ data, const u8* __restrict__ maskBits, const HuffGeom& g, int mode, i64 v)
{
  const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
  if (mode == IEM_Huffman)
  {
    const i64 k = v / g.nDepth;
    if (maskBits && !maskBit(maskBits, k)) return -1;
    return off + (int)data[v];
  }
  const i64 nPix = (i64)g.nRows * g.nCols;
  const int iD = (int)(v / nPix);
  const i64 k = v - (i64)iD * nPix;
  if (maskBits && !maskBit(maskBits, k)) return -1;
  const int i = (int)(k / g.nCols), j = (int)(k - (i64)i * g.nCols);
  const T val = data[k * g.nDepth + iD];
  T pred = 0;
  if (j > 0 && (!maskBits || maskBit(maskBits, k - 1))) pred = data[(k - 1) * g.nDepth + iD];
  else if (i > 0 && (!maskBits || maskBit(maskBits, k - g.nCols))) pred = data[(k - g.nCols) * g.nDepth + iD];
  else if (maskBits)
  {
    const i64 kp = prevValidPixel(maskBits, k);
    if (kp >= 0) pred = data[kp * g.nDepth + iD];
  }
  const T delta = (T)(val - pred);
  return off + (int)delta;
}

// The same symbol for consecutive stream elements without dividing again: where element v sits (depth plane, pixel,
// row, column) is worked out once and then stepped.
struct HuffCursor
{
  i64 v, nPix;
  i64 k;        // pixel
  int iD, i, j; // depth plane (delta mode) resp. depth index (plain mode), row, column
  int mode;
  __device__ __forceinline__ void init(const HuffGeom& g, int mode_, i64 v0)
  {
    mode = mode_; v = v0; nPix = (i64)g.nRows * g.nCols;
    if (mode == IEM_Huffman) { k = v0 / g.nDepth; iD = (int)(v0 - k * g.nDepth); }
    else { iD = (int)(v0 / nPix); k = v0 - (i64)iD * nPix; }
    i = (int)(k / g.nCols); j = (int)(k - (i64)i * g.nCols);
  }
  __device__ __forceinline__ void step(const HuffGeom& g)
  {
    v++;
    if (mode == IEM_Huffman) { if (++iD < g.nDepth) return; iD = 0; }
    k++;
    if (++j == g.nCols) { j = 0; i++; }
    if (mode != IEM_Huffman && k == nPix) { k = 0; i = 0; j = 0; iD++; }
  }
};

template<class T>
__device__ __forceinline__ int huffSymbolAt(const T* __restrict__ data, const u8* __restrict__ maskBits, const HuffGeom& g, const HuffCursor& c)
{
  const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
  if (maskBits && !maskBit(maskBits, c.k)) return -1;
  const T val = data[c.k * g.nDepth + c.iD];
  if (c.mode == IEM_Huffman) return off + (int)val;
  T pred = 0;
  if (c.j > 0 && (!maskBits || maskBit(maskBits, c.k - 1))) pred = data[(c.k - 1) * g.nDepth + c.iD];
  else if (c.i > 0 && (!maskBits || maskBit(maskBits, c.k - g.nCols))) pred = data[(c.k - g.nCols) * g.nDepth + c.iD];
  else if (maskBits)
  {
    const i64 kp = prevValidPixel(maskBits, c.k);
    if (kp >= 0) pred = data[kp * g.nDepth + c.iD];
  }
  const T delta = (T)(val - pred);
  return off + (int)delta;
}

// ---- histograms ---------------------------------------------------------------------------------
template<class T>
__global__ void __launch_bounds__(256) k_huff_histo(const T* __restrict__ data, const u8* __restrict__ maskBits, HuffGeom g, u32* __restrict__ histos)
{
  __shared__ u32 s_h[2][256];
  s_h[0][threadIdx.x] = 0; s_h[1][threadIdx.x] = 0;
  __syncthreads();
  // thread t of the grid takes pixels t, t + T, ... ; both symbols of a (pixel, depth) come from the same bytes
  const i64 nPix = (i64)g.nRows * g.nCols;
  const i64 stride = (i64)gridDim.x * 256;
  const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < nPix; k += stride)
  {
    if (maskBits && !maskBit(maskBits, k)) continue;
    const int i = (int)(k / g.nCols), j = (int)(k - (i64)i * g.nCols);
    const bool left = j > 0 && (!maskBits || maskBit(maskBits, k - 1));
    const bool up = !left && i > 0 && (!maskBits || maskBit(maskBits, k - g.nCols));
    const i64 kp = left ? k - 1 : (up ? k - g.nCols : (maskBits ? prevValidPixel(maskBits, k) : -1));
    for (int m = 0; m < g.nDepth; m++)
    {
      const T val = data[k * g.nDepth + m];
      const T pred = (kp >= 0) ? data[kp * g.nDepth + m] : (T)0;
      atomicAdd(&s_h[0][off + (int)val], 1u);
      atomicAdd(&s_h[1][off + (int)(T)(val - pred)], 1u);
    }
  }
  __syncthreads();
  if (s_h[0][threadIdx.x]) atomicAdd(&histos[threadIdx.x], s_h[0][threadIdx.x]);
  if (s_h[1][threadIdx.x]) atomicAdd(&histos[256 + threadIdx.x], s_h[1][threadIdx.x]);
}

// Every pixel valid, at most four values per pixel, rows of a multiple of 8 pixels: a thread takes 8 consecutive pixels
// (8-byte loads; the predecessors of pixels 1 .. 7 are in its registers, the one of pixel 0 comes with one more load) and
// counts into one of 16 copies of the histograms (a smooth image sends most lanes of a wave to the same few delta
// bins: same-address LDS atomics are served one after the other).
static const int kHistoCopies = 16;

template<class T, int D>
__global__ void __launch_bounds__(256) k_huff_histo_dense(const T* __restrict__ data, HuffGeom g, u32* __restrict__ histos)
{
  __shared__ u32 s_h[2][256 * kHistoCopies];
  for (int i = threadIdx.x; i < 256 * kHistoCopies; i += 256) { s_h[0][i] = 0; s_h[1][i] = 0; }
  __syncthreads();
  const u8* bytes = reinterpret_cast<const u8*>(data);
  const i64 nGroups = ((i64)g.nRows * g.nCols) >> 3;
  const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
  const u32 copy = threadIdx.x & (u32)(kHistoCopies - 1);
  for (i64 grp = (i64)blockIdx.x * 256 + threadIdx.x; grp < nGroups; grp += (i64)gridDim.x * 256)
  {
    const i64 k0 = grp << 3;
    const int i = (int)(k0 / g.nCols), j = (int)(k0 - (i64)i * g.nCols);
    u64 w[D];
#pragma unroll
    for (int x = 0; x < D; x++) w[x] = *reinterpret_cast<const u64*>(bytes + k0 * D + 8 * x);
    // the pixel in front of pixel 0: its left neighbour (the last D bytes in front of the group), in column 0 the pixel
    // above (the first D bytes of the group one row up), for the first pixel of all: 0
    const bool left = j > 0, up = !left && i > 0;
    const u64 pw = *reinterpret_cast<const u64*>(bytes + (left ? k0 * D - 8 : (up ? (k0 - g.nCols) * D : k0 * D)));
    const u64 pbits = (left || up) ? (pw >> (left ? 8 * (8 - D) : 0)) : 0ull;
#define LERC_PX(c, d) ((T)(u8)(w[((c) * D + (d)) >> 3] >> (8 * (((c) * D + (d)) & 7))))
#pragma unroll
    for (int c = 0; c < 8; c++)
#pragma unroll
      for (int m = 0; m < D; m++)
      {
        const T val = LERC_PX(c, m);
        const T pred = (c > 0) ? LERC_PX(c > 0 ? c - 1 : 0, m) : (T)(u8)(pbits >> (8 * m));
        atomicAdd(&s_h[0][(u32)(off + (int)val) * kHistoCopies + copy], 1u);
        atomicAdd(&s_h[1][(u32)(off + (int)(T)(val - pred)) * kHistoCopies + copy], 1u);
      }
#undef LERC_PX
  }
  __syncthreads();
  u32 n0 = 0, n1 = 0;
  for (int c = 0; c < kHistoCopies; c++) { n0 += s_h[0][threadIdx.x * kHistoCopies + c]; n1 += s_h[1][threadIdx.x * kHistoCopies + c]; }
  if (n0) atomicAdd(&histos[threadIdx.x], n0);
  if (n1) atomicAdd(&histos[256 + threadIdx.x], n1);
}

template<class T>
static bool launchHistoDense(const void* data, const u8* maskBits, const HuffGeom& g, u32* histos, hipStream_t st)
{
  if (maskBits || g.nDepth > 4 || (g.nCols & 7) || ((uintptr_t)data & 7)) return false;
  const i64 nGroups = ((i64)g.nRows * g.nCols) >> 3;
  i64 nb = (nGroups + 256 * 4 - 1) / (256 * 4);
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  const dim3 grid((unsigned)nb), block(256);
  switch (g.nDepth)
  {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_histo_dense<T, 1>), grid, block, 0, st, (const T*)data, g, histos); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_histo_dense<T, 2>), grid, block, 0, st, (const T*)data, g, histos); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_histo_dense<T, 3>), grid, block, 0, st, (const T*)data, g, histos); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_histo_dense<T, 4>), grid, block, 0, st, (const T*)data, g, histos); break;
  }
  return true;
}

void launchHuffHisto(int dt, const void* data, const u8* maskBits, const HuffGeom& g, u32* histos, hipStream_t st)
{
  if (dt == DT_Char ? launchHistoDense<signed char>(data, maskBits, g, histos, st) : launchHistoDense<unsigned char>(data, maskBits, g, histos, st)) return;
  const i64 n = (i64)g.nRows * g.nCols * g.nDepth;
  i64 nb = (n + 256 * 32 - 1) / (256 * 32);
  nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
  if (dt == DT_Char) hipLaunchKernelGGL(k_huff_histo<signed char>, dim3((unsigned)nb), dim3(256), 0, st, (const signed char*)data, maskBits, g, histos);
  else hipLaunchKernelGGL(k_huff_histo<unsigned char>, dim3((unsigned)nb), dim3(256), 0, st, (const unsigned char*)data, maskBits, g, histos);
}

// ---- encode -------------------------------------------------------------------------------------
// codes[s] = (length << 32) | code bits
//
// A workgroup owns 256 runs of kHuffRun stream elements.  All pixels valid: the elements are turned into symbols with the
// loads of eight steps in flight at a time (a thread walking its run straight from global memory touches a different cache
// line in every lane, and a loop that waits for each byte load spends its time on memory latency) and parked in LDS,
// element e of run r at [e][r] (rows 260 bytes apart: conflict-free both ways).  Packing assembles the workgroup's part of
// the bit stream in LDS (ds_or) and stores it in whole words -- only the first and the last word of the span are shared
// with the neighbouring workgroups and go out with an atomic OR; a span longer than the LDS area (more than ten bits per
// symbol on average) is ORed into global memory word by word as the masked path does.
static const int kHuffSpanWords = 10240;
#ifndef LERC_HUFF_STAGE_BATCH
#define LERC_HUFF_STAGE_BATCH 8
#endif
static const int kHuffStageBatch = LERC_HUFF_STAGE_BATCH;

#if defined(LERC_PROBE) && !defined(HIPSIM)
// tuning: per-workgroup time lines (constant-rate counter) of the packing kernel, read by tools/trace_huff.py
static __device__ unsigned long long g_traceH[8 * 8192];
extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_trace_huff(unsigned long long* out, int n)
{ hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_traceH), sizeof(unsigned long long) * (size_t)n); }
#define TRACEH(slot) do { if (PACK && threadIdx.x == 0 && blockIdx.x < 8192u) g_traceH[8 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define TRACEH(slot)
#endif

// ORs a word of the bit stream into memory: the stream begins `mis` bytes into the aligned word stream[0]
__device__ __forceinline__ void orStreamWord(u32* __restrict__ stream, u32 mis, u64 w, u32 v)
{
  if (mis == 0u) { atomicOr(&stream[w], v); return; }
  const u32 sh = 8u * mis;
  if (v << sh) atomicOr(&stream[w], v << sh);
  if (v >> (32u - sh)) atomicOr(&stream[w + 1], v >> (32u - sh));
}

template<class T, bool PACK, int RUN>
__global__ void __launch_bounds__(256)
k_huff_encode(const T* __restrict__ data, const u8* __restrict__ maskBits, HuffGeom g, int mode, const u64* __restrict__ codes,
              u32* __restrict__ runBits, const u64* __restrict__ runBase, u32* __restrict__ stream, u32 mis, u64* __restrict__ cells, DeviceStatus* status)
{
  __shared__ u64 s_codes[256];
  __shared__ u8 s_sym[RUN * 260];
  constexpr int kRunShift = (RUN == 128) ? 7 : 6;
  static_assert(RUN == 128 || RUN == 64, "elements per thread");
  __shared__ u32 s_span[PACK ? kHuffSpanWords * RUN / 128 : 1];
  __shared__ u64 s_wave[4], s_base;
  TRACEH(0);
  s_codes[threadIdx.x] = codes[threadIdx.x];
  const i64 n = (i64)g.nRows * g.nCols * g.nDepth;
  const i64 nRuns = (n + RUN - 1) / RUN;
  const bool staged = (maskBits == nullptr);
  // this workgroup's words of the stream
  u64 spanWord0 = 0;
  u32 spanWords = 0;
  bool inLds = false;
  const bool selfScan = PACK && cells != nullptr;    // positions from a look-back over the workgroups instead of a scan pass
  if (PACK && !selfScan)
  {
    const i64 r0 = (i64)blockIdx.x * 256, r1 = (r0 + 256 < nRuns) ? r0 + 256 : nRuns;
    const u64 bit0 = runBase[r0], bit1 = runBase[r1];
    spanWord0 = bit0 >> 5;
    spanWords = (u32)(((bit1 + 31) >> 5) - spanWord0);
    inLds = staged && spanWords <= (u32)(kHuffSpanWords * RUN / 128);
    if (inLds) for (u32 x = threadIdx.x; x < spanWords; x += 256) s_span[x] = 0u;
  }
  if (staged)
  {
    const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
    const i64 vBase = (i64)blockIdx.x * 256 * RUN, nPix = (i64)g.nRows * g.nCols;
    i64 v = vBase + threadIdx.x;
    // delta mode walks plane by plane: (plane, pixel, row, column) of this thread's first element, then 256 further each time
    i64 k = 0;
    int iD = 0, i = 0, j = 0;
    if (mode != IEM_Huffman && v < n) { iD = (int)(v / nPix); k = v - (i64)iD * nPix; i = (int)(k / g.nCols); j = (int)(k - (i64)i * g.nCols); }
    // The common case -- delta mode, all of the workgroup's elements pixels of ONE plane, rows of at least 256 pixels -- with the
    // bookkeeping in 32 bits relative to the thread's first element: the general form below spends 40 instructions an element on
    // 64-bit indices, and the staging is bound by exactly that (time lines: 27 of a workgroup's 47 us; 10 with this form).
    const i64 kFirst = (mode != IEM_Huffman) ? vBase % nPix : 0;
    const bool plain = mode != IEM_Huffman && vBase + 256 * RUN <= n && kFirst + 256 * RUN <= nPix && g.nCols >= 256
      && (i64)g.nCols * g.nDepth < (1ll << 30) && (i64)256 * RUN * g.nDepth < (1ll << 30);
    if (plain)
    {
      const int D = g.nDepth, rowBytes = g.nCols * D;
      const T* __restrict__ mine = data + (k * D + iD);    // this thread's first element
      int ii = i, jj = j;
      for (int q0 = 0; q0 < RUN; q0 += kHuffStageBatch)
      {
        int oVal[kHuffStageBatch], oPred[kHuffStageBatch], kind[kHuffStageBatch];    // byte offsets from `mine`; kind: 0 = load, 1 = the lane in front has it, 2 = none (value 0)
#pragma unroll
        for (int b = 0; b < kHuffStageBatch; b++)
        {
          oVal[b] = (q0 + b) * 256 * D;
          kind[b] = (jj > 0) ? (((threadIdx.x & 63u) != 0u) ? 1 : 0) : (ii > 0 ? 0 : 2);
          oPred[b] = oVal[b] - ((jj > 0) ? D : rowBytes);
          jj += 256;
          if (jj >= g.nCols) { jj -= g.nCols; ii++; }
        }
        T val[kHuffStageBatch], pred[kHuffStageBatch];
#pragma unroll
        for (int b = 0; b < kHuffStageBatch; b++)
        {
          val[b] = mine[oVal[b]];
          pred[b] = (T)0;
          if (kind[b] == 0) pred[b] = mine[oPred[b]];    // (first lane of a wave, first column of a row: few lanes)
        }
#pragma unroll
        for (int b = 0; b < kHuffStageBatch; b++)
        {
          const int idx = (int)threadIdx.x + 256 * (q0 + b);
          const T left = (T)__shfl_up((int)val[b], 1u);
          const T pp = kind[b] == 1 ? left : pred[b];
          s_sym[(idx & (RUN - 1)) * 260 + (idx >> kRunShift)] = (u8)(off + (int)(T)(val[b] - pp));
        }
      }
    }
    else
    for (int q0 = 0; q0 < RUN; q0 += kHuffStageBatch)
    {
      i64 aVal[kHuffStageBatch], aPred[kHuffStageBatch];    // byte offsets; -1: no such byte (value 0)
#pragma unroll
      for (int b = 0; b < kHuffStageBatch; b++, v += 256)
      {
        aVal[b] = -1; aPred[b] = -1;
        if (v >= n) continue;
        if (mode == IEM_Huffman) { aVal[b] = v; continue; }
        aVal[b] = k * g.nDepth + iD;
        if (j > 0) aPred[b] = (threadIdx.x & 63u) ? -2 : aVal[b] - g.nDepth;    // (-2: the value the lane in front has just loaded)
        else if (i > 0) aPred[b] = aVal[b] - (i64)g.nCols * g.nDepth;
        k += 256; j += 256;
        while (j >= g.nCols) { j -= g.nCols; i++; }
        if (k >= nPix) { while (k >= nPix) { k -= nPix; iD++; } i = (int)(k / g.nCols); j = (int)(k - (i64)i * g.nCols); }
      }
      T val[kHuffStageBatch], pred[kHuffStageBatch];
#pragma unroll
      for (int b = 0; b < kHuffStageBatch; b++)    // (loads of a clamped address: all of them leave before the first use)
      {
        val[b] = data[aVal[b] < 0 ? 0 : aVal[b]];
        pred[b] = (T)0;
        if (aPred[b] >= 0) pred[b] = data[aPred[b]];    // (first lane of a wave, first column of a row: few lanes)
      }
#pragma unroll
      for (int b = 0; b < kHuffStageBatch; b++)
      {
        const int idx = (int)threadIdx.x + 256 * (q0 + b);
        const T left = (T)__shfl_up((int)val[b], 1u);
        const T vv = aVal[b] < 0 ? (T)0 : val[b], pp = aPred[b] == -2 ? left : (aPred[b] < 0 ? (T)0 : pred[b]);
        const int sym = (aVal[b] < 0) ? 0 : off + (int)(T)(vv - pp);
        s_sym[(idx & (RUN - 1)) * 260 + (idx >> kRunShift)] = (u8)sym;
      }
    }
  }
  __syncthreads();
  TRACEH(1);
  const i64 run = (i64)blockIdx.x * 256 + threadIdx.x;
  const i64 v0 = run * RUN;
  const bool active = v0 < n;
  const i64 v1 = (v0 + RUN < n) ? v0 + RUN : n;
  HuffCursor cur;
  if (active && !staged) cur.init(g, mode, v0);
  if (!PACK)
  {
    if (!active) return;
    u32 bits = 0;
    if (staged) for (int e = 0; e < (int)(v1 - v0); e++) bits += (u32)(s_codes[s_sym[e * 260 + threadIdx.x]] >> 32);
    else
      for (i64 v = v0; v < v1; v++, cur.step(g))
      {
        const int s = huffSymbolAt<T>(data, maskBits, g, cur);
        if (s >= 0) bits += (u32)(s_codes[s] >> 32);
      }
    runBits[run] = bits;
    return;
  }
  // MSB-first packing into little-endian u32 words (Huffman.h:218-255): keep a 64-bit window whose top
  // bits are the oldest; whole words are ORed into the span (neighbouring runs share boundary words)
  u64 pos = 0;
  if (selfScan)
  {
    // One pass (all pixels valid): the bits of this thread's run, their prefix over the workgroup, and the workgroup's
    // place in the stream from the cells of the workgroups in front -- each publishes the bits it holds ("aggregate", tag
    // 1) as soon as it knows them and the sum up to and including itself ("inclusive", tag 2) once it has looked back far
    // enough to meet an inclusive cell.  Workgroups start in the order of their index, so the ones looked at are running
    // or done; the cells are zeroed by the host before the launch.
    u64 mine = 0;
    if (active) for (int e = 0; e < (int)(v1 - v0); e++) mine += (u32)(s_codes[s_sym[e * 260 + threadIdx.x]] >> 32);
    u64 inc = mine;
    const int lane = laneId(), wv = waveId();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += t; }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    TRACEH(2);
    u64 before = 0;
    for (int k = 0; k < wv; k++) before += s_wave[k];
    const u64 total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    if (wv == 0)
    {
      const u64 kTag = 62, kValue = (1ull << 62) - 1ull;
      const i64 b = (i64)blockIdx.x;
      if (lane == 0) publish64(&cells[b], ((b == 0 ? 2ull : 1ull) << kTag) | total);
      u64 excl = 0;
      u32 spins = 0;
      for (i64 j = b - 1; j >= 0; )
      {
        const u64 c = (j - lane >= 0) ? observe64(&cells[j - lane]) : (2ull << kTag);    // (in front of workgroup 0: nothing)
        const u32 tag = (u32)(c >> kTag);
        const u64 empty = __ballot(tag == 0u), incl = __ballot(tag == 2u);
        const int firstIncl = incl ? __ffsll((long long)incl) - 1 : 64, firstEmpty = empty ? __ffsll((long long)empty) - 1 : 64;
        if (firstEmpty < firstIncl)
        {
          if (++spins > (1u << 24)) { if (lane == 0) raiseError(status, kFailed, (u32)b); break; }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        excl += waveSum((lane <= firstIncl) ? (c & kValue) : 0ull);
        if (firstIncl < 64) break;
        j -= 64;
      }
      if (lane == 0)
      {
        if (b > 0) publish64(&cells[b], (2ull << kTag) | (excl + total));
        s_base = excl;
      }
    }
    __syncthreads();
    const u64 bit0 = s_base, bit1 = bit0 + total;
    pos = bit0 + before + inc - mine;
    spanWord0 = bit0 >> 5;
    spanWords = (u32)(((bit1 + 31) >> 5) - spanWord0);
    inLds = spanWords <= (u32)(kHuffSpanWords * RUN / 128);
    if (inLds) for (u32 x = threadIdx.x; x < spanWords; x += 256) s_span[x] = 0u;
    __syncthreads();
    TRACEH(3);
  }
  else if (active) pos = runBase[run];
  if (active)
  {
    u64 w = pos >> 5;
    u64 acc = 0;                   // bits of this run for the current word(s), left aligned behind the previous run's bits
    int have = (int)(pos & 31);    // bits already used in the current word (by the previous run)
    for (i64 v = v0; v < v1; v++)
    {
      int s;
      if (staged) s = s_sym[(int)(v - v0) * 260 + threadIdx.x];
      else { s = huffSymbolAt<T>(data, maskBits, g, cur); cur.step(g); }
      if (s < 0) continue;
      const u64 c = s_codes[s];
      const int len = (int)(c >> 32);
      const u64 code = c & 0xFFFFFFFFull;
      acc |= code << (64 - have - len);
      have += len;
      if (have >= 32)
      {
        if (inLds) atomicOr(&s_span[(u32)(w - spanWord0)], (u32)(acc >> 32));
        else orStreamWord(stream, mis, w, (u32)(acc >> 32));
        acc <<= 32;
        have -= 32;
        w++;
      }
    }
    if (have > 0 && (u32)(acc >> 32) != 0u)
    {
      if (inLds) atomicOr(&s_span[(u32)(w - spanWord0)], (u32)(acc >> 32));
      else orStreamWord(stream, mis, w, (u32)(acc >> 32));
    }
  }
  if (!inLds) return;    // (uniform over the workgroup)
  __syncthreads();
  TRACEH(4);
  // The stream lies where the blob has it, `mis` bytes into the aligned word stream[0]: aligned word m holds the last bytes of stream
  // word m - 1 and the first ones of word m.  Whole aligned words are stored where both belong to this workgroup alone; the span's
  // first and last word are shared with the neighbours, and what holds bits of them goes out with an atomic OR.
  const u32 sh = 8u * mis;
  for (u32 x = threadIdx.x; x < spanWords + (mis ? 1u : 0u); x += 256)
  {
    const u32 hi = x < spanWords ? s_span[x] : 0u, lo = (mis && x > 0u) ? s_span[x - 1u] : 0u;
    const u32 word = mis ? ((hi << sh) | (lo >> (32u - sh))) : hi;
    const bool own = mis ? (x >= 2u && x + 1u < spanWords) : (x >= 1u && x + 1u < spanWords);
    if (own) stream[spanWord0 + x] = word;
    else if (word) atomicOr(&stream[spanWord0 + x], word);
  }
  TRACEH(5);
}

void launchHuffRunBits(int dt, const void* data, const u8* maskBits, const HuffGeom& g, int mode, const u64* codes, u32* runBits,
                       hipStream_t st)
{
  const i64 n = (i64)g.nRows * g.nCols * g.nDepth;
  const i64 nRuns = (n + kHuffRun - 1) / kHuffRun;
  const dim3 grid((unsigned)((nRuns + 255) / 256)), block(256);
  if (dt == DT_Char) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<signed char, false, kHuffRun>), grid, block, 0, st, (const signed char*)data, maskBits, g, mode, codes, runBits, (const u64*)nullptr, (u32*)nullptr, 0u, (u64*)nullptr, (DeviceStatus*)nullptr);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<unsigned char, false, kHuffRun>), grid, block, 0, st, (const unsigned char*)data, maskBits, g, mode, codes, runBits, (const u64*)nullptr, (u32*)nullptr, 0u, (u64*)nullptr, (DeviceStatus*)nullptr);
}

void launchHuffPack(int dt, const void* data, const u8* maskBits, const HuffGeom& g, int mode, const u64* codes, const u64* runBase,
                    u32* stream, u32 mis, u64* cells, DeviceStatus* status, hipStream_t st)
{
  const i64 n = (i64)g.nRows * g.nCols * g.nDepth;
  if (cells)    // one pass: no table of runs, so a thread takes kHuffSelfRun elements -- half the LDS, twice the workgroups per CU
  {
    const dim3 grid((unsigned)huffPackCells(n)), block(256);
    if (dt == DT_Char) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<signed char, true, kHuffSelfRun>), grid, block, 0, st, (const signed char*)data, maskBits, g, mode, codes, (u32*)nullptr, runBase, stream, mis, cells, status);
    else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<unsigned char, true, kHuffSelfRun>), grid, block, 0, st, (const unsigned char*)data, maskBits, g, mode, codes, (u32*)nullptr, runBase, stream, mis, cells, status);
    return;
  }
  const i64 nRuns = (n + kHuffRun - 1) / kHuffRun;
  const dim3 grid((unsigned)((nRuns + 255) / 256)), block(256);
  if (dt == DT_Char) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<signed char, true, kHuffRun>), grid, block, 0, st, (const signed char*)data, maskBits, g, mode, codes, (u32*)nullptr, runBase, stream, mis, cells, status);
  else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_huff_encode<unsigned char, true, kHuffRun>), grid, block, 0, st, (const unsigned char*)data, maskBits, g, mode, codes, (u32*)nullptr, runBase, stream, mis, cells, status);
}

// u32 -> u64 exclusive scan (bit offsets can exceed 2^32); out[n] = total
__global__ void __launch_bounds__(256) k_scan64_local(const u32* __restrict__ in, u64* __restrict__ out, u32 n, u64* __restrict__ partial)
{
  __shared__ u64 s_w[4];
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  const u64 v = i < n ? in[i] : 0;
  u64 inc = v;
  const int lane = laneId();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += t; }
  if (lane == 63) s_w[waveId()] = inc;
  __syncthreads();
  u64 base = 0;
  for (int k = 0; k < waveId(); k++) base += s_w[k];
  if (i < n) out[i] = base + inc - v;
  if (threadIdx.x == 255) partial[blockIdx.x] = base + inc;
}

__global__ void __launch_bounds__(64) k_scan64_partials(u64* __restrict__ partial, u32 nPartials, u64* __restrict__ totalOut)
{
  // one wave, sequential over chunks of 64 with a carry
  u64 carry = 0;
  const int lane = laneId();
  for (u32 b = 0; b < nPartials; b += 64)
  {
    const u32 i = b + (u32)lane;
    const u64 v = i < nPartials ? partial[i] : 0;
    u64 inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u64 t = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += t; }
    if (i < nPartials) partial[i] = carry + inc - v;
    carry += __shfl(inc, 63);
  }
  if (lane == 0) *totalOut = carry;
}

__global__ void __launch_bounds__(256) k_scan64_add(u64* __restrict__ out, u32 n, const u64* __restrict__ partial)
{
  const u32 i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) out[i] += partial[blockIdx.x];
}

void launchScan64(const u32* in, u64* out, u32 n, u64* scratch, hipStream_t st)
{
  if (n == 0) { hipMemsetAsync(out, 0, 8, st); return; }
  const u32 nPart = (n + 255) / 256;
  hipLaunchKernelGGL(k_scan64_local, dim3(nPart), dim3(256), 0, st, in, out, n, scratch);
  hipLaunchKernelGGL(k_scan64_partials, dim3(1), dim3(64), 0, st, scratch, nPart, out + n);
  hipLaunchKernelGGL(k_scan64_add, dim3(nPart), dim3(256), 0, st, out, n, (const u64*)scratch);
}

// ---- decode -------------------------------------------------------------------------------------
// 32 stream bits starting at bit position p, MSB first (words beyond `nWords` read as zero)
__device__ __forceinline__ u32 peek32(const u32* __restrict__ stream, u64 nWords, u64 p)
{
  const u64 w = p >> 5;
  const int sh = (int)(p & 31);
  const u32 w0 = w < nWords ? stream[w] : 0u;
  const u32 w1 = (w + 1) < nWords ? stream[w + 1] : 0u;
  return sh ? ((w0 << sh) | (w1 >> (32 - sh))) : w0;
}

// The decoders run one thread per sub-sequence of subWords 32-bit words; read straight from global memory every lane would
// sit in a cache line of its own.  A workgroup of kHuffDecThreads threads therefore stages its slice of the stream in LDS
// with coalesced loads -- kHuffWarmWords words in front of its first sub-sequence (the warm-up of the first round, below),
// its sub-sequences, kHuffTailWords of the next -- as it lies in memory: subWords is odd (huffSubWords), so that the
// lanes of a wave, each reading its own sub-sequence, spread over the banks without any transposition.  The host picks
// subWords such that the workgroups fill the machine in whole rounds (853 workgroups on 768 slots take as long as 1536).
#ifdef LERC_SMALL_GROUPS    // (emulator builds: small rasters still give several workgroups, and warm-ups that do not catch on)
static const int kHuffDecThreads = 16;
static const int kHuffWarmWords = 1, kHuffTailWords = 4;
#else
static const int kHuffDecThreads = 256;
#ifndef LERC_HUFF_WARM
#define LERC_HUFF_WARM 8
#endif
static const int kHuffWarmWords = LERC_HUFF_WARM, kHuffTailWords = 4;
#endif
static const int kHuffStageWords = kHuffDecThreads * kHuffSubWordsMax + kHuffWarmWords + kHuffTailWords;

__device__ __forceinline__ i64 stageOriginWord(u32 subWords) { return (i64)blockIdx.x * kHuffDecThreads * subWords - kHuffWarmWords; }

// (the stream begins `mis` bytes into the aligned word stream[0] -- it lies where the blob has it, behind a code table of any
// length: a word of the stream is put together from two aligned ones; the stream's last word from its bytes, so that nothing
// behind the blob is read)
__device__ __forceinline__ void stageStream(const u32* __restrict__ stream, u32 mis, u64 nWords, u32 subWords, u32* s_str)
{
  const i64 o = stageOriginWord(subWords);
  for (u32 l = threadIdx.x; l < (u32)kHuffDecThreads * subWords + (u32)(kHuffWarmWords + kHuffTailWords); l += kHuffDecThreads)
  {
    const i64 w = o + l;
    u32 v = 0u;
    if (w >= 0 && (u64)w < nWords)
    {
      if (mis == 0u) v = stream[w];
      else if ((u64)w + 1u < nWords) v = __builtin_amdgcn_alignbit(stream[w + 1], stream[w], 8u * mis);
      else
      {
        const u8* bytes = reinterpret_cast<const u8*>(stream) + 4u * (size_t)w + mis;
        v = (u32)bytes[0] | ((u32)bytes[1] << 8) | ((u32)bytes[2] << 16) | ((u32)bytes[3] << 24);
      }
    }
    s_str[l] = v;
  }
}

// the workgroup's LDS copy of the tables: the look-up table as (length << 8) | symbol in 16 bits, 0xFFFF = longer than
// the LUT width; the longer codes (sorted by length) as code word and (length << 8) | symbol
struct HuffLdsTable
{
  u16 lut[1 << kHuffLutBits];
  u32 longCode[256];
  u16 longLenSym[256];
  int nLong;
};

__device__ __forceinline__ void stageLut(const HuffDecodeTable* __restrict__ t, HuffLdsTable& s)
{
  for (int i = threadIdx.x; i < (1 << kHuffLutBits); i += kHuffDecThreads)
  {
    const u32 e = t->lut[i];
    s.lut[i] = (e == 0xFFFFFFFFu) ? (u16)0xFFFFu : (u16)(((e >> 16) << 8) | (e & 0xFFu));
  }
  const int nLong = t->nLong;
  for (int i = threadIdx.x; i < nLong; i += kHuffDecThreads)
  {
    s.longCode[i] = t->longCode[i];
    s.longLenSym[i] = (u16)(((u32)t->longLen[i] << 8) | ((u32)t->longSym[i] & 0xFFu));
  }
  if (threadIdx.x == 0) s.nLong = nLong;
}

// Bit reader over the staged slice.  It keeps no window: the 32 bits at the current position are put together from two
// neighbouring LDS words every time (one ds_read2 and a funnel shift).  A 64-bit window refilled "when needed" looks
// cheaper per symbol, but the refill is a branch that some lane of the wave takes in nearly every step, and the
// compiler's mask bookkeeping for it (plus the window's own shifts) came to more instructions than the decoding itself --
// the loop is issue bound, not latency bound (twelve waves per CU hide the two dependent LDS reads).
struct StagedBits
{
  const u32* s_str;
  u32 pos;     // bit position, counted from the slice's first staged word

  __device__ __forceinline__ void start(const u32* str, u32 at) { s_str = str; pos = at; }
  __device__ __forceinline__ u32 top() const
  {
    const u32 w = pos >> 5;
    const u64 x = ((u64)s_str[w] << 32) | s_str[w + 1u];
    return (u32)((x << (pos & 31u)) >> 32);
  }
  __device__ __forceinline__ void skip(int len) { pos += (u32)len; }
};

// returns the code length (0 = no code matches), symbol in sym
__device__ __forceinline__ int decodeOne(const HuffLdsTable& s, u32 top, int& sym)
{
  const u32 e = s.lut[top >> (32 - kHuffLutBits)];
  if (e != 0xFFFFu) { sym = (int)(e & 0xFFu); return (int)(e >> 8); }
  const int nLong = s.nLong;
  for (int i = 0; i < nLong; i++)
  {
    const u32 ls = s.longLenSym[i];
    const int len = (int)(ls >> 8);
    if ((top >> (32 - len)) == s.longCode[i]) { sym = (int)(ls & 0xFFu); return len; }
  }
  return 0;
}

// one thread per sub-sequence of subWords * 32 bits: decode from starts[t] up to the end of the
// sub-sequence; exits[t] = first code word position at or beyond it, counts[t] = symbols decoded.
// First round (warm != 0): nobody knows where a code word starts near the sub-sequence's first bit, but a decoder set
// down anywhere falls into step with the true code words within a few symbols -- so the thread starts kHuffWarmWords
// words early and takes the first code word boundary at or behind its first bit as starts[t].  Where the warm-up did not
// catch on, the sub-sequence in front ends somewhere else than this one begins: the workgroup sees that in LDS and lets
// such threads start over from their predecessor's exit until its sub-sequences fit together.  If the chain of exits
// (k_huff_chain) then fits across the workgroups as well, which is the normal case, that was the only round.
__global__ void __launch_bounds__(kHuffDecThreads)
k_huff_sync(const u32* __restrict__ stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* __restrict__ table, u32 nSub, u32 subWords,
            u64* __restrict__ starts, u64* __restrict__ prevStarts, u64* __restrict__ exits, u32* __restrict__ counts,
            u32* __restrict__ bad, int warm)
{
  __shared__ HuffLdsTable s_tab;
  __shared__ u32 s_str[kHuffStageWords];
  __shared__ u32 s_exit[kHuffDecThreads];
  __shared__ u32 s_any;
  const u32 t = blockIdx.x * (u32)kHuffDecThreads + threadIdx.x;
  u64 s = (t < nSub) ? starts[t] : 0;
  bool todo = t < nSub && prevStarts[t] != s;    // else: unchanged since the last round
  if (threadIdx.x == 0) s_any = 0;
  __syncthreads();
  if (todo) s_any = 1;
  __syncthreads();
  if (!s_any) return;    // (after the first round most workgroups have nothing to redo)
  stageLut(table, s_tab);
  stageStream(stream, mis, nWords, subWords, s_str);
  __syncthreads();
  const u64 origin = (u64)(stageOriginWord(subWords) + kHuffWarmWords) * 32u;    // bit position of the slice's first sub-sequence
  const u64 subBits = (u64)subWords * 32u;
  const u64 end = min((u64)(t + 1) * subBits, streamBits);
  const u32 endLocal = (t < nSub) ? (u32)(end - origin) + kHuffWarmWords * 32u : 0u;
  u32 exitLocal = 0;    // (local positions count from the slice's first staged word)
  bool broken = false;
  for (int pass = 0; ; pass++)
  {
    if (todo)
    {
      StagedBits in;
      if (warm && t > 0 && pass == 0)
      {
        const u32 first = (u32)(s - origin) + kHuffWarmWords * 32u;
        in.start(s_str, first - kHuffWarmWords * 32u);
        while (in.pos < first)
        {
          int sym;
          const int len = decodeOne(s_tab, in.top(), sym);
          if (len == 0) { in.start(s_str, first); break; }
          in.skip(len);
        }
        s = origin + in.pos - kHuffWarmWords * 32u;
      }
      else in.start(s_str, (u32)(s - origin) + kHuffWarmWords * 32u);
      u32 n = 0;
      broken = false;
      while (in.pos < endLocal)
      {
        int sym;
        const int len = decodeOne(s_tab, in.top(), sym);
        if (len == 0) { broken = true; break; }
        in.skip(len);
        n++;
      }
      exitLocal = in.pos;
      starts[t] = s;
      prevStarts[t] = s;
      exits[t] = origin + in.pos - kHuffWarmWords * 32u;
      counts[t] = n;
    }
    else if (t < nSub && pass == 0) exitLocal = (u32)(exits[t] - origin) + kHuffWarmWords * 32u;
    // does every sub-sequence of the workgroup begin where the one in front ended?
    __syncthreads();
    s_exit[threadIdx.x] = exitLocal;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    todo = false;
    if (threadIdx.x > 0 && t < nSub)
    {
      const u32 mine = (u32)(s - origin) + kHuffWarmWords * 32u;
      const u32 before = s_exit[threadIdx.x - 1];
      // (a predecessor that ran far beyond its end -- a broken stream -- is left to the chain check)
      if (before != mine && before < endLocal) { todo = true; s = origin + before - kHuffWarmWords * 32u; s_any = 1; }
    }
    __syncthreads();
    if (!s_any || pass >= 64) break;
  }
  if (broken) atomicOr(bad, 1u);    // (no code word matched: the symbol count comes out short and the host refuses the blob)
}

__global__ void __launch_bounds__(256)
k_huff_chain(u32 nSub, u64* __restrict__ starts, const u64* __restrict__ exits, u32* __restrict__ changed)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t + 1 >= nSub) return;
  const u64 e = exits[t];
  if (starts[t + 1] != e) { starts[t + 1] = e; atomicOr(changed, 1u); }
}

// pixel index of every valid pixel, in scan order (rank -> pixel)
__global__ void __launch_bounds__(256) k_valid_index(const u8* __restrict__ maskBits, const u32* __restrict__ groupBase, i64 nPix, u32* __restrict__ validIdx)
{
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k >= nPix || !maskBit(maskBits, k)) return;
  u32 r = groupBase[k >> 5];
  for (i64 j = (k >> 5) << 5; j < k; j++) r += maskBit(maskBits, j) ? 1u : 0u;
  validIdx[r] = (u32)k;
}

void launchValidIndex(const u8* maskBits, const u32* groupBase, i64 nPix, u32* validIdx, hipStream_t st)
{
  hipLaunchKernelGGL(k_valid_index, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, st, maskBits, groupBase, nPix, validIdx);
}

// second pass: write symbol r of the stream to its pixel (raw deltas in delta mode)
template<class T>
__global__ void __launch_bounds__(kHuffDecThreads)
k_huff_emit(const u32* __restrict__ stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* __restrict__ table, u32 nSub, u32 subWords,
            const u64* __restrict__ starts, const u64* __restrict__ symBase, HuffGeom g, int mode, u64 nSymbols, u32 numValid,
            const u32* __restrict__ validIdx, int rankOrder, T* __restrict__ out)
{
  __shared__ HuffLdsTable s_tab;
  __shared__ u32 s_str[kHuffStageWords];
  stageLut(table, s_tab);
  stageStream(stream, mis, nWords, subWords, s_str);
  __syncthreads();
  const u32 t = blockIdx.x * (u32)kHuffDecThreads + threadIdx.x;
  if (t >= nSub) return;
  const int off = (DtOf<T>::v == DT_Char) ? 128 : 0;
  const u64 origin = (u64)(stageOriginWord(subWords) + kHuffWarmWords) * 32u;
  const u64 end = min((u64)(t + 1) * (u64)subWords * 32u, streamBits);
  const u32 endLocal = (u32)(end - origin) + kHuffWarmWords * 32u;
  StagedBits in;
  in.start(s_str, (u32)(starts[t] - origin) + kHuffWarmWords * 32u);
  u64 r = symBase[t];
  if (rankOrder)
  {
    // symbol r is byte r of the output: eight at a time once r is a multiple of 8 (every lane writes a stream of its own,
    // a cache line apart from its neighbours': the fewer, wider stores the better)
    const bool wide = ((size_t)out & 7u) == 0;
    u64 acc = 0;        // the pending bytes, oldest lowest -- once eight have come in
    int pending = 0;    // bytes r - pending .. r - 1 wait in acc's top bytes
    while (in.pos < endLocal && r < nSymbols)
    {
      int sym;
      const int len = decodeOne(s_tab, in.top(), sym);
      if (len == 0) break;
      in.skip(len);
      const u32 v = (u32)(sym - off) & 255u;
      if (!wide || (pending == 0 && (r & 7u) != 0)) out[r] = (T)(u8)v;
      else
      {
        acc = (acc >> 8) | ((u64)v << 56);
        if (++pending == 8) { *reinterpret_cast<u64*>(out + (r - 7)) = acc; pending = 0; }
      }
      r++;
    }
    for (int j = 0; j < pending; j++) out[r - (u64)pending + j] = (T)(u8)(acc >> (8 * (8 - pending + j)));
    return;
  }
  // rank r -> (valid pixel q, depth m): divided once, then stepped
  u64 q, m;
  if (mode == IEM_Huffman) { q = r / (u64)g.nDepth; m = r - q * (u64)g.nDepth; }
  else { m = r / numValid; q = r - m * numValid; }
  while (in.pos < endLocal && r < nSymbols)
  {
    int sym;
    const int len = decodeOne(s_tab, in.top(), sym);
    if (len == 0) break;
    in.skip(len);
    const i64 k = validIdx ? (i64)validIdx[q] : (i64)q;
    out[k * g.nDepth + (i64)m] = (T)(sym - off);
    r++;
    if (mode == IEM_Huffman) { if (++m == (u64)g.nDepth) { m = 0; q++; } }
    else if (++q == numValid) { q = 0; m++; }
  }
}

// Undo of the delta predictor (Lerc2.cpp:2499-2524 / :2555-2583): one wave per depth plane walks the
// plane in scan order 64 pixels at a time.  A valid pixel continues from the nearest valid pixel
// before it (left neighbour, or the "previous value" rule) unless its left neighbour is missing and
// the pixel above is valid, in which case it restarts from the pixel above.
template<class T>
__global__ void __launch_bounds__(64) k_huff_undelta(T* __restrict__ data, const u8* __restrict__ maskBits, HuffGeom g)
{
  const int iD = (int)blockIdx.x;
  const int lane = laneId();
  const u64 le = laneMaskLt() | (1ull << lane);
  u32 carry = 0;    // value of the last valid pixel so far (prevVal starts at 0)
  // row by row, so that a pixel and the pixel above it never share a 64-lane chunk
  for (int i = 0; i < g.nRows; i++)
    for (int j0 = 0; j0 < g.nCols; j0 += 64)
    {
      const int j = j0 + lane;
      const bool inb = j < g.nCols;
      const i64 k = (i64)i * g.nCols + j;
      const bool valid = inb && (!maskBits || maskBit(maskBits, k));
      const bool leftOk = valid && j > 0 && (!maskBits || maskBit(maskBits, k - 1));
      const bool fromAbove = valid && !leftOk && i > 0 && (!maskBits || maskBit(maskBits, k - g.nCols));
      const u32 d = valid ? (u32)(u8)data[k * g.nDepth + iD] : 0u;
      const u32 above = fromAbove ? (u32)(u8)data[(k - g.nCols) * g.nDepth + iD] : 0u;
      // inclusive prefix sum of the deltas over the lanes
      u32 s = d;
#pragma unroll
      for (int dd = 1; dd < 64; dd <<= 1) { const u32 tt = __shfl_up(s, (unsigned)dd); if (lane >= dd) s += tt; }
      const u64 heads = __ballot(fromAbove);
      const u64 mine = heads & le;
      const int h = mine ? 63 - __clzll((long long)mine) : 0;
      const u32 sH = __shfl(s, h), dH = __shfl(d, h), aH = __shfl(above, h);
      u32 val = mine ? (aH + s - (sH - dH)) : (carry + s);
      val &= 0xFFu;
      if (valid) data[k * g.nDepth + iD] = (T)(u8)val;
      const u64 vmask = __ballot(valid);
      if (vmask)
      {
        const int last = 63 - __clzll((long long)vmask);
        carry = __shfl(val, last);
      }
    }
}

// All pixels valid: a pixel continues from its left neighbour and the first pixel of a row from the one above it
// (the first pixel of the plane from 0), so column 0 is a running sum down the rows and every row then a running sum
// of its own -- one workgroup per depth plane resp. per (row, plane), 8-bit wrap-around arithmetic.
__device__ __forceinline__ u32 undeltaWorkgroupScan(u32 v, u32* s_wave, u32& total)
{
  const int lane = laneId(), w = waveId();
  u32 inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(inc, (unsigned)d); if (lane >= d) inc += t; }
  if (lane == 63) s_wave[w] = inc;
  __syncthreads();
  u32 before = 0;
  for (int i = 0; i < w; i++) before += s_wave[i];
  total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
  __syncthreads();
  return before + inc - v;    // exclusive
}

template<class T>
__global__ void __launch_bounds__(256) k_huff_undelta_col0(T* __restrict__ data, HuffGeom g)
{
  __shared__ u32 s_wave[4];
  const int iD = (int)blockIdx.x;
  u32 carry = 0;
  for (int i0 = 0; i0 < g.nRows; i0 += 256)
  {
    const int i = i0 + (int)threadIdx.x;
    const i64 at = (i64)i * g.nCols * g.nDepth + iD;
    const u32 d = (i < g.nRows) ? (u32)(u8)data[at] : 0u;
    u32 total;
    const u32 before = undeltaWorkgroupScan(d, s_wave, total);
    if (i < g.nRows) data[at] = (T)(u8)(carry + before + d);
    carry += total;
  }
}

template<class T>
__global__ void __launch_bounds__(256) k_huff_undelta_rows(T* __restrict__ data, HuffGeom g)
{
  __shared__ u32 s_wave[4];
  const int iD = (int)blockIdx.y;
  T* row = data + (i64)blockIdx.x * g.nCols * g.nDepth + iD;
  u32 carry = 0;
  for (int j0 = 0; j0 < g.nCols; j0 += 1024)
  {
    const int j = j0 + (int)threadIdx.x * 4;
    u32 v[4];
    for (int k = 0; k < 4; k++) v[k] = (j + k < g.nCols) ? (u32)(u8)row[(i64)(j + k) * g.nDepth] : 0u;
    for (int k = 1; k < 4; k++) v[k] += v[k - 1];
    u32 total;
    const u32 before = carry + undeltaWorkgroupScan(v[3], s_wave, total);
    for (int k = 0; k < 4; k++) if (j + k < g.nCols) row[(i64)(j + k) * g.nDepth] = (T)(u8)(before + v[k]);
    carry += total;
  }
}

// Delta mode, every pixel valid, 1 < nDepth <= kHuffInterleaveMax: the decoder leaves the deltas plane by plane
// ([nDepth][nPix], the order of the stream) -- scattering them to their pixels from the decoder's threads would keep more
// half-written cache lines open than the L2 holds.  Column 0 is summed in the planes; then a workgroup per row sums
// the row of every plane and writes the interleaved pixels out of LDS in whole words.
static const int kHuffInterleaveMax = 16;

// (a workgroup of 1024 threads per plane, a thread takes a stretch of consecutive rows: all its loads -- one byte a row, a row apart: a
// cache line each -- are in flight together, and ONE scan over the threads' sums follows; 256 rows a round with a scan each took sixteen
// dependent rounds for 4096 rows, most of the predictor's 58 us)
__global__ void __launch_bounds__(1024) k_huff_undelta_col0_planar(u8* __restrict__ planar, HuffGeom g)
{
  __shared__ u32 s_wave[16];
  constexpr int kPer = 8;                            // rows a thread and round
  u8* plane = planar + (i64)blockIdx.x * g.nRows * g.nCols;
  const int lane = laneId(), w = waveId();
  u32 carry = 0;
  for (int i0 = 0; i0 < g.nRows; i0 += 1024 * kPer)
  {
    const int iMine = i0 + (int)threadIdx.x * kPer;
    u32 d[kPer];
#pragma unroll
    for (int k = 0; k < kPer; k++) d[k] = (iMine + k < g.nRows) ? (u32)plane[(i64)(iMine + k) * g.nCols] : 0u;
#pragma unroll
    for (int k = 1; k < kPer; k++) d[k] += d[k - 1];
    u32 inc = d[kPer - 1];
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) { const u32 t = __shfl_up(inc, (unsigned)s); if (lane >= s) inc += t; }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    u32 before = carry + inc - d[kPer - 1], total = 0;
    for (int i = 0; i < 16; i++) { const u32 t = s_wave[i]; if (i < w) before += t; total += t; }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kPer; k++) if (iMine + k < g.nRows) plane[(i64)(iMine + k) * g.nCols] = (u8)(before + d[k]);
    carry += total;
  }
}

__global__ void __launch_bounds__(256) k_huff_undelta_rows_interleave(const u8* __restrict__ planar, u8* __restrict__ out, HuffGeom g)
{
  __shared__ u32 s_wave[4];
  __shared__ u32 s_carry[kHuffInterleaveMax];
  __shared__ u32 s_out[1024 * kHuffInterleaveMax / 4];
  u8* s_bytes = reinterpret_cast<u8*>(s_out);
  const i64 nPix = (i64)g.nRows * g.nCols, row0 = (i64)blockIdx.x * g.nCols;
  if (threadIdx.x < (u32)kHuffInterleaveMax) s_carry[threadIdx.x] = 0;
  __syncthreads();
  for (int j0 = 0; j0 < g.nCols; j0 += 1024)
  {
    const int j = j0 + (int)threadIdx.x * 4;
    for (int iD = 0; iD < g.nDepth; iD++)
    {
      const u8* src = planar + (i64)iD * nPix + row0;
      u32 v[4];
      for (int k = 0; k < 4; k++) v[k] = (j + k < g.nCols) ? (u32)src[j + k] : 0u;
      for (int k = 1; k < 4; k++) v[k] += v[k - 1];
      const u32 carry = s_carry[iD];
      u32 total;
      const u32 before = carry + undeltaWorkgroupScan(v[3], s_wave, total);
      for (int k = 0; k < 4; k++) s_bytes[((int)threadIdx.x * 4 + k) * g.nDepth + iD] = (u8)(before + v[k]);
      if (threadIdx.x == 0) s_carry[iD] = carry + total;
    }
    __syncthreads();
    const int nBytes = min(1024, g.nCols - j0) * g.nDepth;
    u8* dst = out + (row0 + j0) * g.nDepth;
    if ((((size_t)dst) & 3u) == 0)
    {
      for (int w = threadIdx.x; w < nBytes / 4; w += 256) reinterpret_cast<u32*>(dst)[w] = s_out[w];
      for (int b = (nBytes & ~3) + (int)threadIdx.x; b < nBytes; b += 256) dst[b] = s_bytes[b];
    }
    else for (int b = threadIdx.x; b < nBytes; b += 256) dst[b] = s_bytes[b];
    __syncthreads();
  }
}

bool huffPlanarDecode(int imageMode, const u8* maskBits, int nDepth)
{
  return imageMode == IEM_DeltaHuffman && !maskBits && nDepth > 1 && nDepth <= kHuffInterleaveMax;
}

void launchHuffUndeltaPlanar(u8* planar, void* out, const HuffGeom& g, hipStream_t st)
{
  hipLaunchKernelGGL(k_huff_undelta_col0_planar, dim3(g.nDepth), dim3(1024), 0, st, planar, g);
  hipLaunchKernelGGL(k_huff_undelta_rows_interleave, dim3(g.nRows), dim3(256), 0, st, (const u8*)planar, (u8*)out, g);
}

void launchHuffUndelta(int dt, void* data, const u8* maskBits, const HuffGeom& g, hipStream_t st)
{
  if (!maskBits)
  {
    if (dt == DT_Char)
    {
      hipLaunchKernelGGL(k_huff_undelta_col0<signed char>, dim3(g.nDepth), dim3(256), 0, st, (signed char*)data, g);
      hipLaunchKernelGGL(k_huff_undelta_rows<signed char>, dim3(g.nRows, g.nDepth), dim3(256), 0, st, (signed char*)data, g);
    }
    else
    {
      hipLaunchKernelGGL(k_huff_undelta_col0<unsigned char>, dim3(g.nDepth), dim3(256), 0, st, (unsigned char*)data, g);
      hipLaunchKernelGGL(k_huff_undelta_rows<unsigned char>, dim3(g.nRows, g.nDepth), dim3(256), 0, st, (unsigned char*)data, g);
    }
    return;
  }
  if (dt == DT_Char) hipLaunchKernelGGL(k_huff_undelta<signed char>, dim3(g.nDepth), dim3(64), 0, st, (signed char*)data, maskBits, g);
  else hipLaunchKernelGGL(k_huff_undelta<unsigned char>, dim3(g.nDepth), dim3(64), 0, st, (unsigned char*)data, maskBits, g);
}

void launchHuffSync(const u32* stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* table, u32 nSub, u32 subWords, u64* starts,
                    u64* prevStarts, u64* exits, u32* counts, u32* bad, bool firstRound, hipStream_t st)
{
  hipLaunchKernelGGL(k_huff_sync, dim3((nSub + kHuffDecThreads - 1) / kHuffDecThreads), dim3(kHuffDecThreads), 0, st, stream, mis, nWords, streamBits, table, nSub, subWords,
                     starts, prevStarts, exits, counts, bad, firstRound ? 1 : 0);
}

// Words per sub-sequence for a stream of that many bits: odd (LDS banks, see above), between kHuffSubWordsMin and
// kHuffSubWordsMax, and such that the workgroups -- `slots` of them run at the same time -- come in full rounds.
u32 huffSubWords(u64 streamBits, u32 slots)
{
  const u64 words = (streamBits + 31) / 32;
  if (slots == 0) slots = 1;
  u32 best = (u32)kHuffSubWordsMin;
  u64 bestCost = ~0ull;
  for (u32 w = (u32)kHuffSubWordsMin; w <= (u32)kHuffSubWordsMax; w += 2)
  {
    const u64 nSub = (words + w - 1) / w, nWG = (nSub + kHuffDecThreads - 1) / kHuffDecThreads;
    const u64 cost = ((nWG + slots - 1) / slots) * w;
    if (cost < bestCost) { bestCost = cost; best = w; }
  }
  return best;
}

void launchHuffChain(u32 nSub, u64* starts, const u64* exits, u32* changed, hipStream_t st)
{
  hipLaunchKernelGGL(k_huff_chain, dim3((nSub + 255) / 256), dim3(256), 0, st, nSub, starts, exits, changed);
}

void launchHuffEmit(int dt, const u32* stream, u32 mis, u64 nWords, u64 streamBits, const HuffDecodeTable* table, u32 nSub, u32 subWords, const u64* starts,
                    const u64* symBase, const HuffGeom& g, int mode, u64 nSymbols, u32 numValid, const u32* validIdx, bool planar,
                    void* out, hipStream_t st)
{
  const dim3 grid((nSub + kHuffDecThreads - 1) / kHuffDecThreads), block(kHuffDecThreads);
  // symbol r goes to byte r: one value per pixel, or pixel-interleaved symbols (not delta mode), or planes wanted
  const int rankOrder = (!validIdx && (g.nDepth == 1 || mode == IEM_Huffman || planar)) ? 1 : 0;
  if (dt == DT_Char) hipLaunchKernelGGL(k_huff_emit<signed char>, grid, block, 0, st, stream, mis, nWords, streamBits, table, nSub, subWords, starts, symBase, g, mode, nSymbols, numValid, validIdx, rankOrder, (signed char*)out);
  else hipLaunchKernelGGL(k_huff_emit<unsigned char>, grid, block, 0, st, stream, mis, nWords, streamBits, table, nSub, subWords, starts, symBase, g, mode, nSymbols, numValid, validIdx, rankOrder, (unsigned char*)out);
}

__global__ void __launch_bounds__(256) k_init_starts(u64* __restrict__ starts, u64* __restrict__ prevStarts, u32 nSub, u32 subWords)
{
  const u32 t = blockIdx.x * 256u + threadIdx.x;
  if (t >= nSub) return;
  starts[t] = (u64)t * subWords * 32u;
  prevStarts[t] = ~0ull;
}

void launchHuffInitStarts(u64* starts, u64* prevStarts, u32 nSub, u32 subWords, hipStream_t st)
{
  hipLaunchKernelGGL(k_init_starts, dim3((nSub + 255) / 256), dim3(256), 0, st, starts, prevStarts, nSub, subWords);
}

}    // namespace lerc
