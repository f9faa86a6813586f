// misc_kernels.hip -- the full-image sweeps around the block loop, as HBM-bound HIP kernels:
//   exclusive scan (block sizes -> block offsets), Fletcher32 partial sums, validity-mask
//   construction, per-depth min / max + float diagnostics, raw "one sweep" copy, const fill, widen.
// Reference counterparts: Lerc2::ComputeChecksumFletcher32 (Lerc2.cpp:1037-1064),
// Lerc::FilterNoDataAndNaN (Lerc.cpp:1378-1552, the noData-free part), Lerc2::ComputeMinMaxRanges
// (Lerc2.cpp:1404-1470), Lerc2::TryRaiseMaxZError (:1233-1318), Write/ReadDataOneSweep (:1343-1400),
// FillConstImage (:2681-2721), Lerc::Convert byte<->bit mask (Lerc.cpp:959-995).
#include "kernels.h"
#include <limits>
#include <algorithm>
#include "wave_utils.h"

namespace lerc {

// ================================================================================================
// exclusive scan
// ================================================================================================
// returns the exclusive prefix of `v` over the 256 threads of the workgroup; total in `total`
__device__ __forceinline__ u32 blockExclusiveScan256(u32 v, u32& total)
{
  __shared__ u32 s_w[4];
  const u32 inc = waveInclusiveScan(v);
  if (laneId() == 63) s_w[waveId()] = inc;
  __syncthreads();
  u32 base = 0;
  for (int i = 0; i < waveId(); i++) base += s_w[i];
  total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  __syncthreads();
  return base + inc - v;
}

__global__ void __launch_bounds__(256) k_scan_local(const u32* __restrict__ in, u32* __restrict__ out, u32 n, u32* __restrict__ partial)
{
  const u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;
  u32 a[4];
#pragma unroll
  for (int i = 0; i < 4; i++) a[i] = (base + i < n) ? in[base + i] : 0u;
  const u32 s = a[0] + a[1] + a[2] + a[3];
  u32 total;
  u32 ex = blockExclusiveScan256(s, total);
#pragma unroll
  for (int i = 0; i < 4; i++) { if (base + i < n) out[base + i] = ex; ex += a[i]; }
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

__global__ void __launch_bounds__(256) k_scan_partials(u32* __restrict__ partial, u32 nPartials, u32* __restrict__ totalOut)
{
  u32 carry = 0;
  for (u32 b = 0; b < nPartials; b += 256)
  {
    const u32 i = b + threadIdx.x;
    const u32 v = (i < nPartials) ? partial[i] : 0u;
    u32 total;
    const u32 ex = blockExclusiveScan256(v, total);
    if (i < nPartials) partial[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *totalOut = carry;
}

__global__ void __launch_bounds__(256) k_scan_add(u32* __restrict__ out, u32 n, const u32* __restrict__ partial)
{
  const u32 add = partial[blockIdx.x];
  const u32 base = blockIdx.x * 1024u + threadIdx.x * 4u;
#pragma unroll
  for (int i = 0; i < 4; i++) if (base + i < n) out[base + i] += add;
}

// one launch for moderate n: thread t owns a contiguous run of 4 * ceil(n / 4096) elements, moved as 16-byte vectors
// (in and out are 16-byte aligned and hold at least that many rounded-up elements: the callers' arrays have slack)
__global__ void __launch_bounds__(1024) k_scan_single(const u32* __restrict__ in, u32* __restrict__ out, u32 n)
{
  scanSingleWorkgroup(in, out, n);
}

void launchExclusiveScan(const u32* in, u32* out, u32 n, u32* scratch, hipStream_t stream)
{
  if (n == 0) { hipMemsetAsync(out, 0, 4, stream); return; }
  if (n <= (1u << 14) && in != out    /* one workgroup: ~0.45 us per 1024 elements; the three launches cost ~15 us */
      && ((uintptr_t)in & 15) == 0 && ((uintptr_t)out & 15) == 0)
  {
    hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, in, out, n);
    return;
  }
  const u32 nPart = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_local, dim3(nPart), dim3(256), 0, stream, in, out, n, scratch);
  hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(256), 0, stream, scratch, nPart, out + n);
  hipLaunchKernelGGL(k_scan_add, dim3(nPart), dim3(256), 0, stream, out, n, (const u32*)scratch);
}

// ================================================================================================
// Fletcher32 (Lerc2.cpp:1037-1064) as two linear sums mod 65535.
//   words w_i = b[2i] << 8 | b[2i+1] (odd tail byte counts as b << 8), N = ceil(len / 2)
//   s1 = 0xffff + sum w_i,   s2 = 0xffff (N + 1) + sum (N - i) w_i      (mod 65535, 0 -> 0xffff)
// Every byte contributes independently: c = (p even ? 256 : 1) * b to A = sum w and (p >> 1) * c to
// B = sum i * w, so the kernel is a plain streaming reduction; partials are combined on the host.
// ================================================================================================
static const int kFletcherBlocks = 512;

// (1024 threads a workgroup: 512 workgroups of 256 threads were two waves a SIMD, and a wave that asks for one vector at a time is
// latency; a unit that begins at an even byte -- every band's checksummed bytes do: they begin 14 bytes into a 16-byte aligned blob -- is
// summed by two byte permutes and 16-bit dot products (wave_utils.h: fletcherUnit) instead of sixteen byte extractions)
__global__ void __launch_bounds__(1024) k_fletcher(const u8* __restrict__ bytes, u32 len, u64* __restrict__ partials)
{
  __shared__ u64 s_a[16], s_b[16];
  u64 A = 0, B = 0;
  const u32 stride = gridDim.x * blockDim.x;
  // 16 bytes per load from the first 16-byte aligned address on; byte p counts as (byte << 8) when p is even, with
  // weight p >> 1 -- inside a vector the weights are (q >> 1) + small constants, so the 64-bit product is paid once
  const u32 head = min(len, (u32)((16u - ((u32)(uintptr_t)bytes & 15u)) & 15u));
  const u32 nVec = (len - head) >> 4;
  const uint4* vec = reinterpret_cast<const uint4*>(bytes + head);
  if ((head & 1u) == 0u)
  {
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nVec; i += stride)
    {
      u32 a32 = 0;
      fletcherUnit(vec[i], (u64)((head + (i << 4)) >> 1), a32, B);
      A += a32;
    }
  }
  else
    for (u32 i = blockIdx.x * blockDim.x + threadIdx.x; i < nVec; i += stride)
    {
      const u32 q = head + (i << 4), odd = q & 1u;
      const uint4 x = vec[i];
      const u32 w[4] = { x.x, x.y, x.z, x.w };
      u32 sumC = 0, inner = 0;
#pragma unroll
      for (u32 j = 0; j < 16u; j++)
      {
        const u32 byte = (w[j >> 2] >> (8u * (j & 3u))) & 255u;
        const u32 c = byte << (((odd + j) & 1u) ? 0u : 8u);
        sumC += c;
        inner += ((odd + j) >> 1) * c;
      }
      A += sumC;
      B += (u64)(q >> 1) * sumC + inner;
    }
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    for (u32 p = 0; p < head; p++) { const u32 c = (u32)bytes[p] << ((p & 1u) ? 0 : 8); A += c; B += (u64)(p >> 1) * c; }
    for (u32 p = head + (nVec << 4); p < len; p++) { const u32 c = (u32)bytes[p] << ((p & 1u) ? 0 : 8); A += c; B += (u64)(p >> 1) * c; }
  }
  A %= 65535u; B %= 65535u;
  A = waveSum(A); B = waveSum(B);
  if (laneId() == 0) { s_a[waveId()] = A; s_b[waveId()] = B; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    u64 a = 0, b = 0;
    for (u32 k = 0; k < blockDim.x / 64u; k++) { a += s_a[k]; b += s_b[k]; }
    partials[2 * blockIdx.x] = a % 65535u;
    partials[2 * blockIdx.x + 1] = b % 65535u;
  }
}

// acc: 2 * kFletcherBlocks u64 words
void launchFletcher(const u8* blob, u32 len, u64* acc, hipStream_t stream)
{
  hipLaunchKernelGGL(k_fletcher, dim3(kFletcherBlocks), dim3(len >= (1u << 22) ? 1024 : 256), 0, stream, blob, len, acc);    // (small blobs: no more threads than vectors)
}

// folds launchFletcher's partials and writes the finished checksum (four bytes, any alignment) -- so that an encoder need
// not bring the sums to the host and send the header field back
__global__ void __launch_bounds__(64) k_fletcher_patch(const u64* __restrict__ partials, u32 len, u8* __restrict__ dst, u32 moreA, u32 moreB)
{
  u64 A = 0, B = 0;
  for (int i = laneId(); i < kFletcherPartials / 2; i += 64) { A += partials[2 * i]; B += partials[2 * i + 1]; }
  A = waveSum(A); B = waveSum(B);
  if (laneId() != 0) return;
  A += moreA; B += moreB;
  const u64 N = ((u64)len + 1) / 2;
  A %= 65535u; B %= 65535u;
  u64 s1 = A;
  u64 s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
  if (s1 == 0) s1 = 0xffff;
  if (s2 == 0) s2 = 0xffff;
  const u32 cs = (u32)((s2 << 16) | s1);
  for (int i = 0; i < 4; i++) dst[i] = (u8)(cs >> (8 * i));
}

void launchFletcherPatch(const u64* partials, u32 len, u8* dst, hipStream_t stream)
{
  hipLaunchKernelGGL(k_fletcher_patch, dim3(1), dim3(64), 0, stream, partials, len, dst, 0u, 0u);
}

// the same where the partials cover the front part of the `len` bytes only and the terms of the rest are known already
// (sums = sum of words mod 65535 | (sum of word index * word mod 65535) << 16, indices counted from the front)
void launchFletcherPatchWith(const u64* partials, u32 sums, u32 len, u8* dst, hipStream_t stream)
{
  hipLaunchKernelGGL(k_fletcher_patch, dim3(1), dim3(64), 0, stream, partials, len, dst, sums & 0xFFFFu, sums >> 16);
}

u32 fletcherFinish(u64 A, u64 B, u32 len)
{
  const u64 N = ((u64)len + 1) / 2;
  A %= 65535u; B %= 65535u;
  u64 s1 = A;
  u64 s2 = ((N % 65535u) * A + 65535u - B) % 65535u;
  if (s1 == 0) s1 = 0xffff;
  if (s2 == 0) s2 = 0xffff;
  return (u32)((s2 << 16) | s1);
}

// ================================================================================================
// validity mask
// ================================================================================================
template<class T> __device__ __forceinline__ bool isNaNT(T) { return false; }
template<> __device__ __forceinline__ bool isNaNT<float>(float v) { return v != v; }
template<> __device__ __forceinline__ bool isNaNT<double>(double v) { return v != v; }

// four pixels per lane (one 32-bit load of the byte mask, one 128-bit load of a float raster), two lanes make a mask
// byte; a fixed grid strides over the raster so that the valid pixel count costs one atomic per wave of the grid
template<class T>
__global__ void __launch_bounds__(256) k_build_mask(const T* __restrict__ data, const u8* __restrict__ byteMask, i64 nPix,
                                                    int nDepth, u8* __restrict__ maskBits, BandStats* stats)
{
  constexpr bool isFlt = DtOf<T>::v >= DT_Float;
  const i64 nBytes = (nPix + 7) >> 3;
  const int lane = laneId();
  const bool maskWords = byteMask && (((size_t)byteMask) & 3u) == 0;
  const bool dataQuads = isFlt && nDepth == 1 && (((size_t)data) & (4 * sizeof(T) - 1)) == 0;
  u32 cnt = 0;
  bool sawNaN = false, sawMixed = false;
  for (i64 base = (i64)blockIdx.x * 1024; base < nBytes * 8; base += (i64)gridDim.x * 1024)    // whole waves: uniform trip count
  {
    const i64 k0 = base + (i64)threadIdx.x * 4;
    const bool whole = k0 + 3 < nPix;
    u32 mask4 = 0x01010101u;    // byte q = pixel k0 + q is valid so far
    if (byteMask)
    {
      if (whole && maskWords) mask4 = *reinterpret_cast<const u32*>(byteMask + k0);
      else { mask4 = 0; for (int q = 0; q < 4; q++) if (k0 + q < nPix && byteMask[k0 + q]) mask4 |= 1u << (8 * q); }
    }
    T vals[4] = { T(0), T(0), T(0), T(0) };
    if (dataQuads && whole && mask4)
    {
      struct alignas(4 * sizeof(T)) Quad { T v[4]; };
      const Quad qd = *reinterpret_cast<const Quad*>(data + k0);
      for (int q = 0; q < 4; q++) vals[q] = qd.v[q];
    }
    u32 nib = 0, tailNib = 0;    // bit 8 >> q: pixel k0 + q valid resp. behind the raster's end
#pragma unroll
    for (int q = 0; q < 4; q++)
    {
      const i64 k = k0 + q;
      const bool inb = k < nPix;
      bool valid = inb && ((mask4 >> (8 * q)) & 0xFFu) != 0;
      if (valid && isFlt)
      {
        int nBad = 0;
        if (dataQuads && whole) nBad = isNaNT(vals[q]) ? 1 : 0;
        else for (int m = 0; m < nDepth; m++) nBad += isNaNT(data[k * nDepth + m]) ? 1 : 0;
        if (nBad > 0) sawNaN = true;
        if (nBad == nDepth) valid = false;
        else if (nBad > 0) sawMixed = true;
      }
      if (valid) nib |= 8u >> q;
      if (!inb) tailNib |= 8u >> q;    // tail bits stay set, like BitMask::SetAllValid + SetInvalid (Lerc.cpp:959-975)
    }
    const u32 mine = nib | tailNib, other = __shfl_xor(mine, 1);
    if (!(lane & 1))
    {
      const i64 byteIdx = k0 >> 3;    // pixel 8 j + i is bit 0x80 >> i of byte j
      if (byteIdx < nBytes) maskBits[byteIdx] = (u8)((mine << 4) | other);
    }
    cnt += (u32)__popc(nib);
  }
  cnt = waveSum(cnt);
  const bool anyNaN = __any(sawNaN), anyMixed = __any(sawMixed);
  if (lane == 0)
  {
    if (cnt) atomicAdd(&stats->numValid, cnt);
    if (anyNaN) atomicOr(&stats->hasNaN, 1u);
    if (anyMixed) atomicOr(&stats->mixedNaN, 1u);
  }
}

// One value per pixel, everything 16-byte aligned, a whole number of 16-pixel groups: a lane takes 16 pixels -- one 16-byte load
// of the byte mask, the pixels themselves (float types: NaN is "not valid", Lerc.cpp:959-975) in 16-byte loads -- and writes two
// bytes of bits.  The form above (four pixels a lane, a byte store from every other lane) runs at 1.5 TB/s on an 8192 x 8192 band.
template<class T>
__global__ void __launch_bounds__(256) k_build_mask16(const T* __restrict__ data, const u8* __restrict__ byteMask, i64 nGroups,
                                                      u8* __restrict__ maskBits, BandStats* stats)
{
  constexpr bool isFlt = DtOf<T>::v >= DT_Float;
  constexpr int PV = 16 / (int)sizeof(T);    // pixels per 16-byte load of the data
  struct alignas(16) Vec { T v[PV]; };
  u32 cnt = 0;
  bool sawNaN = false;
  for (i64 g = (i64)blockIdx.x * 256 + threadIdx.x; g < nGroups; g += (i64)gridDim.x * 256)
  {
    u32 bits = 0xFFFFu;    // bit q: pixel 16 g + q valid
    if (byteMask)
    {
      const uint4 m = reinterpret_cast<const uint4*>(byteMask)[g];
      const u32 w[4] = { m.x, m.y, m.z, m.w };
      bits = 0;
#pragma unroll
      for (int q = 0; q < 16; q++) if ((w[q >> 2] >> (8 * (q & 3))) & 0xFFu) bits |= 1u << q;
    }
    if (isFlt && bits)
    {
      const Vec* src = reinterpret_cast<const Vec*>(data) + g * (16 / PV);
#pragma unroll
      for (int j = 0; j < 16 / PV; j++)
      {
        if (!((bits >> (j * PV)) & ((1u << PV) - 1u))) continue;
        const Vec x = src[j];
#pragma unroll
        for (int q = 0; q < PV; q++)
          if (((bits >> (j * PV + q)) & 1u) && isNaNT(x.v[q])) { sawNaN = true; bits &= ~(1u << (j * PV + q)); }
      }
    }
    cnt += (u32)__popc(bits);
    // pixel 8 j + i is bit 0x80 >> i of byte j
    const u32 lo = __brev(bits & 0xFFu) >> 24, hi = __brev((bits >> 8) & 0xFFu) >> 24;
    reinterpret_cast<u16*>(maskBits)[g] = (u16)(lo | (hi << 8));
  }
  // (one addition per workgroup: sixteen thousand waves adding to one address took longer than reading the band)
  __shared__ u32 s_cnt, s_nan;
  if (threadIdx.x == 0) { s_cnt = 0u; s_nan = 0u; }
  __syncthreads();
  cnt = waveSum(cnt);
  const bool anyNaN = __any(sawNaN);
  if (laneId() == 0) { if (cnt) atomicAdd(&s_cnt, cnt); if (anyNaN) s_nan = 1u; }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    if (s_cnt) atomicAdd(&stats->numValid, s_cnt);
    if (s_nan && !__hip_atomic_load(&stats->hasNaN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats->hasNaN, 1u);
  }
}

void launchBuildMask(int dt, const void* data, const u8* byteMask, int nRows, int nCols, int nDepth, u8* maskBits,
                     BandStats* stats, hipStream_t stream)
{
  const i64 nPix = (i64)nRows * nCols;
  if (nDepth == 1 && (nPix & 15) == 0 && nPix >= 4096 && ((uintptr_t)data & 15) == 0 && ((uintptr_t)byteMask & 15) == 0 && ((uintptr_t)maskBits & 1) == 0)
  {
    const i64 nGroups = nPix >> 4;
    const dim3 gv((unsigned)std::min<i64>((nGroups + 256 * 4 - 1) / (256 * 4), 2048)), bv(256);
    switch (dt)
    {
      case DT_Float: hipLaunchKernelGGL(k_build_mask16<float>, gv, bv, 0, stream, (const float*)data, byteMask, nGroups, maskBits, stats); break;
      case DT_Double: hipLaunchKernelGGL(k_build_mask16<double>, gv, bv, 0, stream, (const double*)data, byteMask, nGroups, maskBits, stats); break;
      default: hipLaunchKernelGGL(k_build_mask16<u8>, gv, bv, 0, stream, (const u8*)data, byteMask, nGroups, maskBits, stats); break;
    }
    return;
  }
  const dim3 grid((unsigned)std::min<i64>((nPix + 1023) / 1024, 4096)), block(256);
  switch (dt)
  {
    case DT_Float: hipLaunchKernelGGL(k_build_mask<float>, grid, block, 0, stream, (const float*)data, byteMask, nPix, nDepth, maskBits, stats); break;
    case DT_Double: hipLaunchKernelGGL(k_build_mask<double>, grid, block, 0, stream, (const double*)data, byteMask, nPix, nDepth, maskBits, stats); break;
    default: hipLaunchKernelGGL(k_build_mask<u8>, grid, block, 0, stream, (const u8*)data, byteMask, nPix, 1, maskBits, stats); break;
  }
}

__global__ void __launch_bounds__(256) k_block_valid_counts(const u8* __restrict__ maskBits, BandParams p, u16* __restrict__ nValidBlk)
{
  const int pos = (int)(blockIdx.x * 256 + threadIdx.x);
  if (pos >= p.nTV * p.nTH) return;
  const int it = pos / p.nTH, jt = pos - it * p.nTH;
  const int i0 = it * p.mb, j0 = jt * p.mb;
  const int i1 = min(p.nRows, i0 + p.mb), j1 = min(p.nCols, j0 + p.mb);
  int n = 0;
  for (int i = i0; i < i1; i++)
    for (int j = j0; j < j1; j++) n += (!maskBits || maskBit(maskBits, (i64)i * p.nCols + j)) ? 1 : 0;
  nValidBlk[pos] = (u16)n;
}

void launchBlockValidCounts(const u8* maskBits, const BandParams& p, u16* nValidBlk, hipStream_t stream)
{
  const int nPos = p.nTV * p.nTH;
  hipLaunchKernelGGL(k_block_valid_counts, dim3((nPos + 255) / 256), dim3(256), 0, stream, maskBits, p, nValidBlk);
}

__global__ void __launch_bounds__(256) k_bits_to_bytes(const u8* __restrict__ maskBits, u8* __restrict__ byteMask, i64 nPix)
{
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k < nPix) byteMask[k] = maskBits ? (maskBit(maskBits, k) ? 1 : 0) : 1;
}

// eight pixels (one byte of bits) per thread and one 8-byte store, for byte masks that are 8-byte aligned
__global__ void __launch_bounds__(256) k_bits_to_bytes8(const u8* __restrict__ maskBits, u8* __restrict__ byteMask, i64 nGroups)
{
  const i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
  if (g >= nGroups) return;
  const u32 x = maskBits[g];
  u64 v = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) v |= (u64)((x >> (7 - i)) & 1u) << (8 * i);
  reinterpret_cast<u64*>(byteMask)[g] = v;
}

void launchBitsToBytes(const u8* maskBits, u8* byteMask, i64 nPix, hipStream_t stream)
{
  if (maskBits && ((uintptr_t)byteMask & 7u) == 0u && nPix >= 8)
  {
    const i64 nGroups = nPix >> 3, done = nGroups << 3;
    hipLaunchKernelGGL(k_bits_to_bytes8, dim3((unsigned)((nGroups + 255) / 256)), dim3(256), 0, stream, maskBits, byteMask, nGroups);
    if (done < nPix)    // (the last byte's few pixels: pixel k of the tail is pixel done + k of the raster, done is a multiple of 8)
      hipLaunchKernelGGL(k_bits_to_bytes, dim3(1), dim3(256), 0, stream, maskBits + (done >> 3), byteMask + done, nPix - done);
    return;
  }
  hipLaunchKernelGGL(k_bits_to_bytes, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, stream, maskBits, byteMask, nPix);
}

// ================================================================================================
// per-depth min / max (+ float diagnostics)
// Order-preserving 64-bit keys so that integer atomics implement min / max for every type.
// ================================================================================================
template<class T> struct Key;
template<> struct Key<signed char>    { static __device__ __host__ u64 enc(signed char v) { return (u64)((i64)v + (1ll << 62)); } };
template<> struct Key<unsigned char>  { static __device__ __host__ u64 enc(unsigned char v) { return (u64)v + (1ull << 62); } };
template<> struct Key<short>          { static __device__ __host__ u64 enc(short v) { return (u64)((i64)v + (1ll << 62)); } };
template<> struct Key<unsigned short> { static __device__ __host__ u64 enc(unsigned short v) { return (u64)v + (1ull << 62); } };
template<> struct Key<int>            { static __device__ __host__ u64 enc(int v) { return (u64)((i64)v + (1ll << 62)); } };
template<> struct Key<unsigned int>   { static __device__ __host__ u64 enc(unsigned int v) { return (u64)v + (1ull << 62); } };
template<> struct Key<float>
{
  static __device__ __host__ u64 enc(float v) { u32 b; memcpy(&b, &v, 4); b = (b & 0x80000000u) ? ~b : (b | 0x80000000u); return b; }
};
template<> struct Key<double>
{
  static __device__ __host__ u64 enc(double v) { u64 b; memcpy(&b, &v, 8); return (b >> 63) ? ~b : (b | (1ull << 63)); }
};

u64 statKeyInitMin() { return ~0ull; }
u64 statKeyInitMax() { return 0ull; }

// inverse of Key<T>::enc: returns the raw little-endian bits of the T value (low dtSize bytes)
u64 statKeyToRawBits(int dt, u64 key)
{
  switch (dt)
  {
    case DT_Float: { u32 b = (u32)key; b = (b & 0x80000000u) ? (b & 0x7fffffffu) : ~b; return b; }
    case DT_Double: { return (key >> 63) ? (key & ~(1ull << 63)) : ~key; }
    default: { const i64 v = (i64)key - (1ll << 62); return (u64)v; }
  }
}

double statKeyToDouble(int dt, u64 key)
{
  const u64 raw = statKeyToRawBits(dt, key);
  switch (dt)
  {
    case DT_Float: { u32 b = (u32)raw; float f; memcpy(&f, &b, 4); return (double)f; }
    case DT_Double: { double d; memcpy(&d, &raw, 8); return d; }
    default: return (double)(i64)raw;
  }
}

static const int kStatsMaxDepthLds = 256;

template<class T>
__global__ void __launch_bounds__(256)
k_band_stats(const T* __restrict__ data, const u8* __restrict__ maskBits, i64 nPix, int nDepth, double maxZErr, u32 raiseMask,
             u64* __restrict__ mins, u64* __restrict__ maxs, BandStats* stats)
{
  __shared__ u64 s_min[kStatsMaxDepthLds], s_max[kStatsMaxDepthLds];
  __shared__ u64 s_raise[9];
  constexpr bool isFlt = (DtOf<T>::v >= DT_Float);
  const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
  const bool ldsRanges = nDepth <= kStatsMaxDepthLds;    // (more values per pixel than that: straight to the global ranges)
  for (int i = threadIdx.x; i < nDepth && ldsRanges; i += 256) { s_min[i] = ~0ull; s_max[i] = 0ull; }
  if (threadIdx.x < 9) s_raise[threadIdx.x] = 0ull;
  __syncthreads();

  // The grid's stride is cut to a multiple of nDepth (the few threads beyond it idle), so that a thread stays with one
  // of the nDepth values of a pixel and keeps its range in registers; its pixel index advances by a constant.
  const i64 nElem = nPix * nDepth;
  const i64 first = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 stride = (i64)gridDim.x * 256;
  if (nDepth > 1) stride -= stride % nDepth;    // (the launch has at least nDepth threads)
  const int m = (nDepth > 1) ? (int)(first % nDepth) : 0;
  const i64 kStep = stride / nDepth;
  bool sawNaN = false, sawFrac = false;
  u64 kMin = ~0ull, kMax = 0ull;
  double rerr[9];
#pragma unroll
  for (int c = 0; c < 9; c++) rerr[c] = 0;

  i64 k = (nDepth > 1) ? first / nDepth : first;
  for (i64 t = first; t < nElem && first < stride; t += stride, k += kStep)
  {
    if (maskBits && !maskBit(maskBits, k)) continue;
    const T v = data[t];
    if (isFlt && isNaNT(v)) { sawNaN = true; continue; }
    const u64 key = Key<T>::enc(v);
    kMin = key < kMin ? key : kMin; kMax = key > kMax ? key : kMax;
    if (isFlt)
    {
      const double x = (double)v;
      if (!sawFrac && !(v == (T)floor(x + 0.5))) sawFrac = true;    // Lerc.h:271 IsInt
      if (raiseMask)
      {
        // Lerc2.cpp:1269-1276: candidates in increasing factor order, stop at the first exact hit
#pragma unroll
        for (int c = 0; c < 9; c++)
        {
          if (!((raiseMask >> c) & 1u)) continue;
          const double z = x * facCand[c];
          if (z == (double)(int)z) break;
          const double dlt = fabs(floor(z + 0.5) - z);
          rerr[c] = dlt > rerr[c] ? dlt : rerr[c];
        }
      }
    }
  }
  if (nDepth == 1)
  {
    kMin = waveMin(kMin); kMax = waveMax(kMax);
    if (laneId() == 0) { atomicMin(&s_min[0], kMin); atomicMax(&s_max[0], kMax); }
  }
  else if (kMin <= kMax)
  {
    if (ldsRanges) { atomicMin(&s_min[m], kMin); atomicMax(&s_max[m], kMax); }
    else { atomicMin(&mins[m], kMin); atomicMax(&maxs[m], kMax); }
  }
  if (isFlt && raiseMask)
  {
#pragma unroll
    for (int c = 0; c < 9; c++)
    {
      u64 b; double r = rerr[c]; memcpy(&b, &r, 8);    // non-negative doubles order like their bit patterns
      b = waveMax(b);
      if (laneId() == 0 && b) atomicMax(&s_raise[c], b);
    }
  }
  const bool anyNaN = __any(sawNaN), anyFrac = __any(sawFrac);
  if (laneId() == 0)
  {
    if (anyNaN) atomicOr(&stats->hasNaN, 1u);
    if (anyFrac) atomicOr(&stats->notAllInt, 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nDepth && ldsRanges; i += 256)
  {
    if (s_min[i] != ~0ull) atomicMin(&mins[i], s_min[i]);
    if (s_max[i] != 0ull) atomicMax(&maxs[i], s_max[i]);
  }
  if (threadIdx.x < 9 && s_raise[threadIdx.x])
    atomicMax(reinterpret_cast<u64*>(&stats->raiseErr[threadIdx.x]), s_raise[threadIdx.x]);
  (void)maxZErr;
}

// One value per pixel, 16 bits a value or more, rows of whole vectors: 16 bytes of pixels per load and their validity bits by
// the byte -- the form above, a value per lane and step with a byte load of the mask for each, reads an 8192 x 8192 float32
// band at 0.9 TB/s, and with the block stream of masked bands down to 170 us it was the longest kernel of a masked encode.
template<class T>
__global__ void __launch_bounds__(256)
k_band_stats_vec(const T* __restrict__ data, const u8* __restrict__ maskBits, i64 nVec, u32 raiseMask,
                 u64* __restrict__ mins, u64* __restrict__ maxs, BandStats* stats)
{
  constexpr int V = 16 / (int)sizeof(T);    // 2, 4 or 8 pixels: a whole number of them per mask byte
  constexpr bool isFlt = (DtOf<T>::v >= DT_Float);
  __shared__ u64 s_min, s_max, s_raise[9];
  __shared__ u32 s_flags;
  const int facCand[9] = { 1, 2, 10, 20, 100, 200, 1000, 2000, 10000 };
  if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_flags = 0u; }
  if (threadIdx.x < 9) s_raise[threadIdx.x] = 0ull;
  __syncthreads();
  bool sawNaN = false, sawFrac = false;
  u64 kMin = ~0ull, kMax = 0ull;
  double rerr[9];
#pragma unroll
  for (int c = 0; c < 9; c++) rerr[c] = 0;
  struct alignas(16) Vec { T v[V]; };
  const Vec* vec = reinterpret_cast<const Vec*>(data);
  for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < nVec; t += (i64)gridDim.x * 256)
  {
    const i64 k0 = t * V;    // first pixel of the vector
    u32 vm = (1u << V) - 1u;
    if (maskBits) vm = (__brev(((u32)maskBits[k0 >> 3] << ((u32)k0 & 7u)) & 0xFFu) >> 24) & ((1u << V) - 1u);    // bit q: pixel k0 + q
    if (!vm) continue;
    const Vec x = vec[t];
#pragma unroll
    for (int q = 0; q < V; q++)
    {
      if (!((vm >> q) & 1u)) continue;
      const T v = x.v[q];
      if (isFlt && isNaNT(v)) { sawNaN = true; continue; }
      const u64 key = Key<T>::enc(v);
      kMin = key < kMin ? key : kMin; kMax = key > kMax ? key : kMax;
      if (isFlt)
      {
        const double xd = (double)v;
        if (!sawFrac && !(v == (T)floor(xd + 0.5))) sawFrac = true;    // Lerc.h:271 IsInt
        if (raiseMask)
        {
#pragma unroll
          for (int c = 0; c < 9; c++)    // Lerc2.cpp:1269-1276: candidates in increasing factor order, stop at the first exact hit
          {
            if (!((raiseMask >> c) & 1u)) continue;
            const double z = xd * facCand[c];
            if (z == (double)(int)z) break;
            const double dlt = fabs(floor(z + 0.5) - z);
            rerr[c] = dlt > rerr[c] ? dlt : rerr[c];
          }
        }
      }
    }
  }
  kMin = waveMin(kMin); kMax = waveMax(kMax);
  if (laneId() == 0) { atomicMin(&s_min, kMin); atomicMax(&s_max, kMax); }
  if (isFlt && raiseMask)
  {
#pragma unroll
    for (int c = 0; c < 9; c++)
    {
      u64 b; double r = rerr[c]; memcpy(&b, &r, 8);    // non-negative doubles order like their bit patterns
      b = waveMax(b);
      if (laneId() == 0 && b) atomicMax(&s_raise[c], b);
    }
  }
  // (the two flags: one atomic per WORKGROUP, and none once the flag stands -- sixteen thousand waves adding to one address
  // were most of this kernel's time)
  const bool anyNaN = __any(sawNaN), anyFrac = __any(sawFrac);
  if (laneId() == 0 && (anyNaN || anyFrac)) atomicOr(&s_flags, (anyNaN ? 1u : 0u) | (anyFrac ? 2u : 0u));
  __syncthreads();
  if (threadIdx.x == 0)
  {
    if (s_min != ~0ull && s_min < __hip_atomic_load(&mins[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[0], s_min);
    if (s_max != 0ull && s_max > __hip_atomic_load(&maxs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[0], s_max);
    if ((s_flags & 1u) && !__hip_atomic_load(&stats->hasNaN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats->hasNaN, 1u);
    if ((s_flags & 2u) && !__hip_atomic_load(&stats->notAllInt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats->notAllInt, 1u);
  }
  if (threadIdx.x < 9 && s_raise[threadIdx.x])
    atomicMax(reinterpret_cast<u64*>(&stats->raiseErr[threadIdx.x]), s_raise[threadIdx.x]);
}

// 8-bit values, every pixel valid: 16 bytes per load; a thread's loads are a multiple of nDepth vectors apart, so byte b
// of every vector it sees belongs to the same value of a pixel, and it keeps sixteen running ranges that are sorted into
// the per-depth ranges once at the end.
template<class T>
__global__ void __launch_bounds__(256)
k_band_stats_bytes(const T* __restrict__ data, i64 nVec, int nDepth, u64* __restrict__ mins, u64* __restrict__ maxs)
{
  __shared__ int s_min[kStatsMaxDepthLds], s_max[kStatsMaxDepthLds];
  for (int i = threadIdx.x; i < nDepth; i += 256) { s_min[i] = 0x7FFFFFFF; s_max[i] = -0x7FFFFFFF; }
  __syncthreads();
  const i64 first = (i64)blockIdx.x * 256 + threadIdx.x;
  i64 stride = (i64)gridDim.x * 256;
  stride -= stride % nDepth;    // (the launch has at least nDepth threads)
  int mn[16], mx[16];
#pragma unroll
  for (int b = 0; b < 16; b++) { mn[b] = 0x7FFFFFFF; mx[b] = -0x7FFFFFFF; }
  const uint4* vec = reinterpret_cast<const uint4*>(data);
  if (first < stride)
    for (i64 t = first; t < nVec; t += stride)
    {
      const uint4 x = vec[t];
      const u32 w[4] = { x.x, x.y, x.z, x.w };
#pragma unroll
      for (int b = 0; b < 16; b++)
      {
        const int v = (int)(T)(u8)(w[b >> 2] >> (8 * (b & 3)));
        mn[b] = v < mn[b] ? v : mn[b]; mx[b] = v > mx[b] ? v : mx[b];
      }
    }
  const int phase = (int)((first * 16) % nDepth);
#pragma unroll
  for (int b = 0; b < 16; b++)
    if (mn[b] <= mx[b])
    {
      // (an atomic only where it changes something: sixteen of them a thread on nDepth addresses, and the workgroups' on the band's,
      // are otherwise what this kernel's time is made of)
      const int m = (phase + b) % nDepth;
      if (mn[b] < *(volatile int*)&s_min[m]) atomicMin(&s_min[m], mn[b]);
      if (mx[b] > *(volatile int*)&s_max[m]) atomicMax(&s_max[m], mx[b]);
    }
  __syncthreads();
  for (int i = threadIdx.x; i < nDepth; i += 256)
    if (s_min[i] <= s_max[i])
    {
      const u64 kMin = Key<T>::enc((T)s_min[i]), kMax = Key<T>::enc((T)s_max[i]);
      if (kMin < __hip_atomic_load(&mins[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[i], kMin);
      if (kMax > __hip_atomic_load(&maxs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[i], kMax);
    }
}

void launchBandStats(int dt, const void* data, const u8* maskBits, int nRows, int nCols, int nDepth, u32 raiseMask,
                     u64* mins, u64* maxs, BandStats* stats, hipStream_t stream)
{
  const double maxZErr = 0;
  const i64 nPix = (i64)nRows * nCols;
  const i64 nElem = nPix * nDepth;
  const i64 perBlock = nElem < (1 << 20) ? 256 : 256 * 16;    // small inputs (e.g. one row): one element per thread
  i64 nBlocks = (nElem + perBlock - 1) / perBlock;
  if (nBlocks > 4096) nBlocks = 4096;
  if (nBlocks < ((i64)nDepth + 255) / 256) nBlocks = ((i64)nDepth + 255) / 256;    // a thread per value of a pixel at least
  if (nBlocks < 1) nBlocks = 1;
  const dim3 grid((unsigned)nBlocks), block(256);
  if ((dt == DT_Char || dt == DT_Byte) && !maskBits && nDepth <= kStatsMaxDepthLds && (nElem & 15) == 0 && nElem >= (1 << 12) && ((uintptr_t)data & 15) == 0
      && nBlocks * 256 >= nDepth)
  {
    const i64 nVec = nElem >> 4;
    i64 nb = (nVec + 256 * 8 - 1) / (256 * 8);
    nb = nb > 4096 ? 4096 : (nb < ((i64)nDepth + 255) / 256 ? ((i64)nDepth + 255) / 256 : nb);
    if (dt == DT_Char) hipLaunchKernelGGL(k_band_stats_bytes<signed char>, dim3((unsigned)nb), block, 0, stream, (const signed char*)data, nVec, nDepth, mins, maxs);
    else hipLaunchKernelGGL(k_band_stats_bytes<unsigned char>, dim3((unsigned)nb), block, 0, stream, (const unsigned char*)data, nVec, nDepth, mins, maxs);
    return;
  }
  if (nDepth == 1 && dt > DT_Byte && nElem >= 4096 && ((uintptr_t)data & 15) == 0 && (nElem * dtSize(dt)) % 16 == 0)
  {
    const i64 nVec = nElem * dtSize(dt) / 16;
    // (small inputs -- the one row TryRaiseMaxZError's candidates are pruned on: a vector per thread, or a single workgroup
    // tests nine candidates on every element of the row by itself: 45 us for 8192 floats)
    const i64 perWG = nVec < (1 << 16) ? 256 : 256 * 8;
    const dim3 gv((unsigned)std::min<i64>((nVec + perWG - 1) / perWG, 4096));
    switch (dt)
    {
      case DT_Short:  hipLaunchKernelGGL(k_band_stats_vec<short>, gv, block, 0, stream, (const short*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
      case DT_UShort: hipLaunchKernelGGL(k_band_stats_vec<unsigned short>, gv, block, 0, stream, (const unsigned short*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
      case DT_Int:    hipLaunchKernelGGL(k_band_stats_vec<int>, gv, block, 0, stream, (const int*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
      case DT_UInt:   hipLaunchKernelGGL(k_band_stats_vec<unsigned int>, gv, block, 0, stream, (const unsigned int*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
      case DT_Float:  hipLaunchKernelGGL(k_band_stats_vec<float>, gv, block, 0, stream, (const float*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
      default:        hipLaunchKernelGGL(k_band_stats_vec<double>, gv, block, 0, stream, (const double*)data, maskBits, nVec, raiseMask, mins, maxs, stats); break;
    }
    return;
  }
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_band_stats<signed char>, grid, block, 0, stream, (const signed char*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_Byte:   hipLaunchKernelGGL(k_band_stats<unsigned char>, grid, block, 0, stream, (const unsigned char*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_Short:  hipLaunchKernelGGL(k_band_stats<short>, grid, block, 0, stream, (const short*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_UShort: hipLaunchKernelGGL(k_band_stats<unsigned short>, grid, block, 0, stream, (const unsigned short*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_Int:    hipLaunchKernelGGL(k_band_stats<int>, grid, block, 0, stream, (const int*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_UInt:   hipLaunchKernelGGL(k_band_stats<unsigned int>, grid, block, 0, stream, (const unsigned int*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_Float:  hipLaunchKernelGGL(k_band_stats<float>, grid, block, 0, stream, (const float*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    case DT_Double: hipLaunchKernelGGL(k_band_stats<double>, grid, block, 0, stream, (const double*)data, maskBits, nPix, nDepth, maxZErr, raiseMask, mins, maxs, stats); break;
    default: break;
  }
}

// ================================================================================================
// Mask AND statistics in one read of the band (a band with a byte mask, one value a pixel, 16- and 32-bit types): what k_build_mask16
// and k_band_stats_vec make in a read of the band each -- the bit mask (minus pixels that are NaN), the count of valid pixels, the
// range of the valid pixels, "not all integers" -- for the case that no TryRaiseMaxZError candidate survives the first row (then
// the statistics need nothing else, Lerc2.cpp:1233-1318; else launchBandStats runs as before).  A wave takes 1024 / 2048 pixels a
// round: every load instruction reads the band (16 bytes a lane) and the byte mask (the same pixels' bytes) on end, the lanes
// whose bits share a 32-bit word of the bit mask (pixel 8 j + i is bit 0x80 >> i of byte j) OR theirs together and one of them stores.
// The range in the pixels' own type (float: v_min / v_max; NaN never gets there), keys at the very end.
// ================================================================================================
template<class T>
__global__ void __launch_bounds__(1024)
k_mask_stats(const T* __restrict__ data, const u8* __restrict__ byteMask, i64 nVec, u32* __restrict__ maskWords,
             u64* __restrict__ mins, u64* __restrict__ maxs, BandStats* stats)
{
  constexpr int V = 16 / (int)sizeof(T);           // 4 or 8 pixels a vector
  constexpr int L = 32 / V;                        // lanes a word of the bit mask
  constexpr bool isFlt = (DtOf<T>::v >= DT_Float);
  static_assert(V == 4 || V == 8, "16- and 32-bit types");
  struct alignas(16) Vec { T v[V]; };
  struct alignas(V) MaskBytes { u8 b[V]; };
  __shared__ u64 s_min, s_max;
  __shared__ u32 s_cnt, s_flags;
  if (threadIdx.x == 0) { s_min = ~0ull; s_max = 0ull; s_cnt = 0u; s_flags = 0u; }
  __syncthreads();
  const int lane = laneId();
  const Vec* vec = reinterpret_cast<const Vec*>(data);
  const MaskBytes* mb = reinterpret_cast<const MaskBytes*>(byteMask);
  T mn = std::numeric_limits<T>::max(), mx = std::numeric_limits<T>::lowest();
  if (isFlt) { mn = (T)__builtin_huge_valf(); mx = (T)(-__builtin_huge_valf()); }
  bool sawNaN = false, sawFrac = false;
  u32 cnt = 0;
  // (workgroups of sixteen waves, two to a CU: the chip is full of waves -- a wave has a round's loads in flight, no more -- and the
  // workgroups' atomics on the statistics' addresses, 40 ns each across the XCDs, are five hundred instead of four thousand)
  const i64 nWaves = blockDim.x / 64;
  const i64 waveStride = (i64)gridDim.x * nWaves * 256;    // vectors: a wave takes 4 x 64 a round
  for (i64 base = ((i64)blockIdx.x * nWaves + waveId()) * 256; base < nVec; base += waveStride)
  {
    Vec x[4];
    MaskBytes m[4];
#pragma unroll
    for (int j = 0; j < 4; j++)    // (all loads of the round in flight together)
    {
      const i64 t = base + j * 64 + lane;
      if (t < nVec) { m[j] = mb[t]; x[j] = vec[t]; }
      else { for (int q = 0; q < V; q++) { m[j].b[q] = 0; x[j].v[q] = T(0); } }
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
    {
      const i64 t = base + j * 64 + lane;
      u32 bits = 0;    // pixel q of the vector: bit 7 - q (V == 8), bit 3 - q (V == 4)
#pragma unroll
      for (int q = 0; q < V; q++)
      {
        bool valid = m[j].b[q] != 0;
        const T v = x[j].v[q];
        if (isFlt && valid && isNaNT(v)) { sawNaN = true; valid = false; }    // (one value a pixel: NaN is "not valid", Lerc.cpp:959-975)
        if (valid)
        {
          bits |= 1u << (V - 1 - q);
          mn = v < mn ? v : mn; mx = v > mx ? v : mx;
          if (isFlt) sawFrac = sawFrac || !(v == (T)__builtin_truncf((float)v));    // Lerc.h:271 IsInt: (T)floor(x + 0.5) == x, i.e. x is a whole number
        }
      }
      cnt += (u32)__popc(bits);
      // the word's other lanes: lane i of L holds byte i (V == 8) or a nibble of byte i / 2, the high one first (V == 4)
      const int i = lane & (L - 1);
      u32 word = V == 8 ? bits << (8 * i) : bits << (8 * (i >> 1) + ((i & 1) ? 0 : 4));
      word |= __shfl_xor(word, 1);
      word |= __shfl_xor(word, 2);
      if (L == 8) word |= __shfl_xor(word, 4);
      if (i == 0 && t < nVec) maskWords[t / L] = word;
    }
  }
  {
    u64 kMin = waveMin(Key<T>::enc(mn)), kMax = waveMax(Key<T>::enc(mx));
    cnt = waveSum(cnt);
    const bool anyNaN = __any(sawNaN), anyFrac = __any(sawFrac);
    if (lane == 0)
    {
      atomicMin(&s_min, kMin); atomicMax(&s_max, kMax);
      if (cnt) atomicAdd(&s_cnt, cnt);
      if (anyNaN || anyFrac) atomicOr(&s_flags, (anyNaN ? 1u : 0u) | (anyFrac ? 2u : 0u));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)    // (one addition a workgroup: sixteen thousand waves adding to one address take longer than reading the band)
  {
    if (s_cnt)
    {
      atomicAdd(&stats->numValid, s_cnt);
      // (a workgroup without a valid pixel has no range; one whose range lies inside what stands already adds nothing -- atomics of
      // all workgroups on one address, across the XCDs, are what such a kernel's time is made of)
      if (s_min < __hip_atomic_load(&mins[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mins[0], s_min);
      if (s_max > __hip_atomic_load(&maxs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&maxs[0], s_max);
    }
    if ((s_flags & 1u) && !__hip_atomic_load(&stats->hasNaN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats->hasNaN, 1u);
    if ((s_flags & 2u) && !__hip_atomic_load(&stats->notAllInt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&stats->notAllInt, 1u);
  }
}

// true: enqueued (the band qualifies).  maskBits, mins / maxs (initialised: launchStatsInit), stats (zeroed) as for the two kernels it stands for
bool launchMaskStats(int dt, const void* data, const u8* byteMask, int nRows, int nCols, int nDepth, u8* maskBits,
                     u64* mins, u64* maxs, BandStats* stats, hipStream_t stream)
{
  static const bool on = []() { const char* e = getenv("LERC_AMD_MASK_STATS"); return !e || atoi(e) != 0; }();
  const i64 nPix = (i64)nRows * nCols;
  const int tb = dtSize(dt);
  if (!on || !byteMask || nDepth != 1 || (tb != 2 && tb != 4) || (nPix & 31) != 0 || nPix < 4096 || ((uintptr_t)data & 15) != 0 || ((uintptr_t)byteMask & 15) != 0
      || ((uintptr_t)maskBits & 3) != 0)
    return false;
  const i64 nVec = nPix * tb / 16;
  static const int maxWG = []() { const char* e = getenv("LERC_AMD_MASK_STATS_WG"); return e ? std::max(1, atoi(e)) : 512; }();
  const dim3 grid((unsigned)std::min<i64>((nVec + 4095) / 4096, maxWG)), block(1024);
  switch (dt)
  {
    case DT_Short:  hipLaunchKernelGGL(k_mask_stats<short>, grid, block, 0, stream, (const short*)data, byteMask, nVec, (u32*)maskBits, mins, maxs, stats); break;
    case DT_UShort: hipLaunchKernelGGL(k_mask_stats<unsigned short>, grid, block, 0, stream, (const unsigned short*)data, byteMask, nVec, (u32*)maskBits, mins, maxs, stats); break;
    case DT_Int:    hipLaunchKernelGGL(k_mask_stats<int>, grid, block, 0, stream, (const int*)data, byteMask, nVec, (u32*)maskBits, mins, maxs, stats); break;
    case DT_UInt:   hipLaunchKernelGGL(k_mask_stats<unsigned int>, grid, block, 0, stream, (const unsigned int*)data, byteMask, nVec, (u32*)maskBits, mins, maxs, stats); break;
    case DT_Float:  hipLaunchKernelGGL(k_mask_stats<float>, grid, block, 0, stream, (const float*)data, byteMask, nVec, (u32*)maskBits, mins, maxs, stats); break;
    default: return false;
  }
  return true;
}

// ================================================================================================
// noData values (Lerc::FilterNoDataAndNaN with a noData value, Lerc.cpp:1378-1552; integer types :1241-1374):
// one sweep over a private copy of the band.  Per valid pixel: count the depth values that are the noData value (or
// NaN); all of them -> the pixel leaves the mask; some -> the blob will have to carry a noData value.  NaNs turn into
// the noData value (nDepth > 1) or 0 (nDepth 1).  Min / max / "all integers" over the remaining values.
// ================================================================================================
template<class T>
__global__ void __launch_bounds__(256)
k_nodata_scan(T* __restrict__ data, u8* __restrict__ maskBytes, i64 nPix, int nDepth, T orig, NoDataScan* res)
{
  constexpr bool isFlt = (DtOf<T>::v >= DT_Float);
  u64 kMin = ~0ull, kMax = 0ull;
  u32 flags = 0;    // 1 NaN seen, 2 noData left in a valid pixel, 4 mask modified, 8 a fractional value seen
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < nPix; k += (i64)gridDim.x * 256)
  {
    if (!maskBytes[k]) continue;
    int bad = 0;
    for (int m = 0; m < nDepth; m++)
    {
      const T z = data[k * nDepth + m];
      if (isFlt && isNaNT(z))
      {
        flags |= 1u; bad++;
        if (nDepth > 1) data[k * nDepth + m] = orig; else data[k * nDepth + m] = T(0);
      }
      else if (z == orig) bad++;
      else
      {
        const u64 key = Key<T>::enc(z);
        kMin = key < kMin ? key : kMin; kMax = key > kMax ? key : kMax;
        if (isFlt && !(z == (T)floor((double)z + 0.5))) flags |= 8u;    // Lerc.h:271 IsInt
      }
    }
    if (bad == nDepth) { maskBytes[k] = 0; flags |= 4u; }
    else if (bad > 0) flags |= 2u;
  }
  kMin = waveMin(kMin); kMax = waveMax(kMax);
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) flags |= __shfl_xor(flags, d);
  if (laneId() == 0)
  {
    if (kMin != ~0ull) atomicMin(&res->minKey, kMin);
    if (kMax != 0ull) atomicMax(&res->maxKey, kMax);
    if (flags) atomicOr(&res->flags, flags);
  }
}

// data == from -> to, in valid pixels only (byte mask, bit mask, or neither)
template<class T>
__global__ void __launch_bounds__(256)
k_nodata_remap(T* __restrict__ data, const u8* __restrict__ maskBytes, const u8* __restrict__ maskBits, i64 nPix, int nDepth, T from, T to)
{
  for (i64 k = (i64)blockIdx.x * 256 + threadIdx.x; k < nPix; k += (i64)gridDim.x * 256)
  {
    if (maskBytes && !maskBytes[k]) continue;
    if (maskBits && !maskBit(maskBits, k)) continue;
    for (int m = 0; m < nDepth; m++) if (data[k * nDepth + m] == from) data[k * nDepth + m] = to;
  }
}

void launchNoDataScan(int dt, void* data, u8* maskBytes, i64 nPix, int nDepth, double orig, NoDataScan* res, hipStream_t stream)
{
  NoDataScan init; init.minKey = ~0ull; init.maxKey = 0ull; init.flags = 0; init.pad = 0;
  hipMemcpyAsync(res, &init, sizeof(init), hipMemcpyHostToDevice, stream);
  hipStreamSynchronize(stream);    // `init` is a local
  const dim3 grid((unsigned)std::min<i64>((nPix + 255) / 256, 8192)), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_nodata_scan<signed char>, grid, block, 0, stream, (signed char*)data, maskBytes, nPix, nDepth, (signed char)orig, res); break;
    case DT_Byte:   hipLaunchKernelGGL(k_nodata_scan<unsigned char>, grid, block, 0, stream, (unsigned char*)data, maskBytes, nPix, nDepth, (unsigned char)orig, res); break;
    case DT_Short:  hipLaunchKernelGGL(k_nodata_scan<short>, grid, block, 0, stream, (short*)data, maskBytes, nPix, nDepth, (short)orig, res); break;
    case DT_UShort: hipLaunchKernelGGL(k_nodata_scan<unsigned short>, grid, block, 0, stream, (unsigned short*)data, maskBytes, nPix, nDepth, (unsigned short)orig, res); break;
    case DT_Int:    hipLaunchKernelGGL(k_nodata_scan<int>, grid, block, 0, stream, (int*)data, maskBytes, nPix, nDepth, (int)orig, res); break;
    case DT_UInt:   hipLaunchKernelGGL(k_nodata_scan<unsigned int>, grid, block, 0, stream, (unsigned int*)data, maskBytes, nPix, nDepth, (unsigned int)orig, res); break;
    case DT_Float:  hipLaunchKernelGGL(k_nodata_scan<float>, grid, block, 0, stream, (float*)data, maskBytes, nPix, nDepth, (float)orig, res); break;
    case DT_Double: hipLaunchKernelGGL(k_nodata_scan<double>, grid, block, 0, stream, (double*)data, maskBytes, nPix, nDepth, orig, res); break;
    default: break;
  }
}

void launchNoDataRemap(int dt, void* data, const u8* maskBytes, const u8* maskBits, i64 nPix, int nDepth, double from, double to, hipStream_t stream)
{
  const dim3 grid((unsigned)std::min<i64>((nPix + 255) / 256, 8192)), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_nodata_remap<signed char>, grid, block, 0, stream, (signed char*)data, maskBytes, maskBits, nPix, nDepth, (signed char)from, (signed char)to); break;
    case DT_Byte:   hipLaunchKernelGGL(k_nodata_remap<unsigned char>, grid, block, 0, stream, (unsigned char*)data, maskBytes, maskBits, nPix, nDepth, (unsigned char)from, (unsigned char)to); break;
    case DT_Short:  hipLaunchKernelGGL(k_nodata_remap<short>, grid, block, 0, stream, (short*)data, maskBytes, maskBits, nPix, nDepth, (short)from, (short)to); break;
    case DT_UShort: hipLaunchKernelGGL(k_nodata_remap<unsigned short>, grid, block, 0, stream, (unsigned short*)data, maskBytes, maskBits, nPix, nDepth, (unsigned short)from, (unsigned short)to); break;
    case DT_Int:    hipLaunchKernelGGL(k_nodata_remap<int>, grid, block, 0, stream, (int*)data, maskBytes, maskBits, nPix, nDepth, (int)from, (int)to); break;
    case DT_UInt:   hipLaunchKernelGGL(k_nodata_remap<unsigned int>, grid, block, 0, stream, (unsigned int*)data, maskBytes, maskBits, nPix, nDepth, (unsigned int)from, (unsigned int)to); break;
    case DT_Float:  hipLaunchKernelGGL(k_nodata_remap<float>, grid, block, 0, stream, (float*)data, maskBytes, maskBits, nPix, nDepth, (float)from, (float)to); break;
    case DT_Double: hipLaunchKernelGGL(k_nodata_remap<double>, grid, block, 0, stream, (double*)data, maskBytes, maskBits, nPix, nDepth, from, to); break;
    default: break;
  }
}

// ================================================================================================
// bit plane statistics for the integer "bit plane" mode (Lerc2::TryBitPlaneCompression, Lerc2.cpp:1071-1229):
// for every valid pixel and its valid right / lower neighbour, per depth, how often each bit of a XOR b is set.
// counts[m * 32 + s] = tally of bit s in depth m, counts[nDepth * 32] = number of neighbour pairs.
// The all-valid nDepth == 1 case of the reference leaves the last row and column out altogether (:1091-1103).
// ================================================================================================
template<class T>
__global__ void __launch_bounds__(256)
k_bitplane_counts(const T* __restrict__ data, const u8* __restrict__ maskBits, int nRows, int nCols, int nDepth, u32* __restrict__ counts)
{
  constexpr int nBits = 8 * (int)sizeof(T);
  const i64 nPix = (i64)nRows * nCols;
  const bool simple = (nDepth == 1 && !maskBits);
  const int lane = laneId();
  u32 pairs = 0;
  for (i64 base = (i64)blockIdx.x * 256; base < nPix; base += (i64)gridDim.x * 256)    // uniform trip count: ballots inside
  {
    const i64 k = base + threadIdx.x;
    const bool in = k < nPix;
    const int i = in ? (int)(k / nCols) : 0, j = in ? (int)(k - (i64)i * nCols) : 0;
    bool hori, vert;
    if (simple) hori = vert = in && i < nRows - 1 && j < nCols - 1;
    else
    {
      const bool valid = in && (!maskBits || maskBit(maskBits, k));
      hori = valid && j < nCols - 1 && (!maskBits || maskBit(maskBits, k + 1));
      vert = valid && i < nRows - 1 && (!maskBits || maskBit(maskBits, k + nCols));
    }
    pairs += (hori ? 1u : 0u) + (vert ? 1u : 0u);
    for (int m = 0; m < nDepth; m++)
    {
      const i64 at = k * nDepth + m;
      const u32 a = (hori || vert) ? (u32)data[at] : 0u;
      const u32 xh = hori ? (a ^ (u32)data[at + nDepth]) : 0u;
      const u32 xv = vert ? (a ^ (u32)data[at + (i64)nDepth * nCols]) : 0u;
#pragma unroll
      for (int s = 0; s < nBits; s++)
      {
        const u32 c = (u32)__popcll(__ballot((xh >> s) & 1u)) + (u32)__popcll(__ballot((xv >> s) & 1u));
        if (lane == 0 && c) atomicAdd(&counts[m * 32 + s], c);
      }
    }
  }
  pairs = waveSum(pairs);
  if (lane == 0 && pairs) atomicAdd(&counts[nDepth * 32], pairs);
}

void launchBitPlaneCounts(int dt, const void* data, const u8* maskBits, int nRows, int nCols, int nDepth, u32* counts, hipStream_t stream)
{
  hipMemsetAsync(counts, 0, ((size_t)nDepth * 32 + 1) * 4, stream);
  const i64 nPix = (i64)nRows * nCols;
  const unsigned nBlocks = (unsigned)std::min<i64>((nPix + 255) / 256, 4096);
  const dim3 grid(nBlocks), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_bitplane_counts<signed char>, grid, block, 0, stream, (const signed char*)data, maskBits, nRows, nCols, nDepth, counts); break;
    case DT_Byte:   hipLaunchKernelGGL(k_bitplane_counts<unsigned char>, grid, block, 0, stream, (const unsigned char*)data, maskBits, nRows, nCols, nDepth, counts); break;
    case DT_Short:  hipLaunchKernelGGL(k_bitplane_counts<short>, grid, block, 0, stream, (const short*)data, maskBits, nRows, nCols, nDepth, counts); break;
    case DT_UShort: hipLaunchKernelGGL(k_bitplane_counts<unsigned short>, grid, block, 0, stream, (const unsigned short*)data, maskBits, nRows, nCols, nDepth, counts); break;
    case DT_Int:    hipLaunchKernelGGL(k_bitplane_counts<int>, grid, block, 0, stream, (const int*)data, maskBits, nRows, nCols, nDepth, counts); break;
    case DT_UInt:   hipLaunchKernelGGL(k_bitplane_counts<unsigned int>, grid, block, 0, stream, (const unsigned int*)data, maskBits, nRows, nCols, nDepth, counts); break;
    default: break;
  }
}

// ================================================================================================
// raw one-sweep copy / const fill / widen
// ================================================================================================
// wordBase[g] = number of valid pixels before pixel 32 * g (exclusive scan of per-group popcounts)
__global__ void __launch_bounds__(256) k_mask_group_counts(const u8* __restrict__ maskBits, i64 nPix, u32* __restrict__ counts)
{
  const i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
  const i64 nGroups = (nPix + 31) >> 5;
  if (g >= nGroups) return;
  u32 c = 0;
  for (int j = 0; j < 32; j++) { const i64 k = g * 32 + j; if (k < nPix && maskBit(maskBits, k)) c++; }
  counts[g] = c;
}

__global__ void __launch_bounds__(256)
k_one_sweep(bool encode, const u8* __restrict__ src, u8* __restrict__ dst, const u8* __restrict__ maskBits,
            const u32* __restrict__ groupBase, i64 nPix, int pixelBytes)
{
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k >= nPix) return;
  i64 r = k;
  if (maskBits)
  {
    if (!maskBit(maskBits, k)) return;
    r = groupBase[k >> 5];
    for (i64 j = (k >> 5) << 5; j < k; j++) r += maskBit(maskBits, j) ? 1 : 0;
  }
  const u8* s = src + (encode ? k : r) * pixelBytes;
  u8* d = dst + (encode ? r : k) * pixelBytes;
  for (int b = 0; b < pixelBytes; b++) d[b] = s[b];
}

void launchOneSweep(bool encode, const void* src, void* dst, const u8* maskBits, const u32* groupBase, i64 nPix,
                    int pixelBytes, hipStream_t stream)
{
  hipLaunchKernelGGL(k_one_sweep, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, stream, encode, (const u8*)src, (u8*)dst,
                     maskBits, groupBase, nPix, pixelBytes);
}

void launchMaskGroupCounts(const u8* maskBits, i64 nPix, u32* counts, hipStream_t stream)
{
  const i64 nGroups = (nPix + 31) >> 5;
  hipLaunchKernelGGL(k_mask_group_counts, dim3((unsigned)((nGroups + 255) / 256)), dim3(256), 0, stream, maskBits, nPix, counts);
}

__global__ void __launch_bounds__(256)
k_fill(u8* __restrict__ dst, const u8* __restrict__ pixel, int pixelBytes, const u8* __restrict__ maskBits, i64 nPix)
{
  const i64 k = (i64)blockIdx.x * 256 + threadIdx.x;
  if (k >= nPix) return;
  const bool valid = !maskBits || maskBit(maskBits, k);
  for (int b = 0; b < pixelBytes; b++) dst[k * pixelBytes + b] = valid ? pixel[b] : (u8)0;
}

void launchFill(void* dst, const void* pixel, int pixelBytes, const u8* maskBits, i64 nPix, hipStream_t stream)
{
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((nPix + 255) / 256)), dim3(256), 0, stream, (u8*)dst, (const u8*)pixel, pixelBytes, maskBits, nPix);
}

template<class T>
__global__ void __launch_bounds__(256) k_widen(const T* __restrict__ src, double* __restrict__ dst, i64 n)
{
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = (double)src[i];
}

void launchWidenToDouble(int dt, const void* src, double* dst, i64 n, hipStream_t stream)
{
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  switch (dt)
  {
    case DT_Char:   hipLaunchKernelGGL(k_widen<signed char>, grid, block, 0, stream, (const signed char*)src, dst, n); break;
    case DT_Byte:   hipLaunchKernelGGL(k_widen<unsigned char>, grid, block, 0, stream, (const unsigned char*)src, dst, n); break;
    case DT_Short:  hipLaunchKernelGGL(k_widen<short>, grid, block, 0, stream, (const short*)src, dst, n); break;
    case DT_UShort: hipLaunchKernelGGL(k_widen<unsigned short>, grid, block, 0, stream, (const unsigned short*)src, dst, n); break;
    case DT_Int:    hipLaunchKernelGGL(k_widen<int>, grid, block, 0, stream, (const int*)src, dst, n); break;
    case DT_UInt:   hipLaunchKernelGGL(k_widen<unsigned int>, grid, block, 0, stream, (const unsigned int*)src, dst, n); break;
    case DT_Float:  hipLaunchKernelGGL(k_widen<float>, grid, block, 0, stream, (const float*)src, dst, n); break;
    default: break;
  }
}

// ---- small moves by kernels instead of copy commands ---------------------------------------------------------------------
// Between two kernels of a stream a hipMemcpyAsync / hipMemsetAsync of a few bytes is a command of its own -- for host memory
// one of the copy engine's, with a hand-over either way -- and a copy into PAGEABLE host memory keeps the calling thread until
// it is through: a band's statistics used to come home in five of them.  Here a kernel sets the inputs of the statistics
// kernels (k_stats_init), and one gathers their results in a block of pinned host memory that the host reads after its wait
// (k_words_gather); k_bytes_scatter moves up to four short byte strings out of pinned host memory to where the band wants
// them (header and ranges, the Huffman mode's code table and code words).
struct WordsGather { const u32* src[5]; u32 n[5]; };    // words of each source, written one behind the other
__global__ void __launch_bounds__(256) k_words_gather(WordsGather g, u32* __restrict__ dst)
{
  u32 at = 0;
  for (int k = 0; k < 5; k++)
  {
    if (g.src[k]) for (u32 i = threadIdx.x; i < g.n[k]; i += 256u) dst[at + i] = g.src[k][i];
    at += g.n[k];
  }
}
__global__ void __launch_bounds__(256) k_stats_init(u64* __restrict__ mins, u64* __restrict__ maxs, u32 nDepth, u64 initMin, u64 initMax,
                                                    u32* __restrict__ zeroA, u32 nA, u32* __restrict__ zeroB, u32 nB)
{
  for (u32 i = threadIdx.x; i < nDepth; i += 256u) { mins[i] = initMin; maxs[i] = initMax; }
  for (u32 i = threadIdx.x; i < nA; i += 256u) zeroA[i] = 0u;
  for (u32 i = threadIdx.x; i < nB; i += 256u) zeroB[i] = 0u;
}
struct BytesScatter { u8* dst[4]; const u8* src[4]; u32 n[4]; };
__global__ void __launch_bounds__(256) k_bytes_scatter(BytesScatter g)
{
  for (int k = 0; k < 4; k++)
    if (g.dst[k]) for (u32 i = threadIdx.x; i < g.n[k]; i += 256u) g.dst[k][i] = g.src[k][i];
}

void launchStatsInit(u64* mins, u64* maxs, int nDepth, u32* zeroA, u32 nWordsA, u32* zeroB, u32 nWordsB, hipStream_t stream)
{
  hipLaunchKernelGGL(k_stats_init, dim3(1), dim3(256), 0, stream, mins, maxs, (u32)nDepth, statKeyInitMin(), statKeyInitMax(), zeroA, nWordsA, zeroB, nWordsB);
}
void launchWordsGather(const u32* const src[5], const u32 nWords[5], u32* dstPinned, hipStream_t stream)
{
  WordsGather g;
  for (int k = 0; k < 5; k++) { g.src[k] = src[k]; g.n[k] = nWords[k]; }
  hipLaunchKernelGGL(k_words_gather, dim3(1), dim3(256), 0, stream, g, dstPinned);
}
void launchBytesScatter(u8* const dst[4], const u8* const srcPinned[4], const u32 n[4], hipStream_t stream)
{
  BytesScatter g;
  for (int k = 0; k < 4; k++) { g.dst[k] = dst[k]; g.src[k] = srcPinned[k]; g.n[k] = n[k]; }
  hipLaunchKernelGGL(k_bytes_scatter, dim3(1), dim3(256), 0, stream, g);
}

}    // namespace lerc
