// wave_utils.h -- wave64 helpers used by all kernels (CDNA4: a wavefront is 64 lanes).
#pragma once
#include "lerc_common.h"

// Phase probes (tuning tool, `make probe` only): thread 0 of every workgroup adds the shader cycles between
// consecutive PROBE(i) points to slot i of a per-file device array; tools/probe_phases.py prints the totals.
// The product build compiles them to nothing.
#if defined(LERC_PROBE) && !defined(HIPSIM)
#define PROBE_DEFINE(tag) \
  static __device__ unsigned long long g_probe[32]; \
  extern "C" __attribute__((visibility("default"))) void lerc_amd_probe_##tag(unsigned long long* out, int reset) \
  { hipDeviceSynchronize(); hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(g_probe)); \
    if (reset) { unsigned long long z[32] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)); } }
#ifdef LERC_PROBE_TRACE_ONLY    // (the per-workgroup time line alone: the phase sums' atomics on a few addresses distort it)
#define PROBE_BEGIN
#define PROBE(i)
#else
#define PROBE_BEGIN unsigned long long probeT_ = clock64()
#define PROBE(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = clock64(); atomicAdd(&g_probe[i], n_ - probeT_); probeT_ = n_; } } while (0)
#endif
#define PROBE_DRAIN asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")    // so that a phase is charged its own loads
#else
#define PROBE_DEFINE(tag)
#define PROBE_BEGIN
#define PROBE(i)
#define PROBE_DRAIN
#endif

namespace lerc {

__device__ __forceinline__ int laneId() { return (int)(threadIdx.x & 63); }
__device__ __forceinline__ int waveId() { return (int)(threadIdx.x >> 6); }
__device__ __forceinline__ u64 laneMaskLt() { return (1ull << laneId()) - 1ull; }

// Orders LDS traffic between the lanes of ONE wave.  A wave issues its LDS instructions in order,
// so no hardware barrier is needed -- only the compiler must not move accesses across this point.
__device__ __forceinline__ void waveSync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Waits until this wave's stores (and loads) have been acknowledged.  Between write-through (agent scope) stores of a
// partial result and the arrival ticket that lets another workgroup read them: inline assembly, because the compiler
// drops a builtin wait it believes redundant (MI355X_MICROARCH.md, hand-off rules).
__device__ __forceinline__ void drainVmem()
{
#ifndef HIPSIM
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
// partial results handed from workgroup to workgroup inside one launch: write-through stores / loads that bypass the
// (per XCD, not coherent) L2, so that no release fence -- which would write back every dirty line of the L2 -- is needed
__device__ __forceinline__ void publish64(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 observe64(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// draws an arrival ticket; true for the workgroup that draws the last of `expected` (call by ONE thread, after
// drainVmem() + __syncthreads())
__device__ __forceinline__ bool lastArrival(u32* ticket, u32 expected)
{
  return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == expected - 1u;
}

// ------------------------------------------------------------------------------------------------
// 8- and 16-byte vector accesses with the "non-temporal" hint (streamed once: do not displace what other kernels of the
// call still want from the caches)
// ------------------------------------------------------------------------------------------------
typedef u32 u32x2_t __attribute__((ext_vector_type(2)));
typedef u32 u32x4_t __attribute__((ext_vector_type(4)));
template<class Vec> __device__ __forceinline__ void storeStreaming(Vec* dst, const Vec& v)
{
  static_assert(sizeof(Vec) == 8 || sizeof(Vec) == 16, "vector store");
#ifdef HIPSIM
  *dst = v;
#else
  if (sizeof(Vec) == 16) { u32x4_t x; memcpy(&x, &v, 16); __builtin_nontemporal_store(x, reinterpret_cast<u32x4_t*>(dst)); }
  else { u32x2_t x; memcpy(&x, &v, 8); __builtin_nontemporal_store(x, reinterpret_cast<u32x2_t*>(dst)); }
#endif
}
template<class Vec> __device__ __forceinline__ Vec loadStreaming(const Vec* src)
{
  static_assert(sizeof(Vec) == 8 || sizeof(Vec) == 16, "vector load");
#ifdef HIPSIM
  return *src;
#else
  Vec v;
  if (sizeof(Vec) == 16) { const u32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src)); memcpy(&v, &x, 16); }
  else { const u32x2_t x = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(src)); memcpy(&v, &x, 8); }
  return v;
#endif
}

// The same 16 bytes at an address that is a multiple of 4 only (rasters whose row pitch is no multiple of 16 bytes: 8190 floats): one
// dwordx4 instruction all the same -- the hardware takes vector accesses at dword alignment, they may touch two lines
typedef u32x4_t u32x4_a4_t __attribute__((aligned(4)));
template<class Vec> __device__ __forceinline__ void storeStreamingA4(void* dst, const Vec& v)
{
  static_assert(sizeof(Vec) == 16, "vector store");
#ifdef HIPSIM
  memcpy(dst, &v, 16);
#else
  u32x4_t x; memcpy(&x, &v, 16); __builtin_nontemporal_store(x, reinterpret_cast<u32x4_a4_t*>(dst));
#endif
}
template<class Vec> __device__ __forceinline__ Vec loadStreamingA4(const void* src)
{
  static_assert(sizeof(Vec) == 16, "vector load");
  Vec v;
#ifdef HIPSIM
  memcpy(&v, src, 16);
#else
  const u32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const u32x4_a4_t*>(src)); memcpy(&v, &x, 16);
#endif
  return v;
}

// ------------------------------------------------------------------------------------------------
// Cross-lane moves that stay in the VALU (no LDS traffic, unlike ds_bpermute behind __shfl*):
// DPP (data parallel primitives) inside a row of 16 lanes, v_permlane{16,32}_swap across rows (gfx950).
// A lane whose DPP source does not exist keeps its own value.
// ------------------------------------------------------------------------------------------------
enum : int
{
  kDppQuadXor1 = 0xB1,         // quad_perm [1,0,3,2]
  kDppQuadXor2 = 0x4E,         // quad_perm [2,3,0,1]
  kDppRowShr1 = 0x111,         // lane i <- lane i - 1 inside its row
  kDppRowRor1 = 0x121, kDppRowRor2 = 0x122, kDppRowRor4 = 0x124, kDppRowRor8 = 0x128,
  kDppWaveShr1 = 0x138,        // lane i <- lane i - 1 across the whole wave
  kDppRowMirror = 0x140, kDppRowHalfMirror = 0x141
};

template<int CTRL> __device__ __forceinline__ u32 dppMov(u32 v)
{
#ifdef HIPSIM
  const int lane = laneId();
  int src = lane;
  if (CTRL < 0x100) src = (lane & ~3) | ((CTRL >> (2 * (lane & 3))) & 3);
  else if (CTRL >= 0x111 && CTRL <= 0x11F) src = ((lane & 15) >= (CTRL - 0x110)) ? lane - (CTRL - 0x110) : lane;
  else if (CTRL >= 0x121 && CTRL <= 0x12F) src = (lane & ~15) | ((lane - (CTRL - 0x120)) & 15);
  else if (CTRL == kDppWaveShr1) src = lane > 0 ? lane - 1 : lane;
  else if (CTRL == kDppRowMirror) src = (lane & ~15) | (15 - (lane & 15));
  else if (CTRL == kDppRowHalfMirror) src = (lane & ~7) | (7 - (lane & 7));
  return __shfl(v, src);
#else
  // the shifts keep a lane's own value where its source lane does not exist; the permutations (every lane has a
  // source) are written with "bound_ctrl", the form hipcc folds into the consuming VALU instruction (v_add_u32_dpp ...)
  if (CTRL == kDppRowShr1 || CTRL == kDppWaveShr1) return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false);
  return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
#endif
}

template<int CTRL, class X> __device__ __forceinline__ X dppMovT(X v)
{
  static_assert(sizeof(X) == 4 || sizeof(X) == 8, "32- or 64-bit payloads");
  u32 w[sizeof(X) / 4];
  memcpy(w, &v, sizeof(X));
#pragma unroll
  for (int i = 0; i < (int)(sizeof(X) / 4); i++) w[i] = dppMov<CTRL>(w[i]);
  memcpy(&v, w, sizeof(X));
  return v;
}

// op(v of this lane, v of lane ^ 16) resp. lane ^ 32 for a symmetric op
template<int DIST, class X, class Op> __device__ __forceinline__ X combineXor(X v, Op op)
{
#ifdef HIPSIM
  return op(v, __shfl_xor(v, DIST));
#else
  u32 w[sizeof(X) / 4], a[sizeof(X) / 4], b[sizeof(X) / 4];
  memcpy(w, &v, sizeof(X));
#pragma unroll
  for (int i = 0; i < (int)(sizeof(X) / 4); i++)
  {
    // both copies hold v; the swap leaves this lane's value in one result and its partner's in the other
    if (DIST == 16) { auto r = __builtin_amdgcn_permlane16_swap(w[i], w[i], false, false); a[i] = r[0]; b[i] = r[1]; }
    else { auto r = __builtin_amdgcn_permlane32_swap(w[i], w[i], false, false); a[i] = r[0]; b[i] = r[1]; }
  }
  X xa, xb;
  memcpy(&xa, a, sizeof(X)); memcpy(&xb, b, sizeof(X));
  return op(xa, xb);
#endif
}

// all-reduce over groups of G consecutive lanes (G = 8, 16, 32 or 64), result in every lane of the group
template<int G, class X, class Op> __device__ __forceinline__ X groupReduce(X v, Op op)
{
  static_assert(G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "group size");
  if (G == 2) return op(v, dppMovT<kDppQuadXor1>(v));
  if (G == 4)
  {
    v = op(v, dppMovT<kDppQuadXor1>(v));
    v = op(v, dppMovT<kDppQuadXor2>(v));
    return v;
  }
  if (G == 8)
  {
    v = op(v, dppMovT<kDppQuadXor1>(v));
    v = op(v, dppMovT<kDppQuadXor2>(v));
    v = op(v, dppMovT<kDppRowHalfMirror>(v));
    return v;
  }
  v = op(v, dppMovT<kDppRowRor8>(v));
  v = op(v, dppMovT<kDppRowRor4>(v));
  v = op(v, dppMovT<kDppRowRor2>(v));
  v = op(v, dppMovT<kDppRowRor1>(v));
  if (G >= 32) v = combineXor<16>(v, op);
  if (G >= 64) v = combineXor<32>(v, op);
  return v;
}

// Floating point minima / maxima go through v_min_f32 / v_max_f32 (one instruction, and the DPP move of a reduction
// step folds into it) instead of compare + select.  The two differ only where a NaN is involved, and a band with a
// NaN in it leaves the streaming kernels for the general path before any of these results is used.
struct OpMin
{
  template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return b < a ? b : a; }
  __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fminf(a, b); }
  __device__ __forceinline__ double operator()(double a, double b) const { return __builtin_fmin(a, b); }
};
struct OpMax
{
  template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return b > a ? b : a; }
  __device__ __forceinline__ float operator()(float a, float b) const { return __builtin_fmaxf(a, b); }
  __device__ __forceinline__ double operator()(double a, double b) const { return __builtin_fmax(a, b); }
};
struct OpSum { template<class X> __device__ __forceinline__ X operator()(X a, X b) const { return a + b; } };

// minimum and maximum of float values over groups of 16 lanes (a DPP row) in one go: v_min_f32 / v_max_f32 with the
// lane permutation as an operand modifier, four steps of two instructions (hipcc does not fold the move into a
// floating point minimum itself, because it wants to quiet signalling NaNs first)
__device__ __forceinline__ void rowMinMax(float& mn, float& mx)
{
#ifdef HIPSIM
  mn = groupReduce<16>(mn, OpMin()); mx = groupReduce<16>(mx, OpMax());
#else
  // s_nop 1: a DPP operand must not have been written by the VALU instruction right in front (2 wait states)
  asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_min_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_min_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_ror:2 row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_min_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf"
      : "+v"(mn), "+v"(mx));
#endif
}

// minimum and maximum of a lane's four float values: v_min3 / v_max3 + one step, four instructions (through fminf / fmaxf hipcc puts a
// v_max x, x in front of the first two operands to quiet signalling NaNs: six); results as OpMin / OpMax give them
__device__ __forceinline__ void laneMinMax4(float a, float b, float c, float d, float& mn, float& mx)
{
#ifdef HIPSIM
  mn = __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(c, d)); mx = __builtin_fmaxf(__builtin_fmaxf(a, b), __builtin_fmaxf(c, d));
#else
  float t, u;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(a), "v"(b), "v"(c));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u) : "v"(a), "v"(b), "v"(c));
  asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(t), "v"(d));
  asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(u), "v"(d));
#endif
}

// ... of a lane's eight values: a chain of v_min3 / v_max3, eight instructions
__device__ __forceinline__ void laneMinMax8(const float (&x)[8], float& mn, float& mx)
{
#ifdef HIPSIM
  mn = x[0]; mx = x[0];
  for (int k = 1; k < 8; k++) { mn = __builtin_fminf(mn, x[k]); mx = __builtin_fmaxf(mx, x[k]); }
#else
  float t, u;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(x[0]), "v"(x[1]), "v"(x[2]));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u) : "v"(x[0]), "v"(x[1]), "v"(x[2]));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(x[3]), "v"(x[4]));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u) : "v"(u), "v"(x[3]), "v"(x[4]));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(x[5]), "v"(x[6]));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(u) : "v"(u), "v"(x[5]), "v"(x[6]));
  asm("v_min_f32 %0, %1, %2" : "=v"(mn) : "v"(t), "v"(x[7]));
  asm("v_max_f32 %0, %1, %2" : "=v"(mx) : "v"(u), "v"(x[7]));
#endif
}
// minimum and maximum over groups of 8 lanes (half a DPP row): neighbour, neighbouring pair, the mirrored half -- three steps of two
__device__ __forceinline__ void halfRowMinMax(float& mn, float& mx)
{
#ifdef HIPSIM
  mn = groupReduce<8>(mn, OpMin()); mx = groupReduce<8>(mx, OpMax());
#else
  asm("s_nop 1\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_min_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\tv_min_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf"
      : "+v"(mn), "+v"(mx));
#endif
}

// whole-wave all-reduces (32- and 64-bit payloads)
template<class T> __device__ __forceinline__ T waveMin(T v) { return groupReduce<64>(v, OpMin()); }
template<class T> __device__ __forceinline__ T waveMax(T v) { return groupReduce<64>(v, OpMax()); }
template<class T> __device__ __forceinline__ T waveSum(T v) { return groupReduce<64>(v, OpSum()); }

// signed char is not a shuffle payload type everywhere: widen small integers
template<class T> struct ShflT { typedef T type; };
template<> struct ShflT<signed char> { typedef int type; };
template<> struct ShflT<unsigned char> { typedef int type; };
template<> struct ShflT<short> { typedef int type; };
template<> struct ShflT<unsigned short> { typedef int type; };

template<class T> __device__ __forceinline__ T waveMinT(T v) { return (T)waveMin((typename ShflT<T>::type)v); }
template<class T> __device__ __forceinline__ T waveMaxT(T v) { return (T)waveMax((typename ShflT<T>::type)v); }

// OR `nbits` (<= 32) of `value` into a little-endian bit stream held in 32-bit words (LDS).
__device__ __forceinline__ void orBits(u32* words, u32 bitPos, u32 value, int nbits)
{
  u32 w = bitPos >> 5, sh = bitPos & 31;
  atomicOr(&words[w], value << sh);
  if (sh + (u32)nbits > 32) atomicOr(&words[w + 1], value >> (32 - sh));
}

// Inclusive sum scan over the wave.  In the VALU: four shifts inside a row of 16 lanes and two row broadcasts (lane 15 of a row into
// the next row, lane 31 into the upper half), six v_add_u32_dpp -- through __shfl_up every step is an LDS permute with its round
// trip, a select and an add, and the scan sits on the critical path of the encoder's plan wave and the decoder's list.
__device__ __forceinline__ u32 waveInclusiveScan(u32 v)
{
#ifdef HIPSIM
  const int lane = laneId();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { u32 t = __shfl_up(v, (unsigned)d); if (lane >= d) v += t; }
  return v;
#else
  // (a lane without a source adds the "old" operand, 0)
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);    // row_shr:1
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);    // row_shr:2
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);    // row_shr:4
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);    // row_shr:8
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
  v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
  return v;
#endif
}

// Exclusive scan of in[0 .. n) into out[0 .. n], out[n] = total, by ONE workgroup of 1024 threads: thread t owns a
// contiguous run of 4 * ceil(n / 4096) elements, moved as 16-byte vectors (in and out are 16-byte aligned and
// readable up to the next multiple of 4 elements: the callers' arrays have slack)
__device__ __forceinline__ void scanSingleWorkgroup(const u32* __restrict__ in, u32* __restrict__ out, u32 n)
{
  __shared__ u32 s_w[16];
  const u32 per4 = (n + 4095u) / 4096u;                   // vectors per thread
  const u32 begin = threadIdx.x * per4 * 4u;
  const u32 lastVec = (n - 1u) & ~3u;                      // clamped loads: no data-dependent exit, so they batch
  u32 sum = 0;
#pragma unroll 8
  for (u32 q = 0; q < per4; q++)
  {
    const u32 i = begin + q * 4u;
    const uint4 x = *reinterpret_cast<const uint4*>(in + min(i, lastVec));
    sum += (i < n ? x.x : 0u) + (i + 1 < n ? x.y : 0u) + (i + 2 < n ? x.z : 0u) + (i + 3 < n ? x.w : 0u);
  }
  const u32 inc = waveInclusiveScan(sum);
  if (laneId() == 63) s_w[waveId()] = inc;
  __syncthreads();
  u32 run = inc - sum;
  for (int i = 0; i < waveId(); i++) run += s_w[i];
#pragma unroll 8
  for (u32 q = 0; q < per4; q++)
  {
    const u32 i = begin + q * 4u;
    const uint4 x = *reinterpret_cast<const uint4*>(in + min(i, lastVec));
    uint4 o;
    o.x = run; run += (i < n ? x.x : 0u);
    o.y = run; run += (i + 1 < n ? x.y : 0u);
    o.z = run; run += (i + 2 < n ? x.z : 0u);
    o.w = run; run += (i + 3 < n ? x.w : 0u);
    if (i + 3 < n) *reinterpret_cast<uint4*>(out + i) = o;
    else if (i < n) { out[i] = o.x; if (i + 1 < n) out[i + 1] = o.y; if (i + 2 < n) out[i + 2] = o.z; }
  }
  // every thread behind the one that holds element n - 1 carries the total too
  if (threadIdx.x == 1023) out[n] = run;
}

// Fletcher32 terms (Lerc2.cpp:1037-1064) of one 16-byte unit whose first byte is byte 2 * k0 of the checksummed range
// blob[14 ..): the checksum works on big-endian 16-bit words w, A = sum w, B = sum index * w.  Per dword: the bytes of both
// halves swapped (one v_perm), then two dot products of 16-bit pairs -- the words themselves, and the words weighted with
// their index inside the unit (0 .. 7): 12 instructions a unit (byte dot products with even / odd masks took 16 + 6).
__device__ __forceinline__ u32 dot2u16(u32 a, u32 b, u32 c)
{
#ifdef HIPSIM
  return lercsim_udot2(a, b, c);
#else
  typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
  u16x2_t x, y;
  __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
  return __builtin_amdgcn_udot2(x, y, c, false);
#endif
}
// Fletcher sums are wanted mod 65535 only, and 2^16 = 1 (mod 65535): a sum's 16-bit digits added up are congruent to it.  64 bits
// come down to less than 2^18, 32 bits to less than 2^17 -- small enough for a wave's (a workgroup's) sums to stay in 32 bits, where a
// reduction step is one DPP add, and no division anywhere.
__device__ __forceinline__ u32 fold65535(u32 x) { return (x & 0xFFFFu) + (x >> 16); }
__device__ __forceinline__ u32 fold65535(u64 x) { const u32 lo = (u32)x, hi = (u32)(x >> 32); return (lo & 0xFFFFu) + (lo >> 16) + (hi & 0xFFFFu) + (hi >> 16); }
__device__ __forceinline__ void fletcherUnit(const uint4& x, u64 k0, u32& A, u64& B)
{
  const u32 w0 = __builtin_amdgcn_perm(x.x, x.x, 0x02030001u), w1 = __builtin_amdgcn_perm(x.y, x.y, 0x02030001u);
  const u32 w2 = __builtin_amdgcn_perm(x.z, x.z, 0x02030001u), w3 = __builtin_amdgcn_perm(x.w, x.w, 0x02030001u);
  u32 a = dot2u16(w0, 0x00010001u, 0u), bw = dot2u16(w0, 0x00010000u, 0u);
  a = dot2u16(w1, 0x00010001u, a); bw = dot2u16(w1, 0x00030002u, bw);
  a = dot2u16(w2, 0x00010001u, a); bw = dot2u16(w2, 0x00050004u, bw);
  a = dot2u16(w3, 0x00010001u, a); bw = dot2u16(w3, 0x00070006u, bw);    // a < 2^19, bw < 2^21
  A += a;
  B += k0 * a + bw;
}

// first error wins
__device__ __forceinline__ void raiseError(DeviceStatus* st, u32 code, u32 where)
{
  if (atomicCAS(&st->error, 0u, code) == 0u) st->errorWhere = where;
}

}    // namespace lerc
