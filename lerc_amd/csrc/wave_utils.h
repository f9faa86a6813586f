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
#define PROBE_BEGIN unsigned long long probeT_ = clock64()
#define PROBE(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = clock64(); atomicAdd(&g_probe[i], n_ - probeT_); probeT_ = n_; } } while (0)
#else
#define PROBE_DEFINE(tag)
#define PROBE_BEGIN
#define PROBE(i)
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

template<class T> __device__ __forceinline__ T waveMin(T v)
{
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { T o = __shfl_xor(v, m); v = (o < v) ? o : v; }
  return v;
}
template<class T> __device__ __forceinline__ T waveMax(T v)
{
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) { T o = __shfl_xor(v, m); v = (o > v) ? o : v; }
  return v;
}
template<class T> __device__ __forceinline__ T waveSum(T v)
{
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m);
  return v;
}

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

// first error wins
__device__ __forceinline__ void raiseError(DeviceStatus* st, u32 code, u32 where)
{
  if (atomicCAS(&st->error, 0u, code) == 0u) st->errorWhere = where;
}

}    // namespace lerc
