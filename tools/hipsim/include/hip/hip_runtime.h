/*
 * tools/hipsim -- a tiny SIMT emulator standing in for <hip/hip_runtime.h> so that the *unmodified*
 * kernel sources under lerc_amd/csrc can be compiled with plain g++ and unit-tested in a container
 * that has no GPU.  TEST INFRASTRUCTURE ONLY:
 *   - it is never on the include path of the product build (hipcc uses the real ROCm header);
 *   - the resulting tests/_sim/liblerc_amd_sim.so is loaded only by `-m "not gpu"` tests and is never
 *     looked at by lerc_amd/ (the product fails loudly without a GPU);
 *   - it is not a performance or compatibility layer: it exists to catch indexing / protocol bugs
 *     (and, under -fsanitize=address, out-of-bounds accesses) before spending GPU minutes.
 *
 * Model: one workgroup at a time, in blockIdx order; every thread is a ucontext fiber; the 64
 * fibers of a wave rendezvous at wave collectives (__shfl*, __ballot, wave barrier) and all live
 * fibers of the workgroup rendezvous at __syncthreads().  Collectives must be reached by all live
 * lanes of the wave (wave-uniform control flow), which is also what the hardware kernels rely on.
 */
#ifndef LERC_HIPSIM_RUNTIME_H
#define LERC_HIPSIM_RUNTIME_H

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict
#define __align__(n) alignas(n)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define HIPSIM 1
using std::min;
using std::max;

struct dim3
{
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct ushort4 { unsigned short x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{ a, b, c, d }; }
inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{ a, b }; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNoDevice = 100, hipErrorNotReady = 600 };
typedef struct hipsimStream* hipStream_t;
typedef struct hipsimEvent { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; char gcnArchName[256]; };

namespace hipsim {

struct Group { int live = 0, arrived = 0; unsigned gen = 0; };

struct Fiber
{
  void* sp = nullptr;            // saved stack pointer (hand-rolled x86-64 context switch, see hipsim.cpp)
  std::vector<char> stack;
  bool done = true;
  dim3 tid;
  int lin = 0;
  unsigned collGen = 0;          // number of wave collectives this fiber has passed
};

struct State
{
  void* schedSp = nullptr;
  std::vector<Fiber> fibers;
  int cur = -1;
  Group block;
  std::vector<Group> waves;
  uint64_t slots[2][1024];                  // exchange slots, double buffered by collective parity
  unsigned stamp[2][1024];                  // generation that wrote the slot (= "lane took part")
  std::function<void()> body;
  dim3 grid, blk;
};

inline State& S() { static State s; return s; }

extern "C" void hipsim_switch(void** saveSp, void* loadSp);

inline void yield_to_sched()
{
  State& s = S();
  hipsim_switch(&s.fibers[s.cur].sp, s.schedSp);
}

inline void sync_group(Group& g)
{
  unsigned gen = g.gen;
  g.arrived++;
  if (g.arrived >= g.live) { g.arrived = 0; g.gen++; return; }
  while (g.gen == gen) yield_to_sched();
}

inline void fiber_exit_bookkeeping()
{
  State& s = S();
  Fiber& f = s.fibers[s.cur];
  f.done = true;
  Group* gs[2] = { &s.block, &s.waves[f.lin / 64] };
  for (Group* g : gs)
  {
    g->live--;
    if (g->live > 0 && g->arrived >= g->live) { g->arrived = 0; g->gen++; }
  }
}

inline void fiber_main()
{
  State& s = S();
  s.body();
  fiber_exit_bookkeeping();
  yield_to_sched();
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body);

}    // namespace hipsim

// ---- built-in coordinates (plain globals: a single host thread runs one fiber at a time) ----
extern dim3 threadIdx, blockIdx, blockDim, gridDim;
static const int warpSize = 64;

inline int __lane_id() { return hipsim::S().fibers[hipsim::S().cur].lin & 63; }

inline void __syncthreads() { hipsim::sync_group(hipsim::S().block); }

namespace hipsim {
inline Group& myWave() { State& s = S(); return s.waves[s.fibers[s.cur].lin / 64]; }
inline int myLin() { State& s = S(); return s.fibers[s.cur].lin; }
inline int waveBase() { return myLin() & ~63; }
inline int waveLanes() { State& s = S(); int n = (int)(s.blk.x * s.blk.y * s.blk.z) - waveBase(); return n < 64 ? n : 64; }

// One rendezvous per collective: every lane writes (value, generation stamp) into the slot set of
// the current parity, all live lanes of the wave meet, then read.  A lane can be at most one
// collective ahead of the slowest lane, so two slot sets suffice and nothing needs clearing.
inline unsigned collectiveBegin(uint64_t raw)
{
  State& s = S();
  Fiber& f = s.fibers[s.cur];
  const unsigned gen = ++f.collGen, par = gen & 1u;
  s.slots[par][f.lin] = raw;
  s.stamp[par][f.lin] = gen;
  sync_group(myWave());
  return gen;
}

template<class T> inline T exchange(T v, int srcLane, bool valid = true)
{
  static_assert(sizeof(T) <= 8, "exchange payload");
  State& s = S();
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  const unsigned gen = collectiveBegin(raw), par = gen & 1u;
  const int base = waveBase();
  T r = v;
  if (valid && srcLane >= 0 && srcLane < waveLanes() && s.stamp[par][base + srcLane] == gen)
    memcpy(&r, &s.slots[par][base + srcLane], sizeof(T));
  return r;
}
}    // namespace hipsim

template<class T> inline T __shfl(T v, int src, int width = 64)
{
  int lane = __lane_id();
  int s = (lane & ~(width - 1)) + (src & (width - 1));
  return hipsim::exchange(v, s);
}
template<class T> inline T __shfl_xor(T v, int mask, int width = 64)
{
  int lane = __lane_id();
  int s = lane ^ mask;
  bool ok = (s & ~(width - 1)) == (lane & ~(width - 1));
  return hipsim::exchange(v, s, ok);
}
template<class T> inline T __shfl_up(T v, unsigned delta, int width = 64)
{
  int lane = __lane_id();
  int s = lane - (int)delta;
  bool ok = s >= (lane & ~(width - 1));
  return hipsim::exchange(v, s, ok);
}
template<class T> inline T __shfl_down(T v, unsigned delta, int width = 64)
{
  int lane = __lane_id();
  int s = lane + (int)delta;
  bool ok = s <= (lane | (width - 1));
  return hipsim::exchange(v, s, ok);
}

inline unsigned long long __ballot(int pred)
{
  hipsim::State& s = hipsim::S();
  const unsigned gen = hipsim::collectiveBegin(pred ? 1u : 0u), par = gen & 1u;
  const int base = hipsim::waveBase();
  unsigned long long m = 0;
  for (int l = 0; l < hipsim::waveLanes(); l++)
    if (s.stamp[par][base + l] == gen && s.slots[par][base + l]) m |= 1ull << l;
  return m;
}
inline int __any(int p) { return __ballot(p) != 0; }
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return __ballot(p ? 1 : 0); }
inline int __all(int p)
{
  unsigned long long act = __ballot(1);
  return __ballot(p) == act;
}
inline unsigned long long __activemask() { return __ballot(1); }

template<class T> inline T __builtin_amdgcn_readfirstlane(T v)
{
  unsigned long long act = __ballot(1);
  return hipsim::exchange(v, __builtin_ctzll(act));
}
inline int __builtin_amdgcn_ds_bpermute(int byteAddr, int v) { return hipsim::exchange(v, (byteAddr >> 2) & 63); }
inline void __builtin_amdgcn_wave_barrier() { hipsim::sync_group(hipsim::myWave()); }
inline void __builtin_amdgcn_s_barrier() { __syncthreads(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_s_sleep(int) { /* a spinning fiber must let others run */ hipsim::yield_to_sched(); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline void __builtin_amdgcn_s_setprio(int) {}
inline void __threadfence() {}
inline void __threadfence_block() {}
inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned, unsigned) { return (unsigned)__lane_id(); }
inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned, unsigned v) { return v; }

// ---- bit intrinsics ----
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline unsigned __brev(unsigned v) { unsigned r = 0; for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i); return r; }
inline unsigned __funnelshift_r(unsigned lo, unsigned hi, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
inline unsigned __builtin_amdgcn_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> (sh & 31)); }
// v_dot4_u32_u8: sum of the four byte products + c
// (the two 16-bit halves of a and b multiplied pairwise: the product forms of fletcherUnit in wave_utils.h)
inline unsigned lercsim_udot2(unsigned a, unsigned b, unsigned c) { return (a & 0xFFFFu) * (b & 0xFFFFu) + (a >> 16) * (b >> 16) + c; }
inline unsigned __builtin_amdgcn_perm(unsigned a, unsigned b, unsigned sel)
{
  const uint64_t both = ((uint64_t)a << 32) | b;
  unsigned r = 0;
  for (int i = 0; i < 4; i++) r |= (unsigned)((both >> (8 * ((sel >> (8 * i)) & 7u))) & 0xFFu) << (8 * i);
  return r;
}
inline unsigned __builtin_amdgcn_udot4(unsigned a, unsigned b, unsigned c, bool)
{
  for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFFu) * ((b >> (8 * i)) & 0xFFu);
  return c;
}
inline unsigned __builtin_amdgcn_ubfe(unsigned v, unsigned off, unsigned w) { return w == 0 ? 0 : (v >> (off & 31)) & (w >= 32 ? ~0u : ((1u << w) - 1)); }

// ---- atomics (single host thread: plain read-modify-write) ----
template<class T, class U> inline T atomicAdd(T* p, U v) { T o = *p; *p = (T)(o + (T)v); return o; }
template<class T, class U> inline T atomicSub(T* p, U v) { T o = *p; *p = (T)(o - (T)v); return o; }
template<class T, class U> inline T atomicOr(T* p, U v) { T o = *p; *p = (T)(o | (T)v); return o; }
template<class T, class U> inline T atomicAnd(T* p, U v) { T o = *p; *p = (T)(o & (T)v); return o; }
template<class T, class U> inline T atomicXor(T* p, U v) { T o = *p; *p = (T)(o ^ (T)v); return o; }
template<class T, class U> inline T atomicMin(T* p, U v) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template<class T, class U> inline T atomicMax(T* p, U v) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template<class T, class U> inline T atomicExch(T* p, U v) { T o = *p; *p = (T)v; return o; }
template<class T, class U, class V> inline T atomicCAS(T* p, U cmp, V v) { T o = *p; if (o == (T)cmp) *p = (T)v; return o; }
#define __ATOMIC_RELAXED_SIM 0
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_WAVEFRONT 1
#define __HIP_MEMORY_SCOPE_SYSTEM 5
template<class T> inline T __hip_atomic_load(const T* p, int, int) { return *(const volatile T*)p; }
template<class T, class U> inline void __hip_atomic_store(T* p, U v, int, int) { *(volatile T*)p = (T)v; }
template<class T, class U> inline T __hip_atomic_fetch_add(T* p, U v, int, int) { T o = *p; *p = (T)(o + (T)v); return o; }
template<class T, class U> inline T __hip_atomic_fetch_or(T* p, U v, int, int) { T o = *p; *p = (T)(o | (T)v); return o; }
template<class T, class U> inline T __hip_atomic_fetch_max(T* p, U v, int, int) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template<class T, class U> inline T __hip_atomic_fetch_min(T* p, U v, int, int) { T o = *p; if ((T)v < o) *p = (T)v; return o; }

// ---- math helpers the kernels may use ----
inline double __dadd_rn(double a, double b) { return a + b; }
inline double __dmul_rn(double a, double b) { return a * b; }
inline double __dsub_rn(double a, double b) { return a - b; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }

// ---- host runtime subset ----
inline hipError_t hipMalloc(void** p, size_t n)
{
  *p = malloc(n ? n : 1);
  // (LERC_AMD_POISON: fresh device memory is not zero on the real thing either)
  static const char* poison = getenv("LERC_AMD_POISON");
  if (*p && poison) memset(*p, (int)strtol(poison, nullptr, 0) & 255, n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
template<class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template<class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }    // kernels run synchronously here
inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof(*p)); strcpy(p->name, "hipsim"); strcpy(p->gcnArchName, "hipsim"); p->multiProcessorCount = 256; return hipSuccess; }
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipsim error"; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipsimEvent{ 0 }; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
struct hipPointerAttribute_t { int type; int device; void* devicePointer; void* hostPointer; };
enum { hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2, hipMemoryTypeUnregistered = 0 };
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void*) { a->type = hipMemoryTypeUnregistered; return hipSuccess; }

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipsim::launch((grid), (block), [&]() { kernel(__VA_ARGS__); })

#endif
