// Microbenchmark (tuning aid): what each ingredient of the statistics pass costs on 8192 x 8192 f32, added one by one.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o stats_steps stats_steps.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../lerc_amd/csrc/lerc_common.h"
#include "../../lerc_amd/csrc/wave_utils.h"
#include "../../lerc_amd/csrc/block_plan.h"
using namespace lerc;

template<int CTRL> __device__ __forceinline__ float dppf(float v)
{ return __uint_as_float((u32)__builtin_amdgcn_update_dpp((int)__float_as_uint(v), (int)__float_as_uint(v), CTRL, 0xF, 0xF, false)); }
template<int CTRL> __device__ __forceinline__ int dppi(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ float rmin16(float v)
{ v = fminf(v, dppf<0x128>(v)); v = fminf(v, dppf<0x124>(v)); v = fminf(v, dppf<0x122>(v)); v = fminf(v, dppf<0x121>(v)); return v; }
__device__ __forceinline__ float rmax16(float v)
{ v = fmaxf(v, dppf<0x128>(v)); v = fmaxf(v, dppf<0x124>(v)); v = fmaxf(v, dppf<0x122>(v)); v = fmaxf(v, dppf<0x121>(v)); return v; }
__device__ __forceinline__ int rsum16(int v)
{ v += dppi<0x128>(v); v += dppi<0x124>(v); v += dppi<0x122>(v); v += dppi<0x121>(v); return v; }

template<int LEVEL>
__global__ void __launch_bounds__(256) k(const float4* __restrict__ d, int nCols4, float* out, float4* desc)
{
  __shared__ float s_mn[64], s_mx[64];
  __shared__ u32 s_same[64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = lane >> 4, r = (lane & 15) >> 1, h = lane & 1, c = b * 2 + h;
  const int wgPerRow = nCols4 / 128, it = blockIdx.x / wgPerRow, wgc = blockIdx.x % wgPerRow;
  const size_t rowBase = (size_t)(it * 8 + r) * nCols4 + (size_t)wgc * 128;
  float4 v[4];
#pragma unroll
  for (int t = 0; t < 4; t++) v[t] = d[rowBase + (t * 4 + w) * 8 + c];
  float acc = 0;
  u32 flags = 0;
#pragma unroll
  for (int t = 0; t < 4; t++)
  {
    const float4 x = v[t];
    float mn = fminf(fminf(x.x, x.y), fminf(x.z, x.w)), mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
    if (LEVEL >= 1) { mn = rmin16(mn); mx = rmax16(mx); }
    int same = 0;
    if (LEVEL >= 2)
    {
      float prev = dppf<0x138>(x.w);
      if ((lane & 15) == 0) prev = 0;
      same = (x.x == prev) + (x.y == x.x) + (x.z == x.y) + (x.w == x.z);
      if (__any(same > 2)) same = rsum16(same); else same = 0;
    }
    if (LEVEL >= 3)
    {
      if (x.x != x.x || x.y != x.y || x.z != x.z || x.w != x.w) flags |= 1;
      if (!(flags & 2)) { if (x.x != truncf(x.x) || x.y != truncf(x.y) || x.z != truncf(x.z) || x.w != truncf(x.w)) flags |= 2; }
    }
    if (LEVEL >= 4)
    {
      if ((lane & 15) == 0) { const int blk = (t * 4 + w) * 4 + b; s_mn[blk] = mn; s_mx[blk] = mx; s_same[blk] = same; }
    }
    acc += mn + mx + same;
  }
  if (LEVEL >= 4)
  {
    __syncthreads();
    if (w == (int)((blockIdx.x * 2654435761u) >> 30))
    {
      const float mn = s_mn[lane], mx = s_mx[lane];
      const double mv = ((double)mx - (double)mn) * 50.0;
      const u32 q = (u32)(mv + 0.5);
      const int nb = 32 - __clz((int)q);
      if (LEVEL == 5) desc[(size_t)blockIdx.x * 64 + lane] = make_float4(mn, __uint_as_float(7u + 8u * nb), 0, 0);
      if (LEVEL >= 6)
      {
        BandParams p; memset(&p, 0, sizeof(p));
        p.dt = DT_Float; p.maxZErr = 0.01; p.scale = 50.0; p.maxQ = (1u << 30) - 1; p.version = 6;
        const bool quantOk = !(mv > (double)p.maxQ || q == 0);
        const Plan pl = planBlock<float>(p, 64, mn, mx, p.dt, false, mv, quantOk ? q : 0u, 0u);
        desc[(size_t)blockIdx.x * 64 + lane] = make_float4(mn, __uint_as_float((u32)pl.nBytes | ((u32)pl.kind << 16) | ((u32)pl.tc << 19) | ((u32)nb << 24)), 0, 0);
        const u32 total = waveSum((u32)pl.nBytes);
        u32 kb = __float_as_uint(mn); kb = (kb & 0x80000000u) ? ~kb : (kb | 0x80000000u);
        const unsigned long long kMin = waveMin((unsigned long long)kb);
        if (lane == 0) { out[1 + (blockIdx.x & 1023)] = (float)total + (float)kMin; }
      }
      acc += nb;
    }
  }
  if (acc == 123.456f || flags == 77) out[0] = acc;
}

#define RUN(L) { float best = 1e9f; for (int rep = 0; rep < 9; rep++) { const float4* d = dd[rep % 3]; hipEventRecord(a); hipLaunchKernelGGL(k<L>, dim3(nWG), dim3(256), 0, 0, d, n / 4, out, desc); \
  hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; } printf("level %d: %.1f us\n", L, best * 1e3); }

int main()
{
  const int n = 8192;
  const size_t bytes = (size_t)n * n * 4;
  // three input buffers in rotation: one 256 MiB raster alone would be served out of the Infinity Cache
  float4* dd[3]; float* out; float4* desc;
  for (int i = 0; i < 3; i++) { hipMalloc(&dd[i], bytes); hipMemset(dd[i], 0x3f, bytes); } hipMalloc(&out, 8192); hipMalloc(&desc, (size_t)(n / 8) * (n / 8) * 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nWG = (n / 8) * (n / 512);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  return 0;
}
