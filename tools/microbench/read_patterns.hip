// Microbenchmark (tuning aid, not part of the product): how fast can 8192 x 8192 f32 be read with the access
// shapes the streaming kernels use?   hipcc --offload-arch=gfx950 -O3 -o /tmp/read_patterns read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: workgroup = 8 rows x 512 cols; wave instruction = 8 rows x 128 B (lane = r * 8 + c), 4 tiles per wave
__global__ void __launch_bounds__(256) kA(const float4* __restrict__ d, int nCols4, float* out)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 3, c = lane & 7;
  const int wgPerRow = nCols4 / 128, it = blockIdx.x / wgPerRow, wgc = blockIdx.x % wgPerRow;
  const size_t rowBase = (size_t)(it * 8 + r) * nCols4 + (size_t)wgc * 128;
  float s = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) { const float4 v = d[rowBase + (t * 4 + w) * 8 + c]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}
// B: same workgroup footprint, wave instruction = 1 row x 1 KiB (lane = column), 8 rows per wave
__global__ void __launch_bounds__(256) kB(const float4* __restrict__ d, int nCols4, float* out)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wgPerRow = nCols4 / 128, it = blockIdx.x / wgPerRow, wgc = blockIdx.x % wgPerRow;
  float s = 0;
#pragma unroll
  for (int r = 0; r < 8; r++)
  {
    const float4 v = d[(size_t)(it * 8 + r) * nCols4 + (size_t)wgc * 128 + (w & 1) * 64 + lane + 0 * (w >> 1)];
    if ((r & 1) == (w >> 1)) s += v.x + v.y + v.z + v.w;    // 4 waves x 8 loads: every row segment read twice -> halve below
  }
  if (s == 123.456f) out[0] = s;
}
// C: plain linear stream, 4 x 16 B per thread
__global__ void __launch_bounds__(256) kC(const float4* __restrict__ d, size_t n4, float* out)
{
  size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x;
  float s = 0;
#pragma unroll
  for (int t = 0; t < 4; t++) { const float4 v = d[i + t * 256]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}
// D: like A but the workgroup covers 8 rows x 2048 cols (16 tiles per wave): fewer, longer row segments
__global__ void __launch_bounds__(256) kD(const float4* __restrict__ d, int nCols4, float* out)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 3, c = lane & 7;
  const int wgPerRow = nCols4 / 512, it = blockIdx.x / wgPerRow, wgc = blockIdx.x % wgPerRow;
  const size_t rowBase = (size_t)(it * 8 + r) * nCols4 + (size_t)wgc * 512;
  float s = 0;
#pragma unroll 4
  for (int t = 0; t < 16; t++) { const float4 v = d[rowBase + (t * 4 + w) * 8 + c]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}
// E: write pattern of the decoder: 8 rows x 128 B per wave store
__global__ void __launch_bounds__(256) kE(float4* __restrict__ d, int nCols4)
{
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane >> 3, c = lane & 7;
  const int wgPerRow = nCols4 / 128, it = blockIdx.x / wgPerRow, wgc = blockIdx.x % wgPerRow;
  const size_t rowBase = (size_t)(it * 8 + r) * nCols4 + (size_t)wgc * 128;
#pragma unroll
  for (int t = 0; t < 4; t++) d[rowBase + (t * 4 + w) * 8 + c] = make_float4(1, 2, 3, (float)lane);
}

int main()
{
  const int n = 8192;
  const size_t bytes = (size_t)n * n * 4;
  float4* d; float* out;
  CHECK(hipMalloc(&d, bytes)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(d, 0, bytes));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int nWG = (n / 8) * (n / 512);
  for (int variant = 0; variant < 5; variant++)
  {
    float best = 1e9f;
    for (int rep = 0; rep < 8; rep++)
    {
      hipEventRecord(a);
      switch (variant)
      {
        case 0: hipLaunchKernelGGL(kA, dim3(nWG), dim3(256), 0, 0, d, n / 4, out); break;
        case 1: hipLaunchKernelGGL(kB, dim3(nWG), dim3(256), 0, 0, d, n / 4, out); break;
        case 2: hipLaunchKernelGGL(kC, dim3((unsigned)(bytes / 16 / 1024)), dim3(256), 0, 0, d, bytes / 16, out); break;
        case 3: hipLaunchKernelGGL(kD, dim3(nWG / 4), dim3(256), 0, 0, d, n / 4, out); break;
        case 4: hipLaunchKernelGGL(kE, dim3(nWG), dim3(256), 0, 0, d, n / 4); break;
      }
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double moved = (variant == 1) ? 2.0 * bytes : (double)bytes;
    printf("variant %c: %.1f us, %.0f GB/s\n", "ABCDE"[variant], best * 1e3, moved / best / 1e6);
  }
  return 0;
}
