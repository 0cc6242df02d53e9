// Dev microbenchmark: what keeps the fp32 MFMA pipe from saturating?
// build: hipcc --offload-arch=gfx950 -O3 mfma_pipe.hip -o mfma_pipe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* __restrict__ g, float* out, int iters)
{
  __shared__ float lds[2 * (128 * 17 + 16 * 128)];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 2 * (128 * 17 + 16 * 128); i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  float a[2] = {1.0f + lane, 2.0f}, b[2] = {0.5f, 0.25f * lane};
  float rg[10];
  for (int q = 0; q < 10; ++q) rg[q] = 0;
  const float* gp = g + (size_t)blockIdx.x * 4096 + tid;
  for (int it = 0; it < iters; ++it)
  {
    float* cur = lds + (it & 1) * (128 * 17 + 16 * 128);
    if (MODE >= 4)
    {
#pragma unroll
      for (int q = 0; q < 10; ++q) rg[q] = gp[q * 256 + (it & 3) * 64];
    }
    const float* As = cur + ((wave >> 1) * 64 + l31) * 17 + lh;
    const float* Bs = cur + 128 * 17 + lh * 128 + (wave & 1) * 64 + l31;
#pragma unroll
    for (int s = 0; s < 8; ++s)
    {
      if (MODE >= 1)
      {
        a[0] = As[2 * s]; a[1] = As[32 * 17 + 2 * s];
        b[0] = Bs[2 * s * 128]; b[1] = Bs[2 * s * 128 + 32];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (MODE >= 3)
    {
      float* nxt = lds + ((it + 1) & 1) * (128 * 17 + 16 * 128);
#pragma unroll
      for (int q = 0; q < 10; ++q) nxt[(tid + q * 256) % (128 * 17)] = rg[q] + (float)q;
    }
    if (MODE >= 2)
      __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s + rg[0];
}

template <int MODE>
void run(int blocks_per_cu, const float* g, float* out)
{
  const int iters = 2000;
  const int nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, g, out, 100);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, g, out, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double flops = (double)nblk * 4 * iters * 32 * 4096.0;
  printf("mode %d blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks_per_cu, ms, flops / ms / 1e9);
}

int main()
{
  float *g, *out;
  CHECK(hipMalloc(&g, (size_t)256 * 4 * 4096 * 4 + (1 << 20)));
  CHECK(hipMemset(g, 0, 256 * 4 * 4096 * 4));
  CHECK(hipMalloc(&out, 256 * 4 * 256 * 4));
  for (int b = 1; b <= 3; ++b)
  {
    run<0>(b, g, out); run<1>(b, g, out); run<2>(b, g, out); run<3>(b, g, out); run<4>(b, g, out);
  }
  return 0;
}
