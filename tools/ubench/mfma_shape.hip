// Dev microbenchmark (round 3): does the fp32 MFMA SHAPE change what the chip sustains?
// v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 have the same FLOP/clk; the guide reports that
// under DVFS the 16x16 bf16 shape holds a higher clock on random data.  Same 64x64 output tile per
// wave in both arms, operands re-read from LDS (random data) every k step, 1 or 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 mfma_shape.hip -o mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// LDS image: A[64 rows][KT] (row pitch KT+1), B[KT][64 cols]; KT = 32 k per tile, re-used every iteration.
constexpr int KT = 32;
constexpr int APITCH = KT + 1;
constexpr int LDS_PER_WAVE = 64 * APITCH + KT * 64;

template <int SHAPE, int LDSREAD, int NW>
__global__ void __launch_bounds__(NW * 64) k(const float* __restrict__ g, float* out, long long* stamps, int iters)
{
  __shared__ float lds[NW * LDS_PER_WAVE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NW * LDS_PER_WAVE; i += NW * 64) lds[i] = g[(blockIdx.x * 131 + i) & 0xfffff];
  __syncthreads();
  const float* base = lds + wave * LDS_PER_WAVE;
  long long t0 = 0, r0 = 0;
  if (lane == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0;
  if (SHAPE == 32)
  {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
    const int l31 = lane & 31, lh = lane >> 5;
    const float* As = base + l31 * APITCH + lh;
    const float* Bs = base + 64 * APITCH + lh * 64 + l31;
    float a[2] = {g[lane], g[lane + 64]}, b[2] = {g[lane + 128], g[lane + 192]};
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int s2 = 0; s2 < KT / 2; ++s2)
      {
        if (LDSREAD)
        {
          a[0] = As[2 * s2]; a[1] = As[32 * APITCH + 2 * s2];
          b[0] = Bs[2 * s2 * 64]; b[1] = Bs[2 * s2 * 64 + 32];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      // keep magnitudes bounded without touching the pipe balance much: scale every 64 iterations
      if ((it & 63) == 63)
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) acc[i][j] *= 1e-3f;
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  }
  else
  {
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
    const int l15 = lane & 15, lq = lane >> 4;
    const float* As = base + l15 * APITCH + lq;
    const float* Bs = base + 64 * APITCH + lq * 64 + l15;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = g[lane + 64 * i]; b[i] = g[lane + 256 + 64 * i]; }
    for (int it = 0; it < iters; ++it)
    {
#pragma unroll
      for (int s4 = 0; s4 < KT / 4; ++s4)
      {
        if (LDSREAD)
        {
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[i] = As[16 * i * APITCH + 4 * s4]; b[i] = Bs[4 * s4 * 64 + 16 * i]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if ((it & 63) == 63)
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  }
  if (lane == 0)
  {
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamps[(blockIdx.x * NW + wave) * 2] = t1 - t0;
    stamps[(blockIdx.x * NW + wave) * 2 + 1] = r1 - r0;
  }
  out[blockIdx.x * NW * 64 + tid] = s;
}

template <int SHAPE, int LDSREAD, int NW>
double run(int blocks_per_cu, const float* g, float* out, long long* stamps, const char* tag)
{
  const int iters = 3000;
  const int nblk = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<SHAPE, LDSREAD, NW>), dim3(nblk), dim3(NW * 64), 0, 0, g, out, stamps, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<SHAPE, LDSREAD, NW>), dim3(nblk), dim3(NW * 64), 0, 0, g, out, stamps, iters);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  std::vector<long long> h((size_t)nblk * NW * 2);
  CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> clk;
  for (size_t i = 0; i < h.size() / 2; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
  std::sort(clk.begin(), clk.end());
  double flops = (double)nblk * NW * iters * (KT / 2) * 4 * 4096.0;
  double tf = flops / ms / 1e9;
  printf("%-28s shape %2d lds %d waves/WG %d WG/CU %d: %.3f ms  %.1f TFLOP/s  clock %.2f GHz (median)\n", tag, SHAPE, LDSREAD, NW,
         blocks_per_cu, ms, tf, clk[clk.size() / 2]);
  return tf;
}

int main()
{
  float *g, *out; long long* stamps;
  const size_t NG = 1 << 20;
  CHECK(hipMalloc(&g, NG * 4));
  std::vector<float> hg(NG);
  srand(1);
  for (size_t i = 0; i < NG; ++i) hg[i] = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
  CHECK(hipMemcpy(g, hg.data(), NG * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&out, 256 * 4 * 512 * 4));
  CHECK(hipMalloc(&stamps, 256 * 4 * 8 * 2 * 8));
  // interleaved rounds in one process (guide rule 24)
  for (int round = 0; round < 3; ++round)
  {
    printf("--- round %d\n", round);
    run<32, 0, 4>(1, g, out, stamps, "regs only");
    run<16, 0, 4>(1, g, out, stamps, "regs only");
    run<32, 1, 4>(1, g, out, stamps, "lds operands");
    run<16, 1, 4>(1, g, out, stamps, "lds operands");
    run<32, 1, 8>(1, g, out, stamps, "lds operands, 2 waves/SIMD");
    run<16, 1, 8>(1, g, out, stamps, "lds operands, 2 waves/SIMD");
    run<32, 1, 4>(2, g, out, stamps, "lds operands, 2 WG/CU");
    run<16, 1, 4>(2, g, out, stamps, "lds operands, 2 WG/CU");
  }
  return 0;
}
