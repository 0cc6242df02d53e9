// Dev microbenchmark (round 3): do VALU work and fp32 / fp16 MFMA work of DIFFERENT waves on one SIMD overlap?
// One workgroup of 8 waves per CU (waves w and w+4 share SIMD w): waves 0-3 run a VALU stream, waves 4-7 an MFMA stream;
// each arm is timed alone and together.  together ~ max(alone): separate pipes; together ~ sum: one pipe.
// Arms: VALU = packed fp32 FMA | plain fp32 FMA | 32-bit integer add/xor | v_exp_f32;  MFMA = f32 32x32x2 | f16 32x32x16.
// build: hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// valu: 0 none, 1 pk_fma, 2 fma, 3 int, 4 exp;  mfma: 0 none, 1 f32, 2 f16
template <int VALU, int MFMA>
__global__ void __launch_bounds__(512) k(const float* __restrict__ g, float* out, int iters)
{
  __shared__ float hog[30000];   // 120 KB: one workgroup per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  hog[tid] = g[tid];
  __syncthreads();
  float s = hog[(tid * 7) & 511];
  if (wave < 4)
  {
    if (VALU == 1)
    {
      f32x2 a[8];
      for (int i = 0; i < 8; ++i) a[i] = f32x2{g[lane + i], g[lane + 64 + i]};
      const f32x2 m = f32x2{g[lane + 200], g[lane + 300]}, c = f32x2{g[lane + 400], g[lane + 500]};
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
      for (int i = 0; i < 8; ++i) s += a[i][0] + a[i][1];
    }
    else if (VALU == 2)
    {
      float a[8];
      for (int i = 0; i < 8; ++i) a[i] = g[lane + i];
      const float m = g[lane + 200], c = g[lane + 400];
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
      for (int i = 0; i < 8; ++i) s += a[i];
    }
    else if (VALU == 3)
    {
      unsigned a[8];
      for (int i = 0; i < 8; ++i) a[i] = __float_as_uint(g[lane + i]);
      const unsigned m = __float_as_uint(g[lane + 200]);
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = (a[i] + m) ^ (a[i] >> 3);
      for (int i = 0; i < 8; ++i) s += __uint_as_float(a[i] & 0x3fffffff);
    }
    else if (VALU == 4)
    {
      float a[8];
      for (int i = 0; i < 8; ++i) a[i] = g[lane + i] * 1e-3f;
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]) - 1.f;
      for (int i = 0; i < 8; ++i) s += a[i];
    }
  }
  else
  {
    if (MFMA == 1)
    {
      f32x16 acc[2];
      for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
      const float a = g[lane], b = g[lane + 64];
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    else if (MFMA == 2)
    {
      f32x16 acc[2];
      for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
      f16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (_Float16)g[lane + i]; b[i] = (_Float16)g[lane + 64 + i]; }
      for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
  }
  out[blockIdx.x * 512 + tid] = s;
}


// In-wave mix: every wave (NW of them, 1 or 2 per SIMD) issues, per MFMA, NF independent plain fp32 FMAs in program order.
template <int NF, int NW, int MF16>
__global__ void __launch_bounds__(NW * 64) kmix(const float* __restrict__ g, float* out, int iters)
{
  __shared__ float hog[30000];
  const int tid = threadIdx.x, lane = tid & 63;
  hog[tid] = g[tid];
  __syncthreads();
  float s = hog[(tid * 7) & 255];
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  const float a = g[lane], b = g[lane + 64];
  f16x8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)g[lane + i]; bh[i] = (_Float16)g[lane + 64 + i]; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = g[lane + 100 + i];
  const float m = g[lane + 200], c = g[lane + 400];
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
      if (MF16)
        acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[r & 1], 0, 0, 0);
      else
        acc[r & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[r & 1], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NF; ++i) f[i & 7] = __builtin_fmaf(f[i & 7], m, c);
      __builtin_amdgcn_sched_barrier(0);
    }
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 512 + tid] = s;
}

template <int NF, int NW, int MF16>
void mixrun(const float* g, float* out, int iters)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((kmix<NF, NW, MF16>), dim3(256), dim3(NW * 64), 0, 0, g, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kmix<NF, NW, MF16>), dim3(256), dim3(NW * 64), 0, 0, g, out, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("in-wave mix %s: %d waves/SIMD, %2d fma per MFMA: %.3f ms (%.1f cycles per MFMA at 2.4 GHz)\n", MF16 ? "f16 32x32x16" : "f32 32x32x2 ",
      NW / 4, NF, ms / 5, ms / 5 * 2.4e6 / (iters * 16.0));
}

// In-wave cost table: per fp32 MFMA, NF copies of ONE kind of filler instruction in program order (NW/4 waves per SIMD).
// Everything in the loop body is volatile inline asm: the optimiser can neither merge, pack nor move the fillers.
// kind: 0 v_fma_f32, 1 ds_read_b32, 2 ds_read_b64, 3 ds_read_b128, 4 s_add (SALU), 5 v_pk_fma_f32, 6 ds_write_b32, 7 v_exp_f32,
//       8 v_add_u32
template <int KIND, int NF, int NW>
__global__ void __launch_bounds__(NW * 64) kfill(const float* __restrict__ g, float* out, int iters)
{
  __shared__ __attribute__((aligned(16))) float hog[30000];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += NW * 64) hog[i] = g[i & 4095];
  __syncthreads();
  float s = 0.f;
  f32x16 acc[2];
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
  const float a = g[lane], b = g[lane + 64];
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = g[lane + 100 + i];
  const float m = g[lane + 200], c = g[lane + 400];
  f32x2 p2[4];
  for (int i = 0; i < 4; ++i) p2[i] = f32x2{f[i], f[i + 4]};
  const f32x2 m2 = f32x2{m, m}, c2 = f32x2{c, c};
  f32x4 q4[4];
  for (int i = 0; i < 4; ++i) q4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  int sreg = iters;
  unsigned u[8];
  for (int i = 0; i < 8; ++i) u[i] = tid + i;
  // conflict-free LDS addresses: consecutive lanes -> consecutive 4 / 8 / 16 bytes
  const unsigned l4 = (unsigned)(size_t)(hog + tid), l8 = (unsigned)(size_t)(hog + tid * 2), l16 = (unsigned)(size_t)(hog + tid * 4);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
      if (r & 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(a), "v"(b));
#pragma unroll
      for (int i = 0; i < NF; ++i)
      {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 7]) : "v"(m), "v"(c));
        else if (KIND == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(f[i & 7]) : "v"(l4), "i"((i & 7) * 2048));
        else if (KIND == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(p2[i & 3]) : "v"(l8), "i"((i & 3) * 4096));
        else if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q4[i & 3]) : "v"(l16), "i"((i & 3) * 8192));
        else if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
        else if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2[i & 3]) : "v"(m2), "v"(c2));
        else if (KIND == 6) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(l4), "v"(f[i & 7]), "i"((i & 7) * 2048) : "memory");
        else if (KIND == 7) asm volatile("v_exp_f32 %0, %0" : "+v"(f[i & 7]));
        else if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i & 7]) : "v"(tid));
      }
      if (KIND >= 1 && KIND <= 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i] + (float)u[i];
  for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1] + q4[i][0];
  out[blockIdx.x * 512 + tid] = s + (float)sreg;
}

template <int KIND, int NF, int NW>
float fillrun(const float* g, float* out, int iters)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((kfill<KIND, NF, NW>), dim3(256), dim3(NW * 64), 0, 0, g, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((kfill<KIND, NF, NW>), dim3(256), dim3(NW * 64), 0, 0, g, out, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5 * 2.4e6 / (iters * 16.0);   // cycles per MFMA slot at 2.4 GHz
}

template <int KIND, int NW>
void fillrow(const char* name, const float* g, float* out)
{
  const float c0 = fillrun<KIND, 0, NW>(g, out, 2000), c2 = fillrun<KIND, 2, NW>(g, out, 2000), c4 = fillrun<KIND, 4, NW>(g, out, 2000),
              c8 = fillrun<KIND, 8, NW>(g, out, 2000), c16 = fillrun<KIND, 16, NW>(g, out, 2000);
  printf("%d wave/SIMD, per f32 32x32x2 MFMA (%.1f cyc bare): %-13s x2 %+6.1f  x4 %+6.1f  x8 %+6.1f  x16 %+6.1f cycles\n", NW / 4, c0, name,
      c2 - c0, c4 - c0, c8 - c0, c16 - c0);
}

template <int V, int M>
float run(const float* g, float* out, int iters)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<V, M>), dim3(256), dim3(512), 0, 0, g, out, iters);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<V, M>), dim3(256), dim3(512), 0, 0, g, out, iters);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 5;
}

template <int V, int M>
void trio(const char* vn, const char* mn, const float* g, float* out, int iters)
{
  const float v = run<V, 0>(g, out, iters), m = run<0, M>(g, out, iters), b = run<V, M>(g, out, iters);
  printf("%-10s + %-14s  VALU alone %.3f ms, MFMA alone %.3f ms, together %.3f ms  (sum %.3f, max %.3f)\n", vn, mn, v, m, b, v + m,
      v > m ? v : m);
}

int main()
{
  float *g, *out;
  CHECK(hipMalloc(&g, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 512 * 4));
  float h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
  CHECK(hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice));
  const int iters = 4000;
  // per iteration: VALU arm 128 instructions (64 for exp), MFMA arm 16 instructions
  trio<1, 1>("pk_fma_f32", "mfma f32 32x32x2", g, out, iters);
  trio<2, 1>("fma_f32", "mfma f32 32x32x2", g, out, iters);
  trio<3, 1>("int add/xor", "mfma f32 32x32x2", g, out, iters);
  trio<4, 1>("exp_f32", "mfma f32 32x32x2", g, out, iters);
  trio<1, 2>("pk_fma_f32", "mfma f16 32x32x16", g, out, iters);
  trio<2, 2>("fma_f32", "mfma f16 32x32x16", g, out, iters);
  trio<3, 2>("int add/xor", "mfma f16 32x32x16", g, out, iters);
  fillrow<0, 4>("v_fma_f32", g, out); fillrow<8, 4>("v_add_u32", g, out); fillrow<5, 4>("v_pk_fma_f32", g, out); fillrow<7, 4>("v_exp_f32", g, out);
  fillrow<4, 4>("s_add_u32", g, out); fillrow<1, 4>("ds_read_b32", g, out); fillrow<2, 4>("ds_read_b64", g, out);
  fillrow<3, 4>("ds_read_b128", g, out); fillrow<6, 4>("ds_write_b32", g, out);
  fillrow<0, 8>("v_fma_f32", g, out); fillrow<3, 8>("ds_read_b128", g, out);
  return 0;
}
