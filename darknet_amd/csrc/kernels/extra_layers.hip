// extra_layers.hip -- HBM-bound kernels of the sibling-cfg layer kinds:
// global average pooling and scale_channels (squeeze-and-excitation), forward and backward.
// Reference twins: avgpool_layer_kernels.cu:9-44 (CPU: avgpool_layer.cpp:40-72),
// blas_kernels.cu scale_channels_kernel / backward_scale_channels_kernel
// (CPU: scale_channels_layer.c:70-127).  Numerics follow the CPU path: the average is a
// sequential fp32 sum in pixel order divided by h*w (one lane per (image, channel) plane would be
// slow; a wave sums a plane with lane-strided partial sums combined in a FIXED order, so results
// are reproducible run to run; parity with the sequential CPU sum is a rounding-level tolerance).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_internal.h"
#include "dk_device_math.h"

namespace
{
inline hipStream_t S(void* s) { return s ? (hipStream_t)s : get_cuda_stream(); }
inline int grid_for(size_t work, int threads = 256)
{
  size_t b = (work + threads - 1) / threads;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

// one wave per plane: out[plane] = sum(in[plane][0..hw)) / hw
__global__ void avgpool_forward_kernel(const float* __restrict__ in, float* __restrict__ out, size_t planes, int hw)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t p = wave; p < planes; p += nwaves)
  {
    const float* src = in + p * (size_t)hw;
    double acc = 0;   // fp64 partials: closer to the exact mean than any fp32 order
    for (int i = lane; i < hw; i += 64) acc += (double)src[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0)
      out[p] = (float)(acc / (double)hw);
  }
}

// prev_delta[plane][i] += delta[plane] / hw
__global__ void avgpool_backward_kernel(const float* __restrict__ delta, float* __restrict__ prev, size_t total, int hw)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    prev[i] += delta[i / hw] / hw;
}

// out[i] = act(in[scale index] * from[i]); scale_wh = 0: one scale per (image, channel) plane;
// scale_wh = 1: one scale per (image, pixel)
__global__ void scale_channels_kernel(const float* __restrict__ in, const float* __restrict__ from,
    float* __restrict__ out, size_t size, int channel_size, int batch_size, int scale_wh, int act)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < size; i += (size_t)gridDim.x * blockDim.x)
  {
    const size_t si = scale_wh ? (i % channel_size + (i / batch_size) * channel_size) : i / channel_size;
    out[i] = dk_activate(in[si] * from[i], act);
  }
}

// from_delta[i] += in[si] * delta[i]  (elementwise);   in_delta[si] += sum_i delta[i] * from[i]
// (one wave per scale element in the scale_wh = 0 case: the plane is reduced in a fixed order)
__global__ void scale_channels_backward_planes(const float* __restrict__ delta, const float* __restrict__ in,
    const float* __restrict__ from, float* __restrict__ from_delta, float* __restrict__ in_delta, size_t planes,
    int channel_size)
{
  const int lane = threadIdx.x & 63;
  const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
  const size_t nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t p = wave; p < planes; p += nwaves)
  {
    const size_t base = p * (size_t)channel_size;
    const float s = in[p];
    float acc = 0;
    for (int i = lane; i < channel_size; i += 64)
    {
      const float d = delta[base + i];
      acc += d * from[base + i];
      if (from_delta)
        from_delta[base + i] += s * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0 && in_delta)
      in_delta[p] += acc;
  }
}

// scale_wh = 1: the scale tensor is [batch][1][h][w]; every (image, pixel) sums over channels
__global__ void scale_channels_backward_wh(const float* __restrict__ delta, const float* __restrict__ in,
    const float* __restrict__ from, float* __restrict__ from_delta, float* __restrict__ in_delta, int batch,
    int channels, int channel_size)
{
  const size_t total = (size_t)batch * channel_size;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x)
  {
    const size_t b = t / channel_size, px = t - b * channel_size;
    const float s = in[t];
    float acc = 0;
    for (int c = 0; c < channels; ++c)
    {
      const size_t i = (b * channels + c) * (size_t)channel_size + px;
      const float d = delta[i];
      acc += d * from[i];
      if (from_delta)
        from_delta[i] += s * d;
    }
    if (in_delta)
      in_delta[t] += acc;
  }
}
}  // namespace

extern "C" int dk_avgpool_forward(const float* in, float* out, int batch, int c, int h, int w, void* stream)
{
  if (!in || !out || batch < 1 || c < 1 || h < 1 || w < 1)
  {
    fprintf(stderr, "dk_avgpool_forward: invalid arguments\n");
    return 1;
  }
  const size_t planes = (size_t)batch * c;
  hipLaunchKernelGGL(avgpool_forward_kernel, dim3(grid_for(planes * 64)), dim3(256), 0, S(stream), in, out, planes, h * w);
  return 0;
}

extern "C" int dk_avgpool_backward(const float* delta, float* prev_delta, int batch, int c, int h, int w, void* stream)
{
  if (!delta || !prev_delta)
  {
    fprintf(stderr, "dk_avgpool_backward: invalid arguments\n");
    return 1;
  }
  const size_t total = (size_t)batch * c * h * w;
  hipLaunchKernelGGL(avgpool_backward_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), delta, prev_delta, total, h * w);
  return 0;
}

extern "C" int dk_scale_channels_forward(const float* in, const float* from, float* out, int batch, int out_c,
    int out_h, int out_w, int scale_wh, int activation, void* stream)
{
  if (!in || !from || !out)
  {
    fprintf(stderr, "dk_scale_channels_forward: invalid arguments\n");
    return 1;
  }
  const size_t size = (size_t)batch * out_c * out_h * out_w;
  hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(size)), dim3(256), 0, S(stream), in, from, out, size,
      out_h * out_w, out_c * out_h * out_w, scale_wh, activation);
  return 0;
}

extern "C" int dk_scale_channels_backward(const float* delta, const float* in, const float* from, float* from_delta,
    float* in_delta, int batch, int out_c, int out_h, int out_w, int scale_wh, void* stream)
{
  if (!delta || !in || !from)
  {
    fprintf(stderr, "dk_scale_channels_backward: invalid arguments\n");
    return 1;
  }
  const int cs = out_h * out_w;
  if (scale_wh)
    hipLaunchKernelGGL(scale_channels_backward_wh, dim3(grid_for((size_t)batch * cs)), dim3(256), 0, S(stream), delta,
        in, from, from_delta, in_delta, batch, out_c, cs);
  else
    hipLaunchKernelGGL(scale_channels_backward_planes, dim3(grid_for((size_t)batch * out_c * 64)), dim3(256), 0,
        S(stream), delta, in, from, from_delta, in_delta, (size_t)batch * out_c, cs);
  return 0;
}


// ---- [dropout], train mode (src/dropout_layer_kernels.cu: cuda_random + yoloswag420blazeit360noscope; CPU
// dropout_layer.c:90-104) ------------------------------------------------------------------------------------
// rand[i] uniform in [0, 1); x[i] = rand[i] < probability ? 0 : x[i] * scale, in place; the backward pass applies the
// same mask and scale to the delta.  The reference draws from cuRAND (GPU) or rand() (CPU): neither stream can be
// reproduced, so the draw here is a counter-based hash of (seed, i) -- same distribution, "parity unpinned".
namespace
{
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long i)
{
  unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (i + 1);   // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0, 1)
}
__global__ void dropout_forward_kernel(float* __restrict__ x, float* __restrict__ rnd, size_t n, float prob, float scale,
    unsigned long long seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    const float r = hash_uniform(seed, i);
    rnd[i] = r;
    x[i] = (r < prob) ? 0.f : x[i] * scale;
  }
}
__global__ void random_uniform_kernel(float* __restrict__ x, size_t n, unsigned long long seed)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = hash_uniform(seed, i);
}
__global__ void dropout_backward_kernel(float* __restrict__ delta, const float* __restrict__ rnd, size_t n, float prob,
    float scale)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    delta[i] = (rnd[i] < prob) ? 0.f : delta[i] * scale;
}
}  // namespace

extern "C" int dk_dropout_forward(float* x, float* rnd, size_t n, float probability, float scale,
    unsigned long long seed, void* stream)
{
  if (n == 0)
    return 0;
  if (!x || !rnd)
    return 1;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 65535u) g = 65535u;
  hipLaunchKernelGGL(dropout_forward_kernel, dim3(g), dim3(256), 0, stream ? (hipStream_t)stream : get_cuda_stream(), x, rnd, n,
      probability, scale, seed);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_dropout_backward(float* delta, const float* rnd, size_t n, float probability, float scale, void* stream)
{
  if (n == 0)
    return 0;
  if (!delta || !rnd)
    return 1;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 65535u) g = 65535u;
  hipLaunchKernelGGL(dropout_backward_kernel, dim3(g), dim3(256), 0, stream ? (hipStream_t)stream : get_cuda_stream(), delta,
      rnd, n, probability, scale);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// Uniform [0, 1) fill of a device array: what dark_cuda.c:464-477 (`cuda_random`) gets from cuRAND, here the
// counter-based hash of (seed, i) the dropout layer uses.
extern "C" int dk_random_uniform(float* x, size_t n, unsigned long long seed, void* stream)
{
  if (n == 0)
    return 0;
  if (!x)
    return 1;
  unsigned g = (unsigned)((n + 255) / 256);
  if (g > 65535u) g = 65535u;
  hipLaunchKernelGGL(random_uniform_kernel, dim3(g), dim3(256), 0, stream ? (hipStream_t)stream : get_cuda_stream(), x, n, seed);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
