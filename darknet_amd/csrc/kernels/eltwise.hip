// eltwise.hip -- the HBM-bound elementwise kernels of the forward path:
// route copy, shortcut add, upsample, bias/scale, activations, fill/copy/axpy/scal.
// All are grid-stride, 16 bytes per lane where alignment allows (coalesced
// 1 KiB per wave instruction), <= 2048 blocks of 256 threads.
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
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// dst[j*dst_stride + i] = src[j*src_stride + i], i < part, j < batch
template <int VEC>
__global__ void copy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int part,
    int batch, size_t src_stride, size_t dst_stride)
{
  const int pv = part / VEC;
  const size_t total = (size_t)pv * batch;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(i / pv);
    const int e = (int)(i - (size_t)j * pv);
    if (VEC == 4)
      ((float4*)(dst + j * dst_stride))[e] = ((const float4*)(src + j * src_stride))[e];
    else
      dst[j * dst_stride + e] = src[j * src_stride + e];
  }
}

template <int VEC>
__global__ void shortcut_kernel(const float* __restrict__ a, const float* __restrict__ b,
    float* __restrict__ out, size_t n, int act)
{
  const size_t nv = n / VEC;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < nv;
       i += (size_t)gridDim.x * blockDim.x)
  {
    if (VEC == 4)
    {
      float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i], r;
      r.x = dk_activate(x.x + y.x, act);
      r.y = dk_activate(x.y + y.y, act);
      r.z = dk_activate(x.z + y.z, act);
      r.w = dk_activate(x.w + y.w, act);
      ((float4*)out)[i] = r;
    }
    else
      out[i] = dk_activate(a[i] + b[i], act);
  }
}

// one thread per OUTPUT element pair: out row = 2 floats per input float for stride 2
__global__ void upsample_kernel(const float* __restrict__ in, float* __restrict__ out, int w, int h,
    size_t planes, int stride, float scale, int c, size_t out_bstride)
{
  const int ow = w * stride, oh = h * stride;
  const size_t total = planes * (size_t)oh * ow;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int ox = (int)(i % ow);
    const size_t t = i / ow;
    const int oy = (int)(t % oh);
    const size_t pl = t / oh;
    const size_t b = pl / c;
    out[b * out_bstride + ((pl - b * c) * oh + oy) * (size_t)ow + ox] =
        scale * in[(pl * h + oy / stride) * w + ox / stride];
  }
}

__global__ void bias_kernel(float* __restrict__ out, const float* __restrict__ v, int n, int size,
    size_t total, int mul)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int f = (int)((i / size) % n);
    out[i] = mul ? out[i] * v[f] : out[i] + v[f];
  }
}

__global__ void activate_kernel(float* __restrict__ x, size_t n, int act)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] = dk_activate(x[i], act);
}

__global__ void mish_kernel(const float* __restrict__ x, size_t n, float* __restrict__ act_in,
    float* __restrict__ out)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const float v = x[i];
    if (act_in)
      act_in[i] = v;
    out[i] = dk_mish(v);
  }
}

__global__ void fill_kernel(size_t n, float a, float* __restrict__ x)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] = a;
}
__global__ void axpy_kernel(size_t n, float a, const float* __restrict__ x, float* __restrict__ y)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    y[i] += a * x[i];
}
__global__ void scal_kernel(size_t n, float a, float* __restrict__ x)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] *= a;
}
__global__ void constrain_kernel(size_t n, float a, float* __restrict__ x)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    x[i] = fminf(a, fmaxf(-a, x[i]));
}
}  // namespace

extern "C" int dk_route_copy(const float* src, int input_size, int groups, int group_id, int batch,
    float* dst, int outputs, int offset, void* stream)
{
  if (!src || !dst || groups < 1 || input_size % groups)
  {
    fprintf(stderr, "dk_route_copy: invalid arguments\n");
    return 1;
  }
  const int part = input_size / groups;
  if (part == 0 || batch == 0)
    return 0;
  const float* s = src + (size_t)part * group_id;
  float* d = dst + offset;
  const bool v4 = (part % 4 == 0) && (input_size % 4 == 0) && (outputs % 4 == 0) && al16(s) && al16(d);
  if (v4)
    hipLaunchKernelGGL(copy2d_kernel<4>, dim3(grid_for((size_t)part / 4 * batch)), dim3(256), 0,
        S(stream), s, d, part, batch, (size_t)input_size, (size_t)outputs);
  else
    hipLaunchKernelGGL(copy2d_kernel<1>, dim3(grid_for((size_t)part * batch)), dim3(256), 0,
        S(stream), s, d, part, batch, (size_t)input_size, (size_t)outputs);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_shortcut_forward(const float* in, const float* from, float* out, size_t total,
    int activation, void* stream)
{
  if (!in || !from || !out)
  {
    fprintf(stderr, "dk_shortcut_forward: null pointer\n");
    return 1;
  }
  if (total == 0)
    return 0;
  if (total % 4 == 0 && al16(in) && al16(from) && al16(out))
    hipLaunchKernelGGL(shortcut_kernel<4>, dim3(grid_for(total / 4)), dim3(256), 0, S(stream), in,
        from, out, total, activation);
  else
    hipLaunchKernelGGL(shortcut_kernel<1>, dim3(grid_for(total)), dim3(256), 0, S(stream), in,
        from, out, total, activation);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_upsample_forward(const float* in, int w, int h, int c, int batch, int stride,
    float scale, float* out, void* stream)
{
  return dk_upsample_forward_strided(in, w, h, c, batch, stride, scale, out, 0, stream);
}

int dk_upsample_forward_strided(const float* in, int w, int h, int c, int batch, int stride,
    float scale, float* out, size_t out_batch_stride, void* stream)
{
  if (!in || !out || stride < 1)
  {
    fprintf(stderr, "dk_upsample_forward: invalid arguments\n");
    return 1;
  }
  const size_t planes = (size_t)c * batch;
  const size_t total = planes * h * w * stride * stride;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), in, out, w,
      h, planes, stride, scale, c,
      out_batch_stride ? out_batch_stride : (size_t)c * h * w * stride * stride);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_add_bias(float* out, const float* biases, int batch, int n, int size, void* stream)
{
  const size_t total = (size_t)batch * n * size;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(bias_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), out, biases, n,
      size, total, 0);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_scale_bias(float* out, const float* scales, int batch, int n, int size, void* stream)
{
  const size_t total = (size_t)batch * n * size;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(bias_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), out, scales, n,
      size, total, 1);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_activate_array(float* x, size_t n, int activation, void* stream)
{
  if (n == 0 || activation == DK_LINEAR)
    return 0;
  hipLaunchKernelGGL(activate_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), x, n, activation);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_activate_array_mish(const float* x, size_t n, float* activation_input, float* out,
    void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(mish_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), x, n,
      activation_input, out);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_fill(size_t n, float alpha, float* x, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), n, alpha, x);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
extern "C" int dk_copy(size_t n, const float* x, float* y, void* stream)
{
  if (n == 0)
    return 0;
  CHECK_HIP(hipMemcpyAsync(y, x, n * sizeof(float), hipMemcpyDeviceToDevice, S(stream)));
  return 0;
}
extern "C" int dk_axpy(size_t n, float alpha, const float* x, float* y, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), n, alpha, x, y);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
extern "C" int dk_scal(size_t n, float alpha, float* x, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(scal_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), n, alpha, x);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// constrain_ongpu (src/blas_kernels.cu:450-455, :874; CPU twin constrain_cpu blas.c:408): x = min(a, max(-a, x));
// what `clip=` applies to a layer's weights after every update (convolutional_kernels.cu:919-920)
extern "C" int dk_constrain(size_t n, float alpha, float* x, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(constrain_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), n, alpha, x);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
