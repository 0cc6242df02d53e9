// train_ops.hip -- the bandwidth-bound kernels of the training path: batch-norm
// forward (batch statistics) and backward, activation gradients, bias gradient,
// backward of maxpool / route / shortcut / upsample / yolo, the fused SGD update
// and the weight transpose used by the data-gradient convolution.
//
// Parity target = the reference's CPU functions (the GPU twins differ in eps,
// rolling momentum and variance denominator -- SURVEY.md section 8a quirks 1-2):
//   mean_cpu / variance_cpu / normalize_cpu      src/blas.c:164-218
//   ForwardBatchnormLayer                        src/batchnorm_layer.cpp:206-238
//   backward_scale_cpu / mean_delta_cpu / variance_delta_cpu / normalize_delta_cpu
//                                                src/batchnorm_layer.cpp:92-165
//   gradient_array / gradient_array_mish         src/activations.c:401-452
//   backward_bias                                src/convolutional_layer.cpp:946-957
//   BackwardMaxpoolLayer                         src/maxpool_layer.cpp:312-324
//   BackwardRouteLayer                           src/route_layer.c:106-122
//   BackwardShortcutCpu                          src/blas.c:101-129
//   upsample_cpu (forward = 0)                   src/blas.c:382-406
//   UpdateConvolutionalLayer                     src/convolutional_layer.cpp:1382-1399
// Per-channel reductions are two-stage (slices of a channel per workgroup: wave
// shuffles + LDS tree in double, fp64 atomics into a per-channel scratch, then a
// finalize kernel); the summation order differs from the CPU's sequential fp32
// loop -- see tests/util.py TRAIN_ATOL_RMS.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>
#include <vector>
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
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

constexpr int RT = 512;  // threads per reduction workgroup

// per-device fp64 scratch for the two-stage channel reductions (4 doubles per channel).
// Zero on entry by contract: it is cleared once when allocated and every finalize kernel clears
// the entries it has consumed, so no per-call memset is needed (there used to be ~300 per step).
// (Keyed by device AND stream: replicas that share a device run on streams of their own -- dk_set_thread_stream --
// and must not share reduction scratch.)
double* chan_scratch(int filters, hipStream_t st)
{
  struct Buf { double* p = nullptr; int cap = 0; };
  static std::map<std::pair<int, void*>, Buf> bufs;
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  Buf& b = bufs[{cuda_get_device(), (void*)st}];
  if (b.cap < filters)
  {
    if (b.p)
    {
      CHECK_HIP(hipStreamSynchronize(st));
      CHECK_HIP(hipFree(b.p));
    }
    b.cap = filters < 4096 ? 4096 : filters;
    CHECK_HIP(hipMalloc((void**)&b.p, (size_t)b.cap * 4 * sizeof(double)));
    CHECK_HIP(hipMemsetAsync(b.p, 0, (size_t)b.cap * 4 * sizeof(double), st));
  }
  return b.p;
}

// Scratch of the FUSED statistics paths (dk_bn_forward_train, dk_bn_act_backward): the consumer kernel
// (bn_apply / bn_act_delta) finishes the statistics itself, so nobody is left to clear the sums; every call takes a
// fresh slice of a per-device ring instead, and the ring is cleared as a whole when it wraps (one memset every few
// steps instead of two tiny finalize launches per layer and pass: 217 launches per yolov4 step).
double* chan_ring_take(size_t doubles, hipStream_t st)
{
  struct Ring { double* p = nullptr; size_t off = 0; };
  static std::map<std::pair<int, void*>, Ring> rings;   // per (device, stream), see chan_scratch
  static std::mutex mu;
  constexpr size_t CAP = (size_t)1 << 20;   // 8 MB
  if (doubles > CAP)
    return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  Ring& r = rings[{cuda_get_device(), (void*)st}];
  if (!r.p)
  {
    CHECK_HIP(hipMalloc((void**)&r.p, CAP * sizeof(double)));
    CHECK_HIP(hipMemsetAsync(r.p, 0, CAP * sizeof(double), st));
    r.off = 0;
  }
  if (r.off + doubles > CAP)
  {
    CHECK_HIP(hipMemsetAsync(r.p, 0, CAP * sizeof(double), st));
    r.off = 0;
  }
  double* p = r.p + r.off;
  r.off += doubles;
  return p;
}

// split a channel's batch*spatial elements so that ~2048 workgroups exist in total
inline void chan_split(int batch, int filters, int spatial, int* chunks, size_t* slice)
{
  const size_t n = (size_t)batch * spatial;
  size_t c = 2048 / (size_t)(filters > 0 ? filters : 1);
  if (c < 1) c = 1;
  const size_t maxc = (n + 2047) / 2048;  // at least ~2048 elements per workgroup
  if (c > maxc) c = maxc;
  *slice = (n + c - 1) / c;
  *chunks = (int)((n + *slice - 1) / *slice);
}

// Work split of the plane-walking reductions (round 2b): a workgroup of RP threads takes `ipw` whole
// (image, channel) planes of ONE channel, or one slice of a plane when a plane alone is large, so that it
// has >= ~8 K elements to stream (several independent 16-byte loads in flight per thread) and pays for ONE
// block reduction; small layers trade elements per workgroup (down to ~2 K) for >= ~1024 workgroups.
// grid = (cps * image groups, filters).
constexpr int RP = 256;
inline void plane_split2(int batch, int filters, int spatial, int* cps, int* slice, int* ipw)
{
  const long long n = (long long)batch * spatial;          // elements per channel
  const int f = filters > 0 ? filters : 1;
  long long wpc = n / 8192;                                 // workgroups per channel
  long long fill = 1024 / f;
  if (fill > n / 2048) fill = n / 2048;
  if (wpc < fill) wpc = fill;
  if (wpc < 1) wpc = 1;
  const long long per = (n + wpc - 1) / wpc;                // elements per workgroup
  if (per >= spatial)
  {
    int k = (int)(per / spatial);
    if (k < 1) k = 1;
    if (k > batch) k = batch;
    *ipw = k;
    *cps = 1;
    *slice = (spatial + 3) & ~3;
  }
  else
  {
    int c = (int)((spatial + per - 1) / per);
    int sl = (spatial + c - 1) / c;
    sl = (sl + 3) & ~3;
    *ipw = 1;
    *slice = sl;
    *cps = (spatial + sl - 1) / sl;
  }
}

// sums NV doubles over the workgroup (RP threads) and adds them to dst[0..NV) with fp64 atomics:
// wave shuffles, one LDS exchange, one barrier
template <int NV, bool STORE = false>
__device__ __forceinline__ void block_reduce_atomic(double (&v)[NV], double* __restrict__ dst)
{
  __shared__ double sh[NV * (RP / 64)];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_down(v[k], o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0)
  {
#pragma unroll
    for (int k = 0; k < NV; ++k) sh[k * (RP / 64) + w] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < NV)
  {
    double r = 0;
#pragma unroll
    for (int i = 0; i < RP / 64; ++i) r += sh[threadIdx.x * (RP / 64) + i];
    if (STORE)
      dst[threadIdx.x] = r;     // deterministic mode: this workgroup's own slot
    else
      atomicAdd(&dst[threadIdx.x], r);
  }
}

// The NV sums of channel f, for the first wave of a consumer workgroup (all 64 lanes call; the result is valid on
// lane 0).  nwg == 0: the atomics layout [f][NV]; nwg > 0 (deterministic mode): the partials [f][nwg][NV] of the nwg
// producer workgroups, added lane-strided and then across the wave -- a fixed order for a fixed nwg.
template <int NV>
__device__ __forceinline__ void chan_sums(const double* __restrict__ sums, int f, int nwg, double (&out)[NV])
{
  if (nwg == 0)
  {
#pragma unroll
    for (int k = 0; k < NV; ++k) out[k] = sums[NV * f + k];
    return;
  }
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < NV; ++k) out[k] = 0;
  for (int w = lane; w < nwg; w += 64)
#pragma unroll
    for (int k = 0; k < NV; ++k) out[k] += sums[((size_t)f * nwg + w) * NV + k];
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) out[k] += __shfl_down(out[k], o, 64);
}

__device__ __forceinline__ double block_sum(double v, double* sh)
{
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0)
    sh[w] = v;
  __syncthreads();
  double r = 0;
  if (threadIdx.x == 0)
  {
    for (int i = 0; i < RT / 64; ++i) r += sh[i];
    sh[0] = r;
  }
  __syncthreads();
  r = sh[0];
  return r;
}

// element (b, f, i) of a [batch][filters][spatial] tensor, for a flat per-channel index t
__device__ __forceinline__ size_t chan_index(size_t t, int f, int filters, int spatial)
{
  const size_t b = t / spatial, i = t - b * spatial;
  return (b * filters + f) * (size_t)spatial + i;
}

// ---- batch-norm forward -------------------------------------------------------
// Statistics in two stages so that low-channel layers still fill the chip: stage 1,
// grid (chunks, filters): every workgroup reduces a slice of one channel to
// {sum, sum of squares} in double and adds them to a per-channel scratch with fp64
// atomics; stage 2 (inside bn_apply_kernel since round 2b): mean = S/N, variance = (Q - S*S/N)/(N-1)
// (the reference's N-1 denominator, src/blas.c:186; evaluated in double, so the
// one-pass form loses nothing), rolling statistics .9/.1.
__global__ void __launch_bounds__(RP) bn_partial_kernel(const float* __restrict__ x, int batch,
    int filters, int spatial, int cps, int slice, int ipw, double* __restrict__ scratch, int det)
{
  const int f = blockIdx.y;
  const int bg = blockIdx.x / cps, ch = blockIdx.x - bg * cps;
  const int b0 = bg * ipw;
  const int b1 = (b0 + ipw < batch) ? b0 + ipw : batch;
  const int i0 = ch * slice;
  const int i1 = (i0 + slice < spatial) ? i0 + slice : spatial;
  double acc[2] = {0, 0};
  const bool vec = (spatial & 3) == 0 && (((uintptr_t)x) & 15) == 0;
  for (int b = b0; b < b1; ++b)
  {
    const float* const px = x + ((size_t)b * filters + f) * spatial;
    if (vec)
    {
      int i = i0 + 4 * (int)threadIdx.x;
      // four independent 16-byte loads in flight per thread
      for (; i + 12 * RP < i1; i += 16 * RP)
      {
        const float4 v0 = *(const float4*)(px + i);
        const float4 v1 = *(const float4*)(px + i + 4 * RP);
        const float4 v2 = *(const float4*)(px + i + 8 * RP);
        const float4 v3 = *(const float4*)(px + i + 12 * RP);
        const float4 vv[4] = {v0, v1, v2, v3};
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
          acc[0] += (double)vv[k].x + (double)vv[k].y + (double)vv[k].z + (double)vv[k].w;
          acc[1] += (double)vv[k].x * vv[k].x + (double)vv[k].y * vv[k].y + (double)vv[k].z * vv[k].z + (double)vv[k].w * vv[k].w;
        }
      }
      for (; i < i1; i += 4 * RP)
      {
        const float4 v = *(const float4*)(px + i);
        acc[0] += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        acc[1] += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
    else
    {
      int i = i0 + (int)threadIdx.x;
      for (; i + RP < i1; i += 2 * RP)
      {
        const double v0 = px[i], v1 = px[i + RP];
        acc[0] += v0 + v1;
        acc[1] += v0 * v0 + v1 * v1;
      }
      for (; i < i1; i += RP)
      {
        const double v = px[i];
        acc[0] += v;
        acc[1] += v * v;
      }
    }
  }
  if (det)
    block_reduce_atomic<2, true>(acc, scratch + ((size_t)f * gridDim.x + blockIdx.x) * 2);
  else
    block_reduce_atomic<2>(acc, scratch + 2 * f);
}

// normalize_cpu (eps 1e-6) + scale_bias + add_bias + activation, one pass.
// grid (batch*filters, chunks): one (image, channel) plane per blockIdx.x -> channel constants are
// scalars, no per-element division, 16-byte accesses when the planes allow it.
__device__ __forceinline__ float bn_apply_one(float v, float m, float div, float sc, float bi, int act,
    float* xn_out, float* pre_out)
{
  float xn = (v - m) / div;
  *xn_out = xn;
  xn = xn * sc;
  xn = xn + bi;
  *pre_out = xn;
  return dk_activate(xn, act);
}

// sums != nullptr (train): the channel's {sum, sum of squares} from bn_partial_kernel; mean and variance are formed
// here (mean = S/N, variance = (Q - S*S/N)/(N-1) in double), and the workgroup of image 0 / chunk 0 stores them and moves the
// rolling statistics (mean / variance / rolling_* are outputs then)
__global__ void bn_apply_kernel(const float* __restrict__ raw, float* __restrict__ x_save,
    float* __restrict__ x_norm, float* __restrict__ act_in, float* __restrict__ out,
    float* __restrict__ mean, float* __restrict__ variance,
    const float* __restrict__ scales, const float* __restrict__ biases, int filters, int spatial,
    int act, int vec, const double* __restrict__ sums, int batch, float* __restrict__ rolling_mean,
    float* __restrict__ rolling_variance, int nwg)
{
  const int plane = blockIdx.x;
  const int f = plane % filters;
  float m, var;
  if (sums)
  {
    __shared__ float stat[2];
    double sq[2] = {0, 0};
    if (threadIdx.x < 64)
      chan_sums<2>(sums, f, nwg, sq);
    if (threadIdx.x == 0)
    {
      const double n = (double)batch * spatial;
      const double s = sq[0], q = sq[1];
      const float mm = (float)(s / n);
      double v = (q - s * s / n) / (n - 1);
      if (v < 0)
        v = 0;
      const float vv = (float)v;
      stat[0] = mm;
      stat[1] = vv;
      if (plane < filters && blockIdx.y == 0)
      {
        mean[f] = mm;
        variance[f] = vv;
        // rolling = .9*rolling + .1*batch: scal_cpu then axpy_cpu, batchnorm_layer.cpp:221-224
        float rm = rolling_mean[f] * .9f;
        rm += .1f * mm;
        rolling_mean[f] = rm;
        float rv = rolling_variance[f] * .9f;
        rv += .1f * vv;
        rolling_variance[f] = rv;
      }
    }
    __syncthreads();
    m = stat[0];
    var = stat[1];
  }
  else
  {
    m = mean[f];
    var = variance[f];
  }
  const float div = sqrtf(var + .000001f), sc = scales[f], bi = biases[f];
  const size_t base = (size_t)plane * spatial;
  if (vec)
  {
    auto four = [&](const float4 v, int i) {
      float4 xn, pre, o;
      o.x = bn_apply_one(v.x, m, div, sc, bi, act, &xn.x, &pre.x);
      o.y = bn_apply_one(v.y, m, div, sc, bi, act, &xn.y, &pre.y);
      o.z = bn_apply_one(v.z, m, div, sc, bi, act, &xn.z, &pre.z);
      o.w = bn_apply_one(v.w, m, div, sc, bi, act, &xn.w, &pre.w);
      if (x_save)
        *(float4*)(x_save + base + i) = v;
      if (x_norm)
        *(float4*)(x_norm + base + i) = xn;
      if (act_in)
        *(float4*)(act_in + base + i) = pre;
      *(float4*)(out + base + i) = o;
    };
    const int step = 4 * gridDim.y * blockDim.x;
    int i = 4 * (blockIdx.y * blockDim.x + threadIdx.x);
    // two 16-byte loads in flight per thread
    for (; i + step < spatial; i += 2 * step)
    {
      const float4 va = *(const float4*)(raw + base + i);
      const float4 vb = *(const float4*)(raw + base + i + step);
      four(va, i);
      four(vb, i + step);
    }
    for (; i < spatial; i += step) four(*(const float4*)(raw + base + i), i);
    return;
  }
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < spatial; i += gridDim.y * blockDim.x)
  {
    const float v = raw[base + i];
    float xn, pre;
    const float o = bn_apply_one(v, m, div, sc, bi, act, &xn, &pre);
    if (x_save)
      x_save[base + i] = v;
    if (x_norm)
      x_norm[base + i] = xn;
    if (act_in)
      act_in[base + i] = pre;
    out[base + i] = o;
  }
}

// ---- activation gradient -------------------------------------------------------
__global__ void gradient_kernel(const float* __restrict__ y, const float* __restrict__ act_in,
    float* __restrict__ delta, size_t n, int act)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
  {
    // gradient_array / gradient_array_mish / gradient_array_swish, activations.c:401-452
    const float pre = (act == DK_MISH || act == 16) ? act_in[i] : 0.f;
    delta[i] *= dk_act_gradient(y[i], pre, act);
  }
}

// ---- per-channel reductions of the backward pass ----------------------------------
// mode 0: bias_updates[f] += sum(delta)                      (backward_bias)
// mode 1: batch-norm: scale_updates[f] += sum(delta*x_norm); bias_updates[f] += sum(delta);
//         mean_delta[f] = sum(delta*scale) * (-1/sqrt(var+1e-5));
//         variance_delta[f] = sum(delta*scale*(x-mean)) * (-.5*pow(var+1e-5,-1.5))
// Same two-stage scheme as the forward statistics (grid (chunks, filters) + fp64 atomics).
__global__ void __launch_bounds__(RT) chan_partial_kernel(const float* __restrict__ delta,
    const float* __restrict__ x, const float* __restrict__ x_norm, const float* __restrict__ mean,
    const float* __restrict__ scales, int batch, int filters, int spatial, size_t slice,
    double* __restrict__ scratch, int mode)
{
  __shared__ double sh[RT / 64];
  const int f = blockIdx.y;
  const size_t n = (size_t)batch * spatial;
  const size_t t0 = blockIdx.x * slice;
  size_t t1 = t0 + slice;
  if (t1 > n)
    t1 = n;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  const float sc = mode ? scales[f] : 1.f;
  const float m = mode ? mean[f] : 0.f;
  for (size_t t = t0 + threadIdx.x; t < t1; t += RT)
  {
    const size_t idx = chan_index(t, f, filters, spatial);
    const float d = delta[idx];
    s0 += d;
    if (mode)
    {
      s1 += d * x_norm[idx];
      const float ds = d * sc;  // scale_bias(delta, scales) rounds here
      s2 += ds;
      s3 += ds * (x[idx] - m);
    }
  }
  s0 = block_sum(s0, sh);
  if (mode)
  {
    s1 = block_sum(s1, sh);
    s2 = block_sum(s2, sh);
    s3 = block_sum(s3, sh);
  }
  if (threadIdx.x == 0)
  {
    atomicAdd(&scratch[4 * f + 0], s0);
    if (mode)
    {
      atomicAdd(&scratch[4 * f + 1], s1);
      atomicAdd(&scratch[4 * f + 2], s2);
      atomicAdd(&scratch[4 * f + 3], s3);
    }
  }
}

__global__ void chan_finalize_kernel(double* __restrict__ scratch,
    const float* __restrict__ variance, int filters, float* __restrict__ bias_updates,
    float* __restrict__ scale_updates, float* __restrict__ mean_delta,
    float* __restrict__ variance_delta, int mode)
{
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= filters)
    return;
  const double s0 = scratch[4 * f + 0], s1 = scratch[4 * f + 1], s2 = scratch[4 * f + 2], s3 = scratch[4 * f + 3];
  scratch[4 * f + 0] = 0;   // leave the scratch zeroed for the next reduction
  scratch[4 * f + 1] = 0;
  scratch[4 * f + 2] = 0;
  scratch[4 * f + 3] = 0;
  if (bias_updates)
    bias_updates[f] += (float)s0;
  if (mode)
  {
    scale_updates[f] += (float)s1;
    const float var = variance[f];
    float md = (float)s2;
    md *= (-1. / sqrtf(var + .00001f));
    mean_delta[f] = md;
    float vd = (float)s3;
    vd *= -.5 * powf(var + .00001f, (float)(-3. / 2.));
    variance_delta[f] = vd;
  }
}

// normalize_delta_cpu, batchnorm_layer.cpp:147-165 (after delta *= scale)
__global__ void bn_delta_kernel(float* __restrict__ delta, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ variance,
    const float* __restrict__ mean_delta, const float* __restrict__ variance_delta,
    const float* __restrict__ scales, int batch, int filters, int spatial, size_t total)
{
  const int nb = spatial * batch;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int f = (int)((i / spatial) % filters);
    const float ds = delta[i] * scales[f];
    delta[i] = ds * 1. / (sqrtf(variance[f]) + .00001f) +
               variance_delta[f] * 2. * (x[i] - mean[f]) / nb + mean_delta[f] / nb;
  }
}


// ---- fused activation-gradient + batch-norm backward ---------------------------------
// Nothing but x (the raw conv output) was saved in the forward pass: x_norm, the
// pre-activation input and the activation gradient are recomputed with the forward's
// own float operations (same values as the stored ones would have), so the backward
// of activation + batch norm reads {delta, x} twice and writes delta once instead of
// the separate gradient_array / reductions / normalize_delta passes (9 streams -> 5)
// and the forward writes only its output (5 streams -> 2).
struct BnRecompute
{
  float xn, d;  // x_norm, delta * activation gradient
};

__device__ __forceinline__ BnRecompute bn_recompute(float x, float delta, float mean, float div_fwd,
    float scale, float bias, int act)
{
  BnRecompute r;
  r.xn = (x - mean) / div_fwd;  // div_fwd = sqrtf(variance + 1e-6f): the forward's divisor
  float a = r.xn * scale;
  a = a + bias;
  float g;
  if (act == (DK_MISH | DK_ACT_FAST))
  {
    // gradient_array_mish in closed form (one exp, two reciprocals): with e = exp(a),
    // w = e(e+2): tanh(softplus(a)) = w/(w+2), 1-exp(-softplus(a)) = e/(1+e).
    // Differs from the libm chain below by ~1e-7 relative; DK_FAST_MISH=0 selects that.
    const float e = __expf(a);
    const float w = e * (e + 2.f);
    const float tsp = (a > 20.f) ? 1.f : w * __builtin_amdgcn_rcpf(w + 2.f);
    const float grad_sp = (a > 20.f) ? 1.f : e * __builtin_amdgcn_rcpf(1.f + e);
    g = a * ((1 - tsp * tsp) * grad_sp) + tsp;
  }
  else if ((act & 0xff) == DK_MISH)
  {
    const float sp = dk_softplus(a, 20.f);
    const float grad_sp = 1 - expf(-sp);
    const float tsp = tanhf(sp);
    const float grad_tsp = (1 - tsp * tsp) * grad_sp;
    g = a * grad_tsp + tsp;
  }
  else if (act == DK_LEAKY)
    g = (dk_leaky(a) > 0) ? 1 : .1f;
  else if (act == DK_LOGISTIC)
  {
    const float y = dk_logistic(a);
    g = (1 - y) * y;
  }
  else if (act == DK_RELU)
    g = (a * (a > 0.f) > 0);
  else if (act == DK_LINEAR)
    g = 1;
  else
    g = dk_act_gradient(dk_activate(a, act), a, act);   // the rarer kinds: recompute the output, then gradient()
  r.d = delta * g;
  return r;
}

__global__ void __launch_bounds__(RP) bn_act_partial_kernel(const float* __restrict__ delta,
    const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ variance,
    const float* __restrict__ scales, const float* __restrict__ biases, int batch, int filters,
    int spatial, int cps, int slice, int ipw, double* __restrict__ scratch, int act, int det)
{
  const int f = blockIdx.y;
  const int bg = blockIdx.x / cps, ch = blockIdx.x - bg * cps;
  const int b0 = bg * ipw;
  const int b1 = (b0 + ipw < batch) ? b0 + ipw : batch;
  const int i0 = ch * slice;
  const int i1 = (i0 + slice < spatial) ? i0 + slice : spatial;
  double acc[4] = {0, 0, 0, 0};
  const float sc = scales[f], bi = biases[f], m = mean[f];
  const float div = sqrtf(variance[f] + .000001f);
  auto one = [&](float xv, float dv) {
    const BnRecompute r = bn_recompute(xv, dv, m, div, sc, bi, act);
    acc[0] += r.d;
    acc[1] += r.d * r.xn;
    const float ds = r.d * sc;
    acc[2] += ds;
    acc[3] += ds * (xv - m);
  };
  const bool vec = (spatial & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)delta)) & 15) == 0;
  for (int b = b0; b < b1; ++b)
  {
    const size_t base = ((size_t)b * filters + f) * spatial;
    if (vec)
    {
      int i = i0 + 4 * (int)threadIdx.x;
      // two planes' worth of 16-byte loads (x and delta, two positions) in flight per thread
      for (; i + 4 * RP < i1; i += 8 * RP)
      {
        const float4 xa = *(const float4*)(x + base + i);
        const float4 da = *(const float4*)(delta + base + i);
        const float4 xb = *(const float4*)(x + base + i + 4 * RP);
        const float4 db = *(const float4*)(delta + base + i + 4 * RP);
        one(xa.x, da.x); one(xa.y, da.y); one(xa.z, da.z); one(xa.w, da.w);
        one(xb.x, db.x); one(xb.y, db.y); one(xb.z, db.z); one(xb.w, db.w);
      }
      for (; i < i1; i += 4 * RP)
      {
        const float4 xv = *(const float4*)(x + base + i);
        const float4 dv = *(const float4*)(delta + base + i);
        one(xv.x, dv.x); one(xv.y, dv.y); one(xv.z, dv.z); one(xv.w, dv.w);
      }
    }
    else
    {
      int i = i0 + (int)threadIdx.x;
      for (; i + RP < i1; i += 2 * RP)
      {
        const float x0 = x[base + i], d0 = delta[base + i], x1 = x[base + i + RP], d1 = delta[base + i + RP];
        one(x0, d0);
        one(x1, d1);
      }
      for (; i < i1; i += RP) one(x[base + i], delta[base + i]);
    }
  }
  if (det)
    block_reduce_atomic<4, true>(acc, scratch + ((size_t)f * gridDim.x + blockIdx.x) * 4);
  else
    block_reduce_atomic<4>(acc, scratch + 4 * f);
}

// grid (batch*filters, chunks): one (image, channel) plane per blockIdx.x -> no per-element division.
// normalize_delta_cpu's expression  ds * 1. / (sqrt(var) + .00001f) + vd * 2. * (x - m) / nb + md / nb  is
// evaluated in double there (the literals promote it); here the three per-channel factors are formed once in
// double and the element costs one fp64 multiply and one fma instead of two fp64 divisions -- the double
// result differs by <= 2 ulp(double) before the final rounding to float.
// sums: the channel's four sums from bn_act_partial_kernel; mean_delta / variance_delta are formed here exactly as
// chan_finalize_kernel forms them, and the workgroup of image 0 / chunk 0 stores them and accumulates bias_updates /
// scale_updates (all four are outputs).
__global__ void bn_act_delta_kernel(float* __restrict__ delta, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ variance,
    float* __restrict__ mean_delta, float* __restrict__ variance_delta,
    const float* __restrict__ scales, const float* __restrict__ biases, int batch, int filters,
    int spatial, int act, int vec, const double* __restrict__ sums, float* __restrict__ bias_updates,
    float* __restrict__ scale_updates, int nwg)
{
  const int plane = blockIdx.x;
  const int f = plane % filters;
  const int nb = spatial * batch;
  const float sc = scales[f], bi = biases[f], m = mean[f], var = variance[f];
  const float div = sqrtf(var + .000001f);
  __shared__ float fin[2];
  double s4[4] = {0, 0, 0, 0};
  if (threadIdx.x < 64)
    chan_sums<4>(sums, f, nwg, s4);
  if (threadIdx.x == 0)
  {
    float md = (float)s4[2];
    md *= (-1. / sqrtf(var + .00001f));
    float vd = (float)s4[3];
    vd *= -.5 * powf(var + .00001f, (float)(-3. / 2.));
    fin[0] = md;
    fin[1] = vd;
    if (plane < filters && blockIdx.y == 0)
    {
      mean_delta[f] = md;
      variance_delta[f] = vd;
      bias_updates[f] += (float)s4[0];
      scale_updates[f] += (float)s4[1];
    }
  }
  __syncthreads();
  const double ka = 1. / (double)(sqrtf(var) + .00001f);
  const double kb = (double)fin[1] * 2. / (double)nb;
  const double kc = (double)(fin[0] / nb);
  float* dp = delta + (size_t)plane * spatial;
  const float* xp = x + (size_t)plane * spatial;
  auto one = [&](float xv, float dv) -> float {
    const BnRecompute r = bn_recompute(xv, dv, m, div, sc, bi, act);
    const float ds = r.d * sc;
    return (float)((double)ds * ka + kb * (double)(xv - m) + kc);
  };
  if (vec)
  {
    const int step = 4 * gridDim.y * blockDim.x;
    int i = 4 * (blockIdx.y * blockDim.x + threadIdx.x);
    for (; i + step < spatial; i += 2 * step)
    {
      const float4 xa = *(const float4*)(xp + i);
      const float4 da = *(const float4*)(dp + i);
      const float4 xb = *(const float4*)(xp + i + step);
      const float4 db = *(const float4*)(dp + i + step);
      float4 oa, ob;
      oa.x = one(xa.x, da.x); oa.y = one(xa.y, da.y); oa.z = one(xa.z, da.z); oa.w = one(xa.w, da.w);
      ob.x = one(xb.x, db.x); ob.y = one(xb.y, db.y); ob.z = one(xb.z, db.z); ob.w = one(xb.w, db.w);
      *(float4*)(dp + i) = oa;
      *(float4*)(dp + i + step) = ob;
    }
    for (; i < spatial; i += step)
    {
      const float4 xv = *(const float4*)(xp + i);
      const float4 dv = *(const float4*)(dp + i);
      float4 o;
      o.x = one(xv.x, dv.x); o.y = one(xv.y, dv.y); o.z = one(xv.z, dv.z); o.w = one(xv.w, dv.w);
      *(float4*)(dp + i) = o;
    }
    return;
  }
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < spatial; i += gridDim.y * blockDim.x)
    dp[i] = one(xp[i], dp[i]);
}

// ---- backward of the glue layers ----------------------------------------------------
__global__ void maxpool_bwd_kernel(const float* __restrict__ delta, const int* __restrict__ indexes,
    size_t n, float* __restrict__ prev_delta)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int idx = indexes[i];
    if (idx >= 0)
      atomicAdd(&prev_delta[idx], delta[i]);  // stride-1 SPP windows overlap
  }
}

// The same gradient as a gather (deterministic mode): every INPUT element visits the windows that contain it, in
// row-major window order, and adds the deltas of those whose argmax it is.  No atomics; an input element of a
// k x k / stride-1 pool checks k*k windows (the SPP block: 25 / 81 / 169 on 19x19 maps).
__global__ void maxpool_bwd_gather_kernel(const float* __restrict__ delta, const int* __restrict__ indexes,
    size_t total_in, int h, int w, int out_h, int out_w, int size, int stride_x, int stride_y, int pad,
    float* __restrict__ prev_delta)
{
  const int off = pad / 2;
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total_in; id += (size_t)gridDim.x * blockDim.x)
  {
    const int ix = (int)(id % w);
    const size_t t = id / w;
    const int iy = (int)(t % h);
    const size_t plane = t / h;
    // windows (oy, ox) with oy*stride - off <= iy < oy*stride - off + size
    int oy0 = (iy + off - size + stride_y) / stride_y, oy1 = (iy + off) / stride_y;
    int ox0 = (ix + off - size + stride_x) / stride_x, ox1 = (ix + off) / stride_x;
    if (iy + off - size + 1 <= 0) oy0 = 0;
    if (ix + off - size + 1 <= 0) ox0 = 0;
    if (oy1 > out_h - 1) oy1 = out_h - 1;
    if (ox1 > out_w - 1) ox1 = out_w - 1;
    const int* ip = indexes + plane * (size_t)out_h * out_w;
    const float* dp = delta + plane * (size_t)out_h * out_w;
    float s = 0.f;
    for (int oy = oy0; oy <= oy1; ++oy)
      for (int ox = ox0; ox <= ox1; ++ox)
        if (ip[oy * out_w + ox] == (int)id)
          s += dp[oy * out_w + ox];
    prev_delta[id] += s;
  }
}

// dst[j*dst_stride + i] += alpha * src[j*src_stride + i]
__global__ void axpy2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int part,
    int batch, size_t src_stride, size_t dst_stride, float alpha)
{
  const size_t total = (size_t)part * batch;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(i / part);
    const int e = (int)(i - (size_t)j * part);
    dst[j * dst_stride + e] += alpha * src[j * src_stride + e];
  }
}

__global__ void shortcut_bwd_kernel(const float* __restrict__ delta, size_t n,
    float* __restrict__ prev_delta, float* __restrict__ from_delta)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const float d = delta[i];
    prev_delta[i] += d;
    from_delta[i] += d;
  }
}

// prev[in] += scale*out over the stride x stride block, in the reference's (j, i) order
__global__ void upsample_bwd_kernel(const float* __restrict__ delta, float* __restrict__ prev, int w,
    int h, size_t planes, int stride, float scale)
{
  const size_t total = planes * (size_t)h * w;
  const int ow = w * stride;
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total;
       id += (size_t)gridDim.x * blockDim.x)
  {
    const int x = (int)(id % w);
    const size_t t = id / w;
    const int y = (int)(t % h);
    const size_t pl = t / h;
    float acc = prev[id];
    const float* o = delta + (pl * h * stride + (size_t)y * stride) * ow + (size_t)x * stride;
    for (int j = 0; j < stride; ++j)
      for (int i = 0; i < stride; ++i) acc += scale * o[(size_t)j * ow + i];
    prev[id] = acc;
  }
}

// ---- SGD with momentum and decay, one pass (UpdateConvolutionalLayer) -------------------
__global__ void sgd_kernel(float* __restrict__ w, float* __restrict__ wu, size_t n, float decay_b,
    float lr_b, float momentum, int use_decay)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
  {
    float u = wu[i];
    float x = w[i];
    if (use_decay)
      u += decay_b * x;   // axpy(-decay*batch, weights -> weight_updates)
    x += lr_b * u;        // axpy(lr/batch, weight_updates -> weights)
    u *= momentum;        // scal(momentum, weight_updates)
    w[i] = x;
    wu[i] = u;
  }
}

// The same update for MANY tensors in one launch (UpdateNetworkGpu: ~330 tensors per step, most
// of them a few hundred floats -- one launch each is pure launch overhead).  Work is cut into
// chunks of SGD_CHUNK elements; chunk c belongs to tensor ent[c].x starting at element ent[c].y.
constexpr int SGD_CHUNK = 4096;
struct DkSgdTensor
{
  float* w;
  float* wu;
  unsigned long long n;
  float lr_scale;     // the layer's learning_rate_scale
  int use_decay;
};

__global__ void sgd_multi_kernel(const DkSgdTensor* __restrict__ tensors, const int2* __restrict__ chunks,
    float decay_b, float lr_init, int batch, float momentum)
{
  const int2 ch = chunks[blockIdx.x];
  const DkSgdTensor t = tensors[ch.x];
  const size_t lo = (size_t)ch.y * SGD_CHUNK;
  size_t hi = lo + SGD_CHUNK;
  if (hi > t.n)
    hi = t.n;
  const float lr = (lr_init * t.lr_scale) / batch;   // the per-layer path's float operations, in its order
  for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
  {
    float u = t.wu[i];
    float x = t.w[i];
    if (t.use_decay)
      u += decay_b * x;
    x += lr * u;
    u *= momentum;
    t.w[i] = x;
    t.wu[i] = u;
  }
}

// Wt[c][(m, t)] = W[m][c][t]  (t = kh*size + kw), per group
__global__ void transpose_w_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C,
    int ss, size_t total, int flip)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int t = (int)(i % ss);
    const size_t r = i / ss;
    const int m = (int)(r % M);
    const int c = (int)(r / M);
    wt[i] = w[((size_t)m * C + c) * ss + (flip ? ss - 1 - t : t)];
  }
}
// wt[c][t * M + m] = w[m][c][t]
__global__ void transpose_w_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C,
    int ss, size_t total)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int m = (int)(i % M);
    const size_t r = i / M;
    const int t = (int)(r % ss);
    const int c = (int)(r / ss);
    wt[i] = w[((size_t)m * C + c) * ss + t];
  }
}
}  // namespace

// adam_update_gpu (src/blas_kernels.cu:99-134) in one pass, the reference's seven launches
// (scal m, scal v, axpy decay, axpy m, mul d, axpy v, adam_kernel, fill d) per element in the same order:
//   m *= B1; v *= B2; d += (-decay*batch)*w; m += (1-B1)*d; d *= d; v += (1-B2)*d;
//   w += rate * (m / (1 - B1^t)) / (sqrt(v / (1 - B2^t)) + eps); d = 0
// Like the reference's function it applies the decay term to whatever tensor it is given (biases and scales too).
namespace
{
__global__ void adam_update_kernel(float* __restrict__ w, float* __restrict__ d, float* __restrict__ m,
    float* __restrict__ v, float B1, float B2, float eps, float decay_batch, float rate, float one_b1, float one_b2,
    float c1, float c2, size_t n)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
  {
    float mi = m[i] * B1;
    float vi = v[i] * B2;
    const float wi = w[i];
    float di = d[i];
    di = di + decay_batch * wi;
    mi = mi + one_b1 * di;
    di = di * di;
    vi = vi + one_b2 * di;
    const float mhat = mi / c1;
    const float vhat = vi / c2;
    w[i] = wi + rate * mhat / (sqrtf(vhat) + eps);
    m[i] = mi;
    v[i] = vi;
    d[i] = 0.f;
  }
}
}  // namespace

extern "C" int dk_adam_update(float* w, float* d, float* m, float* v, float B1, float B2, float eps, float decay,
    float rate, size_t n, int batch, int t, void* stream)
{
  if (n == 0)
    return 0;
  if (!w || !d || !m || !v)
  {
    fprintf(stderr, "dk_adam_update: null pointer\n");
    return 1;
  }
  const float c1 = 1.f - powf(B1, (float)t), c2 = 1.f - powf(B2, (float)t);   // adam_kernel: 1.f - powf(B, t)
  hipLaunchKernelGGL(adam_update_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), w, d, m, v, B1, B2, eps,
      -decay * batch, rate, 1 - B1, 1 - B2, c1, c2, n);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_bn_forward_train(const float* raw, float* x_save, float* x_norm, float* act_in,
    float* out, float* mean, float* variance, float* rolling_mean, float* rolling_variance,
    const float* scales, const float* biases, int batch, int filters, int spatial, int activation,
    int train, void* stream)
{
  const size_t total = (size_t)batch * filters * spatial;
  if (total == 0)
    return 0;
  // the same mish as the inference epilogue's default (one hardware exp and reciprocal instead of the libm
  // chain, dk_device_math.h): with the libm form this pass was ALU-bound, not bandwidth-bound
  if (activation == DK_MISH && dk_fast_mish_enabled())
    activation |= DK_ACT_FAST;
  hipStream_t st = S(stream);
  double* sums = nullptr;
  int nwg = 0;
  if (train)
  {
    int cps, slice, ipw;
    plane_split2(batch, filters, spatial, &cps, &slice, &ipw);
    const int gx = cps * ((batch + ipw - 1) / ipw);
    nwg = dk_deterministic() ? gx : 0;   // deterministic mode: one slot per producer workgroup, summed in order
    sums = chan_ring_take((size_t)2 * filters * (nwg ? nwg : 1), st);
    if (!sums)
    {
      fprintf(stderr, "dk_bn_forward_train: too many channels\n");
      return 1;
    }
    hipLaunchKernelGGL(bn_partial_kernel, dim3(gx, filters), dim3(RP), 0, st, raw, batch,
        filters, spatial, cps, slice, ipw, sums, nwg ? 1 : 0);
    CHECK_HIP(hipPeekAtLastError());
  }
  {
    float* const xs = train ? x_save : nullptr;
    float* const xn = train ? x_norm : nullptr;
    const uintptr_t al = (uintptr_t)raw | (uintptr_t)out | (uintptr_t)xs | (uintptr_t)xn | (uintptr_t)act_in;
    const int vec = ((spatial & 3) == 0 && (al & 15) == 0) ? 1 : 0;
    int gx = (spatial + (vec ? 8191 : 1023)) / (vec ? 8192 : 1024);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(batch * filters, gx), dim3(256), 0, st, raw, xs, xn, act_in, out,
        train ? mean : rolling_mean, train ? variance : rolling_variance, scales, biases, filters, spatial,
        activation, vec, (const double*)sums, batch, rolling_mean, rolling_variance, nwg);
  }
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_gradient_array(const float* y, const float* activation_input, float* delta,
    size_t n, int activation, void* stream)
{
  if (n == 0 || activation == DK_LINEAR)
    return 0;
  if ((activation == DK_MISH || activation == 16) && !activation_input)
  {
    fprintf(stderr, "dk_gradient_array: mish / swish need the saved pre-activation\n");
    return 1;
  }
  hipLaunchKernelGGL(gradient_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), y,
      activation_input, delta, n, activation);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_backward_bias(float* bias_updates, const float* delta, int batch, int n, int size,
    void* stream)
{
  if ((size_t)batch * n * size == 0)
    return 0;
  hipStream_t st = S(stream);
  double* scratch = chan_scratch(n, st);
  int chunks;
  size_t slice;
  chan_split(batch, n, size, &chunks, &slice);
  if (dk_deterministic())
  {
    chunks = 1;   // one workgroup per channel: its block sum has a fixed order, the single atomic lands on zero
    slice = (size_t)batch * size;
  }
  hipLaunchKernelGGL(chan_partial_kernel, dim3(chunks, n), dim3(RT), 0, st, delta, nullptr, nullptr,
      nullptr, nullptr, batch, n, size, slice, scratch, 0);
  hipLaunchKernelGGL(chan_finalize_kernel, dim3((n + 255) / 256), dim3(256), 0, st, scratch, nullptr,
      n, bias_updates, nullptr, nullptr, nullptr, 0);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_bn_backward(float* delta, const float* x, const float* x_norm, const float* mean,
    const float* variance, const float* scales, float* mean_delta, float* variance_delta,
    float* scale_updates, float* bias_updates, int batch, int filters, int spatial, void* stream)
{
  const size_t total = (size_t)batch * filters * spatial;
  if (total == 0)
    return 0;
  hipStream_t st = S(stream);
  double* scratch = chan_scratch(filters, st);
  int chunks;
  size_t slice;
  chan_split(batch, filters, spatial, &chunks, &slice);
  if (dk_deterministic())
  {
    chunks = 1;
    slice = (size_t)batch * spatial;
  }
  hipLaunchKernelGGL(chan_partial_kernel, dim3(chunks, filters), dim3(RT), 0, st, delta, x, x_norm,
      mean, scales, batch, filters, spatial, slice, scratch, 1);
  hipLaunchKernelGGL(chan_finalize_kernel, dim3((filters + 255) / 256), dim3(256), 0, st, scratch,
      variance, filters, bias_updates, scale_updates, mean_delta, variance_delta, 1);
  CHECK_HIP(hipPeekAtLastError());
  hipLaunchKernelGGL(bn_delta_kernel, dim3(grid_for(total)), dim3(256), 0, st, delta, x,
      mean, variance, mean_delta, variance_delta, scales, batch, filters, spatial, total);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// gradient_array + backward_batchnorm in one (see bn_recompute): `delta` holds the
// gradient w.r.t. the layer's OUTPUT on entry and w.r.t. the raw conv output x on exit.
extern "C" int dk_bn_act_backward(float* delta, const float* x, const float* mean,
    const float* variance, const float* scales, const float* biases, float* mean_delta,
    float* variance_delta, float* scale_updates, float* bias_updates, int batch, int filters,
    int spatial, int activation, void* stream)
{
  const size_t total = (size_t)batch * filters * spatial;
  if (total == 0)
    return 0;
  hipStream_t st = S(stream);
  if (activation == DK_MISH && dk_fast_mish_enabled())
    activation |= DK_ACT_FAST;
  int cps, slice, ipw;
  plane_split2(batch, filters, spatial, &cps, &slice, &ipw);
  const int gpx = cps * ((batch + ipw - 1) / ipw);
  const int nwg = dk_deterministic() ? gpx : 0;
  double* sums = chan_ring_take((size_t)4 * filters * (nwg ? nwg : 1), st);
  if (!sums)
  {
    fprintf(stderr, "dk_bn_act_backward: too many channels\n");
    return 1;
  }
  hipLaunchKernelGGL(bn_act_partial_kernel, dim3(gpx, filters), dim3(RP), 0, st, delta, x, mean,
      variance, scales, biases, batch, filters, spatial, cps, slice, ipw, sums, activation, nwg ? 1 : 0);
  const int vec = ((spatial & 3) == 0 && ((((uintptr_t)delta) | ((uintptr_t)x)) & 15) == 0) ? 1 : 0;
  // one 256-thread workgroup streams up to 8 K elements of its plane (two 16-byte positions in flight per thread)
  int gx = vec ? (spatial + 8191) / 8192 : (spatial + 1023) / 1024;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(bn_act_delta_kernel, dim3(batch * filters, gx), dim3(256), 0, st, delta, x, mean,
      variance, mean_delta, variance_delta, scales, biases, batch, filters, spatial, activation, vec,
      (const double*)sums, bias_updates, scale_updates, nwg);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_maxpool_backward(const float* delta, const int* indexes, size_t n,
    float* prev_delta, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), delta, indexes,
      n, prev_delta);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// deterministic form (needs the layer's geometry): see maxpool_bwd_gather_kernel
extern "C" int dk_maxpool_backward_gather(const float* delta, const int* indexes, int batch, int c, int h, int w, int out_h,
    int out_w, int size, int stride_x, int stride_y, int pad, float* prev_delta, void* stream)
{
  const size_t total_in = (size_t)batch * c * h * w;
  if (total_in == 0)
    return 0;
  if (total_in >= ((size_t)1 << 31))
  {
    fprintf(stderr, "dk_maxpool_backward_gather: tensor too large for the int argmax indices\n");
    return 1;
  }
  hipLaunchKernelGGL(maxpool_bwd_gather_kernel, dim3(grid_for(total_in)), dim3(256), 0, S(stream), delta, indexes,
      total_in, h, w, out_h, out_w, size, stride_x, stride_y, pad, prev_delta);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_route_backward(const float* delta, int outputs, int offset, int input_size,
    int groups, int group_id, int batch, float* src_delta, void* stream)
{
  const int part = input_size / groups;
  if (part == 0 || batch == 0)
    return 0;
  hipLaunchKernelGGL(axpy2d_kernel, dim3(grid_for((size_t)part * batch)), dim3(256), 0, S(stream),
      delta + offset, src_delta + (size_t)part * group_id, part, batch, (size_t)outputs,
      (size_t)input_size, 1.0f);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_shortcut_backward(const float* delta, size_t n, float* prev_delta,
    float* from_delta, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(shortcut_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), delta, n,
      prev_delta, from_delta);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_upsample_backward(const float* delta, int w, int h, int c, int batch, int stride,
    float scale, float* prev_delta, void* stream)
{
  const size_t planes = (size_t)c * batch;
  if (planes * h * w == 0)
    return 0;
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for(planes * h * w)), dim3(256), 0, S(stream),
      delta, prev_delta, w, h, planes, stride, scale);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_sgd_update(float* weights, float* weight_updates, size_t n, int batch,
    float learning_rate, float momentum, float decay, int use_decay, void* stream)
{
  if (n == 0)
    return 0;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n)), dim3(256), 0, S(stream), weights,
      weight_updates, n, -decay * batch, learning_rate / batch, momentum, use_decay);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// Multi-tensor form: `plan` is an opaque device-side plan built by dk_sgd_plan_create from host
// arrays (weights[i], updates[i], counts[i], lr_scales[i], use_decay[i]); the float operations per
// element are exactly dk_sgd_update's with learning_rate * lr_scale[i] (computed in float: the
// reference multiplies learning_rate_init * l->learning_rate_scale in float before dividing by batch).
struct DkSgdPlan
{
  DkSgdTensor* tensors;
  int2* chunks;
  int nchunks;
};

void* dk_sgd_plan_create(int ntensors, float* const* weights, float* const* updates, const size_t* counts,
    const float* lr_scales, const int* use_decay)
{
  std::vector<DkSgdTensor> ht(ntensors);
  std::vector<int2> hc;
  for (int i = 0; i < ntensors; ++i)
  {
    ht[i].w = weights[i];
    ht[i].wu = updates[i];
    ht[i].n = counts[i];
    ht[i].lr_scale = lr_scales[i];
    ht[i].use_decay = use_decay[i];
    for (size_t c = 0; c * SGD_CHUNK < counts[i]; ++c) hc.push_back(make_int2(i, (int)c));
  }
  DkSgdPlan* p = new DkSgdPlan();
  p->nchunks = (int)hc.size();
  CHECK_HIP(hipMalloc((void**)&p->tensors, sizeof(DkSgdTensor) * (ntensors > 0 ? ntensors : 1)));
  CHECK_HIP(hipMalloc((void**)&p->chunks, sizeof(int2) * (hc.size() ? hc.size() : 1)));
  CHECK_HIP(hipMemcpy(p->tensors, ht.data(), sizeof(DkSgdTensor) * ntensors, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(p->chunks, hc.data(), sizeof(int2) * hc.size(), hipMemcpyHostToDevice));
  return p;
}

void dk_sgd_plan_destroy(void* plan)
{
  DkSgdPlan* p = (DkSgdPlan*)plan;
  if (!p)
    return;
  (void)hipFree(p->tensors);
  (void)hipFree(p->chunks);
  delete p;
}

int dk_sgd_update_multi(void* plan, int batch, float learning_rate, float momentum, float decay, void* stream)
{
  DkSgdPlan* p = (DkSgdPlan*)plan;
  if (!p || !p->nchunks)
    return 0;
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(p->nchunks), dim3(256), 0, S(stream), p->tensors, p->chunks,
      -decay * batch, learning_rate, batch, momentum);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_transpose_weights(const float* w, float* wt, int M, int C, int size, void* stream)
{
  const size_t total = (size_t)M * C * size * size;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(transpose_w_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), w, wt, M,
      C, size * size, total, 0);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// wt[c][(t, m)] = w[m][c][t]: the contraction index of the data gradient ordered tap-major, for the parity-class
// form of stride-2 layers (dk_conv_backward_data_tapmajor)
extern "C" int dk_transpose_weights_tapmajor(const float* w, float* wt, int M, int C, int size, void* stream)
{
  const size_t total = (size_t)M * C * size * size;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(transpose_w_tapmajor_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), w, wt, M, C,
      size * size, total);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// wt[c][(m, t)] = w[m][c][ss-1-t]: the weights of the convolution that computes the data
// gradient of a stride-1 "same" convolution (taps rotated by 180 degrees)
extern "C" int dk_transpose_weights_flip(const float* w, float* wt, int M, int C, int size, void* stream)
{
  const size_t total = (size_t)M * C * size * size;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(transpose_w_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), w, wt, M,
      C, size * size, total, 1);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
