// nms.hip -- NmsSort on the device: the candidate records of dk_yolo_compact
// (predictors whose objectness passed the threshold) get their boxes and class probabilities
// computed and the per-class greedy / DIoU suppression applied in HBM, so only surviving boxes
// need to cross PCIe.
//
// Reference twins (Ravicmoon/darknet src/): GetYoloBox / GetYoloDetections yolo_layer.cpp:139-148,
// :794-834 (box = ((col + x) / lw, (row + y) / lh, exp(w) * anchor_w / net_w, exp(h) * anchor_h / net_h),
// prob = objectness * class score if > thresh else 0); NmsSort box.cpp:393-419 (per class: sort by
// that class' probability, descending; every later box whose IoU -- or DIoU, Box::Diou :98-113 --
// with an earlier box of non-zero probability exceeds the threshold loses its probability);
// Box::Iou :36-63.  The IoU arithmetic is the host's float operation sequence; box w/h go through
// the device's expf (vs glibc's on the host path: last-bit differences), so a suppression decision
// can differ from the host's only for an IoU within rounding of the threshold.
//
// One workgroup per (class, image): its candidates with a non-zero probability for the class are
// gathered into LDS, sorted (bitonic; descending probability, ties in scan order), and swept
// greedily -- the sweep is sequential in the kept boxes, parallel over the boxes they suppress.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <stdio.h>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_internal.h"

namespace
{
inline hipStream_t S(void* s) { return s ? (hipStream_t)s : get_cuda_stream(); }
constexpr int NMS_MAXC = 4096;   // candidates of one (image, class) the sweep holds in LDS

__device__ __forceinline__ float overlap(float x1, float w1, float x2, float w2)
{
  const float l1 = x1 - w1 / 2, l2 = x2 - w2 / 2;
  const float left = l1 > l2 ? l1 : l2;
  const float r1 = x1 + w1 / 2, r2 = x2 + w2 / 2;
  const float right = r1 < r2 ? r1 : r2;
  return right - left;
}

__device__ __forceinline__ float box_iou(const float4 a, const float4 b)
{
  const float w = overlap(a.x, a.z, b.x, b.z);
  const float h = overlap(a.y, a.w, b.y, b.w);
  const float I = (w < 0 || h < 0) ? 0.f : w * h;
  const float U = a.z * a.w + b.z * b.w - I;
  if (fabsf(I) < FLT_EPSILON || fabsf(U) < FLT_EPSILON)
    return 0;
  return I / U;
}

__device__ __forceinline__ float box_diou(const float4 a, const float4 b, float beta)
{
  const float left = fminf(a.x - a.z / 2.0f, b.x - b.z / 2.0f);
  const float right = fmaxf(a.x + a.z / 2.0f, b.x + b.z / 2.0f);
  const float top = fminf(a.y - a.w / 2.0f, b.y - b.w / 2.0f);
  const float bottom = fmaxf(a.y + a.w / 2.0f, b.y + b.w / 2.0f);
  const float w = right - left, h = bottom - top;
  const float c = w * w + h * h;
  const float iou = box_iou(a, b);
  if (fabsf(c) < FLT_EPSILON)
    return iou;
  const float d = (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y);
  return iou - powf(d / c, beta);
}

// records: [tag, image, loc, x, y, w, h, obj, cls...]; on exit x..h hold the BOX and cls[j] the
// thresholded probability obj * cls[j] (GetYoloDetections); heads[tag] describes the yolo layer
__global__ void nms_prepare_kernel(float* __restrict__ records, int count, int rec, int classes,
    const DkYoloHead* __restrict__ heads, int net_w, int net_h, float thresh)
{
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x)
  {
    float* r = records + (size_t)k * rec;
    const DkYoloHead hd = heads[__float_as_int(r[0])];
    const int loc = __float_as_int(r[2]);
    const int wh = hd.lw * hd.lh;
    const int n = loc / wh, i = loc - n * wh;
    const int col = i % hd.lw, row = i / hd.lw;
    const float bx = (col + r[3]) / hd.lw;
    const float by = (row + r[4]) / hd.lh;
    const float bw = expf(r[5]) * hd.anchor_w[n] / net_w;
    const float bh = expf(r[6]) * hd.anchor_h[n] / net_h;
    r[3] = bx; r[4] = by; r[5] = bw; r[6] = bh;
    const float obj = r[7];
    for (int j = 0; j < classes; ++j)
    {
      const float prob = obj * r[8 + j];
      r[8 + j] = (prob > thresh) ? prob : 0.f;
    }
  }
}

__global__ void __launch_bounds__(256) nms_class_kernel(float* __restrict__ records, int count, int rec,
    float nms_thresh, int nms_kind, float beta, int* __restrict__ overflow)
{
  __shared__ float s_prob[NMS_MAXC];
  __shared__ unsigned long long s_key[NMS_MAXC];   // (tag << 40 | loc << 16 ...) scan order for ties, low bits: slot
  __shared__ int s_slot[NMS_MAXC];
  __shared__ int s_n;
  const int cls = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  if (tid == 0)
    s_n = 0;
  __syncthreads();
  for (int k = tid; k < count; k += blockDim.x)
  {
    const float* r = records + (size_t)k * rec;
    if (__float_as_int(r[1]) != img)
      continue;
    const float p = r[8 + cls];
    if (fabsf(p) < FLT_EPSILON)
      continue;
    const int at = atomicAdd(&s_n, 1);
    if (at < NMS_MAXC)
    {
      s_prob[at] = p;
      s_slot[at] = k;
      s_key[at] = ((unsigned long long)(unsigned)__float_as_int(r[0]) << 32) | (unsigned)__float_as_int(r[2]);
    }
  }
  __syncthreads();
  int n = s_n;
  if (n > NMS_MAXC)
  {
    if (tid == 0)
      atomicAdd(overflow, 1);
    return;   // the host falls back to NmsSort on the pulled records
  }
  if (n < 2)
    return;
  // bitonic sort of (prob desc, key asc) over the next power of two; padding sorts last
  int np = 1;
  while (np < n) np <<= 1;
  for (int k = n + tid; k < np; k += blockDim.x)
  {
    s_prob[k] = -1.f;
    s_key[k] = ~0ull;
    s_slot[k] = -1;
  }
  __syncthreads();
  auto before = [&](int a, int b) {   // element a sorts before element b
    return s_prob[a] > s_prob[b] || (s_prob[a] == s_prob[b] && s_key[a] < s_key[b]);
  };
  for (int size = 2; size <= np; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1)
    {
      for (int t = tid; t < np / 2; t += blockDim.x)
      {
        const int lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
        const bool up = ((lo & size) == 0);
        if (before(hi, lo) == up)
        {
          const float p = s_prob[lo]; s_prob[lo] = s_prob[hi]; s_prob[hi] = p;
          const unsigned long long q = s_key[lo]; s_key[lo] = s_key[hi]; s_key[hi] = q;
          const int z = s_slot[lo]; s_slot[lo] = s_slot[hi]; s_slot[hi] = z;
        }
      }
      __syncthreads();
    }
  // greedy sweep: s_prob[j] = 0 marks a suppressed box
  for (int i = 0; i < n - 1; ++i)
  {
    if (s_prob[i] != 0.f)   // block-uniform: every thread reads the same LDS word after the barrier
    {
      const float* ri = records + (size_t)s_slot[i] * rec;
      const float4 a = make_float4(ri[3], ri[4], ri[5], ri[6]);
      for (int j = i + 1 + tid; j < n; j += blockDim.x)
      {
        if (s_prob[j] == 0.f)
          continue;
        const float* rj = records + (size_t)s_slot[j] * rec;
        const float4 b = make_float4(rj[3], rj[4], rj[5], rj[6]);
        const float v = nms_kind == 0 ? box_iou(a, b) : box_diou(a, b, beta);
        if (v > nms_thresh)
          s_prob[j] = 0.f;
      }
    }
    __syncthreads();
  }
  for (int j = tid; j < n; j += blockDim.x)
    if (s_prob[j] == 0.f)
      records[(size_t)s_slot[j] * rec + 8 + cls] = 0.f;
}
}  // namespace

extern "C" int dk_nms_records(float* records, int count, int classes, const DkYoloHead* heads_dev, int net_w,
    int net_h, int batch, float thresh, float nms_thresh, int nms_kind, float beta, int* overflow_dev, void* stream)
{
  if (!records || !heads_dev || !overflow_dev || classes < 1 || batch < 1)
  {
    fprintf(stderr, "dk_nms_records: invalid arguments\n");
    return 1;
  }
  if (count <= 0)
    return 0;
  const int rec = 3 + 5 + classes;
  hipStream_t st = S(stream);
  int blocks = (count + 255) / 256;
  if (blocks > 1024)
    blocks = 1024;
  hipLaunchKernelGGL(nms_prepare_kernel, dim3(blocks), dim3(256), 0, st, records, count, rec, classes, heads_dev,
      net_w, net_h, thresh);
  hipLaunchKernelGGL(nms_class_kernel, dim3(classes, batch), dim3(256), 0, st, records, count, rec, nms_thresh,
      nms_kind, beta, overflow_dev);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
