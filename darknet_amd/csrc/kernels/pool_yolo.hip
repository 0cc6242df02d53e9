// pool_yolo.hip -- maxpool forward and the fused YOLO decode.
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_device_math.h"
#include "dk_internal.h"

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

// Definition of ForwardMaxpoolLayer's generic loop (src/maxpool_layer.cpp:255-297)
// == forward_maxpool_layer_kernel (src/maxpool_layer_kernels.cu:58-101) with the
// CPU's -FLT_MAX padding value: one thread per output, window scanned row-major,
// strict '>' (first maximum wins), index = flat input index.
__global__ void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y,
    int* __restrict__ indexes, size_t total, int c, int h, int w, int out_h, int out_w, int size,
    int stride_x, int stride_y, int pad, size_t out_bstride)
{
  const int w_off = -pad / 2, h_off = -pad / 2;
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total;
       id += (size_t)gridDim.x * blockDim.x)
  {
    const int j = (int)(id % out_w);
    size_t t = id / out_w;
    const int i = (int)(t % out_h);
    const size_t plane = t / out_h;  // b*c + k
    const float* src = x + plane * h * w;
    float max = -FLT_MAX;
    int max_i = -1;
    const int h0 = h_off + i * stride_y, w0 = w_off + j * stride_x;
    for (int n = 0; n < size; ++n)
    {
      const int cur_h = h0 + n;
      if ((unsigned)cur_h >= (unsigned)h)
        continue;
      for (int m = 0; m < size; ++m)
      {
        const int cur_w = w0 + m;
        if ((unsigned)cur_w >= (unsigned)w)
          continue;
        const float val = src[cur_h * w + cur_w];
        if (val > max)
        {
          max = val;
          max_i = (int)(plane * h * w) + cur_h * w + cur_w;
        }
      }
    }
    const size_t bi = plane / c;
    y[bi * out_bstride + (plane - bi * c) * (size_t)out_h * out_w + (size_t)i * out_w + j] = max;
    if (indexes)
      indexes[id] = max_i;
  }
}

// Stride-1 pools on small planes (the SPP block: 5/9/13 windows on 19x19): one block
// per (b, c) plane held in LDS, separable scan.  Row pass: per (row, ox) the window
// maximum and its FIRST column (strict '>'); column pass: strict '>' over the rows ->
// the first row holding the maximum and that row's first column, i.e. exactly the
// element the reference's row-major window scan keeps (value and index identical).
__global__ void __launch_bounds__(256) maxpool_plane_kernel(const float* __restrict__ x,
    float* __restrict__ y, int* __restrict__ indexes, int c, int h, int w, int out_h, int out_w,
    int size, int pad, size_t out_bstride)
{
  extern __shared__ float sm[];
  float* plane_s = sm;                     // [h][w]
  float* rmax = sm + h * w;                // [h][out_w]
  int* ridx = (int*)(rmax + h * out_w);    // [h][out_w] column of the row maximum (-1: empty window)
  const size_t plane = blockIdx.x;
  const float* src = x + plane * (size_t)h * w;
  for (int i = threadIdx.x; i < h * w; i += blockDim.x) plane_s[i] = src[i];
  __syncthreads();
  const int off = -pad / 2;
  for (int i = threadIdx.x; i < h * out_w; i += blockDim.x)
  {
    const int r = i / out_w, j = i - r * out_w;
    float max = -FLT_MAX;
    int mi = -1;
    for (int m = 0; m < size; ++m)
    {
      const int cw = off + j + m;
      if ((unsigned)cw >= (unsigned)w)
        continue;
      const float v = plane_s[r * w + cw];
      if (v > max)
      {
        max = v;
        mi = cw;
      }
    }
    rmax[i] = max;
    ridx[i] = mi;
  }
  __syncthreads();
  const size_t bi = plane / c;
  float* dst = y + bi * out_bstride + (plane - bi * c) * (size_t)out_h * out_w;
  for (int i = threadIdx.x; i < out_h * out_w; i += blockDim.x)
  {
    const int oy = i / out_w, j = i - oy * out_w;
    float max = -FLT_MAX;
    int max_i = -1;
    for (int n = 0; n < size; ++n)
    {
      const int ch = off + oy + n;
      if ((unsigned)ch >= (unsigned)h)
        continue;
      const float v = rmax[ch * out_w + j];
      if (v > max)
      {
        max = v;
        max_i = (int)(plane * h * w) + ch * w + ridx[ch * out_w + j];
      }
    }
    dst[i] = max;
    if (indexes)
      indexes[plane * (size_t)out_h * out_w + i] = max_i;
  }
}

// ForwardYoloLayerGpu decode (src/yolo_layer.cpp:836-853) in one launch.
// entry e of anchor a at location loc lives at b*outputs + a*(5+cls)*wh + e*wh + loc.
__global__ void yolo_decode_kernel(const float* __restrict__ in, float* __restrict__ out,
    size_t total, int wh, int entries, float scale_x_y, float beta)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int e = (int)((i / wh) % entries);
    float v = in[i];
    if (e < 2)
    {
      v = dk_logistic(v);
      v = v * scale_x_y + beta;  // scal_add_cpu, src/blas.c:252-256
    }
    else if (e >= 4)
      v = dk_logistic(v);
    out[i] = v;
  }
}
// ForwardGaussianYoloLayerGpu decode (src/gaussian_yolo_layer.cpp:934-966) in one launch: entry e of anchor
// a lives at b*outputs + a*(9+cls)*wh + e*wh + loc; logistic on entries 0-3, 5, 7 and 8.., scale_x_y on the two
// mu planes (entries 0 and 2), entries 4 and 6 (mu of w, h) pass through.
__global__ void gaussian_yolo_decode_kernel(const float* __restrict__ in, float* __restrict__ out, size_t total, int wh,
    int entries, float scale_x_y, float beta)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
  {
    const int e = (int)((i / wh) % entries);
    float v = in[i];
    if (e != 4 && e != 6)
      v = dk_logistic(v);
    if (e == 0 || e == 2)
      v = v * scale_x_y + beta;  // scal_add_cpu, src/blas.c:252-256
    out[i] = v;
  }
}
// Candidate compaction of GetYoloDetections (src/yolo_layer.cpp:794-834): every predictor
// whose objectness exceeds thresh appends one record {layer tag, image, loc = n*wh + i,
// x, y, w, h, objectness, class scores (raw decoded values)} to a shared list.  The host
// sorts the few records into the reference's scan order and applies the reference's box
// arithmetic, so only candidates cross PCIe instead of the whole head.
__global__ void yolo_compact_kernel(const float* __restrict__ out, size_t total, int wh, int n_anchors,
    int entries, float thresh, int tag, float* __restrict__ records, int* __restrict__ counter, int cap)
{
  const int rec = 3 + entries;
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total;
       id += (size_t)gridDim.x * blockDim.x)
  {
    const int loc = (int)(id % ((size_t)n_anchors * wh));
    const int b = (int)(id / ((size_t)n_anchors * wh));
    const int n = loc / wh, i = loc - n * wh;
    const float* base = out + ((size_t)b * n_anchors + n) * entries * wh + i;
    const float obj = base[4 * (size_t)wh];
    if (!(obj > thresh))
      continue;
    const int slot = atomicAdd(counter, 1);
    if (slot >= cap)
      continue;  // the host sees counter > cap and falls back to the full head
    float* r = records + (size_t)slot * rec;
    r[0] = __int_as_float(tag);
    r[1] = __int_as_float(b);
    r[2] = __int_as_float(loc);
    for (int e = 0; e < entries; ++e) r[3 + e] = base[(size_t)e * wh];
  }
}

// Mat2Image (src/visualize.cpp:26-55): interleaved u8 rows -> planar float / 255
__global__ void u8_hwc_to_chw_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst,
    int w, int h, int c, size_t step, size_t image_bytes, size_t total)
{
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total;
       id += (size_t)gridDim.x * blockDim.x)
  {
    const int x = (int)(id % w);
    size_t t = id / w;
    const int y = (int)(t % h);
    t /= h;
    const int k = (int)(t % c);
    const size_t b = t / c;
    dst[id] = src[b * image_bytes + (size_t)y * step + (size_t)x * c + k] / 255.0f;
  }
}
// cv::resize(src, dst, Size(w, h)) with the default INTER_LINEAR on 8-bit interleaved frames, then (optionally)
// cv::cvtColor(RGB2BGR), then Mat2Image -- the input step of the reference's ProcImage (src/yolo_core.cpp:104-112)
// in ONE pass.  The resize is OpenCV's generic fixed-point path (modules/imgproc/src/resize.cpp, 4.x:
// resizeGeneric_ with HResizeLinear / VResizeLinear for uchar): 11-bit coefficients
// ialpha = cvRound((1-fx)*2048), cvRound(fx*2048) with fx = (float)((dx+0.5)*scale-0.5) - floor, left/right columns
// clamped with fx = 0, rows clamped; dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2 with
// S = s[sx]*a0 + s[sx+1]*a1; an exact 2x shrink takes OpenCV's area shortcut (sum of the 2x2 block + 2) >> 2.
// OpenCV is not in this image and the reference ships no frames, so this restatement is checked against
// oracle/orc_resize.py only ("parity unpinned" against an OpenCV build; IPP builds differ anyway).
__global__ void resize_u8_to_chw_kernel(const unsigned char* __restrict__ src, int sw, int sh, size_t sstep,
    size_t simage, float* __restrict__ dst, int w, int h, int c, int swap_rb, double scale_x, double scale_y,
    int area2, size_t total)
{
  for (size_t id = blockIdx.x * (size_t)blockDim.x + threadIdx.x; id < total; id += (size_t)gridDim.x * blockDim.x)
  {
    const int dx = (int)(id % w);
    size_t t = id / w;
    const int dy = (int)(t % h);
    const size_t b = t / h;
    const unsigned char* const img = src + b * simage;
    int out[4];
    if (area2)
    {
      const unsigned char* r0 = img + (size_t)(2 * dy) * sstep + (size_t)(2 * dx) * c;
      const unsigned char* r1 = r0 + sstep;
      for (int k = 0; k < c; ++k) out[k] = (r0[k] + r0[c + k] + r1[k] + r1[c + k] + 2) >> 2;
    }
    else
    {
      float fx = (float)(((double)dx + 0.5) * scale_x - 0.5);
      int sx = (int)floorf(fx);
      fx -= (float)sx;
      if (sx < 0) { fx = 0.f; sx = 0; }
      if (sx >= sw - 1) { fx = 0.f; sx = sw - 1; }
      const int sx1 = (sx + 1 < sw) ? sx + 1 : sw - 1;
      const int a0 = __float2int_rn((1.f - fx) * 2048.f), a1 = __float2int_rn(fx * 2048.f);
      float fy = (float)(((double)dy + 0.5) * scale_y - 0.5);
      const int sy = (int)floorf(fy);
      fy -= (float)sy;
      const int b0 = __float2int_rn((1.f - fy) * 2048.f), b1 = __float2int_rn(fy * 2048.f);
      const int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
      const int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
      const unsigned char* r0 = img + (size_t)y0 * sstep;
      const unsigned char* r1 = img + (size_t)y1 * sstep;
      for (int k = 0; k < c; ++k)
      {
        const int S0 = r0[sx * c + k] * a0 + r0[sx1 * c + k] * a1;
        const int S1 = r1[sx * c + k] * a0 + r1[sx1 * c + k] * a1;
        int v = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
        out[k] = v < 0 ? 0 : (v > 255 ? 255 : v);
      }
    }
    const size_t plane = (size_t)w * h;
    float* const o = dst + b * plane * c + (size_t)dy * w + dx;
    for (int k = 0; k < c; ++k)
    {
      const int kk = (swap_rb && c >= 3 && k < 3) ? 2 - k : k;   // cvtColor(RGB2BGR) swaps channels 0 and 2
      o[(size_t)kk * plane] = (float)out[k] / 255.0f;
    }
  }
}
}  // namespace

extern "C" int dk_image_resize_u8_to_chw(const unsigned char* src_hwc, int src_w, int src_h, size_t src_row_step,
    float* chw, int batch, int w, int h, int c, int swap_rb, void* stream)
{
  if (!src_hwc || !chw || c < 1 || c > 4 || src_w < 1 || src_h < 1 || w < 1 || h < 1 || src_row_step < (size_t)src_w * c)
  {
    fprintf(stderr, "dk_image_resize_u8_to_chw: invalid arguments\n");
    return 1;
  }
  const size_t total = (size_t)batch * h * w;
  if (total == 0)
    return 0;
  // cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale (resize.cpp)
  const double scale_x = 1.0 / ((double)w / (double)src_w), scale_y = 1.0 / ((double)h / (double)src_h);
  const int area2 = (src_w == 2 * w && src_h == 2 * h) ? 1 : 0;
  hipLaunchKernelGGL(resize_u8_to_chw_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), src_hwc, src_w, src_h,
      src_row_step, src_row_step * src_h, chw, w, h, c, swap_rb, scale_x, scale_y, area2, total);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_yolo_compact(const float* decoded, int batch, int lw, int lh, int n_anchors,
    int classes, float thresh, int tag, float* records, int* counter, int cap, void* stream)
{
  if (!decoded || !records || !counter || cap < 1)
  {
    fprintf(stderr, "dk_yolo_compact: invalid arguments\n");
    return 1;
  }
  const size_t total = (size_t)batch * n_anchors * lw * lh;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(yolo_compact_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), decoded, total,
      lw * lh, n_anchors, classes + 5, thresh, tag, records, counter, cap);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_image_u8_to_chw(const unsigned char* hwc, float* chw, int batch, int w, int h, int c,
    size_t row_step, void* stream)
{
  if (!hwc || !chw || row_step < (size_t)w * c)
  {
    fprintf(stderr, "dk_image_u8_to_chw: invalid arguments\n");
    return 1;
  }
  const size_t total = (size_t)batch * c * h * w;
  if (total == 0)
    return 0;
  hipLaunchKernelGGL(u8_hwc_to_chw_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), hwc, chw, w, h,
      c, row_step, row_step * h, total);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_maxpool_forward(const float* x, float* y, int* indexes, int batch, int c, int h,
    int w, int size, int stride_x, int stride_y, int pad, void* stream)
{
  return dk_maxpool_forward_strided(x, y, indexes, batch, c, h, w, size, stride_x, stride_y, pad, 0, stream);
}

int dk_maxpool_forward_strided(const float* x, float* y, int* indexes, int batch, int c, int h,
    int w, int size, int stride_x, int stride_y, int pad, size_t out_batch_stride, void* stream)
{
  if (!x || !y || size < 1 || stride_x < 1 || stride_y < 1)
  {
    fprintf(stderr, "dk_maxpool_forward: invalid arguments\n");
    return 1;
  }
  const int out_w = (w + pad - size) / stride_x + 1;
  const int out_h = (h + pad - size) / stride_y + 1;
  const size_t total = (size_t)batch * c * out_h * out_w;
  if (total == 0)
    return 0;
  if ((size_t)batch * c * h * w >= ((size_t)1 << 31))
  {
    fprintf(stderr, "dk_maxpool_forward: input too large for int indexes\n");
    return 1;
  }
  const size_t obs = out_batch_stride ? out_batch_stride : (size_t)c * out_h * out_w;
  const size_t lds = ((size_t)h * w + 2 * (size_t)h * out_w) * sizeof(float);
  if (stride_x == 1 && stride_y == 1 && size >= 3 && lds <= 48 * 1024)
  {
    hipLaunchKernelGGL(maxpool_plane_kernel, dim3((unsigned)(batch * c)), dim3(256), lds, S(stream), x, y,
        indexes, c, h, w, out_h, out_w, size, pad, obs);
    CHECK_HIP(hipPeekAtLastError());
    return 0;
  }
  hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), x, y, indexes,
      total, c, h, w, out_h, out_w, size, stride_x, stride_y, pad, obs);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_yolo_forward(const float* in, float* out, int batch, int lw, int lh,
    int n_anchors, int classes, float scale_x_y, void* stream)
{
  if (!in || !out)
  {
    fprintf(stderr, "dk_yolo_forward: null pointer\n");
    return 1;
  }
  const int entries = classes + 4 + 1;
  const size_t total = (size_t)batch * n_anchors * entries * lw * lh;
  if (total == 0)
    return 0;
  const float beta = (float)(-0.5 * (scale_x_y - 1));  // yolo_layer.cpp:400
  hipLaunchKernelGGL(yolo_decode_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), in, out,
      total, lw * lh, entries, scale_x_y, beta);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

extern "C" int dk_gaussian_yolo_forward(const float* in, float* out, int batch, int lw, int lh, int n_anchors,
    int classes, float scale_x_y, void* stream)
{
  if (!in || !out)
  {
    fprintf(stderr, "dk_gaussian_yolo_forward: invalid arguments\n");
    return 1;
  }
  const int entries = classes + 8 + 1;
  const size_t total = (size_t)batch * n_anchors * entries * lw * lh;
  if (total == 0)
    return 0;
  const float beta = (float)(-0.5 * (scale_x_y - 1));
  hipLaunchKernelGGL(gaussian_yolo_decode_kernel, dim3(grid_for(total)), dim3(256), 0, S(stream), in, out, total,
      lw * lh, entries, scale_x_y, beta);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}
