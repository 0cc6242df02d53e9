// conv_wgrad.hip -- weight gradient of the convolution on fp32 MFMA.
//
// Replaces the reference's per-image im2col_gpu_ext + cublasSgemm(NT, beta = 1)
// (src/convolutional_kernels.cu:757-781; CPU: im2col_cpu_ext + gemm(0,1,...),
// src/convolutional_layer.cpp:1345-1356):
//     dW[m][k] += sum_n delta[m][n] * col[k][n]      n = (image, oy, ox), k = (c,kh,kw)
// The contraction runs over n, which is huge (batch*oh*ow) while the output is
// small (n x k*k*c), so the n range is split over many workgroups, each
// accumulates a 64x64 tile of dW in MFMA accumulators over its slice and adds it
// to dW with float atomics (dW accumulates across images and subdivisions in
// the reference anyway: beta = 1).  col is gathered on the fly (no im2col
// buffer): a thread owns 4 fixed taps k and walks the pixels.
// Round-1 kernel: correctness first (64x64 tile, 16 pixels per step).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "dark_hip.h"
#include "dk_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace
{
struct WgradArgs
{
  const float* x;
  const float* delta;
  float* dw;
  unsigned x_bytes, d_bytes;
  int C, H, W, Ctot;   // channels per group, input dims, total channels
  int M, Mtot;         // filters per group / total
  int K;               // C*size*size
  int OH, OW, OHW;
  int N;               // batch*OHW
  int size, stride_x, stride_y, pad, dil;
  int tiles_m, tiles_k, nsplit, chunks_per_split, groups;
};

constexpr unsigned OOB = 0x80000000u;
constexpr int BM = 64, BKO = 64, NC = 16, LS = NC + 1, T = 256;

__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

__global__ void __launch_bounds__(T) conv_wgrad_f32(const WgradArgs p)
{
  __shared__ float As[BM * LS];   // delta tile  [m][n]
  __shared__ float Bs[BKO * LS];  // col tile    [k][n]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wk = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  int id = blockIdx.x;
  const int split = id % p.nsplit;
  id /= p.nsplit;
  const int tile_k = id % p.tiles_k;
  id /= p.tiles_k;
  const int tile_m = id % p.tiles_m;
  const int g = id / p.tiles_m;
  const int m0 = tile_m * BM, k0 = tile_k * BKO;

  const int HW = p.H * p.W;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)p.delta, 0, p.d_bytes, 0x00020000);

  const int nl = tid % NC;   // pixel within the chunk
  const int r0 = tid / NC;   // row 0..15 (+16j)
  // fixed taps of this thread
  int tap_c[4], tap_dy[4], tap_dx[4];
  bool tap_ok[4];
  const int ss = p.size * p.size;
#pragma unroll
  for (int j = 0; j < 4; ++j)
  {
    const int k = k0 + r0 + 16 * j;
    tap_ok[j] = k < p.K;
    const int kk = tap_ok[j] ? k : 0;
    const int c = kk / ss, t = kk - c * ss, kh = t / p.size, kw = t - kh * p.size;
    tap_c[j] = c;
    tap_dy[j] = kh * p.dil - p.pad;
    tap_dx[j] = kw * p.dil - p.pad;
  }
  bool row_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) row_ok[j] = (m0 + r0 + 16 * j) < p.M;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  const int nchunks = (p.N + NC - 1) / NC;
  int ch_begin = split * p.chunks_per_split;
  int ch_end = ch_begin + p.chunks_per_split;
  if (ch_end > nchunks)
    ch_end = nchunks;

  for (int ch = ch_begin; ch < ch_end; ++ch)
  {
    const int n = ch * NC + nl;
    const bool nv = n < p.N;
    const int nn = nv ? n : 0;
    const int b = nn / p.OHW;
    const int pix = nn - b * p.OHW;
    const int oy = pix / p.OW, ox = pix - oy * p.OW;
    float ra[4], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      const int m = m0 + r0 + 16 * j;
      const unsigned off = (unsigned)((b * p.Mtot + g * p.M + m) * p.OHW + pix) * 4u;
      ra[j] = ld_buf(dr, (nv && row_ok[j]) ? off : OOB);
      const int iy = oy * p.stride_y + tap_dy[j], ix = ox * p.stride_x + tap_dx[j];
      const bool ok = nv && tap_ok[j] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const unsigned xo = (unsigned)((b * p.Ctot + g * p.C + tap_c[j]) * HW + iy * p.W + ix) * 4u;
      rb[j] = ld_buf(xr, ok ? xo : OOB);
    }
    __syncthreads();  // previous chunk's MFMAs have read the tiles
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      As[(r0 + 16 * j) * LS + nl] = ra[j];
      Bs[(r0 + 16 * j) * LS + nl] = rb[j];
    }
    __syncthreads();
    // A operand: lane (i = m, kk = n) ; B operand: lane (kk = n, j = k)
    const float* ap = As + (wm * 32 + l31) * LS + lh;
    const float* bp = Bs + (wk * 32 + l31) * LS + lh;
#pragma unroll
    for (int s = 0; s < NC / 2; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
  }

  // C/D: col (= k) = lane&31, row (= m) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int k = k0 + wk * 32 + l31;
  if (k < p.K)
  {
#pragma unroll
    for (int r = 0; r < 16; ++r)
    {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < p.M)
        atomicAdd(&p.dw[((size_t)g * p.M + m) * p.K + k], acc[r]);
    }
  }
}
}  // namespace

extern "C" int dk_conv_backward_weights(const DkConvDesc* d, const float* x, const float* delta,
    float* weight_updates, void* stream)
{
  if (!d || !x || !delta || !weight_updates || d->groups < 1)
  {
    fprintf(stderr, "dk_conv_backward_weights: invalid arguments\n");
    return 1;
  }
  const int pad = d->pad * d->dilation;
  const int keff = d->dilation * (d->size - 1) + 1;
  const int OH = (d->h + 2 * pad - keff) / d->stride_y + 1;
  const int OW = (d->w + 2 * pad - keff) / d->stride_x + 1;
  const int C = d->c / d->groups, M = d->n / d->groups, K = C * d->size * d->size;
  const size_t in_img = (size_t)d->c * d->h * d->w, out_img = (size_t)d->n * OH * OW;
  int chunk = d->batch;
  const size_t lim = (size_t)1 << 29;
  if (in_img * chunk >= lim || out_img * chunk >= lim)
  {
    chunk = (int)((lim - 1) / (in_img > out_img ? in_img : out_img));
    if (chunk < 1)
    {
      fprintf(stderr, "dk_conv_backward_weights: one image exceeds the addressing window\n");
      return 1;
    }
  }
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  for (int b0 = 0; b0 < d->batch; b0 += chunk)
  {
    const int nb = (d->batch - b0 < chunk) ? d->batch - b0 : chunk;
    WgradArgs a;
    a.x = x + (size_t)b0 * in_img;
    a.delta = delta + (size_t)b0 * out_img;
    a.dw = weight_updates;
    a.x_bytes = (unsigned)(in_img * nb * 4);
    a.d_bytes = (unsigned)(out_img * nb * 4);
    a.C = C; a.H = d->h; a.W = d->w; a.Ctot = d->c;
    a.M = M; a.Mtot = d->n; a.K = K;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW;
    a.N = nb * OH * OW;
    a.size = d->size; a.stride_x = d->stride_x; a.stride_y = d->stride_y;
    a.pad = pad; a.dil = d->dilation;
    a.tiles_m = (M + BM - 1) / BM;
    a.tiles_k = (K + BKO - 1) / BKO;
    a.groups = d->groups;
    const int nchunks = (a.N + NC - 1) / NC;
    const long long tiles = (long long)a.tiles_m * a.tiles_k * d->groups;
    long long want = (2048 + tiles - 1) / tiles;  // ~8 workgroups per CU in total
    if (want < 1) want = 1;
    if (want > nchunks) want = nchunks;
    a.chunks_per_split = (int)((nchunks + want - 1) / want);
    if (a.chunks_per_split < 8 && nchunks >= 8)
      a.chunks_per_split = 8;
    a.nsplit = (nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    const long long nblk = tiles * a.nsplit;
    hipLaunchKernelGGL(conv_wgrad_f32, dim3((unsigned)nblk), dim3(T), 0, st, a);
    CHECK_HIP(hipPeekAtLastError());
  }
  return 0;
}
