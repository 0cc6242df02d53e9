// conv_igemm_f16.hip -- forward convolution with fp16 OPERANDS and fp32 accumulation
// (BASELINE config C5).  Behavioural spec = the reference's CUDNN_HALF branch
// (src/convolutional_kernels.cu:357-456 with cuda_f32_to_f16 / cuda_f16_to_f32
// :202-235): for eligible layers (size > 1, c % 8 == 0, n % 8 == 0, groups == 1, not
// layer 0) input activations and weights are rounded to fp16 (round-to-nearest-even),
// products are accumulated in fp32, the result stays fp32; bias / activation in fp32.
// Here nothing is converted in HBM: the gather loads fp32, converts in registers and
// stages fp16 tiles in LDS; the contraction is v_mfma_f32_32x32x16_f16 (16 x the
// fp32 MFMA rate), so this kernel is bound by the gather, not by the matrix pipe.
//
// LDS images (k contiguous, rows padded to 40 halves = 80 B so that the 16 lanes of a
// ds_read_b128 group hit 16 different 4-bank slots):
//   A: [BM][40] halves, B: [BN][40] halves; a lane (l31, lh) reads 8 halves at
//   [row l31][16*ks + 8*lh] = the MFMA's k = 8*lh + j operand slice of k-step ks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_device_math.h"
#include "dk_internal.h"
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
    conv_igemm_f16(const ConvArgs p)
{
  constexpr int BK = 32;
  constexpr int NWN = BN / WN;
  constexpr int NW = (BM / WM) * NWN;
  constexpr int T = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int LS = BK + 8;                  // row stride in halves
  constexpr int A_H = BM * LS, B_H = BN * LS, STAGE = A_H + B_H;
  constexpr int B_GROUPS = T / BN;
  constexpr int PB = BK / B_GROUPS;           // taps per thread (multiple of 8)
  constexpr int KQ = BK / 4;                  // float4 per A row
  constexpr int AV_ROWS = T / KQ;
  constexpr int PAV = (BM + AV_ROWS - 1) / AV_ROWS;
  static_assert(T % BN == 0 && BN % 64 == 0 && PB % 8 == 0, "B gather mapping");

  __shared__ __attribute__((aligned(16))) _Float16 lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l31 = lane & 31, lh = lane >> 5;

  int g, tile_m, tile_n;
  if (!conv_block_tile(p, g, tile_m, tile_n))
    return;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const int HW = p.H * p.W;
  const int K = p.K;
  const float* wg = p.w + (size_t)g * p.M * K;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, p.w_bytes, 0x00020000);

  // ---- B column of this thread
  const int bn_l = tid % BN;
  const int bk_g = __builtin_amdgcn_readfirstlane(tid / BN);
  unsigned xbase4, nmask = 0xFFFFFFFFu;
  {
    const int n = n0 + bn_l;
    const bool nv = n < p.N;
    const int nn = nv ? n : 0;
    const int b = fdiv(nn, p.OHW, p.inv_OHW);
    const int pix = nn - b * p.OHW;
    const int oy = fdiv(pix, p.OW, p.inv_OW), ox = pix - oy * p.OW;
    const int iy0 = oy * p.stride_y - p.pad, ix0 = ox * p.stride_x - p.pad;
    xbase4 = (unsigned)((b * p.Ctot + g * p.C) * HW + iy0 * p.W + ix0) * 4u;
    if (nv)
    {
      unsigned ok_bits = 0;
      for (int kh = 0; kh < p.size; ++kh)
        for (int kw = 0; kw < p.size; ++kw)
        {
          const bool ok = (unsigned)(iy0 + kh * p.dil) < (unsigned)p.H &&
                          (unsigned)(ix0 + kw * p.dil) < (unsigned)p.W;
          ok_bits |= (ok ? 1u : 0u) << (kh * p.size + kw);
        }
      nmask = ~ok_bits;
    }
  }
  // ---- A rows of this thread
  const int aq = tid % KQ, av_r = tid / KQ;
  unsigned aoff[PAV];
#pragma unroll
  for (int j = 0; j < PAV; ++j)
  {
    const int ml = av_r + j * AV_ROWS;
    const int m = m0 + ml;
    aoff[j] = (ml < BM && m < p.M) ? (unsigned)(m * K + aq * 4) * 4u : OOB;
  }

  float4 ra[PAV];
  float rb[PB];

  auto load_tile = [&](int k0) {
    const unsigned kinv = (k0 + aq * 4 < K) ? 0u : OOB;
#pragma unroll
    for (int j = 0; j < PAV; ++j) ra[j] = ld_buf4(wr, (aoff[j] + (unsigned)k0 * 4u) | kinv);
    const int2* kp = p.ktab + (k0 + bk_g * PB);
    int2 kt[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) kt[j] = kp[j];
#pragma unroll
    for (int j = 0; j < PB; ++j)
    {
      const unsigned inv = (nmask << kt[j].y) & OOB;
      rb[j] = ld_buf(xr, (xbase4 + (unsigned)kt[j].x) | inv);
    }
  };

  auto store_tile = [&](_Float16* st) {
    _Float16* As = st;
    _Float16* Bs = st + A_H;
#pragma unroll
    for (int j = 0; j < PAV; ++j)
    {
      const int ml = av_r + j * AV_ROWS;
      if (PAV * AV_ROWS == BM || ml < BM)
      {
        half4 h;
        h[0] = (_Float16)ra[j].x; h[1] = (_Float16)ra[j].y;   // v_cvt_f16_f32: round to nearest even
        h[2] = (_Float16)ra[j].z; h[3] = (_Float16)ra[j].w;
        *(half4*)&As[ml * LS + aq * 4] = h;
      }
    }
#pragma unroll
    for (int q = 0; q < PB / 8; ++q)
    {
      half8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (_Float16)rb[8 * q + e];
      *(half8*)&Bs[bn_l * LS + bk_g * PB + 8 * q] = h;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (K + BK - 1) / BK;
  load_tile(0);
  store_tile(lds);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt)
  {
    const _Float16* cur = lds + (kt & 1) * STAGE;
    const bool more = (kt + 1) < nkt;
    if (more)
      load_tile((kt + 1) * BK);
    const _Float16* As = cur + (wm * WM + l31) * LS + 8 * lh;
    const _Float16* Bs = cur + A_H + (wn * WN + l31) * LS + 8 * lh;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks)
    {
      half8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const half8*)&As[i * 32 * LS + 16 * ks];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const half8*)&Bs[j * 32 * LS + 16 * ks];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more)
      store_tile(lds + ((kt + 1) & 1) * STAGE);
    __syncthreads();
  }

  // ---- epilogue (fp32), shared with the fp32 kernel
  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, m0, n0, g, wm, wn, l31, lh);
}

// Eligibility rule of the reference's fp16 path (src/convolutional_kernels.cu:361-365);
// `layer_index` is the layer's position in the network (layer 0 stays fp32).
extern "C" int dk_conv_half_eligible(const DkConvDesc* d, int layer_index)
{
  return d->size > 1 && d->c % 8 == 0 && d->n % 8 == 0 && d->groups == 1 && layer_index != 0 &&
         d->size * d->size <= 31;
}

extern "C" int dk_conv_forward_half(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream)
{
  return dk_conv_forward_half_strided(d, x, weights, biases, y, residual, activation_input, stream, 0);
}

int dk_conv_forward_half_strided(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream,
    int out_ctot)
{
  if (!d || !x || !weights || !y || !dk_conv_half_eligible(d, 1))
  {
    fprintf(stderr, "dk_conv_forward_half: layer is not eligible for the fp16-operand path\n");
    return 1;
  }
  const int pad = d->pad * d->dilation;
  const int keff = d->dilation * (d->size - 1) + 1;
  const int OH = (d->h + 2 * pad - keff) / d->stride_y + 1;
  const int OW = (d->w + 2 * pad - keff) / d->stride_x + 1;
  const int C = d->c, M = d->n, K = C * d->size * d->size;
  if (out_ctot && (out_ctot < d->n || residual || activation_input))
    return 1;
  const int Mtot = out_ctot ? out_ctot : d->n;
  const size_t in_img = (size_t)d->c * d->h * d->w, out_img = (size_t)Mtot * OH * OW;
  int chunk = d->batch;
  const size_t lim_in = (size_t)1 << 29, lim_out = (size_t)1 << 30;
  if (in_img * chunk >= lim_in || out_img * chunk >= lim_out)
  {
    chunk = (int)((lim_in - 1) / in_img);
    const int c2 = (int)((lim_out - 1) / out_img);
    if (c2 < chunk)
      chunk = c2;
    if (chunk < 1)
      return 1;
  }
  const int2* ktab = dk_conv_ktab(d, K, C, 0);
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  static int fast = -1;
  if (fast < 0)
  {
    const char* e = getenv("DK_FAST_MISH");
    fast = (e && !atoi(e)) ? 0 : 1;
  }
  for (int b0 = 0; b0 < d->batch; b0 += chunk)
  {
    const int nb = (d->batch - b0 < chunk) ? d->batch - b0 : chunk;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x + (size_t)b0 * in_img;
    a.w = weights;
    a.bias = biases;
    a.y = y + (size_t)b0 * out_img;
    a.residual = residual ? residual + (size_t)b0 * out_img : nullptr;
    a.act_in = activation_input ? activation_input + (size_t)b0 * out_img : nullptr;
    a.ktab = ktab;
    a.x_bytes = (unsigned)(in_img * nb * 4);
    a.w_bytes = (unsigned)((size_t)M * K * 4);
    a.y_bytes = (unsigned)((out_img * (nb - 1) + (size_t)d->n * OH * OW) * 4);
    a.C = C; a.H = d->h; a.W = d->w; a.Ctot = d->c;
    a.M = M; a.Mtot = Mtot; a.K = K;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW;
    a.N = nb * OH * OW;
    a.size = d->size; a.stride_x = d->stride_x; a.stride_y = d->stride_y;
    a.pad = pad; a.dil = d->dilation;
    a.act = d->activation;
    if (d->activation == DK_MISH && fast)
      a.act |= DK_ACT_FAST;
    a.groups = 1;
    a.mode = 0;
    // tile choice: 128x128 unless M is small
    const bool small_m = M <= 64;
    const int bm = small_m ? 64 : 128, bn = 128;
    a.tiles_m = (M + bm - 1) / bm;
    a.tiles_n = (a.N + bn - 1) / bn;
    conv_args_finish(a);
    const long long nblk = conv_pick_partition(a, (size_t)M * K * sizeof(float), bm);
    DkProfScope prof;
    dk_prof_begin(prof, st);
    if (small_m)
      hipLaunchKernelGGL((conv_igemm_f16<64, 128, 32, 64>), dim3((unsigned)nblk), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv_igemm_f16<128, 128, 64, 64>), dim3((unsigned)nblk), dim3(256), 0, st, a);
    CHECK_HIP(hipPeekAtLastError());
    if (prof.e0)
      dk_prof_end(prof, st, dk_prof_named_slot(small_m ? "conv_igemm_f16<64, 128, 32, 64>" : "conv_igemm_f16<128, 128, 64, 64>"),
          2.0 * (double)M * K * (double)a.N / 1e9);
  }
  return 0;
}
