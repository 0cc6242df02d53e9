// conv_stem.hip -- the first two convolutions of the yolov4 family in ONE kernel (gfx950):
//   layer 0: 3x3 / stride 1 / pad 1, C0 <= 4 input channels -> 32 filters (yolov4 / yolov4-csp: 3 -> 32 at 608 / 512)
//   layer 1: 3x3 / stride 2 / pad 1, 32 -> M1 filters (a multiple of 64)
// Reference: two ForwardConvolutionalLayer calls (src/convolutional_layer.cpp:1128-1305) with the 32-channel full-resolution
// tensor written to and read back from memory in between -- at 608x608 b=16 that tensor is 757 MB: layer 0 is a 757 MB
// write at K = 27 (0.32 ms, 2.9 TB/s) and layer 1 re-reads it (0.61 ms).  Nothing else reads layer 0's output, so here a
// workgroup computes the patch of layer-0 OUTPUT its layer-1 tile needs straight into LDS and multiplies from there:
//
//   tile = 2 output rows x 32 output columns of layer 1 (64 pixels) x 64 filters; 256 threads = 4 waves (wm, wn): 32
//   filters x the 32 pixels of row wn.
//   phase 0  the 7 x 67 x C0 input patch -> LDS (zero outside the image);
//   phase 1  layer 0 on the VALU: the 5 x 65 positions x 32 channels the tile's taps touch, each a k-ascending fmaf chain
//            over (c, kh, kw) from 0, then + bias, then the activation -- the same chain, order and rounding the fp32 MFMA
//            kernels produce (cdna_hip_programming.md: v_mfma_f32_32x32x2_f32 == fmaf chain), so the fused result is BITWISE
//            the unfused one; positions outside the image are layer 1's zero padding, not a convolution of padded input;
//   phase 2  layer 1 on the MFMA pipe: the B fragment of tap (c, kh, kw) is the lane's patch position + a constant
//            (conv3x3_direct.hip's formulation with stride-2 lane positions), the filters are staged 4 channels at a time
//            (double-buffered, register prefetch); k ascending as everywhere;
//   epilogue bias + activation, 32 consecutive pixels per store instruction.
// Two workgroups fit a CU (67 KB of LDS each): one's VALU phase runs beside the other's MFMA phase.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "conv_common.h"
#include "dark_hip.h"
#include "dk_device_math.h"
#include "dk_internal.h"

namespace
{
constexpr int CM = 32;               // filters of layer 0 = channels of layer 1
constexpr int TOW = 32, TOH = 2;     // layer-1 output tile
constexpr int AROWS = 2 * TOH + 1;   // 5 rows of layer-0 output
constexpr int ACOLS = 2 * TOW + 1;   // 65 columns
constexpr int AP = 66;               // row pitch of the layer-0 patch (floats)
constexpr int ACAP = AROWS * AP;     // floats per channel
constexpr int XROWS = AROWS + 2, XCOLS = ACOLS + 2, XP = 68;   // input patch 7 x 67, pitch 68
constexpr int SBM = 64;              // filters of layer 1 per workgroup
constexpr int SCK = 4, SKS = SCK * 9;   // channels / k per stage of the filter tile
constexpr int SAS = SKS + 1;            // padded row of the filter stage
constexpr int SA_FLOATS = (SBM * SAS + 3) / 4 * 4;

struct StemArgs
{
  const float* x;       // [b][C0][H][W]
  const float* w0;      // [32][C0][3][3]
  const float* b0;      // [32] or NULL
  const float* w1;      // [M1][32][3][3]
  const float* b1;      // [M1] or NULL
  float* y;             // [b][M1tot][H/2][W/2]
  unsigned x_bytes, w1_bytes, y_bytes;
  int C0, H, W, OH, OW, M1, M1tot, batch;
  int act0, act1;
  int abl;              // DK_STEM_ABL timing diagnostics (results are garbage by construction)
  int tiles_x, tiles_y, tiles_m;
  double inv_tiles_x, inv_tiles_xy;
};

template <int P, int CAP>
constexpr int stem_tap_off(int k)
{
  return (k / 9) * CAP + ((k % 9) / 3) * P + (k % 3);
}

// C0: input channels of layer 0 (compile time: the filters of a wave's 8 channels are then batched scalar loads and the
// chain has no per-term branch); ACT >= 0: both layers' activation, known at compile time; ACT < 0: p.act0 / p.act1.
template <int C0, int ACT>
__global__ void __launch_bounds__(256) conv_stem_f32(const StemArgs p)
{
  // the input patch lives in the SECOND filter stage buffer: it is dead when phase 1 ends and that buffer is first written
  // inside the stage loop, behind phase 1's closing barrier
  static_assert(4 * XROWS * XP <= SA_FLOATS, "input patch does not fit the filter stage it aliases");
  __shared__ __attribute__((aligned(16))) float as[CM * ACAP];
  __shared__ __attribute__((aligned(16))) float ws[2 * SA_FLOATS];
  float* const xs = ws + SA_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;

  // workgroup -> (filter tile, image, tile row, tile column); consecutive workgroups walk a tile row
  int bid = blockIdx.x;
  const int tile_m = bid % p.tiles_m;
  bid /= p.tiles_m;
  const int bt = fdiv(bid, p.tiles_x * p.tiles_y, p.inv_tiles_xy);
  const int r2 = bid - bt * p.tiles_x * p.tiles_y;
  const int ty = fdiv(r2, p.tiles_x, p.inv_tiles_x);
  const int tx = r2 - ty * p.tiles_x;
  const int oy0 = ty * TOH, ox0 = tx * TOW;
  const int ay0 = 2 * oy0 - 1, ax0 = 2 * ox0 - 1;   // layer-0 output (row, column) of patch position (0, 0)
  const int H = p.H, W = p.W;
  const int m0 = tile_m * SBM;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000);

  // ---- phase 0: input patch rows ay0 - 1 .. ay0 + 5, columns ax0 - 1 .. ax0 + 65 (all loads in flight, then the stores) ---
  {
    constexpr int XN = C0 * XROWS * XP, XPASS = (XN + 255) / 256;
    float xin[XPASS];
#pragma unroll
    for (int j = 0; j < XPASS; ++j)
    {
      const int i = tid + j * 256;
      const int c = i / (XROWS * XP);
      const int rem = i - c * XROWS * XP;
      const int r = rem / XP, col = rem - r * XP;
      const int iy = ay0 - 1 + r, ix = ax0 - 1 + col;
      const bool ok = i < XN && col < XCOLS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W && !(p.abl & 4);
      xin[j] = ld_buf(xr, ok ? (unsigned)(((bt * C0 + c) * H + iy) * W + ix) * 4u : OOB);
    }
#pragma unroll
    for (int j = 0; j < XPASS; ++j)
      if (tid + j * 256 < XN)
        xs[tid + j * 256] = xin[j];
  }
  // first filter stage of layer 1 into registers meanwhile
  constexpr int AQ = SBM * 9;                 // float4 slots of a filter stage
  constexpr int PA = (AQ + 255) / 256;
  unsigned aofs[PA];
  int a_lds[PA];
#pragma unroll
  for (int jj = 0; jj < PA; ++jj)
  {
    const int u = tid + jj * 256;
    const int row = u / 9, q4 = u - row * 9;
    const bool ok = u < AQ && (m0 + row) < p.M1;
    aofs[jj] = ok ? (unsigned)((m0 + row) * (CM * 9) + q4 * 4) * 4u : OOB;
    a_lds[jj] = row * SAS + q4 * 4;
  }
  float ra[PA * 4];
  auto load_w = [&](int st) {
#pragma unroll
    for (int jj = 0; jj < PA; ++jj)
    {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)aofs[jj], st * (SKS * 4), 0);
      ra[4 * jj + 0] = __uint_as_float(v.x);
      ra[4 * jj + 1] = __uint_as_float(v.y);
      ra[4 * jj + 2] = __uint_as_float(v.z);
      ra[4 * jj + 3] = __uint_as_float(v.w);
    }
  };
  auto store_w = [&](float* stg) {
#pragma unroll
    for (int jj = 0; jj < PA; ++jj)
      if ((jj + 1) * 256 <= AQ || tid < AQ - jj * 256)
      {
#pragma unroll
        for (int e = 0; e < 4; ++e) stg[a_lds[jj] + e] = ra[4 * jj + e];
      }
  };
  load_w(0);
  __syncthreads();

  // ---- phase 1: layer 0 at the 5 x 65 patch positions; wave q computes channels 8q .. 8q+7 (its filters are wave-uniform:
  // scalar loads), 64 positions per pass, six passes ---------------------------------------------------------------------
  {
    const int act0 = ACT >= 0 ? ACT : p.act0;
    constexpr int K0 = C0 * 9;
    const int q = __builtin_amdgcn_readfirstlane(wave);
    const float* const wq = p.w0 + (size_t)(q * 8) * K0;
    float bq[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) bq[m] = p.b0 ? p.b0[q * 8 + m] : 0.f;
    for (int pos = lane; pos < ((p.abl & 1) ? 0 : AROWS * ACOLS); pos += 64)
    {
      const int r = pos / ACOLS, col = pos - r * ACOLS;
      const int ay = ay0 + r, ax = ax0 + col;
      float* const dst = as + (q * 8) * ACAP + r * AP + col;
      if ((unsigned)ay >= (unsigned)H || (unsigned)ax >= (unsigned)W)
      {
        // layer 1's zero padding
#pragma unroll
        for (int m = 0; m < 8; ++m) dst[m * ACAP] = 0.f;
        continue;
      }
      float xv[K0];
#pragma unroll
      for (int c = 0; c < C0; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t)
          xv[c * 9 + t] = xs[(c * XROWS + r + t / 3) * XP + col + t % 3];
#pragma unroll
      for (int m = 0; m < 8; ++m)
      {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < K0; ++k)
          s = __builtin_fmaf(wq[m * K0 + k], xv[k], s);   // (w, x) as the MFMA's (A, B); k ascending from 0
        dst[m * ACAP] = dk_activate(s + bq[m], act0);
      }
    }
  }
  store_w(ws);
  __syncthreads();

  // ---- phase 2: layer 1, 8 stages of 4 channels ------------------------------------------------------------------------
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr int D1 = 1, D2 = AP - 2, D3 = ACAP - 2 * AP - 2;
  const int lpos = (2 * wn) * AP + 2 * l31;
  const int lb[3] = {lpos + lh * D1, lpos + lh * D2, lpos + lh * D3};
  for (int st = 0; st < ((p.abl & 2) ? 0 : CM / SCK); ++st)
  {
    const float* cur = ws + (st & 1) * SA_FLOATS;
    const bool more = st + 1 < CM / SCK;
    if (more)
      load_w(st + 1);
    const float* As = cur + (wm * 32 + l31) * SAS + lh;
    const float* Ps = as + st * SCK * ACAP;
#pragma unroll
    for (int s = 0; s < SKS / 2; ++s)
    {
      const int o0 = stem_tap_off<AP, ACAP>(2 * s);
      const int dd = stem_tap_off<AP, ACAP>(2 * s + 1) - o0;
      const int x = (dd == D1) ? 0 : (dd == D2) ? 1 : 2;
      const float a = As[2 * s];
      const float b = Ps[lb[x] + o0];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (more)
      store_w(ws + ((st + 1) & 1) * SA_FLOATS);
    __syncthreads();
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  const int oy = oy0 + wn, ox = ox0 + l31;
  const bool pv = oy < p.OH && ox < p.OW;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
  const unsigned row_bytes = (unsigned)(p.OH * p.OW) * 4u;
  const unsigned pbase = (unsigned)((bt * p.M1tot) * p.OH * p.OW + oy * p.OW + ox) * 4u;
  const int act1 = ACT >= 0 ? ACT : p.act1;
#pragma unroll
  for (int r = 0; r < 16; ++r)
  {
    const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    const bool ok = pv && m < p.M1 && !(p.abl & 8);
    float v = acc[r] + ((p.b1 && m < p.M1) ? p.b1[m] : 0.f);
    v = dk_activate(v, act1);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, (int)(ok ? pbase + (unsigned)m * row_bytes : 0xFFFFFFF0u), 0, 0);
  }
}
}  // namespace

// 1 when the pair (d0, d1) is the stem this kernel computes
extern "C" int dk_conv_stem_applicable(const DkConvDesc* d0, const DkConvDesc* d1)
{
  if (!d0 || !d1)
    return 0;
  const bool l0 = d0->size == 3 && d0->stride_x == 1 && d0->stride_y == 1 && d0->pad == 1 && d0->dilation == 1 && d0->groups == 1 &&
                  d0->c >= 1 && d0->c <= 4 && d0->n == CM;
  const bool l1 = d1->size == 3 && d1->stride_x == 2 && d1->stride_y == 2 && d1->pad == 1 && d1->dilation == 1 && d1->groups == 1 &&
                  d1->c == CM && d1->n % SBM == 0 && d1->h == d0->h && d1->w == d0->w && d1->batch == d0->batch;
  return l0 && l1 && d0->h % 2 == 0 && d0->w % 2 == 0 && (size_t)d0->batch * d1->n * (d0->h / 2) * (d0->w / 2) < ((size_t)1 << 30) &&
         (size_t)d0->batch * d0->c * d0->h * d0->w < ((size_t)1 << 29);
}

// y = act1(b1 + conv3x3/s2(act0(b0 + conv3x3/s1(x)))): layers 0 and 1 of the yolov4 family in one launch.  out_ctot: channels of
// the tensor y is a slice of (0: dense), as in dk_conv_forward's strided form.
extern "C" int dk_conv_stem_forward(const DkConvDesc* d0, const DkConvDesc* d1, const float* x, const float* w0, const float* b0,
    const float* w1, const float* b1, float* y, int out_ctot, void* stream)
{
  if (!dk_conv_stem_applicable(d0, d1) || !x || !w0 || !w1 || !y || (out_ctot && out_ctot < d1->n))
  {
    fprintf(stderr, "dk_conv_stem_forward: the layer pair is not the fused stem's\n");
    return 1;
  }
  StemArgs a;
  memset(&a, 0, sizeof(a));
  a.x = x; a.w0 = w0; a.b0 = b0; a.w1 = w1; a.b1 = b1; a.y = y;
  a.C0 = d0->c; a.H = d0->h; a.W = d0->w; a.OH = d0->h / 2; a.OW = d0->w / 2;
  a.M1 = d1->n; a.M1tot = out_ctot ? out_ctot : d1->n; a.batch = d0->batch;
  a.x_bytes = (unsigned)((size_t)d0->batch * d0->c * d0->h * d0->w * 4);
  a.w1_bytes = (unsigned)((size_t)d1->n * CM * 9 * 4);
  a.y_bytes = (unsigned)((((size_t)d0->batch - 1) * a.M1tot + d1->n) * a.OH * a.OW * 4);
  a.act0 = d0->activation; a.act1 = d1->activation;
  if (d0->activation == DK_MISH && dk_fast_mish_enabled()) a.act0 |= DK_ACT_FAST;
  if (d1->activation == DK_MISH && dk_fast_mish_enabled()) a.act1 |= DK_ACT_FAST;
  static const int abl = getenv("DK_STEM_ABL") ? atoi(getenv("DK_STEM_ABL")) : 0;
  a.abl = abl;
  a.tiles_x = (a.OW + TOW - 1) / TOW;
  a.tiles_y = (a.OH + TOH - 1) / TOH;
  a.tiles_m = d1->n / SBM;
  a.inv_tiles_x = 1.0 / a.tiles_x;
  a.inv_tiles_xy = 1.0 / ((double)a.tiles_x * a.tiles_y);
  const long long nblk = (long long)a.tiles_m * a.tiles_x * a.tiles_y * d0->batch;
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  DkProfScope prof;
  dk_prof_begin(prof, st);
  auto launch = [&](auto c0) {
    constexpr int C0 = decltype(c0)::value;
    if (a.act0 == a.act1 && a.act0 == (DK_MISH | DK_ACT_FAST))
      hipLaunchKernelGGL((conv_stem_f32<C0, (DK_MISH | DK_ACT_FAST)>), dim3((unsigned)nblk), dim3(256), 0, st, a);
    else if (a.act0 == a.act1 && a.act0 == DK_LEAKY)
      hipLaunchKernelGGL((conv_stem_f32<C0, DK_LEAKY>), dim3((unsigned)nblk), dim3(256), 0, st, a);
    else
      hipLaunchKernelGGL((conv_stem_f32<C0, -1>), dim3((unsigned)nblk), dim3(256), 0, st, a);
  };
  switch (d0->c)
  {
    case 1: launch(std::integral_constant<int, 1>()); break;
    case 2: launch(std::integral_constant<int, 2>()); break;
    case 3: launch(std::integral_constant<int, 3>()); break;
    default: launch(std::integral_constant<int, 4>()); break;
  }
  CHECK_HIP(hipPeekAtLastError());
  if (prof.e0)
    dk_prof_end(prof, st, dk_prof_named_slot("conv_stem_f32"),
        (2.0 * CM * d0->c * 9 * (double)d0->h * d0->w + 2.0 * d1->n * CM * 9 * (double)a.OH * a.OW) * d0->batch / 1e9);
  return 0;
}
