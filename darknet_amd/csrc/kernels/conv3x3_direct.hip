// conv3x3_direct.hip -- 3x3 / stride 1 / pad 1 convolution as an implicit GEMM whose
// B operand is read straight out of an input PATCH held in LDS (gfx950).
//
// Same contraction as conv_igemm_f32 (the reference's im2col + gemm_nn,
// src/im2col.c:56-104 + src/gemm.c:2223-2239: M = filters, N = batch*oh*ow folded,
// K = (c, kh, kw) ascending, fp32 MFMA 32x32x2, identical accumulation order), but
// the im2col matrix is not gathered element by element.  For a tile of BN
// consecutive output pixels the block stages, per input channel, the few input rows
// those pixels touch as a 2-D patch with a compile-time row pitch P and zero halos;
// the MFMA B fragment of tap (c, kh, kw) is then the lane's patch position plus the
// CONSTANT c*CAP + kh*P + kw, i.e. a ds_read_b32 with an immediate offset:
//
//   * 9 taps re-use one staged element: ~4-6x fewer global loads and LDS staging
//     writes than the gather (which loads every im2col element separately),
//   * no tap table, no per-tap padding mask, no address VALU in the K loop,
//   * one barrier per 36 k (4 channels x 9 taps) instead of one per 16.
//
// "Extended rows": every image gets a zero row above and below (H+2 rows); the
// patch covers the extended rows [R0, R0 + rows_used) where R0 is the extended row
// above the tile's first pixel, so a tile may straddle images (batch folded into N)
// and vertical padding needs no special case.  Horizontal padding = the patch's
// zero columns 0 and W+1.
//
// Eligibility (host): size 3, stride 1, dilation 1, pad 1, groups 1, C % 4 == 0,
// W + 2 <= P and the tile's worst-case row count <= ROWS for one compiled (P, ROWS)
// class; everything else keeps the gather kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "conv_common.h"
#include "dark_hip.h"
#include "dk_device_math.h"
#include "dk_internal.h"

// k-step after which the next stage's tiles are written to LDS (mid-sequence: measured
// 1-4 % faster than after the last MFMA); undefine to store at the end of the stage
#ifndef DK_DIRECT_MIDSTORE
#define DK_DIRECT_MIDSTORE 11
#endif

// diagnostic ablation builds: see conv_igemm.hip (DK_ABL bits: 0 epilogue, 1 LDS staging writes,
// 2 global loads, 3 barriers, 4 LDS fragment reads)
#ifndef DK_ABL
#define DK_ABL 0
#endif

namespace
{
constexpr int CK = 4;        // channels per K stage
constexpr int KS = CK * 9;   // k per stage

// floats of the A region of one stage: BM padded rows
constexpr int direct_a_floats(int bm) { return (bm * (KS + 1) + 3) / 4 * 4; }

template <int P, int CAP>
constexpr int tap_off(int k)
{
  return (k / 9) * CAP + ((k % 9) / 3) * P + (k % 3);
}
}  // namespace

template <int BM, int BN, int WM, int WN, int P, int ROWS>
#ifndef DK_DIRECT_MIN_BLOCKS
#define DK_DIRECT_MIN_BLOCKS 1
#endif
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, DK_DIRECT_MIN_BLOCKS)
conv3x3_direct_f32(const ConvArgs p)
{
  constexpr int NWN = BN / WN;
  constexpr int NW = (BM / WM) * NWN;
  constexpr int T = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AS = KS + 1;               // padded A row stride (floats): conflict-free fragment reads
  constexpr int CAP = ROWS * P;            // patch floats per channel
  constexpr int PPT = (CAP + T - 1) / T;   // patch slots per thread and channel
  constexpr int AQ = BM * 9;               // float4 slots of the A stage
  constexpr int PA = (AQ + T - 1) / T;
  constexpr int A_FLOATS = direct_a_floats(BM);
  constexpr int STAGE = A_FLOATS + CK * CAP;
  constexpr int D1 = 1, D2 = P - 2, D3 = CAP - 2 * P - 2;  // off(2s+1) - off(2s) takes these values

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int l31 = lane & 31, lh = lane >> 5;

  int g_unused, tile_m, tile_n;
  if (!conv_block_tile(p, g_unused, tile_m, tile_n))
    return;
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int H = p.H, W = p.W, HW = H * W, He = H + 2, K = p.K;
  const int nbatch = fdiv(p.N, HW, p.inv_HW);
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // ---- block geometry in extended rows -----------------------------------------
  int R0, rows_used;
  {
    const int b0 = fdiv(n0, HW, p.inv_HW);
    const int oy0 = fdiv(n0 - b0 * HW, W, p.inv_W);
    R0 = b0 * He + oy0;  // extended row of the first pixel is R0 + 1
    const int nl = ((n0 + BN < p.N) ? n0 + BN : p.N) - 1;
    const int bl = fdiv(nl, HW, p.inv_HW);
    const int oyl = fdiv(nl - bl * HW, W, p.inv_W);
    rows_used = bl * He + oyl + 1 - R0 + 2;
  }
  const int used_slots = rows_used * P;

  // ---- per-lane patch positions of the wave's pixel columns ------------------------
  // lb[j][x] = position of tap (0,0) of pixel j's column, plus lh * D_x
  int lb[TN][3];
#pragma unroll
  for (int j = 0; j < TN; ++j)
  {
    int n = n0 + wn * WN + j * 32 + l31;
    n = (n < p.N) ? n : p.N - 1;
    const int b = fdiv(n, HW, p.inv_HW);
    const int q = n - b * HW;
    const int oy = fdiv(q, W, p.inv_W);
    const int ox = q - oy * W;
    const int lpos = (b * He + oy - R0) * P + ox;
    lb[j][0] = lpos + lh * D1;
    lb[j][1] = lpos + lh * D2;
    lb[j][2] = lpos + lh * D3;
  }

  // ---- patch loader: slot i -> (extended row, column), fixed over channels ----------
  unsigned gofs[PPT];
#pragma unroll
  for (int jj = 0; jj < PPT; ++jj)
  {
    const int i = tid + jj * T;
    const int r = i / P;
    const int col = i - r * P - 1;
    const int Rr = R0 + r;
    const int b = fdiv(Rr, He, p.inv_He);
    const int ye = Rr - b * He - 1;
    const bool ok = i < used_slots && (unsigned)col < (unsigned)W && (unsigned)ye < (unsigned)H && b < nbatch;
    gofs[jj] = ok ? (unsigned)((b * p.Ctot * H + ye) * W + col) * 4u : OOB;
  }

  // ---- A loader: slot u -> (row, float4 of the 36-float stage row) -------------------
  unsigned aofs[PA];
  int a_lds[PA];
#pragma unroll
  for (int jj = 0; jj < PA; ++jj)
  {
    const int u = tid + jj * T;
    const int row = u / 9;
    const int q4 = u - row * 9;
    const bool ok = u < AQ && (m0 + row) < p.M;
    aofs[jj] = ok ? (unsigned)((m0 + row) * K + q4 * 4) * 4u : OOB;
    a_lds[jj] = row * AS + q4 * 4;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nst = p.C / CK;

  // The number of patch slot iterations a block needs (its rows * P / T) is block
  // uniform.  Two forms of the K loop, chosen per tile shape by measurement (A/B on
  // yolov4's layers, MI355X): BRANCHY keeps one loop and skips unused slot
  // iterations with scalar branches (3 % faster for the 128-wide tiles), the other
  // dispatches once to a loop specialised on the count (straight-line body, counted
  // vmcnt waits; faster for 64x64).
  constexpr bool BRANCHY = !(BM == 64 && BN == 64);
  auto run = [&](auto njc) {
    constexpr int NJ = decltype(njc)::value;
    float ra[PA * 4];
    float rp[CK * NJ];

    auto load_stage = [&](int st) {
      if (DK_ABL & 4)
      {
#pragma unroll
        for (int j = 0; j < PA * 4; ++j) ra[j] = (float)(st + j);
#pragma unroll
        for (int j = 0; j < CK * NJ; ++j) rp[j] = (float)(st - j);
        return;
      }
      const int ak = st * (KS * 4);  // byte offset of the stage inside a weight row
#pragma unroll
      for (int jj = 0; jj < PA; ++jj)
      {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)aofs[jj], ak, 0);
        ra[4 * jj + 0] = __uint_as_float(v.x);
        ra[4 * jj + 1] = __uint_as_float(v.y);
        ra[4 * jj + 2] = __uint_as_float(v.z);
        ra[4 * jj + 3] = __uint_as_float(v.w);
      }
#pragma unroll
      for (int cc = 0; cc < CK; ++cc)
      {
        const int coff = (st * CK + cc) * HW * 4;
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          if (!BRANCHY || jj * T < used_slots)
            rp[cc * NJ + jj] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xr, (int)gofs[jj], coff, 0));
      }
    };

    auto store_stage = [&](float* stg) {
      if (DK_ABL & 2)
      {
#pragma unroll
        for (int j = 0; j < PA * 4; ++j) asm volatile("" ::"v"(ra[j]));
#pragma unroll
        for (int j = 0; j < CK * NJ; ++j) asm volatile("" ::"v"(rp[j]));
        return;
      }
      float* As = stg;
      float* Ps = stg + A_FLOATS;
#pragma unroll
      for (int jj = 0; jj < PA; ++jj)
        if ((jj + 1) * T <= AQ || tid < AQ - jj * T)  // only the last iteration is partial
        {
#pragma unroll
          for (int e = 0; e < 4; ++e) As[a_lds[jj] + e] = ra[4 * jj + e];
        }
#pragma unroll
      for (int cc = 0; cc < CK; ++cc)
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj)
          if ((!BRANCHY || jj * T < used_slots) && ((jj + 1) * T <= CAP || tid < CAP - jj * T))
            Ps[cc * CAP + tid + jj * T] = rp[cc * NJ + jj];
    };

    load_stage(0);
    store_stage(lds);
    if (!(DK_ABL & 8))
      __syncthreads();

    for (int st = 0; st < nst; ++st)
    {
      float* cur = lds + (st & 1) * STAGE;
      const bool more = (st + 1) < nst;
      if (more)
        load_stage(st + 1);

      const float* As = cur + (wm * WM + l31) * AS + lh;
      const float* Ps = cur + A_FLOATS;
      const float* pj[TN][3];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int x = 0; x < 3; ++x) pj[j][x] = Ps + lb[j][x];

      // MFMA step s contracts k = 2s (lanes 0-31) and 2s+1 (lanes 32-63), k ascending
#pragma unroll
      for (int s = 0; s < KS / 2; ++s)
      {
        const int o0 = tap_off<P, CAP>(2 * s);
        const int dd = tap_off<P, CAP>(2 * s + 1) - o0;
        const int x = (dd == D1) ? 0 : (dd == D2) ? 1 : 2;
        float a[TM], b[TN];
        if (DK_ABL & 16)
        {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i] = (float)(lane + i + s);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j] = (float)(lane - j - s);
        }
        else
        {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[i * 32 * AS + 2 * s];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = pj[j][x][o0];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
#ifdef DK_DIRECT_MIDSTORE
        // the next stage's tiles go to the OTHER buffer: write them in the middle of
        // the MFMA sequence (their loads were issued a half stage ago) instead of after it
        if (s == DK_DIRECT_MIDSTORE && more)
          store_stage(lds + ((st + 1) & 1) * STAGE);
#endif
      }

#ifndef DK_DIRECT_MIDSTORE
      if (more)
        store_stage(lds + ((st + 1) & 1) * STAGE);
#endif
      if (!(DK_ABL & 8))
        __syncthreads();
    }
  };

  const int nj = (used_slots + T - 1) / T;
  static_assert(PPT <= 4, "patch slot dispatch");
  if (BRANCHY)
    run(std::integral_constant<int, PPT>{});
  else if (nj <= 1)
    run(std::integral_constant<int, 1>{});
  else if (nj == 2 || PPT < 3)
    run(std::integral_constant<int, (PPT < 2 ? PPT : 2)>{});
  else if (nj == 3 || PPT < 4)
    run(std::integral_constant<int, (PPT < 3 ? PPT : 3)>{});
  else
    run(std::integral_constant<int, PPT>{});

  if (DK_ABL & 1)
  {
    float sacc = 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
    if (sacc == 123.456f)
      p.y[tid] = sacc;
    return;
  }
  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, m0, n0, 0, wm, wn, l31, lh);
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
namespace
{
struct PitchClass
{
  int p, rows;
};
// (row pitch, rows): sized for the feature maps of 416/512/608 nets with 64..128-pixel tiles
// (w = 13..22 | 23..38 | 39..78 | 79..158)
#ifndef DK_PC0
#define DK_PC0 24, 15
#endif
#ifndef DK_PC1
#define DK_PC1 40, 10
#endif
#ifndef DK_PC2
#define DK_PC2 80, 8
#endif
#ifndef DK_PC3
#define DK_PC3 160, 6
#endif
const PitchClass g_pc[4] = {{DK_PC0}, {DK_PC1}, {DK_PC2}, {DK_PC3}};

struct DirectCfg
{
  int bm, bn, wm, wn;
  const char* name;
  void (*kernel[4])(const ConvArgs);
};

#define DK_DCFG(BM, BN, WM, WN)                                                                \
  {                                                                                            \
    BM, BN, WM, WN, "direct3x3_" #BM "x" #BN "_w" #WM "x" #WN,                                 \
    {                                                                                          \
      conv3x3_direct_f32<BM, BN, WM, WN, DK_PC0>, conv3x3_direct_f32<BM, BN, WM, WN, DK_PC1>,  \
          conv3x3_direct_f32<BM, BN, WM, WN, DK_PC2>, conv3x3_direct_f32<BM, BN, WM, WN, DK_PC3> \
    }                                                                                          \
  }

const DirectCfg g_dcfgs[] = {
    DK_DCFG(128, 128, 64, 64),
    DK_DCFG(128, 64, 64, 32),
    DK_DCFG(64, 128, 32, 64),
    DK_DCFG(64, 64, 32, 32),
};
const int g_ndcfg = sizeof(g_dcfgs) / sizeof(g_dcfgs[0]);

int lds_bytes(const DirectCfg& c, int pc)
{
  return 2 * (direct_a_floats(c.bm) + CK * g_pc[pc].rows * g_pc[pc].p) * (int)sizeof(float);
}

// pitch class for (tile width bn, image w x h) or -1
int pitch_class(int bn, int w, int h)
{
  const int hw = w * h;
  const int rows = (bn - 1 + w - 1) / w + 1 + 2 * ((bn - 1 + hw - 1) / hw) + 2;
  for (int i = 0; i < 4; ++i)
    if (w + 2 <= g_pc[i].p)
      return rows <= g_pc[i].rows ? i : -1;
  return -1;
}
}  // namespace

int dk_conv_direct_num_configs() { return g_ndcfg; }

const char* dk_conv_direct_config_name(int dcfg)
{
  return (dcfg >= 0 && dcfg < g_ndcfg) ? g_dcfgs[dcfg].name : nullptr;
}

const char* dk_conv_direct_kernel_name(int dcfg, int pc)
{
  static thread_local char buf[128];
  if (dcfg < 0 || dcfg >= g_ndcfg || pc < 0 || pc > 3)
    return nullptr;
  const DirectCfg& c = g_dcfgs[dcfg];
  snprintf(buf, sizeof(buf), "conv3x3_direct_f32<%d, %d, %d, %d, %d, %d>", c.bm, c.bn, c.wm, c.wn,
      g_pc[pc].p, g_pc[pc].rows);
  return buf;
}

bool dk_conv_direct_applicable(const DkConvDesc* d, const float* weights, int dcfg)
{
  if (dcfg < 0 || dcfg >= g_ndcfg)
    return false;
  if (d->size != 3 || d->stride_x != 1 || d->stride_y != 1 || d->dilation != 1 || d->pad != 1 ||
      d->groups != 1 || d->c % CK != 0)
    return false;
  if (weights && ((uintptr_t)weights & 15))
    return false;
  return pitch_class(g_dcfgs[dcfg].bn, d->w, d->h) >= 0;
}

// Launches one chunk; `a` was filled by dk_conv_forward_cfg (tiles_m / tiles_n are set here).
// Returns the pitch class used.
int dk_conv_direct_launch(ConvArgs a, int dcfg, hipStream_t st)
{
  const DirectCfg& c = g_dcfgs[dcfg];
  const int pc = pitch_class(c.bn, a.W, a.H);
  a.tiles_m = (a.M + c.bm - 1) / c.bm;
  a.tiles_n = (a.N + c.bn - 1) / c.bn;
  const int bytes = lds_bytes(c, pc);
  dk_set_max_dynamic_lds((const void*)c.kernel[pc], bytes);
  a.groups = 1;
  conv_args_finish(a);
  const long long nblk = conv_pick_partition(a, (size_t)a.M * a.K * sizeof(float), c.bm);
  hipLaunchKernelGGL(c.kernel[pc], dim3((unsigned)nblk), dim3((c.bm / c.wm) * (c.bn / c.wn) * 64),
      bytes, st, a);
  return pc;
}
