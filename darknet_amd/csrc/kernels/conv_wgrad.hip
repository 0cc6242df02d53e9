// conv_wgrad.hip -- weight gradient of the convolution on fp32 MFMA.
//
// Replaces the reference's per-image im2col_gpu_ext + cublasSgemm(NT, beta = 1)
// (src/convolutional_kernels.cu:757-781; CPU: im2col_cpu_ext + gemm(0,1,...),
// src/convolutional_layer.cpp:1345-1356):
//     dW[m][k] += sum_n delta[m][n] * col[k][n]      n = (image, oy, ox), k = (c,kh,kw)
// The contraction runs over n, which is huge (batch*oh*ow) while the output is
// small (n x k*k*c), so the n range is split over many workgroups, each
// accumulates a tile of dW in MFMA accumulators over its slice and adds it
// to dW with float atomics (dW accumulates across images and subdivisions in
// the reference anyway: beta = 1).  col is gathered on the fly (no im2col
// buffer): a thread owns a few fixed taps k and walks the pixels.
// Kernel: block tile (64*TM) x (64*TK) of dW, 4 waves in a 2x2 arrangement (each
// 32*TM x 32*TK), 32 pixels per stage, double-buffered LDS with one barrier per
// stage, the next stage's global loads in flight during the MFMAs.  delta rows are
// read with float4 loads when the image size allows; col taps use the forward
// kernel's branch-free gather (per-stage padding mask of the thread's pixel,
// out-of-range flag ORed into the byte offset, hardware bounds check).
// 3x3 / pad 1 layers of stride 1 or 2 with channels % 32 == 0 and filters % 128 == 0 take conv_wgrad3_f32 further down
// (row-staged operands, all nine taps per workgroup) unless a gather tile is forced or wins the first step's timing.
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

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
  int tiles_m, tiles_k, nsplit, stages_per_split, groups;
  // deterministic mode (dk_set_deterministic): every pixel split stores its tile into its own slice of a workspace
  // ([split][the dW index space], plain stores) and wgrad_reduce_kernel adds the slices in ascending split order
  float* part;
  unsigned long long part_stride;
  int abl;   // DK_WGRAD3_ABL timing diagnostics (results are garbage): 1 no epilogue stores, 2 no staging after the first stage, 4 no MFMAs
};

constexpr unsigned OOB = 0x80000000u;
// LS = 34: rows 8-byte aligned, and 32 consecutive rows read with ds_read_b64 hit 32 distinct bank pairs
// (34 m mod 64 = 2 (17 m mod 32), a bijection), so a lane fetches TWO contraction steps per LDS read
constexpr int NC = 32, LS = NC + 2, T = 256;

__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

// TMAJ: the tap tile is taken from the contraction index re-ordered tap-major, k' = t * C + c (t = kh * size + kw),
// which the host selects when C is a multiple of the tile width: then ALL taps of a workgroup share one (kh, kw), a
// thread's padding test is ONE range check per stage, and its 16 gather offsets are `pixel base + constant` (one
// add each) instead of five VALU operations each -- the kernel spent 5 VALU instructions per MFMA on that
// (rocprofv3: SQ_INSTS_VALU / SQ_INSTS_MFMA = 5.1, MFMA pipe 48 % busy).  dW keeps the reference's [m][c][kh][kw] layout.
template <int TM, int TK, bool AVEC, bool TMAJ>
__global__ void __launch_bounds__(T) conv_wgrad_f32(const WgradArgs p)
{
  constexpr int BM = 64 * TM, BKO = 64 * TK;
  constexpr int A_FL = BM * LS, B_FL = BKO * LS, STAGE = A_FL + B_FL;
  constexpr int PA = AVEC ? BM / 32 : BM / 8;   // delta loads per thread and stage
  constexpr int PB = BKO / 8;                   // gather loads per thread and stage
  extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 stages: [m][n] delta, [k][n] col

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

  const int nstages = (p.N + NC - 1) / NC;
  int st_begin = split * p.stages_per_split;
  int st_end = st_begin + p.stages_per_split;
  if (st_end > nstages)
    st_end = nstages;

  // ---- the thread's fixed taps (B rows) -------------------------------------------
  const int nl = tid & 31;      // pixel within the stage
  const int r8 = tid >> 5;      // 0..7
  const int ss = p.size * p.size;
  unsigned toff[PB];            // byte offset of the tap inside an image-group, or OOB
  int tsh[PB];                  // 31 - tap: shifts the tap's "outside" bit to the sign position
  // TMAJ: the workgroup's tap (kh, kw) and first channel
  const int tm_t = TMAJ ? k0 / p.C : 0, tm_c0 = TMAJ ? k0 - tm_t * p.C : 0;
  const int tm_kh = tm_t / p.size, tm_kw = tm_t - tm_kh * p.size;
#pragma unroll
  for (int j = 0; j < PB; ++j)
  {
    if (TMAJ)
    {
      toff[j] = (unsigned)((tm_c0 + r8 + 8 * j) * HW + tm_kh * p.dil * p.W + tm_kw * p.dil) * 4u;
      tsh[j] = 0;
      continue;
    }
    const int k = k0 + r8 + 8 * j;
    const bool ok = k < p.K;
    const int kk = ok ? k : 0;
    const int c = kk / ss, t = kk - c * ss, kh = t / p.size, kw = t - kh * p.size;
    toff[j] = ok ? (unsigned)(c * HW + kh * p.dil * p.W + kw * p.dil) * 4u : OOB;
    tsh[j] = 31 - t;
  }
  // ---- the thread's fixed delta rows (A rows) ----------------------------------------
  const int aq = AVEC ? (tid & 7) : nl;            // float4 index / pixel index inside the stage
  const int ar = AVEC ? (tid >> 3) : r8;           // first row
  constexpr int AR_STEP = AVEC ? 32 : 8;
  unsigned roff[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j)
  {
    const int m = m0 + ar + AR_STEP * j;
    roff[j] = (m < p.M) ? (unsigned)((g * p.M + m) * p.OHW) * 4u : OOB;
  }

  // ---- pixel state, advanced by NC per stage ------------------------------------------
  // B (and scalar A): pixel n = st*NC + nl -> (b, pix, oy, ox); vector A: n = st*NC + 4*aq -> (bA, pixA)
  int n = st_begin * NC + nl;
  int b = n / p.OHW, pix = n - b * p.OHW;
  int oy = pix / p.OW, ox = pix - oy * p.OW;
  int nA = st_begin * NC + 4 * aq;
  int bA = nA / p.OHW, pixA = nA - bA * p.OHW;

  float ra[AVEC ? PA * 4 : PA], rb[PB];

  auto load_stage = [&]() {
    // delta
    if (AVEC)
    {
      const unsigned base = (nA < p.N) ? (unsigned)(bA * p.Mtot * p.OHW + pixA) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < PA; ++j)
      {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(dr, (int)((base + roff[j]) | ((base | roff[j]) & OOB)), 0, 0);
        ra[4 * j + 0] = __uint_as_float(v.x);
        ra[4 * j + 1] = __uint_as_float(v.y);
        ra[4 * j + 2] = __uint_as_float(v.z);
        ra[4 * j + 3] = __uint_as_float(v.w);
      }
    }
    else
    {
      const unsigned base = (n < p.N) ? (unsigned)(b * p.Mtot * p.OHW + pix) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < PA; ++j) ra[j] = ld_buf(dr, (base + roff[j]) | ((base | roff[j]) & OOB));
    }
    const int iy0 = oy * p.stride_y - p.pad, ix0 = ox * p.stride_x - p.pad;
    if (TMAJ)
    {
      // one tap for the whole workgroup: one range check, then `base + constant` per channel
      const bool in = n < p.N && (unsigned)(iy0 + tm_kh * p.dil) < (unsigned)p.H && (unsigned)(ix0 + tm_kw * p.dil) < (unsigned)p.W;
      const unsigned xb = in ? (unsigned)((b * p.Ctot + g * p.C) * HW + iy0 * p.W + ix0) * 4u : OOB;
#pragma unroll
      for (int j = 0; j < PB; ++j) rb[j] = ld_buf(xr, xb + toff[j]);
      return;
    }
    // col: padding mask of this pixel (bit t set <=> tap t is outside the image)
    unsigned colbits = 0, okbits = 0;
    for (int kw = 0; kw < p.size; ++kw)
      colbits |= ((unsigned)(ix0 + kw * p.dil) < (unsigned)p.W ? 1u : 0u) << kw;
    for (int kh = 0; kh < p.size; ++kh)
      if ((unsigned)(iy0 + kh * p.dil) < (unsigned)p.H)
        okbits |= colbits << (kh * p.size);
    const unsigned nmask = (n < p.N) ? ~okbits : 0xFFFFFFFFu;
    const unsigned xbase = (unsigned)((b * p.Ctot + g * p.C) * HW + iy0 * p.W + ix0) * 4u;
#pragma unroll
    for (int j = 0; j < PB; ++j)
      rb[j] = ld_buf(xr, (xbase + toff[j]) | (((nmask << tsh[j]) | toff[j]) & OOB));
  };

  auto advance = [&]() {
    n += NC;
    pix += NC;
    ox += NC;
    while (ox >= p.OW)
    {
      ox -= p.OW;
      ++oy;
    }
    while (pix >= p.OHW)
    {
      pix -= p.OHW;
      ++b;
      oy = pix / p.OW;
      ox = pix - oy * p.OW;
    }
    if (AVEC)
    {
      nA += NC;
      pixA += NC;
      while (pixA >= p.OHW)
      {
        pixA -= p.OHW;
        ++bA;
      }
    }
  };

  auto store_stage = [&](float* stg) {
    float* As = stg;
    float* Bs = stg + A_FL;
    if (AVEC)
    {
#pragma unroll
      for (int j = 0; j < PA; ++j)
      {
        float2* const dst = (float2*)(As + (ar + 32 * j) * LS + 4 * aq);
        dst[0] = make_float2(ra[4 * j + 0], ra[4 * j + 1]);
        dst[1] = make_float2(ra[4 * j + 2], ra[4 * j + 3]);
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < PA; ++j) As[(ar + 8 * j) * LS + nl] = ra[j];
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) Bs[(r8 + 8 * j) * LS + nl] = rb[j];
  };

  f32x16 acc[TM][TK];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (st_begin < st_end)
  {
    load_stage();
    advance();
    store_stage(lds);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st)
    {
      float* cur = lds + ((st - st_begin) & 1) * STAGE;
      const bool more = (st + 1) < st_end;
      if (more)
      {
        load_stage();
        advance();
      }
      // A operand: lane (row m, pixel); B operand: lane (pixel, col k).  One ds_read_b64 per operand feeds
      // two MFMAs: lanes 0-31 hold pixels (4s, 4s+1) of the stage, lanes 32-63 pixels (4s+2, 4s+3), i.e. the
      // contraction pairs are (4s, 4s+2) then (4s+1, 4s+3) -- the sum over pixels has no prescribed order here
      // (it ends in float atomics across the pixel splits anyway)
      const float* ap = cur + (wm * 32 * TM + l31) * LS + 2 * lh;
      const float* bp = cur + A_FL + (wk * 32 * TK + l31) * LS + 2 * lh;
#pragma unroll
      for (int s = 0; s < NC / 4; ++s)
      {
        float2 a[TM], bb[TK];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const float2*)(ap + i * 32 * LS + 4 * s);
#pragma unroll
        for (int j = 0; j < TK; ++j) bb[j] = *(const float2*)(bp + j * 32 * LS + 4 * s);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].x, bb[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TK; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].y, bb[j].y, acc[i][j], 0, 0, 0);
      }
      if (more)
        store_stage(lds + ((st + 1 - st_begin) & 1) * STAGE);
      __syncthreads();
    }
  }

  // C/D: col (= k) = lane&31, row (= m) = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // TMAJ accumulates into a tap-major workspace [m][t][c] (consecutive lanes = consecutive addresses: the float
  // atomics stay coalesced; scattered to [c][kh][kw] directly they ran 2.4x slower); wgrad_fold_kernel adds the
  // workspace into dW afterwards.  One launch-uniform branch selects plain stores into this split's slice of the
  // partial workspace (deterministic mode) or the atomics; the loops are written out twice so that the default
  // path's epilogue is exactly the code it was.
  auto emit = [&](auto detc) {
    constexpr bool DET = decltype(detc)::value;
    float* const dst = DET ? p.part + (size_t)split * p.part_stride : p.dw;
#pragma unroll
    for (int j = 0; j < TK; ++j)
    {
      const int k = k0 + (wk * TK + j) * 32 + l31;
      if (k >= p.K)
        continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r)
        {
          const int m = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (m < p.M)
          {
            if (DET)
              dst[((size_t)g * p.M + m) * p.K + k] = acc[i][j][r];
            else
              atomicAdd(&dst[((size_t)g * p.M + m) * p.K + k], acc[i][j][r]);
          }
        }
    }
  };
  if (p.part)
    emit(std::true_type{});
  else
    emit(std::false_type{});
}

// ------------------------------------------------------------------------------------------------------------------
// conv_wgrad3_f32 -- the weight gradient of 3x3 / pad 1 layers of stride S = 1 or 2 (round 3).
// The gather kernel above fetches every tap of every pixel as its own 4-byte load (16 per thread and 32-pixel stage, each
// with its own padding test) and re-reads x nine times, once per tap tile; it sits at 50-54 TFLOP/s on these layers.
// Here a workgroup owns 128 filters x (32 channels x ALL 9 taps) of dW and walks output-row segments of SEG pixels:
//   * per stage the delta segment [128][SEG] and the raw input patch [32 ch][3 rows][SEG + halo] are staged with row-wise
//     vector loads (VW floats each: 16 / 8 / 4 bytes by the row alignment W allows; padding = the buffer descriptor's
//     out-of-range zero), double-buffered, one barrier per stage, the next stage's loads in flight during the MFMAs;
//   * wave w owns filters 32 w .. 32 w + 31 and nine 32x32 accumulators, one per tap: the B operand of tap (kh, kw) is the
//     raw patch at the lane's channel + a compile-time offset (conv3x3_direct.hip's formulation), so the nine taps cost
//     nine LDS reads of data that was loaded ONCE -- no im2col, no per-tap gather, no index arithmetic in the loop;
//   * one two-dword LDS read per operand feeds two MFMAs (lanes 0-31 take pixels 4j, 4j+1, lanes 32-63 pixels 4j+2, 4j+3:
//     the contraction pairs are (4j, 4j+2), (4j+1, 4j+3) -- the pixel sum has no prescribed order, as above); row pitches
//     = 2 mod 4 make 32 lanes x 2 dwords hit 64 distinct banks;
//   * epilogue: the tile goes through LDS so that a filter's 32 x 9 gradients leave as 288 CONSECUTIVE floats of
//     dW[m][c][kh][kw] (coalesced float atomics, or plain stores into the pixel split's slice in deterministic mode).
// 10 LDS reads + ~2 global loads per 18 MFMAs instead of the gather kernel's 5 vector instructions per MFMA
// (vector and LDS instructions are not hidden behind fp32 MFMAs on this part).
// S = 2 (the down-sampling layers): the lane's two pixels are two input columns apart, a tap is still a compile-time offset;
// the segment is 20 output pixels there (40 input columns + halo keep two workgroups per CU).
template <int SEG, int VW, int S>
__global__ void __launch_bounds__(T, 2) conv_wgrad3_f32(const WgradArgs p)
{
  constexpr int LSA = SEG + 2;                            // delta row pitch (= 2 mod 4)
  constexpr int HALO = VW;                                // input columns staged before the segment's first pixel
  constexpr int XCH = (HALO + S * (SEG - 1) + 2 + VW - 1) / VW;   // VW-float pieces per input row
  constexpr int XW = XCH * VW;
  constexpr int RP = XW + (6 - XW % 4) % 4;               // input row pitch (= 2 mod 4)
  constexpr int CP = 3 * RP;                              // channel pitch (= 2 mod 4 too)
  constexpr int A_FL = 128 * LSA, B_FL = 32 * CP, STAGE = A_FL + B_FL;
  // piece -> thread: a thread keeps ONE piece column and walks rows with a uniform step, so a piece's address is
  // (per-thread offset) + (pass * uniform step): one add per load, an immediate in the LDS store, no per-piece registers
  constexpr int ACH = SEG / VW;                           // pieces per delta row
  constexpr int ARP = T / ACH;                            // delta rows per pass
  constexpr int PA = (128 + ARP - 1) / ARP;               // passes
  constexpr int XRP = (T / XCH) / 3 * 3;                  // input rows (channel x kh) per pass: a multiple of 3 keeps kh fixed
  constexpr int PB = (96 + XRP - 1) / XRP;
  static_assert(SEG % 4 == 0 && LSA % 4 == 2 && RP % 4 == 2, "pitches");
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 2 stages: [128][LSA] delta, [32][3][RP] input

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;

  int id = blockIdx.x;
  const int split = id % p.nsplit;
  id /= p.nsplit;
  const int tile_c = id % p.tiles_k;
  const int tile_m = id / p.tiles_k;
  const int m0 = tile_m * 128, c0 = tile_c * 32;
  const int nseg = (p.OW + SEG - 1) / SEG;
  const int nstages = p.N / p.OW * nseg;                  // N = images * OH * OW
  const int st_begin = split * p.stages_per_split;
  int st_end = st_begin + p.stages_per_split;
  if (st_end > nstages)
    st_end = nstages;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)p.delta, 0, p.d_bytes, 0x00020000);

  // ---- the thread's piece column and first row ---------------------------------------------------------------------
  const int a_row0 = tid / ACH, a_ch = tid - a_row0 * ACH;
  const bool a_thr = tid < ARP * ACH;
  const unsigned a_off0 = (unsigned)((m0 + a_row0) * p.OHW + a_ch * VW) * 4u;     // relative to (image, row oy, pixel ox0)
  const int a_lds0 = a_row0 * LSA + a_ch * VW;
  const int b_rr0 = tid / XCH, b_xc = tid - b_rr0 * XCH;
  const bool b_thr = tid < XRP * XCH;
  const int b_c0 = b_rr0 / 3, b_kh = b_rr0 - b_c0 * 3;
  const unsigned b_off0 = (unsigned)(((c0 + b_c0) * p.H + b_kh - 1) * p.W + b_xc * VW - HALO) * 4u;   // relative to (row S oy, column S ox0)
  const int b_lds0 = A_FL + b_c0 * CP + b_kh * RP + b_xc * VW;
  const unsigned a_step = (unsigned)(ARP * p.OHW) * 4u, b_step = (unsigned)((XRP / 3) * p.H * p.W) * 4u;

  float ra[PA * VW], rb[PB * VW];
  auto load_piece = [&](__amdgpu_buffer_rsrc_t r, unsigned voff, float* dst) {
    if (VW == 4)
    {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
      dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y); dst[2] = __uint_as_float(v.z); dst[3] = __uint_as_float(v.w);
    }
    else if (VW == 2)
    {
      typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, 0, 0);
      dst[0] = __uint_as_float(v.x); dst[1] = __uint_as_float(v.y);
    }
    else
      dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
  };
  auto load_stage = [&](int s) {
    const int seg = s % nseg;
    const int t = s / nseg;
    const int oy = t % p.OH, b = t / p.OH;
    const int ox0 = seg * SEG;
    const unsigned abase = (unsigned)(b * p.Mtot * p.OHW + oy * p.OW + ox0) * 4u;
    const unsigned bbase = (unsigned)((b * p.Ctot * p.H + S * oy) * p.W + S * ox0) * 4u;
    // padding and ragged ends: an out-of-range offset, the load returns zeros (the uniform part only moves it further out;
    // everything goes through the VECTOR offset: the bounds check does not see a scalar offset)
    const bool aok = a_thr && ox0 + a_ch * VW < p.OW;
    const bool bok = b_thr && (unsigned)(S * oy + b_kh - 1) < (unsigned)p.H && (unsigned)(S * ox0 + b_xc * VW - HALO) < (unsigned)p.W;
    const unsigned av = aok ? a_off0 : OOB, bv = bok ? b_off0 : OOB;
#pragma unroll
    for (int j = 0; j < PA; ++j)
      load_piece(dr, (((j + 1) * ARP <= 128 || a_row0 + j * ARP < 128) ? av : OOB) + (abase + (unsigned)j * a_step), ra + j * VW);
#pragma unroll
    for (int j = 0; j < PB; ++j)
    {
      // b_off0 is "negative" (mod 2^32) for the row above / the columns left of the segment: only the SUM is an offset.
      // The asm hides the addition from the instruction selector, which would otherwise fold the uniform half into the
      // load's scalar offset -- and the bounds check looks at the vector half alone (measured: every (c0 + 0, kh = 0) row zero)
      unsigned voff = (((j + 1) * XRP <= 96 || b_rr0 + j * XRP < 96) ? bv : OOB) + (bbase + (unsigned)j * b_step);
      asm volatile("" : "+v"(voff));
      load_piece(xr, voff, rb + j * VW);
    }
  };
  auto store_stage = [&](float* stg) {
    if (a_thr)
    {
#pragma unroll
      for (int j = 0; j < PA; ++j)
        if ((j + 1) * ARP <= 128 || a_row0 + j * ARP < 128)
        {
#pragma unroll
          for (int e = 0; e < VW; e += 2)
          {
            if (VW >= 2) *(float2*)(stg + a_lds0 + j * (ARP * LSA) + e) = float2{ra[j * VW + e], ra[j * VW + e + 1]};
            else stg[a_lds0 + j * (ARP * LSA)] = ra[j];
          }
        }
    }
    if (b_thr)
    {
#pragma unroll
      for (int j = 0; j < PB; ++j)
        if ((j + 1) * XRP <= 96 || b_rr0 + j * XRP < 96)
        {
#pragma unroll
          for (int e = 0; e < VW; e += 2)
          {
            if (VW >= 2) *(float2*)(stg + b_lds0 + j * ((XRP / 3) * CP) + e) = float2{rb[j * VW + e], rb[j * VW + e + 1]};
            else stg[b_lds0 + j * ((XRP / 3) * CP)] = rb[j];
          }
        }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (st_begin < st_end)
  {
    load_stage(st_begin);
    store_stage(lds);
    __syncthreads();
    for (int st = st_begin; st < st_end; ++st)
    {
      const float* cur = lds + ((st - st_begin) & 1) * STAGE;
      const bool more = st + 1 < st_end;
      if (more && !(p.abl & 2))
        load_stage(st + 1);
      const float* ap = cur + (wave * 32 + l31) * LSA + 2 * lh;
      const float* bp = cur + A_FL + l31 * CP + S * 2 * lh + HALO - 1;
      // operands of pixel group j: the delta pair and, per input row kh, the 3 + S consecutive columns the three kw taps
      // of the lane's two pixels touch (tap kw of pixel 0 = column kw, of pixel 1 = column kw + S): 7 LDS reads per 18
      // MFMAs.  Group j + 1 is fetched before group j is multiplied and pinned there (sched_barrier): with one or two
      // waves per SIMD nothing else covers the LDS latency -- the compiler's own placement (reads one MFMA ahead of their
      // use) left the loop at 56 % MFMA-busy.
      float2 av[2];
      float bv[2][3][3 + S];
      auto fetch = [&](int j, int buf) {
        av[buf] = *(const float2*)(ap + 4 * j);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int e = 0; e < 3 + S; ++e) bv[buf][kh][e] = bp[kh * RP + S * 4 * j + e];
      };
      fetch(0, 0);
      if (!(p.abl & 4))
#pragma unroll
      for (int j = 0; j < SEG / 4; ++j)
      {
        const int cb = j & 1;
        if (j + 1 < SEG / 4)
          fetch(j + 1, cb ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb].x, bv[cb][t / 3][t % 3], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb].y, bv[cb][t / 3][t % 3 + S], acc[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more && !(p.abl & 2))
        store_stage(lds + ((st + 1 - st_begin) & 1) * STAGE);
      __syncthreads();
    }
  }
  if (p.abl & 1)
    return;

  // ---- epilogue: 16 filters of the wave at a time through LDS, so that a filter's 288 gradients leave consecutively ----
  constexpr int EP = 289;                                 // row pitch of the transposition buffer (host: wgrad3_lds_bytes)
  float* const eb = lds + wave * (16 * EP);
  float* const dst = p.part ? p.part + (size_t)split * p.part_stride : p.dw;
  const bool det = p.part != nullptr;
#pragma unroll
  for (int h = 0; h < 2; ++h)
  {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8)
      {
        const int r = 8 * h + r8;
        eb[((r & 3) + 8 * ((r >> 2) & 1) + 4 * lh) * EP + l31 * 9 + t] = acc[t][r];   // C/D: row (r&3)+8(r>>2)+4 lh, col l31
      }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const size_t mrow = (size_t)(m0 + wave * 32 + 16 * h);
    for (int e = lane; e < 16 * 288; e += 64)
    {
      const int lr = e / 288, col = e - lr * 288;
      const float v = eb[lr * EP + col];
      float* const q = dst + ((mrow + lr) * p.C + c0) * 9 + col;
      if (det)
        *q = v;
      else
        atomicAdd(q, v);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// deterministic mode: dW[i] += sum over the pixel splits, ascending, of their partial tiles (tap-major partials are
// un-permuted on the way, as wgrad_fold_kernel does for the atomic form)
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, size_t stride, float* __restrict__ dw,
    int C, int ss, size_t total, int tmaj)
{
  const size_t K = (size_t)C * ss;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
  {
    size_t src = i;
    if (tmaj)
    {
      const size_t m = i / K, r = i - m * K;
      const int c = (int)(r / ss), t = (int)(r - (size_t)c * ss);
      src = m * K + (size_t)t * C + c;
    }
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * stride + src];
    dw[i] += s;
  }
}

// dW[m][c][t] += ws[m][t][c]; ws is left zeroed for the next layer
__global__ void wgrad_fold_kernel(float* __restrict__ ws, float* __restrict__ dw, int C, int ss, size_t total)
{
  const size_t K = (size_t)C * ss;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
  {
    const size_t m = i / K, r = i - m * K;      // r = c * ss + t in dW order
    const int c = (int)(r / ss), t = (int)(r - (size_t)c * ss);
    const size_t src = m * K + (size_t)t * C + c;
    dw[i] += ws[src];
    ws[src] = 0.f;
  }
}

// per-device workspace of the tap-major path (stream-ordered use: one suffices), grown on demand, zero on creation
namespace
{
struct WsBuf { float* p = nullptr; size_t cap = 0; };
std::mutex g_ws_mu;
int ws_device()
{
  int dev = 0;
  CHECK_HIP(hipGetDevice(&dev));
  return dev;
}
}  // namespace

// (both workspaces are keyed by device AND stream: the weight gradients of replicas that share a device run on
// different streams and must not share a workspace)
float* wgrad_workspace(size_t floats, hipStream_t st)
{
  static std::map<std::pair<int, void*>, WsBuf> bufs;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  WsBuf& b = bufs[{ws_device(), (void*)st}];
  if (b.cap < floats)
  {
    if (b.p)
    {
      CHECK_HIP(hipStreamSynchronize(st));
      CHECK_HIP(hipFree(b.p));
    }
    CHECK_HIP(hipMalloc((void**)&b.p, floats * sizeof(float)));
    CHECK_HIP(hipMemsetAsync(b.p, 0, floats * sizeof(float), st));
    b.cap = floats;
  }
  return b.p;
}

// partial-tile workspace of the deterministic form (grown on demand; contents never assumed)
float* wgrad_part_workspace(size_t floats, hipStream_t st)
{
  static std::map<std::pair<int, void*>, WsBuf> bufs;
  std::lock_guard<std::mutex> lk(g_ws_mu);
  WsBuf& b = bufs[{ws_device(), (void*)st}];
  if (b.cap < floats)
  {
    if (b.p)
    {
      CHECK_HIP(hipStreamSynchronize(st));
      CHECK_HIP(hipFree(b.p));
    }
    CHECK_HIP(hipMalloc((void**)&b.p, floats * sizeof(float)));
    b.cap = floats;
  }
  return b.p;
}

typedef void (*WgradKernel)(const WgradArgs);
struct WgradCfg
{
  int tm, tk;
  WgradKernel kernel[4];  // [AVEC + 2 * TMAJ]
};
#define DK_WG(TMV, TKV)                                                                                     \
  {TMV, TKV, {conv_wgrad_f32<TMV, TKV, false, false>, conv_wgrad_f32<TMV, TKV, true, false>,               \
                 conv_wgrad_f32<TMV, TKV, false, true>, conv_wgrad_f32<TMV, TKV, true, true>}}
const WgradCfg g_wcfg[] = {DK_WG(2, 2), DK_WG(1, 2), DK_WG(2, 1), DK_WG(1, 1)};
#undef DK_WG
}  // namespace

// Deterministic reductions (weight gradients here, BN channel sums in train_ops.hip): partial results go to
// workspaces and are added in a fixed order instead of through float / double atomics, so two runs of a step with
// the same kernel choices are bitwise equal.  Off by default (the weight gradient grows by the partial traffic);
// DK_DETERMINISTIC=1 or dk_set_deterministic(1).
static int g_deterministic = -1;
extern "C" void dk_set_deterministic(int on) { g_deterministic = on; }
bool dk_deterministic()
{
  static const bool env_on = getenv("DK_DETERMINISTIC") && atoi(getenv("DK_DETERMINISTIC"));
  return g_deterministic >= 0 ? g_deterministic != 0 : env_on;
}

// Variant knobs of the weight gradient, settable in-process (tests force every variant; the environment variables
// DK_WGRAD_TILE / DK_WGRAD_TMAJ remain as defaults read once): tile shape 0..3 (-1: by the layer), tap-major tiles
// (-1 / 1: where applicable, 0: never), 16-byte delta loads (-1: where the layout allows, 0: never).
static int g_force_tile = -2, g_force_tmaj = -2, g_force_avec = -2;
extern "C" int dk_train_force(int knob, int value)
{
  int* k = knob == 0 ? &g_force_tile : knob == 1 ? &g_force_tmaj : knob == 2 ? &g_force_avec : nullptr;
  if (!k)
    return -3;
  const int old = *k;
  *k = value < 0 ? -2 : value;
  return old == -2 ? -1 : old;
}
static int wgrad_knob_tile()
{
  static const int env = getenv("DK_WGRAD_TILE") ? atoi(getenv("DK_WGRAD_TILE")) : 0;   // 1..3: OR-ed into the choice
  return g_force_tile != -2 ? g_force_tile : (env > 0 && env < 4 ? -10 - env : -1);
}
static bool wgrad_knob_tmaj()
{
  static const bool env_on = !(getenv("DK_WGRAD_TMAJ") && !atoi(getenv("DK_WGRAD_TMAJ")));
  return g_force_tmaj != -2 ? g_force_tmaj != 0 : env_on;
}

// conv_wgrad3_f32 takes 3x3 / pad 1 layers of stride 1, or 2 on even maps, with one group, channels a multiple of 32,
// filters a multiple of 128
bool dk_wgrad3_applicable(const DkConvDesc* d)
{
  static const bool off = getenv("DK_WGRAD3") && !atoi(getenv("DK_WGRAD3"));   // DK_WGRAD3=0: A/B runs against the gather kernel
  if (off)
    return false;
  const bool s1 = d && d->stride_x == 1 && d->stride_y == 1;
  const bool s2 = d && d->stride_x == 2 && d->stride_y == 2 && d->h % 2 == 0 && d->w % 2 == 0;
  return d && d->size == 3 && (s1 || s2) && d->pad == 1 && d->dilation == 1 && d->groups == 1 &&
         d->c % 32 == 0 && d->n % 128 == 0 && d->w >= 4;
}
namespace
{
template <int SEG, int VW, int S>
constexpr int wgrad3_lds_bytes()
{
  constexpr int XCH = (VW + S * (SEG - 1) + 2 + VW - 1) / VW, XW = XCH * VW, RP = XW + (6 - XW % 4) % 4;
  constexpr int stage2 = 2 * (128 * (SEG + 2) + 32 * 3 * RP), ep = 4 * 16 * 289;
  return (stage2 > ep ? stage2 : ep) * (int)sizeof(float);
}
template <int SEG, int VW, int S>
void wgrad3_launch(const WgradArgs& a, long long nblk, hipStream_t st)
{
  constexpr int lds_bytes = wgrad3_lds_bytes<SEG, VW, S>();
  dk_set_max_dynamic_lds((const void*)conv_wgrad3_f32<SEG, VW, S>, lds_bytes);
  hipLaunchKernelGGL((conv_wgrad3_f32<SEG, VW, S>), dim3((unsigned)nblk), dim3(T), lds_bytes, st, a);
}
}  // namespace

extern "C" int dk_conv_backward_weights(const DkConvDesc* d, const float* x, const float* delta,
    float* weight_updates, void* stream)
{
  return dk_conv_backward_weights_cfg(d, x, delta, weight_updates, stream, -1);
}

int dk_conv_backward_weights_cfg(const DkConvDesc* d, const float* x, const float* delta,
    float* weight_updates, void* stream, int cfg_override)
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
  float* ws = nullptr;   // tap-major workspace, when that path is taken
  for (int b0 = 0; b0 < d->batch; b0 += chunk)
  {
    const int nb = (d->batch - b0 < chunk) ? d->batch - b0 : chunk;
    WgradArgs a;
    a.abl = 0;
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
    const int knob = wgrad_knob_tile();
    // configurations 4 / 5: the row-staged 3x3 kernel with ~2 / ~1 workgroups per CU (4 is the default where the kernel
    // applies; a forced tile 0..3 keeps the gather kernel).  The pixel splits are what the epilogue pays for -- every split
    // adds its whole 128 x 288 tile to dW (ablation, [256->256 38x38] b8: 0.046 of 0.183 ms) -- so fewer, longer workgroups
    // win on the layers with few tiles and lose where the second workgroup per CU was covering latency; first-step timing picks.
    const int want3 = cfg_override >= 0 ? cfg_override : knob;
    const bool pick3 = want3 == 4 || want3 == 5 || (cfg_override < 0 && knob == -1);
    if (pick3 && dk_wgrad3_applicable(d))
    {
      const int S3 = d->stride_x;
      const int seg = (OW <= 20 || S3 == 2) ? 20 : 40;
      const int nseg = (OW + seg - 1) / seg;
      const long long nst = (long long)nb * OH * nseg;
      static const int abl3 = getenv("DK_WGRAD3_ABL") ? atoi(getenv("DK_WGRAD3_ABL")) : 0;
      a.abl = abl3;
      a.tiles_m = M / 128;
      a.tiles_k = C / 32;
      a.groups = 1;
      const long long tiles3 = (long long)a.tiles_m * a.tiles_k;
      static const long long target_env = getenv("DK_WGRAD3_BLOCKS") ? atoll(getenv("DK_WGRAD3_BLOCKS")) : 0;
      const long long target = target_env > 0 ? target_env : want3 == 5 ? 256 : 512;
      long long want = (target + tiles3 - 1) / tiles3;
      if (want < 1) want = 1;
      if (want > nst) want = nst;
      a.stages_per_split = (int)((nst + want - 1) / want);
      a.nsplit = (int)((nst + a.stages_per_split - 1) / a.stages_per_split);
      const long long nblk = tiles3 * a.nsplit;
      const bool det = dk_deterministic();
      a.part = nullptr;
      a.part_stride = 0;
      if (det)
      {
        a.part_stride = (unsigned long long)d->n * K;
        a.part = wgrad_part_workspace((size_t)a.nsplit * a.part_stride, st);
      }
      const bool al16 = (((uintptr_t)a.x | (uintptr_t)a.delta) & 15) == 0, al8 = (((uintptr_t)a.x | (uintptr_t)a.delta) & 7) == 0;
      // (a piece must not straddle a row end of EITHER tensor: the delta rows are OW wide)
      const int vw = (d->w % 4 == 0 && OW % 4 == 0 && al16 && g_force_avec != 0) ? 4 : (d->w % 2 == 0 && OW % 2 == 0 && al8 && g_force_avec != 0) ? 2 : 1;
      DkProfScope prof;
      dk_prof_begin(prof, st);
      if (S3 == 2)
      {
        if (vw == 4) wgrad3_launch<20, 4, 2>(a, nblk, st);
        else if (vw == 2) wgrad3_launch<20, 2, 2>(a, nblk, st);
        else wgrad3_launch<20, 1, 2>(a, nblk, st);
      }
      else if (seg == 20)
        wgrad3_launch<20, 1, 1>(a, nblk, st);
      else if (vw == 4)
        wgrad3_launch<40, 4, 1>(a, nblk, st);
      else if (vw == 2)
        wgrad3_launch<40, 2, 1>(a, nblk, st);
      else
        wgrad3_launch<40, 1, 1>(a, nblk, st);
      CHECK_HIP(hipPeekAtLastError());
      if (prof.e0)
      {
        char nm[96];
        snprintf(nm, sizeof(nm), "conv_wgrad3_f32<%d, %d, %d>", seg, (seg == 20 && S3 == 1) ? 1 : vw, S3);
        dk_prof_end(prof, st, dk_prof_named_slot(nm), 2.0 * (double)M * K * (double)a.N / 1e9);
      }
      if (det)
      {
        const size_t total = (size_t)d->n * K;
        unsigned gb = (unsigned)((total + 255) / 256);
        if (gb > 8192u) gb = 8192u;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gb), dim3(256), 0, st, a.part, a.nsplit, (size_t)a.part_stride,
            weight_updates, C, 9, total, 0);
        CHECK_HIP(hipPeekAtLastError());
      }
      continue;
    }
    int ci = (M > 64 ? 0 : 1) + (K > 64 ? 0 : 2);  // 128/64 rows x 128/64 taps
    if (knob <= -11)
      ci |= -10 - knob;   // DK_WGRAD_TILE=1/2/3: 64x128 / 128x64 / 64x64 tiles (tuning experiments)
    // 3x3 layers with 64 (not 128) channels per tap: 64-tap tiles keep the tap-major path available
    if (d->size > 1 && (C % 128) != 0 && (C % 64) == 0)
      ci |= 2;
    if (cfg_override >= 0 && cfg_override < 4)
      ci = cfg_override;
    else if (knob >= 0 && knob < 4 && cfg_override < 0)
      ci = knob;
    const WgradCfg& c = g_wcfg[ci];
    const int BM = 64 * c.tm, BKO = 64 * c.tk;
    a.tiles_m = (M + BM - 1) / BM;
    a.tiles_k = (K + BKO - 1) / BKO;
    a.groups = d->groups;
    const int nstages = (a.N + NC - 1) / NC;
    const long long tiles = (long long)a.tiles_m * a.tiles_k * d->groups;
    static const long long target3 = getenv("DK_WGRAD_BLOCKS") ? atoll(getenv("DK_WGRAD_BLOCKS")) : 1024;
    static const long long target1 = getenv("DK_WGRAD_BLOCKS_1X1") ? atoll(getenv("DK_WGRAD_BLOCKS_1X1")) : 1024;
    const long long target = d->size == 1 ? target1 : target3;
    long long want = (target + tiles - 1) / tiles;  // ~4 workgroups per CU in total
    if (want < 1) want = 1;
    if (want > nstages) want = nstages;
    a.stages_per_split = (int)((nstages + want - 1) / want);
    if (a.stages_per_split < 8 && nstages >= 8)
      a.stages_per_split = 8;
    a.nsplit = (nstages + a.stages_per_split - 1) / a.stages_per_split;
    const long long nblk = tiles * a.nsplit;
    const bool avec = (a.OHW % 4 == 0) && (((uintptr_t)a.delta & 15) == 0) && g_force_avec != 0;
    // tap-major tiles when a tile never straddles two taps (C a multiple of the tile width) and there is more than one tap
    const bool tmaj_on = wgrad_knob_tmaj();
    // (a 1x1 layer is its own tap-major order: same fast gather, no workspace)
    const bool tmaj = tmaj_on && (C % BKO) == 0 && d->pad * d->dilation == (d->size > 1 ? d->pad * d->dilation : 0);
    const int kv = (avec ? 1 : 0) + (tmaj ? 2 : 0);
    const bool det = dk_deterministic();
    a.part = nullptr;
    a.part_stride = 0;
    if (det)
    {
      a.part_stride = (unsigned long long)d->n * K;
      a.part = wgrad_part_workspace((size_t)a.nsplit * a.part_stride, st);
    }
    else if (tmaj && d->size > 1)
    {
      if (!ws)
        ws = wgrad_workspace((size_t)d->n * K, st);
      a.dw = ws;
    }
    const int lds_bytes = 2 * (BM + BKO) * LS * (int)sizeof(float);
    dk_set_max_dynamic_lds((const void*)c.kernel[kv], lds_bytes);
    DkProfScope prof;
    dk_prof_begin(prof, st);
    hipLaunchKernelGGL(c.kernel[kv], dim3((unsigned)nblk), dim3(T), lds_bytes, st, a);
    CHECK_HIP(hipPeekAtLastError());
    if (prof.e0)
    {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv_wgrad_f32<%d, %d, %s, %s>", c.tm, c.tk, avec ? "true" : "false", tmaj ? "true" : "false");
      dk_prof_end(prof, st, dk_prof_named_slot(nm), 2.0 * (double)M * K * d->groups * (double)a.N / 1e9);
    }
    if (det)
    {
      const size_t total = (size_t)d->n * K;
      unsigned gb = (unsigned)((total + 255) / 256);
      if (gb > 8192u) gb = 8192u;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(gb), dim3(256), 0, st, a.part, a.nsplit, (size_t)a.part_stride,
          weight_updates, C, d->size * d->size, total, (tmaj && d->size > 1) ? 1 : 0);
      CHECK_HIP(hipPeekAtLastError());
    }
  }
  if (ws)
  {
    const size_t total = (size_t)d->n * K;
    unsigned gb = (unsigned)((total + 255) / 256);
    if (gb > 4096u) gb = 4096u;
    hipLaunchKernelGGL(wgrad_fold_kernel, dim3(gb), dim3(256), 0, st, ws, weight_updates, C, d->size * d->size, total);
    CHECK_HIP(hipPeekAtLastError());
  }
  return 0;
}
