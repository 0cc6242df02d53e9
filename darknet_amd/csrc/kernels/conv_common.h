// conv_common.h -- argument block and device helpers shared by the implicit-GEMM
// convolution kernels (fp32: conv_igemm.hip, fp16 operands: conv_igemm_f16.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dk_device_math.h"
#include "dk_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs
{
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  const float* residual;
  float* act_in;
  const int2* ktab;  // [Kpad] {element offset of tap k inside one image-group, tap bit index (31 = padding)}
  unsigned x_bytes;  // buffer sizes for the hardware bounds check
  unsigned w_bytes;
  unsigned y_bytes;
  int C, H, W;       // channels per group, input height/width
  int Ctot;          // total input channels
  int M, Mtot;       // filters per group / total
  int K;             // C*size*size
  int OH, OW, OHW;
  int N;             // batch*OHW
  int size, stride_x, stride_y, pad, dil;  // pad = l->pad*dilation
  int act;
  int tiles_m, tiles_n, groups;
  int mode;          // 0 forward gather; 1 data-gradient gather (x = delta, H/W = delta dims, OH/OW = input dims);
                     // 2 data gradient of a stride-2 layer by parity class (see conv_par_pixel)
  // mode 2: the pixel index is class-major, n = cls * par_ncls + r with cls = 2*(oy&1) + (ox&1) and r < par_count
  // running over (image, oy/2, ox/2); par_ncls = par_count rounded up to the tile width, N = 4 * par_ncls
  int par_ncls, par_count, par_hw2, par_w2;
  double inv_par_ncls, inv_par_hw2, inv_par_w2;
  // reciprocals of the divisors the kernels' index arithmetic uses (set by conv_args_finish):
  // an integer division costs ~40 VALU instructions, fdiv() four
  double inv_OHW, inv_OW, inv_tiles_m, inv_per_group, inv_HW, inv_W, inv_He;
  // XCD partition (conv_pick_partition): pm x (8/pm) grid of XCDs over (M tiles x N tiles);
  // pm = 1 is the N-major default (xcd_remap)
  int pm, tm_per, tn_per;
  double inv_tm_per;
  // dual output (two convolutions of the same input in one launch): filters >= m_split go to y2
  // (a tensor with Mtot2 channels); m_split is a multiple of the tile's BM, 0 = single output
  float* y2;
  unsigned y2_bytes;
  int Mtot2, m_split;
  int nwork;         // persistent kernels: number of virtual workgroups (the grid a one-tile-per-block launch would use)
  // Winograd F(2x2,3x3) (conv3x3_wino.hip): N counts 2x2 output tiles, tiles_w per row, tiles_hw per image
  int tiles_w, tiles_hw;
  double inv_tiles_w, inv_tiles_hw;
  // ... raw-patch geometry: tile rows per image, column groups per staged row, rows per channel, loads per thread
  int wino_th, wino_gp, wino_rs, wino_nk, wino_ring, wino_quad;
  double inv_wino_th, inv_wino_gp, inv_wino_rsg;
};

// exact floor(n / d) for 0 <= n < 2^31, d > 0, given inv = 1.0 / d: the double estimate is within
// one of the quotient, one correction step makes it exact
__host__ __device__ __forceinline__ int fdiv(int n, int d, double inv)
{
  int q = (int)(((double)n + 0.5) * inv);
  const int r = n - q * d;
  q += (r >= d) - (r < 0);
  return q;
}

// Bijective XCD remap (8 XCDs, blocks dealt round-robin): block `bid` of `nwg`
// gets a logical id such that ids handled by one XCD are contiguous.
__host__ __device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Which tile does this workgroup compute?  Default: logical ids dealt so that one XCD gets
// a contiguous N range with all its M tiles (the M tiles of a pixel tile share the input in
// that XCD's L2).  When the weights do not fit an L2 (4 MB) that order re-streams them for
// every pixel tile, so the host switches to a pm x (8/pm) partition: each XCD owns 1/pm of
// the filters (a slice that fits its L2) and 1/(8/pm) of the pixels.  Returns false for the
// surplus workgroups of a partition that does not divide evenly.
// (bid, nblk) = (blockIdx.x, gridDim.x); host-callable so that the mapping is unit-tested on the CPU
__host__ __device__ __forceinline__ bool conv_block_tile_of(const ConvArgs& p, int bid, int nblk, int& g,
    int& tile_m, int& tile_n)
{
  if (p.pm > 1)
  {
    const int xcd = bid & 7, li = bid >> 3;
    const int xm = xcd & (p.pm - 1), xn = xcd / p.pm;
    const int tnl = fdiv(li, p.tm_per, p.inv_tm_per);
    const int tml = li - tnl * p.tm_per;
    g = 0;
    tile_m = xm * p.tm_per + tml;
    tile_n = xn * p.tn_per + tnl;
    return tile_m < p.tiles_m && tile_n < p.tiles_n;
  }
  const int per_group = p.tiles_m * p.tiles_n;
  int id = xcd_remap(bid, nblk);
  g = fdiv(id, per_group, p.inv_per_group);
  id -= g * per_group;
  tile_n = fdiv(id, p.tiles_m, p.inv_tiles_m);
  tile_m = id - tile_n * p.tiles_m;
  return true;
}

__device__ __forceinline__ bool conv_block_tile(const ConvArgs& p, int& g, int& tile_m, int& tile_n)
{
  return conv_block_tile_of(p, (int)blockIdx.x, (int)gridDim.x, g, tile_m, tile_n);
}

// Host: choose the XCD partition for a launch whose tiles_m / tiles_n are set; returns the grid size.
inline long long conv_pick_partition(ConvArgs& a, size_t weight_bytes, int tile_rows)
{
  a.pm = 1;
  a.tm_per = a.tiles_m;
  a.tn_per = a.tiles_n;
  const size_t l2_budget = (size_t)3 << 20;  // of the 4 MB per XCD, leave room for the input stream
#ifdef DK_NO_PARTITION
  if (false)
#else
  if (a.groups == 1 && weight_bytes > l2_budget)
#endif
  {
    int pm = 1;
    while (pm < 8 && pm * 2 <= a.tiles_m && weight_bytes / pm > l2_budget) pm *= 2;
    if (pm > 1)
    {
      a.pm = pm;
      a.tm_per = (a.tiles_m + pm - 1) / pm;
      a.tn_per = (a.tiles_n + (8 / pm) - 1) / (8 / pm);
    }
  }
  (void)tile_rows;
  a.inv_tm_per = 1.0 / (a.tm_per > 0 ? a.tm_per : 1);
  return a.pm > 1 ? 8LL * a.tm_per * a.tn_per : (long long)a.tiles_m * a.tiles_n * a.groups;
}

inline void conv_args_finish(ConvArgs& a)
{
  a.inv_OHW = 1.0 / (a.OHW > 0 ? a.OHW : 1);
  a.inv_OW = 1.0 / (a.OW > 0 ? a.OW : 1);
  a.inv_tiles_m = 1.0 / (a.tiles_m > 0 ? a.tiles_m : 1);
  const long long pg = (long long)a.tiles_m * a.tiles_n;
  a.inv_per_group = 1.0 / (pg > 0 ? (double)pg : 1.0);
  a.inv_HW = 1.0 / ((double)a.H * a.W > 0 ? (double)a.H * a.W : 1.0);
  a.inv_W = 1.0 / (a.W > 0 ? a.W : 1);
  a.inv_He = 1.0 / (a.H + 2);
  a.inv_par_ncls = 1.0 / (a.par_ncls > 0 ? a.par_ncls : 1);
  a.inv_par_hw2 = 1.0 / (a.par_hw2 > 0 ? a.par_hw2 : 1);
  a.inv_par_w2 = 1.0 / (a.par_w2 > 0 ? a.par_w2 : 1);
}

__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_buf4(__amdgpu_buffer_rsrc_t r, unsigned byte_off)
{
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
      __uint_as_float(v.w));
}

constexpr unsigned OOB = 0x80000000u;  // ORed into a byte offset: always outside the buffer

// mode 2 (data gradient of a stride-2 convolution): an input pixel only receives the taps whose parity matches
// its own, so the pixels are enumerated parity class by parity class -- every pixel tile then belongs to ONE
// class, its valid taps are the same for the whole workgroup, and (with the contraction index ordered tap-major)
// the K tiles of the other taps are skipped instead of being multiplied by gathered zeros (9 of 36 tap-pixel
// pairs do work for a 3x3 / stride-2 layer).  Returns false for the padding slots at the end of a class.
__device__ __forceinline__ bool conv_par_pixel(const ConvArgs& p, int n, int& b, int& oy, int& ox)
{
  const int cls = fdiv(n, p.par_ncls, p.inv_par_ncls);
  const int r = n - cls * p.par_ncls;
  const bool ok = r < p.par_count;
  const int rr = ok ? r : 0;
  b = fdiv(rr, p.par_hw2, p.inv_par_hw2);
  const int q = rr - b * p.par_hw2;
  const int j = fdiv(q, p.par_w2, p.inv_par_w2);
  const int i = q - j * p.par_w2;
  oy = 2 * j + (cls >> 1);
  ox = 2 * i + (cls & 1);
  return ok;
}

// --------------------------------------------------------------------------
// Shared epilogue of the implicit-GEMM kernels: bias + activation (+ residual,
// + pre-activation store) of the wave's TM x TN MFMA tiles, stored through
// buffer descriptors.  C/D layout of the 32x32 MFMAs: col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Output (and residual / pre-activation)
// are addressed with 32-bit byte offsets (the host keeps one launch's output < 4 GiB).
// --------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, f32x16 (&acc)[TM][TN], int m0,
    int n0, int g, int wm, int wn, int l31, int lh)
{
  // dual-output launches: the M tiles at or above m_split write the second tensor
  const bool second = p.m_split > 0 && m0 >= p.m_split;
  float* const ybuf = second ? p.y2 : p.y;
  const unsigned ybytes = second ? p.y2_bytes : p.y_bytes;
  const int mtot = second ? p.Mtot2 : p.Mtot;
  const int msub = second ? p.m_split : 0;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)ybuf, 0, ybytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? p.y_bytes : 0u, 0x00020000);
  __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc((void*)p.act_in, 0, p.act_in ? p.y_bytes : 0u, 0x00020000);
  const bool has_res = p.residual != nullptr, has_ain = p.act_in != nullptr;
  const int act = p.act;
  bool full_tile = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (p.mode == 2)   // class-major pixels: the tail of a class is padding
    full_tile = full_tile && (n0 - fdiv(n0, p.par_ncls, p.inv_par_ncls) * p.par_ncls + BN <= p.par_count);
  unsigned obase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
  {
    const int n = n0 + wn * WN + j * 32 + l31;
    bool nv = n < p.N;
    const int nn = nv ? n : 0;
    int b, pix;
    if (p.mode == 2)
    {
      int oy, ox;
      nv = conv_par_pixel(p, nn, b, oy, ox) && nv;
      pix = oy * p.OW + ox;
    }
    else
    {
      b = fdiv(nn, p.OHW, p.inv_OHW);
      pix = nn - b * p.OHW;
    }
    obase[j] = nv ? (unsigned)((b * mtot + g * p.M) * p.OHW + pix) * 4u : 0xFFFFFFFFu;
  }
  const unsigned row_bytes = (unsigned)p.OHW * 4u;
  // generic per-element path: edge tiles, exact mish / logistic / relu, pre-activation store
  auto emit_generic = [&](auto check) {
    constexpr bool CHECK = decltype(check)::value;
    // (no `continue` inside the unrolled loops: with early exits the optimizer gives up on the
    // full unroll of the larger tiles, acc is then indexed dynamically and the whole accumulator
    // array -- the K loop's too -- ends up in scratch memory)
#pragma unroll
    for (int i = 0; i < TM; ++i)
    {
#pragma unroll
      for (int r = 0; r < 16; ++r)
      {
        const int m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool mok = !CHECK || m < p.M;
        const float bv = (p.bias && mok) ? p.bias[g * p.M + m] : 0.f;
        const unsigned mo = (unsigned)(m - msub) * row_bytes;
#pragma unroll
        for (int j = 0; j < TN; ++j)
        {
          const bool ok = mok && !(CHECK && obase[j] == 0xFFFFFFFFu);
          float v = acc[i][j][r] + bv;
          // masked elements are sent beyond the end of the buffers: the range check drops them
          const unsigned o = ok ? obase[j] + mo : 0xFFFFFFF0u;
          if (has_ain)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), ar, (int)o, 0, 0);
          v = dk_activate(v, act);
          if (has_res)
            v += ld_buf(rr, o);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), yr, (int)o, 0, 0);
        }
      }
    }
  };
  // full tiles, the three activations the networks use: straight-line code.  The
  // activation and the residual flag are launch constants, dispatched once.  Rows
  // are handled four at a time (one 8-row group of the C/D layout): bias and
  // residual values of the group are fetched first, so the loads overlap instead
  // of being serialised behind the stores; the row part of every address is a
  // scalar offset (soffset), the per-lane part is loop invariant.
  auto emit_fast = [&](auto actc, auto resc) {
    constexpr int ACT = decltype(actc)::value;
    constexpr bool RES = decltype(resc)::value;
    const int mlane = m0 + wm * WM + 4 * lh;
    __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.bias ? p.bias + g * p.M : p.w), 0, p.bias ? (unsigned)p.M * 4u : 0u, 0x00020000);
    unsigned vo[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) vo[j] = obase[j] + (unsigned)(mlane - msub) * row_bytes;
#pragma unroll
    for (int i = 0; i < TM; ++i)
    {
#pragma unroll
      for (int q = 0; q < 4; ++q)
      {
        float bv[4], res[4][TN];
#pragma unroll
        for (int t = 0; t < 4; ++t)
          bv[t] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(br, mlane * 4, (i * 32 + 8 * q + t) * 4, 0));
        if (RES)
        {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              res[t][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                  rr, (int)vo[j], (i * 32 + 8 * q + t) * (int)row_bytes, 0));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int j = 0; j < TN; ++j)
          {
            float v = acc[i][j][4 * q + t] + bv[t];
            if (ACT == DK_MISH)
              v = dk_mish_fast(v);
            else if (ACT == DK_LEAKY)
              v = dk_leaky(v);
            if (RES)
              v += res[t][j];
            __builtin_amdgcn_raw_buffer_store_b32(
                __float_as_uint(v), yr, (int)vo[j], (i * 32 + 8 * q + t) * (int)row_bytes, 0);
          }
      }
    }
  };
  using std::false_type;
  using std::true_type;
  using std::integral_constant;
  if (full_tile && !has_ain)
  {
    if (act == (DK_MISH | DK_ACT_FAST))
    {
      if (has_res) emit_fast(integral_constant<int, DK_MISH>{}, true_type{});
      else emit_fast(integral_constant<int, DK_MISH>{}, false_type{});
    }
    else if (act == DK_LEAKY)
    {
      if (has_res) emit_fast(integral_constant<int, DK_LEAKY>{}, true_type{});
      else emit_fast(integral_constant<int, DK_LEAKY>{}, false_type{});
    }
    else if (act == DK_LINEAR)
    {
      if (has_res) emit_fast(integral_constant<int, DK_LINEAR>{}, true_type{});
      else emit_fast(integral_constant<int, DK_LINEAR>{}, false_type{});
    }
    else
      emit_generic(false_type{});
  }
  else
    emit_generic(true_type{});
}


// host side (conv_igemm.hip): per-shape tap table, created on first use
const int2* dk_conv_ktab(const DkConvDesc* d, int K, int C, int mode);
