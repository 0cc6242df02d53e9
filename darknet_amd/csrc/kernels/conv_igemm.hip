// conv_igemm.hip -- forward convolution as an implicit GEMM on gfx950 fp32 MFMA.
//
// Replaces the reference's no-cuDNN GPU branch of ForwardConvolutionalLayerGpu
// (src/convolutional_kernels.cu:471-532): fill_ongpu, then per image
// im2col_gpu_ext (src/im2col_kernels.cu:1830-1884) + cublasSgemm
// (src/gemm.c:3011-3022), then add_bias_gpu and the activation kernel.  Here it
// is one launch:
//
//   Out[m][n] = act( bias[m] + sum_k W[m][k] * X[k][n] ) (+ residual)
//   m = filter, n = (image b, oy, ox) -- the batch is folded into N,
//   k = (c, kh, kw) in im2col order (src/im2col.c:56-104).
//
// Design (MI355X / CDNA4):
//  * v_mfma_f32_32x32x2_f32: exact fp32, bitwise a k-ordered fmaf chain, so each
//    output element accumulates its K products in ascending k exactly like the
//    reference's gemm_nn (src/gemm.c:2223-2239), with FMA instead of mul+add.
//  * A = weights tile [BM][BK] and B = gathered input tile [BK][BN] are staged
//    in LDS (double buffered); the im2col matrix is never materialised: each
//    thread owns one output pixel column of the B tile, decomposes it into
//    (b, oy, ox) once, and per k adds a wave-uniform offset from a per-layer
//    table (scalar loads) to its base address.  Padding is a per-thread bitmask
//    over the kernel taps; masked loads use buffer_load's bounds check (offset
//    forced out of range -> 0) so there is no divergent branch.
//  * B rows are contiguous in n -> conflict-free ds_write_b32 / ds_read_b32;
//    A rows are padded to BK+1 floats so the 32 lanes of an MFMA A-fragment
//    read hit 32 different banks.
//  * One barrier per K tile; global loads for tile t+1 are issued before the
//    MFMAs of tile t and written to the other LDS buffer after them.
//  * 1-D grid with a bijective XCD-aware remap so that the M-tiles sharing one
//    input tile run on the same XCD (shared L2).
//  * Epilogue fused: bias, LEAKY / MISH / LOGISTIC / RELU / LINEAR, optional
//    residual add (shortcut fusion) and optional pre-activation store.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <string>
#include <type_traits>
#include <mutex>
#include <vector>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "dk_device_math.h"
#include "dk_internal.h"
#include "conv_common.h"


// Diagnostic builds (make EXTRA=-DDK_ABL=<bits> LIB=... OUT=...; never the shipped library): remove one
// part of the kernel to price it (cdna_hip_programming.md section 7, "Ablate").  Results are wrong by
// construction.  bit 0: no epilogue, 1: no LDS staging writes, 2: no global loads, 3: no barriers,
// 4: no LDS fragment reads.
#ifndef DK_ABL
#define DK_ABL 0
#endif
#define DK_BARRIER()      \
  do                      \
  {                       \
    if (!(DK_ABL & 8))    \
      __syncthreads();    \
  } while (0)

// Diagnostic build -DDK_GSTAMP=1 (tools/build_stamp.sh; never the shipped library): lane 0 of every wave of the first
// 512 workgroups stamps s_memtime at the phase boundaries into g_gstamp (read back with dk_gather_stamps_read).
#ifndef DK_GSTAMP
#define DK_GSTAMP 0
#endif
#if DK_GSTAMP
__device__ long long g_gstamp[512 * 16 * 8];
#define GSTAMP(idx)                                                                                     \
  do {                                                                                                  \
    if (blockIdx.x < 512 && (threadIdx.x & 63) == 0)                                                    \
      g_gstamp[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (idx)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define GSTAMP(idx)
#endif

// AVEC: weights rows are read as float4 (needs K % 4 == 0).
// BVEC: 1x1 / stride 1 / pad 0 / OHW % 4 == 0: the B tile is a plain strided
//       matrix, read as float4 along n and written to LDS with ds_write_b128.
// PF: prefetch distance in K tiles (1: the next tile's loads are in flight during the MFMAs;
//     2: two tiles ahead -- small tiles whose K-tile time is far below a load's latency).
template <int BM, int BN, int BK, int WM, int WN, bool AVEC, bool BVEC, int PF = 1>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
    conv_igemm_f32(const ConvArgs p)
{
  constexpr int NWN = BN / WN;
  constexpr int NW = (BM / WM) * NWN;
  constexpr int T = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int AS = BK + 1;                 // padded A row stride (floats)
  constexpr int A_FLOATS = BM * AS;
  constexpr int B_FLOATS = BK * BN;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  // generic B gather: thread = one pixel column, PB consecutive taps
  constexpr int B_GROUPS = T / BN;           // wave-uniform row groups
  constexpr int PB = BK / B_GROUPS;          // taps per thread
  // vector B (1x1): thread = 4 pixels of one row
  constexpr int NQ = BN / 4;
  constexpr int BV_ROWS = T / NQ;
  constexpr int PBV = BK / BV_ROWS;
  // A: scalar (one k per thread) or float4 (4 k per thread)
  constexpr int A_ROWS = T / BK;
  constexpr int PA = (BM + A_ROWS - 1) / A_ROWS;
  constexpr int KQ = BK / 4;
  constexpr int AV_ROWS = T / KQ;
  constexpr int PAV = (BM + AV_ROWS - 1) / AV_ROWS;
  static_assert(T % BN == 0 && BN % 64 == 0 && BK % B_GROUPS == 0, "B gather mapping");
  static_assert(T % BK == 0 && BK % BV_ROWS == 0 && BK % 8 == 0, "tile mapping");
  static_assert((A_FLOATS % 4) == 0 && (STAGE % 4) == 0, "LDS alignment for ds_write_b128");

  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- which tile -------------------------------------------------------
  int g, tile_m, tile_n;
  if (!conv_block_tile(p, g, tile_m, tile_n))
    return;
  GSTAMP(0);
  const int m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int HW = p.H * p.W;
  const int K = p.K;
  const float* wg = p.w + (size_t)g * p.M * K;
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, p.w_bytes, 0x00020000);

  // ---- per-thread B addressing ----------------------------------------------
  const int bn_l = tid % BN;
  const int bk_g = __builtin_amdgcn_readfirstlane(tid / BN);  // wave-uniform (BN % 64 == 0)
  unsigned xbase4 = 0;           // byte offset of (b, group g, iy0, ix0); may wrap "negative"
  unsigned nmask = 0xFFFFFFFFu;  // bit t set <=> tap t is OUTSIDE the image (bit 31: table padding)
  const int bq = tid % NQ;
  const int bv_r = tid / NQ;
  unsigned bv_base = OOB;        // BVEC: byte offset of (b, group g, channel 0, pix) or OOB
  if (BVEC)
  {
    const int n = n0 + bq * 4;
    if (n < p.N)
    {
      const int b = fdiv(n, p.OHW, p.inv_OHW);
      const int pix = n - b * p.OHW;
      bv_base = (unsigned)((b * p.Ctot + g * p.C) * HW + pix) * 4u;
    }
  }
  else
  {
    const int n = n0 + bn_l;
    bool nv = n < p.N;
    const int nn = nv ? n : 0;
    int b, oy, ox;
    if (p.mode == 2)
      nv = conv_par_pixel(p, nn, b, oy, ox) && nv;
    else
    {
      b = fdiv(nn, p.OHW, p.inv_OHW);
      const int pix = nn - b * p.OHW;
      oy = fdiv(pix, p.OW, p.inv_OW);
      ox = pix - oy * p.OW;
    }
    if (p.mode == 0)
    {
      const int iy0 = oy * p.stride_y - p.pad;
      const int ix0 = ox * p.stride_x - p.pad;
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
    else
    {
      // data gradient: this thread's pixel is an INPUT pixel (iy, ix) = (oy, ox) here;
      // tap (kh, kw) contributes delta[(iy + pad - kh*dil)/stride] when divisible and
      // in range.  With q = iy + pad = s*a + r and kh*dil = s*u + v: valid iff v == r,
      // and then the delta row is a - u -> offset = base(a) + table(tap).
      const int qy = oy + p.pad, qx = ox + p.pad;
      const int ay = qy / p.stride_y, ry = qy - ay * p.stride_y;
      const int ax = qx / p.stride_x, rx = qx - ax * p.stride_x;
      xbase4 = (unsigned)((b * p.Ctot + g * p.C) * HW + ay * p.W + ax) * 4u;
      if (nv)
      {
        unsigned ok_bits = 0;
        for (int kh = 0; kh < p.size; ++kh)
          for (int kw = 0; kw < p.size; ++kw)
          {
            const int uy = (kh * p.dil) / p.stride_y, vy = kh * p.dil - uy * p.stride_y;
            const int ux = (kw * p.dil) / p.stride_x, vx = kw * p.dil - ux * p.stride_x;
            const bool ok = vy == ry && vx == rx && (unsigned)(ay - uy) < (unsigned)p.H &&
                            (unsigned)(ax - ux) < (unsigned)p.W;
            ok_bits |= (ok ? 1u : 0u) << (kh * p.size + kw);
          }
        nmask = ~ok_bits;
      }
    }
  }

  // ---- per-thread A addressing (loop invariant row offsets) -------------------
  const int ak_l = tid % BK;
  const int am_r = tid / BK;
  const int aq = tid % KQ;
  const int av_r = tid / KQ;
  unsigned aoff[AVEC ? PAV : PA];
  if (AVEC)
  {
#pragma unroll
    for (int j = 0; j < PAV; ++j)
    {
      const int ml = av_r + j * AV_ROWS;
      const int m = m0 + ml;
      aoff[j] = (ml < BM && m < p.M) ? (unsigned)(m * K + aq * 4) * 4u : OOB;
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < PA; ++j)
    {
      const int ml = am_r + j * A_ROWS;
      const int m = m0 + ml;
      aoff[j] = (ml < BM && m < p.M) ? (unsigned)(m * K + ak_l) * 4u : OOB;
    }
  }

  constexpr int NA = AVEC ? PAV * 4 : PA, NB = BVEC ? PBV * 4 : PB;
  float ra2[PF][NA];
  float rb2[PF][NB];
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, PF - 1>;

  auto load_tile = [&](int k0, auto sc) {
    float(&ra)[NA] = ra2[decltype(sc)::value];
    float(&rb)[NB] = rb2[decltype(sc)::value];
    if (DK_ABL & 4)
    {
#pragma unroll
      for (int j = 0; j < NA; ++j) ra[j] = (float)(k0 + j);
#pragma unroll
      for (int j = 0; j < NB; ++j) rb[j] = (float)(k0 - j);
      return;
    }
    // ---- A (weights [M][K] row-major)
    if (AVEC)
    {
      const unsigned kinv = (k0 + aq * 4 < K) ? 0u : OOB;
#pragma unroll
      for (int j = 0; j < PAV; ++j)
      {
        const float4 v = ld_buf4(wr, (aoff[j] + (unsigned)k0 * 4u) | kinv);
        ra[4 * j + 0] = v.x; ra[4 * j + 1] = v.y; ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w;
      }
    }
    else
    {
      const unsigned kinv = (k0 + ak_l < K) ? 0u : OOB;
#pragma unroll
      for (int j = 0; j < PA; ++j) ra[j] = ld_buf(wr, (aoff[j] + (unsigned)k0 * 4u) | kinv);
    }
    // ---- B
    if (BVEC)
    {
#pragma unroll
      for (int j = 0; j < PBV; ++j)
      {
        const int k = k0 + bv_r + j * BV_ROWS;
        const unsigned kinv = (k < K) ? 0u : OOB;
        const float4 v = ld_buf4(xr, (bv_base + (unsigned)(k * HW) * 4u) | kinv);
        rb[4 * j + 0] = v.x; rb[4 * j + 1] = v.y; rb[4 * j + 2] = v.z; rb[4 * j + 3] = v.w;
      }
    }
    else
    {
      // PB consecutive table entries, wave-uniform address -> one wide scalar load
      const int2* kp = p.ktab + (k0 + bk_g * PB);
      int2 kt[PB];
#pragma unroll
      for (int j = 0; j < PB; ++j) kt[j] = kp[j];
#pragma unroll
      for (int j = 0; j < PB; ++j)
      {
        // kt.y = 31 - tap: shifts the tap's "outside" bit into the sign position
        const unsigned inv = (nmask << kt[j].y) & OOB;
        rb[j] = ld_buf(xr, (xbase4 + (unsigned)kt[j].x) | inv);
      }
    }
  };

  auto store_tile = [&](float* st, auto sc) {
    float(&ra)[NA] = ra2[decltype(sc)::value];
    float(&rb)[NB] = rb2[decltype(sc)::value];
    if (DK_ABL & 2)
    {
#pragma unroll
      for (int j = 0; j < NA; ++j) asm volatile("" ::"v"(ra[j]));
#pragma unroll
      for (int j = 0; j < NB; ++j) asm volatile("" ::"v"(rb[j]));
      return;
    }
    float* As = st;
    float* Bs = st + A_FLOATS;
    if (AVEC)
    {
#pragma unroll
      for (int j = 0; j < PAV; ++j)
      {
        const int ml = av_r + j * AV_ROWS;
        if (PAV * AV_ROWS == BM || ml < BM)
        {
#pragma unroll
          for (int i = 0; i < 4; ++i) As[ml * AS + aq * 4 + i] = ra[4 * j + i];
        }
      }
    }
    else
    {
#pragma unroll
      for (int j = 0; j < PA; ++j)
      {
        const int ml = am_r + j * A_ROWS;
        if (PA * A_ROWS == BM || ml < BM)
          As[ml * AS + ak_l] = ra[j];
      }
    }
    if (BVEC)
    {
#pragma unroll
      for (int j = 0; j < PBV; ++j)
        *(float4*)&Bs[(bv_r + j * BV_ROWS) * BN + bq * 4] =
            make_float4(rb[4 * j + 0], rb[4 * j + 1], rb[4 * j + 2], rb[4 * j + 3]);
    }
    else
    {
#pragma unroll
      for (int j = 0; j < PB; ++j) Bs[(bk_g * PB + j) * BN + bn_l] = rb[j];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int nkt = (K + BK - 1) / BK;
  const int l31 = lane & 31, lh = lane >> 5;

  // mode 2: only the K tiles of this class's taps are visited.  The contraction index is tap-major there
  // (k = t * C + m, C = filters of the layer, a multiple of BK), so a K tile lies inside one tap; par_taps
  // packs the taps whose parity matches the class (4 bits each), kv_* walk them in launch order.
  unsigned long long par_taps = 0;
  int par_tpt = 1, kv_tap = 0, kv_w = 0;
  if (p.mode == 2)
  {
    const int cls = fdiv(n0, p.par_ncls, p.inv_par_ncls);
    const int ry = ((cls >> 1) + p.pad) & 1, rx = ((cls & 1) + p.pad) & 1;
    int nvt = 0;
    for (int kh = 0; kh < p.size; ++kh)
      for (int kw = 0; kw < p.size; ++kw)
        if (((kh * p.dil) & 1) == ry && ((kw * p.dil) & 1) == rx)
          par_taps |= (unsigned long long)(kh * p.size + kw) << (4 * nvt++);
    par_tpt = p.C / BK;
    nkt = nvt * par_tpt;
  }
  // k0 of the kt-th K tile; tiles are requested in ascending order, each exactly once
  auto k0_of = [&](int kt) -> int {
    if (p.mode != 2)
      return kt * BK;
    const int t = (int)((par_taps >> (4 * kv_tap)) & 15ull);
    const int k0 = t * p.C + kv_w * BK;
    if (++kv_w == par_tpt)
    {
      kv_w = 0;
      ++kv_tap;
    }
    return k0;
  };

  // MFMA step s contracts k = 2s (lanes 0-31) and 2s+1 (lanes 32-63): every output
  // accumulates its K products in ascending k, like the reference's gemm_nn
  // (src/gemm.c:2223-2239).
  auto compute = [&](const float* cur) {
    const float* As = cur + (wm * WM + l31) * AS + lh;
    const float* Bs = cur + A_FLOATS + lh * BN + wn * WN + l31;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s)
    {
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
      for (int j = 0; j < TN; ++j) b[j] = Bs[(2 * s) * BN + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  };

  GSTAMP(1);   // index arithmetic done
  if (nkt > 0)
  {
  if (PF == 1)
  {
    load_tile(k0_of(0), I0{});
    store_tile(lds, I0{});
    DK_BARRIER();
    GSTAMP(2);   // first tile in LDS
    for (int kt = 0; kt < nkt; ++kt)
    {
      const bool more = (kt + 1) < nkt;
      if (more)
        load_tile(k0_of(kt + 1), I0{});
      compute(lds + (kt & 1) * STAGE);
      if (more)
        store_tile(lds + ((kt + 1) & 1) * STAGE, I0{});
      DK_BARRIER();
    }
  }
  else
  {
    // two register sets: tile t lives in set t%2 until it is written to LDS buffer t%2
    load_tile(k0_of(0), I0{});
    if (nkt > 1)
      load_tile(k0_of(1), I1{});
    store_tile(lds, I0{});
    DK_BARRIER();
    for (int kt = 0; kt < nkt; kt += 2)
    {
      if (kt + 2 < nkt)
        load_tile(k0_of(kt + 2), I0{});
      compute(lds);
      if (kt + 1 < nkt)
        store_tile(lds + STAGE, I1{});
      DK_BARRIER();
      if (kt + 1 >= nkt)
        break;
      if (kt + 3 < nkt)
        load_tile(k0_of(kt + 3), I1{});
      compute(lds + STAGE);
      if (kt + 2 < nkt)
        store_tile(lds, I0{});
      DK_BARRIER();
    }
  }
  }

  GSTAMP(3);   // K loop done
  if (DK_ABL & 1)
  {
    // keep the accumulators alive without the epilogue
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
  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, m0, n0, g, wm, wn, l31, lh);
  GSTAMP(4);   // every store issued
#if DK_GSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  GSTAMP(5);   // stores drained
#endif
}

#if DK_GSTAMP
extern "C" __attribute__((visibility("default"))) int dk_gather_stamps_read(long long* dst, int n)
{
  const int have = 512 * 16 * 8;
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_gstamp), sizeof(long long) * (n < have ? n : have)));
  return have;
}
#endif

// --------------------------------------------------------------------------
// host side: plans (tap table + tile choice), dispatch, profiling
// --------------------------------------------------------------------------
// conv3x3_direct.hip: the patch-in-LDS kernel for 3x3/s1/p1 layers; its tile
// configurations are numbered after the gather configurations.
int dk_conv_direct_num_configs();
const char* dk_conv_direct_config_name(int dcfg);
const char* dk_conv_direct_kernel_name(int dcfg, int pitch_class);
bool dk_conv_direct_applicable(const DkConvDesc* d, const float* weights, int dcfg);
int dk_conv_direct_launch(ConvArgs a, int dcfg, hipStream_t st);
// conv1x1_dma.hip: LDS-DMA ring GEMM for 1x1/s1 layers; numbered after the direct configurations.
int dk_conv_dma1x1_num_configs();
const char* dk_conv_dma1x1_config_name(int c);
const char* dk_conv_dma1x1_kernel_name(int c);
int dk_conv_dma1x1_bm(int c);
bool dk_conv_dma1x1_applicable(const DkConvDesc* d, const float* x, const float* weights, int c);
void dk_conv_dma1x1_launch(ConvArgs a, int c, hipStream_t st);
// conv3x3_wino.hip: fused Winograd F(2x2,3x3) for 3x3/s1/p1 layers; numbered after the DMA configurations.
// It needs the layer's transformed filters (dk_conv_wino_transform_weights + dk_conv_wino_register).
int dk_conv_wino_num_configs();
const char* dk_conv_wino_config_name(int c);
const char* dk_conv_wino_kernel_name(int c, int variant);
bool dk_conv_wino_applicable(const DkConvDesc* d, int c);
const float* dk_conv_wino_lookup(const float* weights);
int dk_conv_wino_launch(ConvArgs a, int c, hipStream_t st);

namespace
{
struct TileCfg
{
  int bm, bn, bk, wm, wn;
  float eff;  // relative MFMA efficiency used by the heuristic
  const char* name;
  void (*kernel[4])(const ConvArgs);  // [AVEC + 2*BVEC]
  int threads;
  int pf;
};

#define DK_CFG(BM, BN, BK, WM, WN, EFF)                                                     \
  {                                                                                         \
    BM, BN, BK, WM, WN, EFF, #BM "x" #BN "x" #BK "_w" #WM "x" #WN,                          \
        {conv_igemm_f32<BM, BN, BK, WM, WN, false, false>,                                  \
            conv_igemm_f32<BM, BN, BK, WM, WN, true, false>,                                \
            conv_igemm_f32<BM, BN, BK, WM, WN, false, true>,                                \
            conv_igemm_f32<BM, BN, BK, WM, WN, true, true>},                                \
        (BM / WM) * (BN / WN) * 64, 1                                                       \
  }
// prefetch distance 2 (two register sets)
#define DK_CFG_PF2(BM, BN, BK, WM, WN, EFF)                                                 \
  {                                                                                         \
    BM, BN, BK, WM, WN, EFF, #BM "x" #BN "x" #BK "_w" #WM "x" #WN "_pf2",                   \
        {conv_igemm_f32<BM, BN, BK, WM, WN, false, false, 2>,                               \
            conv_igemm_f32<BM, BN, BK, WM, WN, true, false, 2>,                             \
            conv_igemm_f32<BM, BN, BK, WM, WN, false, true, 2>,                             \
            conv_igemm_f32<BM, BN, BK, WM, WN, true, true, 2>},                             \
        (BM / WM) * (BN / WN) * 64, 2                                                       \
  }

const TileCfg g_cfgs[] = {
    DK_CFG(128, 128, 16, 64, 64, 1.00f),
    DK_CFG(64, 128, 16, 32, 64, 0.95f),
    DK_CFG(128, 64, 16, 64, 32, 0.97f),
    DK_CFG(64, 64, 16, 32, 32, 0.92f),
    DK_CFG(32, 128, 16, 32, 32, 0.80f),
    DK_CFG(256, 128, 16, 64, 64, 0.98f),
    DK_CFG(64, 256, 16, 32, 128, 0.95f),
    DK_CFG(128, 64, 32, 64, 32, 0.97f),
    DK_CFG(64, 64, 32, 32, 32, 0.92f),
    DK_CFG_PF2(64, 64, 16, 32, 32, 0.90f),
    DK_CFG_PF2(64, 64, 32, 32, 32, 0.90f),
    DK_CFG_PF2(128, 64, 16, 64, 32, 0.90f),
};
const int g_ncfg = sizeof(g_cfgs) / sizeof(g_cfgs[0]);

int g_forced = -1;

struct Plan
{
  int2* ktab = nullptr;
  int kpad = 0;
};

struct DescKey
{
  int v[12];
  bool operator<(const DescKey& o) const { return memcmp(v, o.v, sizeof(v)) < 0; }
};

std::mutex g_mu;
std::map<std::pair<int, DescKey>, Plan> g_plans;  // (device, desc) -> plan

struct ProfRec
{
  hipEvent_t e0, e1;
  int cfg;
  double gflop;
};
int g_prof_on = 0;
std::vector<ProfRec> g_prof;

bool fast_mish()
{
  static int v = -1;
  if (v < 0)
  {
    const char* e = getenv("DK_FAST_MISH");  // default on; DK_FAST_MISH=0 = the reference's formula
    v = (e && !atoi(e)) ? 0 : 1;
  }
  return v == 1;
}

}  // namespace

bool dk_fast_mish_enabled() { return fast_mish(); }

namespace
{
int out_dim(int in, int pad, int size, int stride) { return (in + 2 * pad - size) / stride + 1; }

int env_cfg()
{
  static int env = -2;
  if (env == -2)
  {
    const char* e = getenv("DK_CONV_CFG");
    env = e ? atoi(e) : -1;
  }
  return env;
}

// heuristic: 3x3/s1/p1 layers take the patch-in-LDS kernel, 64x128 tiles when they
// fill the chip at least twice, else 64x64 (measured on yolov4's layers); -1 = gather
int pick_direct(const DkConvDesc* d, const float* weights, int M, long long N)
{
  const long long t128 = (long long)((M + 63) / 64) * ((N + 127) / 128);
  const int dc = t128 >= 512 ? 2 : 3;
  return dk_conv_direct_applicable(d, weights, dc) ? g_ncfg + dc : -1;
}

int pick_cfg(int M, long long N, int groups)
{
  if (g_forced >= 0 && g_forced < g_ncfg)
    return g_forced;
  const int env = env_cfg();
  if (env >= 0 && env < g_ncfg)
    return env;
  const int CUS = 256;
  int best = 0;
  double best_cost = 1e300;
  for (int i = 0; i < g_ncfg; ++i)
  {
    const TileCfg& c = g_cfgs[i];
    const long long tiles = (long long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * groups;
    const long long rounds = (tiles + CUS - 1) / CUS;
    // below one full round only the busiest CU matters; beyond, work is conserved
    const double cost = (double)rounds * c.bm * c.bn / c.eff;
    if (cost < best_cost * 0.999)
    {
      best_cost = cost;
      best = i;
    }
  }
  return best;
}

Plan& get_plan(const DkConvDesc* d, int K, int C, int mode = 0)
{
  DescKey key;
  memset(&key, 0, sizeof(key));
  key.v[0] = d->c; key.v[1] = d->h; key.v[2] = d->w; key.v[3] = d->groups;
  key.v[4] = d->size; key.v[5] = d->dilation; key.v[6] = mode;
  if (mode >= 1)
  {
    key.v[7] = d->stride_x; key.v[8] = d->stride_y; key.v[9] = d->n;
  }
  int dev = cuda_get_device();
  std::lock_guard<std::mutex> lk(g_mu);
  Plan& pl = g_plans[std::make_pair(dev, key)];
  if (!pl.ktab)
  {
    const int kpad = ((K + 63) / 64) * 64 + 64;
    std::vector<int2> h(kpad);
    const int ss = d->size * d->size;
    for (int k = 0; k < kpad; ++k)
    {
      if (k < K && mode == 0)
      {
        const int c = k / ss, t = k % ss, kh = t / d->size, kw = t % d->size;
        h[k].x = (c * d->h * d->w + kh * d->dilation * d->w + kw * d->dilation) * 4;  // bytes
        h[k].y = 31 - t;  // left shift that brings tap t's bit to the sign position
      }
      else if (k < K && mode == 2)
      {
        // data gradient, tap-major contraction index k = t * (n/groups) + m (parity-class launches)
        const int keff = d->dilation * (d->size - 1) + 1, pd = d->pad * d->dilation;
        const int oh = (d->h + 2 * pd - keff) / d->stride_y + 1;
        const int ow = (d->w + 2 * pd - keff) / d->stride_x + 1;
        const int t = k / C, m = k % C, kh = t / d->size, kw = t % d->size;   // C = filters per group here
        const int uy = (kh * d->dilation) / d->stride_y, ux = (kw * d->dilation) / d->stride_x;
        h[k].x = (m * oh * ow - uy * ow - ux) * 4;
        h[k].y = 31 - t;
      }
      else if (k < K)
      {
        // data gradient: k = (m, kh, kw) over the delta tensor [n/groups][oh][ow]
        const int keff = d->dilation * (d->size - 1) + 1, pd = d->pad * d->dilation;
        const int oh = (d->h + 2 * pd - keff) / d->stride_y + 1;
        const int ow = (d->w + 2 * pd - keff) / d->stride_x + 1;
        const int m = k / ss, t = k % ss, kh = t / d->size, kw = t % d->size;
        const int uy = (kh * d->dilation) / d->stride_y, ux = (kw * d->dilation) / d->stride_x;
        h[k].x = (m * oh * ow - uy * ow - ux) * 4;
        h[k].y = 31 - t;
      }
      else
      {
        h[k].x = 0;
        h[k].y = 0;   // selects bit 31 of the "outside" mask, which is always set -> reads as 0
      }
    }
    CHECK_HIP(hipMalloc((void**)&pl.ktab, kpad * sizeof(int2)));
    CHECK_HIP(hipMemcpy(pl.ktab, h.data(), kpad * sizeof(int2), hipMemcpyHostToDevice));
    pl.kpad = kpad;
    (void)C;
  }
  return pl;
}
}  // namespace

// configuration index space: [0, g_ncfg) gather shapes, then the direct 3x3 shapes, then the
// LDS-DMA 1x1 shapes, then the Winograd 3x3 shapes
static int dma_base() { return g_ncfg + dk_conv_direct_num_configs(); }
static int wino_base() { return dma_base() + dk_conv_dma1x1_num_configs(); }
static int total_cfgs() { return wino_base() + dk_conv_wino_num_configs(); }

extern "C" int dk_conv_force_config(int cfg)
{
  g_forced = cfg;
  return total_cfgs();
}

extern "C" const char* dk_conv_config_name(int cfg)
{
  if (cfg >= wino_base())
    return dk_conv_wino_config_name(cfg - wino_base());
  if (cfg >= dma_base())
    return dk_conv_dma1x1_config_name(cfg - dma_base());
  if (cfg >= g_ncfg)
    return dk_conv_direct_config_name(cfg - g_ncfg);
  return (cfg >= 0 && cfg < g_ncfg) ? g_cfgs[cfg].name : nullptr;
}

bool dk_conv_config_applicable(const DkConvDesc* d, int cfg)
{
  if (cfg < 0)
    return false;
  if (cfg < g_ncfg)
    return true;
  if (cfg >= wino_base())
    return dk_conv_wino_applicable(d, cfg - wino_base());
  if (cfg >= dma_base())
    return dk_conv_dma1x1_applicable(d, nullptr, nullptr, cfg - dma_base());
  return dk_conv_direct_applicable(d, nullptr, cfg - g_ncfg);
}

bool dk_conv_config_is_wino(int cfg) { return cfg >= wino_base() && cfg < total_cfgs(); }
extern "C" int dk_conv_config_can_run(const DkConvDesc* d, int cfg) { return d && dk_conv_config_applicable(d, cfg) ? 1 : 0; }

extern "C" int dk_conv_pick_config(const DkConvDesc* d)
{
  const int pad = d->pad * d->dilation;
  const int eff = d->dilation * (d->size - 1) + 1;
  const int oh = out_dim(d->h, pad, eff, d->stride_y), ow = out_dim(d->w, pad, eff, d->stride_x);
  if (g_forced < 0 && env_cfg() < 0)
  {
    const int dc = pick_direct(d, nullptr, d->n / d->groups, (long long)d->batch * oh * ow);
    if (dc >= 0)
      return dc;
  }
  return pick_cfg(d->n / d->groups, (long long)d->batch * oh * ow, d->groups);
}

extern "C" __attribute__((visibility("default"))) int dk_profile_is_on() { return g_prof_on; }

namespace
{
std::vector<std::string> g_named_slots;   // slot 256 + i
}
bool dk_prof_on() { return g_prof_on != 0; }

int dk_prof_named_slot(const char* kernel_name)
{
  std::lock_guard<std::mutex> lk(g_mu);
  for (size_t i = 0; i < g_named_slots.size(); ++i)
    if (g_named_slots[i] == kernel_name)
      return 256 + (int)i;
  g_named_slots.push_back(kernel_name);
  return 256 + (int)g_named_slots.size() - 1;
}

void dk_prof_begin(DkProfScope& s, void* stream)
{
  if (!g_prof_on)
    return;
  hipEvent_t e;
  CHECK_HIP(hipEventCreate(&e));
  CHECK_HIP(hipEventRecord(e, (hipStream_t)stream));
  s.e0 = e;
}

void dk_prof_end(DkProfScope& s, void* stream, int slot, double gflop)
{
  if (!s.e0)
    return;
  ProfRec pr;
  pr.e0 = (hipEvent_t)s.e0;
  CHECK_HIP(hipEventCreate(&pr.e1));
  CHECK_HIP(hipEventRecord(pr.e1, (hipStream_t)stream));
  pr.cfg = slot;
  pr.gflop = gflop;
  std::lock_guard<std::mutex> lk(g_mu);
  g_prof.push_back(pr);
}

extern "C" void dk_profile_enable(int on)
{
  g_prof_on = on;
  if (on)
  {
    for (auto& r : g_prof)
    {
      (void)hipEventDestroy(r.e0);
      (void)hipEventDestroy(r.e1);
    }
    g_prof.clear();
  }
}

extern "C" int dk_profile_read(double* out, int max_cfgs)
{
  CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < max_cfgs * 3; ++i) out[i] = 0;
  for (auto& r : g_prof)
  {
    float ms = 0;
    CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    if (r.cfg < max_cfgs)
    {
      out[r.cfg * 3 + 0] += 1;
      out[r.cfg * 3 + 1] += r.gflop;
      out[r.cfg * 3 + 2] += ms;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  g_prof.clear();
  const int named = 256 + (int)g_named_slots.size();
  return named > total_cfgs() * 4 ? named : total_cfgs() * 4;
}

// Kernel symbol exactly as rocprofv3 prints it, for profile slot idx = cfg*4 + AVEC + 2*BVEC
// (direct 3x3 configurations: cfg*4 + pitch class).
extern "C" __attribute__((visibility("default"))) const char* dk_conv_kernel_name(int idx)
{
  static thread_local char buf[128];
  if (idx >= 256)
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx - 256 >= (int)g_named_slots.size())
      return nullptr;
    snprintf(buf, sizeof(buf), "%s", g_named_slots[idx - 256].c_str());
    return buf;
  }
  if (idx >= wino_base() * 4)
    return dk_conv_wino_kernel_name(idx / 4 - wino_base(), idx & 3);
  if (idx >= dma_base() * 4)
    return dk_conv_dma1x1_kernel_name(idx / 4 - dma_base());
  if (idx >= g_ncfg * 4)
    return dk_conv_direct_kernel_name(idx / 4 - g_ncfg, idx & 3);
  if (idx < 0)
    return nullptr;
  const TileCfg& c = g_cfgs[idx / 4];
  snprintf(buf, sizeof(buf), "conv_igemm_f32<%d, %d, %d, %d, %d, %s, %s, %d>", c.bm, c.bn, c.bk, c.wm,
      c.wn, (idx & 1) ? "true" : "false", (idx & 2) ? "true" : "false", c.pf);
  return buf;
}

// ---- CPU-testable host logic (tests/test_host_cpu.py) ---------------------------------
// exact reciprocal division used by the kernels' index arithmetic
extern "C" __attribute__((visibility("default"))) int DkTestFdiv(int n, int d) { return fdiv(n, d, 1.0 / d); }

// Tile of workgroup `bid` under the XCD partition the host would pick for (tiles_m, tiles_n, groups,
// weight_bytes): out = {grid size, pm, valid, g, tile_m, tile_n}
extern "C" __attribute__((visibility("default"))) void DkTestBlockTile(int tiles_m, int tiles_n, int groups,
    long long weight_bytes, int bid, int* out)
{
  ConvArgs a;
  memset(&a, 0, sizeof(a));
  a.tiles_m = tiles_m; a.tiles_n = tiles_n; a.groups = groups;
  a.OHW = a.OW = a.H = a.W = 1;
  conv_args_finish(a);
  const long long nblk = conv_pick_partition(a, (size_t)weight_bytes, 64);
  int g = 0, tm = 0, tn = 0;
  const bool ok = bid < nblk && conv_block_tile_of(a, bid, (int)nblk, g, tm, tn);
  out[0] = (int)nblk; out[1] = a.pm; out[2] = ok; out[3] = g; out[4] = tm; out[5] = tn;
}

int dk_conv_num_configs() { return total_cfgs(); }
int dk_conv_num_gather_configs() { return g_ncfg; }

const int2* dk_conv_ktab(const DkConvDesc* d, int K, int C, int mode) { return get_plan(d, K, C, mode).ktab; }

void dk_conv_prepare(const DkConvDesc* d)
{
  const int C = d->c / d->groups;
  (void)get_plan(d, C * d->size * d->size, C);
}

extern "C" int dk_conv_forward(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream)
{
  return dk_conv_forward_cfg(d, x, weights, biases, y, residual, activation_input, stream, -1);
}

int dk_conv_forward_cfg(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream,
    int cfg_override, int out_ctot, const DkConvDual* dual, const float* wino_filters)
{
  if (!d || !x || !weights || !y || d->groups < 1 || d->c % d->groups || d->n % d->groups ||
      d->size < 1 || d->stride_x < 1 || d->stride_y < 1 || d->dilation < 1)
  {
    fprintf(stderr, "dk_conv_forward: invalid descriptor\n");
    return 1;
  }
  if (d->size * d->size > 31)
  {
    fprintf(stderr, "dk_conv_forward: kernel size %d unsupported (size*size must be <= 31)\n", d->size);
    return 1;
  }
  const int pad = d->pad * d->dilation;
  const int keff = d->dilation * (d->size - 1) + 1;
  const int OH = out_dim(d->h, pad, keff, d->stride_y);
  const int OW = out_dim(d->w, pad, keff, d->stride_x);
  if (OH < 1 || OW < 1)
  {
    fprintf(stderr, "dk_conv_forward: empty output\n");
    return 1;
  }
  // The reference sizes l->output with ConvOutHeight/Width (no dilation term,
  // src/convolutional_layer.cpp:87-95) but unfolds with the dilated formula
  // (src/im2col.c:61-64); the two agree for every cfg darknet generates
  // (padding = size/2, or dilation 1).  Refuse the inconsistent combinations.
  if (OH != out_dim(d->h, d->pad, d->size, d->stride_y) ||
      OW != out_dim(d->w, d->pad, d->size, d->stride_x))
  {
    fprintf(stderr, "dk_conv_forward: pad/dilation combination is inconsistent in the reference\n");
    return 1;
  }
  const int C = d->c / d->groups, M = d->n / d->groups, K = C * d->size * d->size;
  const size_t in_img = (size_t)d->c * d->h * d->w;
  if (out_ctot && (out_ctot < d->n || residual || activation_input))
  {
    fprintf(stderr, "dk_conv_forward: a channel-slice output takes no residual / pre-activation\n");
    return 1;
  }
  if (dual && (d->groups != 1 || residual || activation_input || dual->m_split <= 0 ||
                  dual->m_split >= d->n || dual->m_split % 32 || !dual->y2))
  {
    fprintf(stderr, "dk_conv_forward: invalid dual-output request\n");
    return 1;
  }
  const int n1 = dual ? dual->m_split : d->n;           // filters of the first output
  const int n2 = dual ? d->n - dual->m_split : 0;
  const int Mtot = out_ctot ? out_ctot : n1;
  const int Mtot2 = dual ? (dual->out_ctot2 ? dual->out_ctot2 : n2) : 0;
  const size_t out_img = (size_t)Mtot * OH * OW;  // batch stride of the output
  const size_t out_img2 = (size_t)Mtot2 * OH * OW;
  // the gather uses 32-bit byte offsets checked by the buffer descriptor:
  // process the batch in chunks whose input stays below 2 GiB
  const size_t max_elems = (size_t)1 << 29;  // byte offsets stay below 2^31: bit 31 is the "masked" flag
  int chunk = d->batch;
  if (in_img * (size_t)chunk >= max_elems || out_img * (size_t)chunk >= ((size_t)1 << 30) ||
      out_img2 * (size_t)chunk >= ((size_t)1 << 30))
  {
    chunk = (int)((max_elems - 1) / in_img);
    const int c2 = (int)((((size_t)1 << 30) - 1) / out_img);
    if (c2 < chunk)
      chunk = c2;
    if (out_img2)
    {
      const int c3 = (int)((((size_t)1 << 30) - 1) / out_img2);
      if (c3 < chunk)
        chunk = c3;
    }
    if (chunk < 1)
    {
      fprintf(stderr, "dk_conv_forward: one image exceeds the 4 GiB addressing window\n");
      return 1;
    }
  }
  Plan& pl = get_plan(d, K, C);
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();

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
    a.ktab = pl.ktab;
    a.x_bytes = (unsigned)(in_img * nb * sizeof(float));
    a.w_bytes = (unsigned)((size_t)M * K * sizeof(float));
    a.y_bytes = (unsigned)((out_img * (nb - 1) + (size_t)n1 * OH * OW) * sizeof(float));
    a.y2 = dual ? dual->y2 + (size_t)b0 * out_img2 : nullptr;
    a.y2_bytes = dual ? (unsigned)((out_img2 * (nb - 1) + (size_t)n2 * OH * OW) * sizeof(float)) : 0u;
    a.Mtot2 = Mtot2;
    a.m_split = dual ? dual->m_split : 0;
    a.C = C; a.H = d->h; a.W = d->w; a.Ctot = d->c;
    a.M = M; a.Mtot = Mtot; a.K = K;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW;
    a.N = nb * OH * OW;
    a.size = d->size; a.stride_x = d->stride_x; a.stride_y = d->stride_y;
    a.pad = pad; a.dil = d->dilation;
    a.act = d->activation;
    if (d->activation == DK_MISH && fast_mish())
      a.act |= DK_ACT_FAST;
    {
      // timing-only diagnostics (cdna_hip_programming.md section 7: zero-record descriptors drop the
      // loads / stores of ONE buffer while the instruction stream stays): DK_DEBUG_DROP bit 0 = input
      // loads, bit 1 = output stores, bit 2 = weight loads.  Results are garbage by construction.
      static const int drop = getenv("DK_DEBUG_DROP") ? atoi(getenv("DK_DEBUG_DROP")) : 0;
      if (drop & 1) a.x_bytes = 0;
      if (drop & 2) a.y_bytes = a.y2_bytes = 0;
      if (drop & 4) a.w_bytes = 0;
    }
    int want = cfg_override >= 0 ? cfg_override : (g_forced >= 0 ? g_forced : env_cfg());
    const int ntot = dma_base();
    if (want < 0)
      want = pick_direct(d, weights, M, a.N);
    // Winograd: only with registered transformed filters, never with a pre-activation store or a
    // dual output (neither occurs on the inference loads that select it)
    const float* wino_u = (want >= wino_base() && want < total_cfgs() && !activation_input && !dual &&
                              dk_conv_wino_applicable(d, want - wino_base()))
                              ? (wino_filters ? wino_filters : dk_conv_wino_lookup(weights))
                              : nullptr;
    if (wino_u)
    {
      DkProfScope ps;
      dk_prof_begin(ps, st);
      ConvArgs aw = a;
      aw.w = wino_u;
      const int variant = dk_conv_wino_launch(aw, want - wino_base(), st);
      CHECK_HIP(hipPeekAtLastError());
      if (variant >= 0)
      {
        dk_prof_end(ps, st, want * 4 + variant, 2.0 * (double)M * K * (double)a.N / 1e9);
        continue;
      }
      dk_prof_end(ps, st, dk_prof_named_slot("conv3x3_wino_f32 (geometry does not fit: fell back)"), 0.0);
      if (wino_filters)
      {
        // the caller handed over transformed filters only: `weights` need not be the plain filters of this convolution
        fprintf(stderr, "dk_conv_forward: explicit Winograd filters, but the launch geometry does not fit\n");
        return 1;
      }
      want = -1;   // raw-patch geometry does not fit the compiled load counts: take the direct kernel
    }
    if (want < 0)
      want = pick_direct(d, weights, M, a.N);
    if (want >= dma_base() && want < wino_base() && dk_conv_dma1x1_applicable(d, a.x, weights, want - dma_base()) &&
        !(dual && dual->m_split % dk_conv_dma1x1_bm(want - dma_base())))
    {
      ProfRec pr;
      if (g_prof_on)
      {
        CHECK_HIP(hipEventCreate(&pr.e0));
        CHECK_HIP(hipEventCreate(&pr.e1));
        CHECK_HIP(hipEventRecord(pr.e0, st));
      }
      dk_conv_dma1x1_launch(a, want - dma_base(), st);
      CHECK_HIP(hipPeekAtLastError());
      if (g_prof_on)
      {
        CHECK_HIP(hipEventRecord(pr.e1, st));
        pr.cfg = want * 4;
        pr.gflop = 2.0 * (double)M * K * (double)a.N / 1e9;
        g_prof.push_back(pr);
      }
      continue;
    }
    if (want >= g_ncfg && want < ntot && dk_conv_direct_applicable(d, weights, want - g_ncfg))
    {
      ProfRec pr;
      if (g_prof_on)
      {
        CHECK_HIP(hipEventCreate(&pr.e0));
        CHECK_HIP(hipEventCreate(&pr.e1));
        CHECK_HIP(hipEventRecord(pr.e0, st));
      }
      a.groups = 1;
      a.mode = 0;
      const int pc = dk_conv_direct_launch(a, want - g_ncfg, st);
      CHECK_HIP(hipPeekAtLastError());
      if (g_prof_on)
      {
        CHECK_HIP(hipEventRecord(pr.e1, st));
        pr.cfg = want * 4 + pc;
        pr.gflop = 2.0 * (double)M * K * (double)a.N / 1e9;
        g_prof.push_back(pr);
      }
      continue;
    }
    int ci = (cfg_override >= 0 && cfg_override < g_ncfg) ? cfg_override : pick_cfg(M, a.N, d->groups);
    if (dual && dual->m_split % g_cfgs[ci].bm)
      ci = 3;  // 64x64: every dual split is a multiple of 64 rows (checked by the planner)
    const TileCfg& c = g_cfgs[ci];
    a.tiles_m = (M + c.bm - 1) / c.bm;
    a.tiles_n = (a.N + c.bn - 1) / c.bn;
    a.groups = d->groups;
    a.mode = 0;
    conv_args_finish(a);
    const long long nblk = conv_pick_partition(a, (size_t)M * K * sizeof(float), c.bm);
    if (nblk > 0x7fffffffLL)
    {
      fprintf(stderr, "dk_conv_forward: grid too large\n");
      return 1;
    }
    ProfRec pr;
    if (g_prof_on)
    {
      CHECK_HIP(hipEventCreate(&pr.e0));
      CHECK_HIP(hipEventCreate(&pr.e1));
      CHECK_HIP(hipEventRecord(pr.e0, st));
    }
    const bool avec = (K % 4 == 0) && (((uintptr_t)weights & 15) == 0);
    const bool bvec = d->size == 1 && d->stride_x == 1 && d->stride_y == 1 && pad == 0 &&
                      ((OH * OW) % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
    hipLaunchKernelGGL(c.kernel[(avec ? 1 : 0) + (bvec ? 2 : 0)], dim3((unsigned)nblk),
        dim3(c.threads), 0, st, a);
    CHECK_HIP(hipPeekAtLastError());
    if (g_prof_on)
    {
      CHECK_HIP(hipEventRecord(pr.e1, st));
      pr.cfg = ci * 4 + (avec ? 1 : 0) + (bvec ? 2 : 0);
      pr.gflop = 2.0 * (double)M * K * d->groups * (double)a.N / 1e9;
      g_prof.push_back(pr);
    }
  }
  return 0;
}


// Data gradient of the convolution (the reference's gemm(1,0) + col2im_gpu_kernel_ext,
// src/convolutional_kernels.cu:784-812; CPU: gemm TN + col2im_cpu_ext,
// src/convolutional_layer.cpp:1358-1376) as ONE implicit GEMM with the same
// kernel: prev_delta[b][c][iy][ix] = sum_{m,kh,kw} Wt[c][(m,kh,kw)] * delta[b][m][oy][ox]
// with (oy, ox) = ((iy + pad - kh*dil)/stride, ...) where divisible.  No col
// buffer, no atomics; prev_delta is OVERWRITTEN (col2im_cpu_ext zero-fills its
// target first, src/col2im.c:70 -- SURVEY quirk 4).  `wt` is the per-group
// transposed weight matrix produced by dk_transpose_weights.
extern "C" int dk_conv_backward_data(const DkConvDesc* d, const float* delta, const float* wt,
    float* prev_delta, void* stream)
{
  return dk_conv_backward_data_cfg(d, delta, wt, prev_delta, stream, -1, 0);
}

extern "C" int dk_conv_backward_data_tapmajor(const DkConvDesc* d, const float* delta, const float* wt_tapmajor,
    float* prev_delta, void* stream)
{
  if (!d || !dk_conv_dgrad_tapmajor(d))
  {
    fprintf(stderr, "dk_conv_backward_data_tapmajor: the layer does not take the parity-class form\n");
    return 1;
  }
  return dk_conv_backward_data_cfg(d, delta, wt_tapmajor, prev_delta, stream, -1, 1);
}

// true when the data gradient of this layer takes the parity-class form (mode 2 of the gather kernel): stride 2 in
// both directions, even input dimensions, one group, filters a multiple of the deepest K tile
bool dk_conv_dgrad_tapmajor(const DkConvDesc* d)
{
  static const bool on = !(getenv("DK_DGRAD_PARITY") && !atoi(getenv("DK_DGRAD_PARITY")));
  return on && d->stride_x == 2 && d->stride_y == 2 && d->dilation == 1 && d->groups == 1 && d->size > 1 && d->size < 4 &&
         d->h % 2 == 0 && d->w % 2 == 0 && d->n % 32 == 0;
}

int dk_conv_backward_data_cfg(const DkConvDesc* d, const float* delta, const float* wt,
    float* prev_delta, void* stream, int cfg_override, int wt_tapmajor)
{
  if (!d || !delta || !wt || !prev_delta || d->groups < 1 || d->size * d->size > 31)
  {
    fprintf(stderr, "dk_conv_backward_data: invalid arguments\n");
    return 1;
  }
  const int pad = d->pad * d->dilation;
  const int keff = d->dilation * (d->size - 1) + 1;
  const int OHd = out_dim(d->h, pad, keff, d->stride_y);
  const int OWd = out_dim(d->w, pad, keff, d->stride_x);
  const int Cg = d->c / d->groups, Mg = d->n / d->groups, ss = d->size * d->size;
  const int K = Mg * ss;
  const size_t delta_img = (size_t)d->n * OHd * OWd;
  const size_t in_img = (size_t)d->c * d->h * d->w;
  int chunk = d->batch;
  const size_t lim_in = (size_t)1 << 29, lim_out = (size_t)1 << 30;
  if (delta_img * chunk >= lim_in || in_img * chunk >= lim_out)
  {
    chunk = (int)((lim_in - 1) / delta_img);
    const int c2 = (int)((lim_out - 1) / in_img);
    if (c2 < chunk)
      chunk = c2;
    if (chunk < 1)
    {
      fprintf(stderr, "dk_conv_backward_data: one image exceeds the addressing window\n");
      return 1;
    }
  }
  const bool par = wt_tapmajor != 0;
  if (par && !dk_conv_dgrad_tapmajor(d))
  {
    fprintf(stderr, "dk_conv_backward_data: tap-major weights given for a layer that does not take the parity-class form\n");
    return 1;
  }
  Plan& pl = get_plan(d, K, Mg, par ? 2 : 1);
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  for (int b0 = 0; b0 < d->batch; b0 += chunk)
  {
    const int nb = (d->batch - b0 < chunk) ? d->batch - b0 : chunk;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = delta + (size_t)b0 * delta_img;
    a.w = wt;
    a.bias = nullptr;
    a.y = prev_delta + (size_t)b0 * in_img;
    a.ktab = pl.ktab;
    a.x_bytes = (unsigned)(delta_img * nb * sizeof(float));
    a.w_bytes = (unsigned)((size_t)Cg * K * sizeof(float));
    a.y_bytes = (unsigned)(in_img * nb * sizeof(float));
    a.C = Mg; a.H = OHd; a.W = OWd; a.Ctot = d->n;
    a.M = Cg; a.Mtot = d->c; a.K = K;
    a.OH = d->h; a.OW = d->w; a.OHW = d->h * d->w;
    a.N = nb * d->h * d->w;
    a.size = d->size; a.stride_x = d->stride_x; a.stride_y = d->stride_y;
    a.pad = pad; a.dil = d->dilation;
    a.act = DK_LINEAR;
    a.mode = 1;
    const int ci = (cfg_override >= 0 && cfg_override < g_ncfg) ? cfg_override : pick_cfg(Cg, a.N, d->groups);
    const TileCfg& c = g_cfgs[ci];
    if (par)
    {
      a.mode = 2;
      a.par_w2 = d->w / 2;
      a.par_hw2 = (d->h / 2) * (d->w / 2);
      a.par_count = nb * a.par_hw2;
      a.par_ncls = (a.par_count + c.bn - 1) / c.bn * c.bn;
      a.N = 4 * a.par_ncls;
    }
    a.tiles_m = (Cg + c.bm - 1) / c.bm;
    a.tiles_n = (a.N + c.bn - 1) / c.bn;
    a.groups = d->groups;
    const long long nblk = conv_pick_partition(a, (size_t)a.M * a.K * sizeof(float), c.bm);
    const bool avec = (K % 4 == 0) && (((uintptr_t)wt & 15) == 0);
    conv_args_finish(a);
    hipLaunchKernelGGL(c.kernel[avec ? 1 : 0], dim3((unsigned)nblk), dim3(c.threads), 0, st, a);
    CHECK_HIP(hipPeekAtLastError());
  }
  return 0;
}
