// conv3x3_direct_f16.hip -- fp16 operands / fp32 accumulate for 3x3 / stride 1 / pad 1
// layers (config C5), patch-in-LDS formulation (see conv3x3_direct.hip) on
// v_mfma_f32_32x32x16_f16.
//
// Replaces the reference's fp16 branch (cuda_convert_f32_to_f16 + cudnnConvolutionForward
// with CUDNN_DATA_HALF + cuda_convert_f16_to_f32, src/convolutional_kernels.cu:357-456) for
// the layers its eligibility rule admits AND that are 3x3/s1/p1 with C % 16 == 0; other
// eligible layers keep conv_igemm_f16.  Inputs/outputs stay fp32 NCHW in HBM (as in the
// reference, which converts per layer); operands are rounded to fp16 (RNE) on the way into
// LDS, exactly the values the gather kernel and the test oracle use.
//
// K order inside a stage of 16 input channels: k' = tap*16 + c (tap-major), so that the 8
// halves one MFMA lane needs (k' = 8*lh .. 8*lh+7 of tap t) are 8 CHANNELS of ONE patch
// position: the patch is staged as [channel half h][position][8 halves] and the B fragment
// of tap (kh,kw) is one ds_read_b128 at (lane position + kh*P + kw) * 16 bytes -- an
// immediate offset.  The weights are re-laid-out once per layer (inference: weights are
// static) by dk_conv_half_pack_weights into [m][c/16][tap][16] halves, so a stage's A tile
// is a straight 16-byte copy into LDS rows of 304 bytes (conflict-free b128 fragment reads).
// fp32 accumulation order differs from the fp32 path (tap-major inside 16-channel groups);
// the parity oracle for this path is order-insensitive at its tolerance (tests/test_gpu_half.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "conv_common.h"
#include "dark_hip.h"
#include "dk_device_math.h"
#include "dk_internal.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

namespace
{
constexpr int CG = 16;                 // channels per stage
constexpr int A_ROW_BYTES = 9 * CG * 2;  // 288: one filter's stage row in the packed weights
constexpr int A_PITCH = A_ROW_BYTES + 16;  // 304: LDS row pitch (76 words: b128 reads of 8 rows hit 32 banks)
}  // namespace

template <int BM, int BN, int WM, int WN, int P, int ROWS>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
conv3x3_direct_f16(const ConvArgs p)
{
  constexpr int NWN = BN / WN;
  constexpr int NW = (BM / WM) * NWN;
  constexpr int T = NW * 64;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int CAP = ROWS * P;              // patch positions per channel half
  constexpr int PPT = (CAP + T - 1) / T;     // positions per thread
  constexpr int A_BYTES = BM * A_PITCH;
  constexpr int B_BYTES = 2 * CAP * 16;
  constexpr int STAGE = A_BYTES + B_BYTES;   // bytes
  constexpr int AQ = BM * 18;                // 16-byte chunks of the A stage
  constexpr int PA = (AQ + T - 1) / T;

  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

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

  const int H = p.H, W = p.W, HW = H * W, He = H + 2;
  const int nbatch = fdiv(p.N, HW, p.inv_HW);
  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // ---- block geometry in extended rows (see conv3x3_direct.hip)
  int R0, rows_used;
  {
    const int b0 = fdiv(n0, HW, p.inv_HW);
    const int oy0 = fdiv(n0 - b0 * HW, W, p.inv_W);
    R0 = b0 * He + oy0;
    const int nl = ((n0 + BN < p.N) ? n0 + BN : p.N) - 1;
    const int bl = fdiv(nl, HW, p.inv_HW);
    const int oyl = fdiv(nl - bl * HW, W, p.inv_W);
    rows_used = bl * He + oyl + 1 - R0 + 2;
  }
  const int used_slots = rows_used * P;

  // ---- per-lane byte offsets of the wave's pixel columns inside the patch (tap (0,0), half lh)
  int lbyte[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
  {
    int n = n0 + wn * WN + j * 32 + l31;
    n = (n < p.N) ? n : p.N - 1;
    const int b = fdiv(n, HW, p.inv_HW);
    const int q = n - b * HW;
    const int oy = fdiv(q, W, p.inv_W);
    const int ox = q - oy * W;
    lbyte[j] = (lh * CAP + (b * He + oy - R0) * P + ox) * 16;
  }

  // ---- patch loader: position -> global offset (channel 0 of the group), fixed over stages
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

  // ---- A loader: 16-byte chunk u -> (row, chunk) of the packed weights
  const int wrow_bytes = p.C * 18;  // C*9 halves per filter
  unsigned aofs[PA];
  int a_lds[PA];
#pragma unroll
  for (int jj = 0; jj < PA; ++jj)
  {
    const int u = tid + jj * T;
    const int row = u / 18;
    const int ch = u - row * 18;
    const bool ok = u < AQ && (m0 + row) < p.M;
    aofs[jj] = ok ? (unsigned)((m0 + row) * wrow_bytes + ch * 16) : OOB;
    a_lds[jj] = row * A_PITCH + ch * 16;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nst = p.C / CG;
  u32x4 ra[PA];
  float rp[PPT][2][8];

  auto load_stage = [&](int st) {
#pragma unroll
    for (int jj = 0; jj < PA; ++jj)
      ra[jj] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)aofs[jj], st * A_ROW_BYTES, 0);
#pragma unroll
    for (int jj = 0; jj < PPT; ++jj)
      if (jj * T < used_slots)
      {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            rp[jj][h][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                xr, (int)gofs[jj], ((st * CG + h * 8 + e) * HW) * 4, 0));
      }
  };

  auto store_stage = [&](unsigned char* stg) {
#pragma unroll
    for (int jj = 0; jj < PA; ++jj)
      if ((jj + 1) * T <= AQ || tid < AQ - jj * T)
        *(u32x4*)(stg + a_lds[jj]) = ra[jj];
    unsigned char* Ps = stg + A_BYTES;
#pragma unroll
    for (int jj = 0; jj < PPT; ++jj)
      if (jj * T < used_slots && ((jj + 1) * T <= CAP || tid < CAP - jj * T))
      {
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
          half8 v;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (_Float16)rp[jj][h][e];  // v_cvt_f16_f32: round to nearest even
          *(half8*)(Ps + (h * CAP + tid + jj * T) * 16) = v;
        }
      }
  };

  load_stage(0);
  store_stage(lds);
  __syncthreads();

  for (int st = 0; st < nst; ++st)
  {
    const unsigned char* cur = lds + (st & 1) * STAGE;
    const bool more = (st + 1) < nst;
    if (more)
      load_stage(st + 1);

    const unsigned char* As = cur + (wm * WM + l31) * A_PITCH + lh * 16;
    const unsigned char* Ps = cur + A_BYTES;
#pragma unroll
    for (int t = 0; t < 9; ++t)
    {
      const int po = ((t / 3) * P + (t % 3)) * 16;
      half8 a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const half8*)(As + i * 32 * A_PITCH + t * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *(const half8*)(Ps + lbyte[j] + po);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
    }

    if (more)
      store_stage(lds + ((st + 1) & 1) * STAGE);
    __syncthreads();
  }

  conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, m0, n0, 0, wm, wn, l31, lh);
}

// [m][c][3][3] fp32 -> [m][c/16][tap][16] fp16 (round to nearest even)
__global__ void pack_weights_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ wp, int C,
    size_t total)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x)
  {
    const int cl = (int)(i % CG);
    size_t t = i / CG;
    const int tap = (int)(t % 9);
    t /= 9;
    const int cg = (int)(t % (C / CG));
    const size_t m = t / (C / CG);
    wp[i] = (_Float16)w[(m * C + cg * CG + cl) * 9 + tap];
  }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
namespace
{
struct PitchClassH
{
  int p, rows;
};
#define DK_HPC0 24, 15
#define DK_HPC1 40, 10
#define DK_HPC2 80, 8
#define DK_HPC3 160, 6
const PitchClassH g_hpc[4] = {{DK_HPC0}, {DK_HPC1}, {DK_HPC2}, {DK_HPC3}};

typedef void (*HKernel)(const ConvArgs);
struct HCfg
{
  int bm, bn, wm, wn;
  HKernel kernel[4];
};
#define DK_HCFG(BM, BN, WM, WN)                                                                  \
  {                                                                                              \
    BM, BN, WM, WN,                                                                              \
    {                                                                                            \
      conv3x3_direct_f16<BM, BN, WM, WN, DK_HPC0>, conv3x3_direct_f16<BM, BN, WM, WN, DK_HPC1>,  \
          conv3x3_direct_f16<BM, BN, WM, WN, DK_HPC2>, conv3x3_direct_f16<BM, BN, WM, WN, DK_HPC3> \
    }                                                                                            \
  }
// 128x128 tiles (64x64 per wave, half the LDS fragment traffic per MFMA) were measured slower:
// their 104-119 KB of LDS leave one workgroup per CU and the kernel is latency-bound (a stage
// of 16 channels is only 18 x 32 MFMA cycles per wave, far less than a global load's latency).
const HCfg g_hcfg[] = {
    DK_HCFG(64, 128, 32, 64),
};

int h_pitch_class(int bn, int w, int h)
{
  const int hw = w * h;
  const int rows = (bn - 1 + w - 1) / w + 1 + 2 * ((bn - 1 + hw - 1) / hw) + 2;
  for (int i = 0; i < 4; ++i)
    if (w + 2 <= g_hpc[i].p)
      return rows <= g_hpc[i].rows ? i : -1;
  return -1;
}
}  // namespace

// halves needed for the packed weights of a layer, or 0 when the layer cannot take this kernel
extern "C" size_t dk_conv_half_direct_weights_size(const DkConvDesc* d)
{
  if (d->size != 3 || d->stride_x != 1 || d->stride_y != 1 || d->dilation != 1 || d->pad != 1 ||
      d->groups != 1 || d->c % CG != 0 || !dk_conv_half_eligible(d, 1))
    return 0;
  if (h_pitch_class(128, d->w, d->h) < 0)
    return 0;
  return (size_t)d->n * d->c * 9;
}

extern "C" int dk_conv_half_pack_weights(const DkConvDesc* d, const float* weights, void* packed, void* stream)
{
  const size_t total = dk_conv_half_direct_weights_size(d);
  if (!total || !weights || !packed)
  {
    fprintf(stderr, "dk_conv_half_pack_weights: layer does not take the direct fp16 kernel\n");
    return 1;
  }
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  size_t blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pack_weights_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, weights,
      (_Float16*)packed, d->c, total);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// forward with weights packed by dk_conv_half_pack_weights; same epilogue contract as dk_conv_forward
int dk_conv_forward_half_direct(const DkConvDesc* d, const float* x, const void* packed_weights,
    const float* biases, float* y, const float* residual, void* stream, int out_ctot)
{
  if (!dk_conv_half_direct_weights_size(d) || !x || !packed_weights || !y)
  {
    fprintf(stderr, "dk_conv_forward_half_direct: layer does not take the direct fp16 kernel\n");
    return 1;
  }
  if (out_ctot && (out_ctot < d->n || residual))
    return 1;
  const int OH = d->h, OW = d->w;
  const int C = d->c, M = d->n;
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
  hipStream_t st = stream ? (hipStream_t)stream : get_cuda_stream();
  for (int b0 = 0; b0 < d->batch; b0 += chunk)
  {
    const int nb = (d->batch - b0 < chunk) ? d->batch - b0 : chunk;
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x + (size_t)b0 * in_img;
    a.w = (const float*)packed_weights;
    a.bias = biases;
    a.y = y + (size_t)b0 * out_img;
    a.residual = residual ? residual + (size_t)b0 * out_img : nullptr;
    a.x_bytes = (unsigned)(in_img * nb * 4);
    a.w_bytes = (unsigned)((size_t)M * C * 9 * 2);
    a.y_bytes = (unsigned)((out_img * (nb - 1) + (size_t)d->n * OH * OW) * 4);
    a.C = C; a.H = d->h; a.W = d->w; a.Ctot = d->c;
    a.M = M; a.Mtot = Mtot; a.K = C * 9;
    a.OH = OH; a.OW = OW; a.OHW = OH * OW;
    a.N = nb * OH * OW;
    a.size = 3; a.stride_x = a.stride_y = 1; a.pad = 1; a.dil = 1;
    a.act = d->activation;
    if (d->activation == DK_MISH && dk_fast_mish_enabled())
      a.act |= DK_ACT_FAST;
    a.groups = 1;
    const int ci = 0;
    const HCfg& c = g_hcfg[ci];
    const int pc = h_pitch_class(c.bn, d->w, d->h);
    a.tiles_m = (M + c.bm - 1) / c.bm;
    a.tiles_n = (a.N + c.bn - 1) / c.bn;
    conv_args_finish(a);
    const long long nblk = conv_pick_partition(a, (size_t)M * C * 9 * 2, c.bm);
    const int lds_bytes = 2 * (c.bm * A_PITCH + 2 * g_hpc[pc].rows * g_hpc[pc].p * 16);
    dk_set_max_dynamic_lds((const void*)c.kernel[pc], lds_bytes);
    DkProfScope prof;
    dk_prof_begin(prof, st);
    hipLaunchKernelGGL(c.kernel[pc], dim3((unsigned)nblk), dim3((c.bm / c.wm) * (c.bn / c.wn) * 64),
        lds_bytes, st, a);
    CHECK_HIP(hipPeekAtLastError());
    if (prof.e0)
    {
      char nm[96];
      snprintf(nm, sizeof(nm), "conv3x3_direct_f16<%d, %d, %d, %d, %d, %d>", c.bm, c.bn, c.wm, c.wn, g_hpc[pc].p, g_hpc[pc].rows);
      dk_prof_end(prof, st, dk_prof_named_slot(nm), 2.0 * (double)M * C * 9 * (double)a.N / 1e9);
    }
  }
  return 0;
}

extern "C" int dk_conv_forward_half_packed(const DkConvDesc* d, const float* x, const void* packed_weights,
    const float* biases, float* y, const float* residual, void* stream)
{
  return dk_conv_forward_half_direct(d, x, packed_weights, biases, y, residual, stream, 0);
}
