// conv1x1_dma.hip -- 1x1 / stride 1 convolutions as a plain GEMM whose operand tiles
// stream HBM -> LDS by LDS-DMA (buffer_load ... lds) through a ring of stages (gfx950).
//
// Same contraction as conv_igemm_f32 (the reference's gemm_nn over the im2col matrix,
// which for a 1x1 kernel IS the input: src/convolutional_kernels.cu:486-494 skips
// im2col for size 1; src/gemm.c:2223-2239): Out[m][n] = sum_k W[m][k] X[k][n],
// m = filter, n = (image, pixel) with the batch folded in, k = input channel,
// v_mfma_f32_32x32x2_f32 with k ASCENDING -> bit-identical to the gather kernel.
//
// What differs from the gather kernel (conv_igemm.hip), and why:
//  * PERSISTENT workgroups with wave roles: 4 consumer waves (MFMA + epilogue) and one
//    LOADER wave that only issues LDS-DMA.  PMC on the one-tile-per-block kernels showed
//    nothing saturated on these layers (MFMA pipe 41-55 % busy, HBM ~45 %, VALU ~10 %): a
//    wave lives ~9-17 us for 1.7 us of MFMA issue, the rest being prologue index math,
//    first-load latency, barriers and the epilogue, none of which a short-lived block can
//    overlap with its own MFMAs.  Here a block walks many tiles; the loader runs NS-1 K
//    stages ahead ACROSS tile boundaries, so the next tile's operands land while the
//    consumers are in the epilogue, and with two blocks per CU one block's epilogue (VALU,
//    stores) runs beside the other's MFMAs,
//  * both operand tiles go HBM/L2 -> LDS with 16-byte LDS-DMA (no VGPR staging, no
//    ds_write); one s_barrier per K stage of 32 k; the loader's counted `s_waitcnt vmcnt`
//    keeps NS-2 stages in flight across it; consumers never wait on VMEM at all,
//  * A (weights [M][K], k contiguous) cannot be padded under LDS-DMA (a wave instruction
//    writes 1 KiB linearly), so its 16-byte granules are XOR-swizzled on the SOURCE
//    address: LDS granule (m, s) holds W[m][4*(s ^ ((m>>1)&7)) ..+3]; the fragment read
//    is one ds_read_b128 per row and 4 k (conflict-free: 16 lanes -> 16 slots), from
//    which lane half h takes k = 4g+h (step 1) and 4g+2+h (step 2).
//
// Eligibility (host): size 1, stride 1, pad 0, groups 1, K % 32 == 0, (oh*ow) % 4 == 0,
// 16-byte aligned x / weights; everything else keeps the gather kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"
#include "dark_hip.h"
#include "dk_device_math.h"
#include "dk_internal.h"

namespace
{
typedef __attribute__((address_space(3))) void lds_void;

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA wave instruction: 64 lanes x 16 bytes from buffer `rsrc` at voff (per lane) + soff
// (scalar) to LDS bytes [lds_addr, lds_addr + 1024), lane-linear.  Inline asm on purpose: hipcc
// counts the builtin form in its own vmcnt bookkeeping and drains it (vmcnt(0)) before the next
// ds_read of the same array, which would serialise the ring; here the kernel's counted waits are the
// only ones (cdna_hip_programming.md 5.7).  M0 (the LDS destination) is written and restored inside
// the statement; the nops cover the SGPR-write -> VMEM-read and M0-write hazards hipcc does not pad.
__device__ __forceinline__ void dma16(u32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
               "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
               : "memory");
}

__device__ __forceinline__ u32x4_t make_rsrc(const void* base, unsigned bytes)
{
  const unsigned long long a = (unsigned long long)base;
  u32x4_t r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);   // stride 0
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}
}  // namespace

template <int BM, int BN, int NWM, int NS>
__global__ void __launch_bounds__(320) conv1x1_dma_f32(const ConvArgs p)
{
  constexpr int BK = 32;
  constexpr int NWN = 4 / NWM;
  constexpr int WM = BM / NWM, WN = BN / NWN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BK * BN, STAGE = A_FLOATS + B_FLOATS;
  constexpr int JA = BM / 8;    // A DMA instructions per stage (BM*8 granules / 64 lanes), all by the loader wave
  constexpr int JB = BN / 8;    // B: 32 * BN/4 granules / 64
  constexpr int LW = JA + JB;   // vmcnt units per stage
  constexpr int Q = BN / 4;     // B granules per k row
  constexpr int RB = 64 / Q;    // k rows one B instruction covers
  static_assert(WM % 32 == 0 && WN % 32 == 0 && NS >= 3 && NS <= 4 && LW * (NS - 2) <= 63, "tile");
  static_assert(64 % Q == 0 && JA % 2 == 0, "DMA mapping");

  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = p.K, HW = p.OHW;
  const int nst = K / BK;
  const int nwork = p.nwork;
  const int bid = blockIdx.x, grid = gridDim.x;

  if (wave == 4)
  {
    // ------------------------------ loader wave ------------------------------------
    const u32x4_t xr = make_rsrc(p.x, p.x_bytes);
    const u32x4_t wr = make_rsrc(p.w, p.w_bytes);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
    // issue cursor: tile `iv` (virtual workgroup id), stage `is` of it, per-lane source offsets.
    // Rows >= M and pixels >= N lie beyond the end of their buffers: the descriptor's range
    // check returns zeros for them, no explicit masks.
    int iv = bid - grid, is = nst;
    unsigned voffA0 = 0, voffA1 = 0, voffB = 0;
    bool more = true;
    auto next_tile = [&]() {
      for (;;)
      {
        iv += grid;
        if (iv >= nwork)
        {
          more = false;
          return;
        }
        int g, tm, tn;
        if (!conv_block_tile_of(p, iv, nwork, g, tm, tn))
          continue;
        const int m0 = tm * BM, n0 = tn * BN;
        // LDS granule (row m, slot lane&7) <- source granule slot ^ ((m >> 1) & 7); rows 8j + (lane>>3)
        const int r = lane >> 3, sl = lane & 7;
        voffA0 = (unsigned)((m0 + r) * K + 4 * (sl ^ ((lane >> 4) & 7))) * 4u;
        voffA1 = (unsigned)((m0 + 8 + r) * K + 4 * (sl ^ ((4 + (lane >> 4)) & 7))) * 4u;
        const int n = n0 + 4 * (lane % Q);
        const int b = fdiv(n, HW, p.inv_OHW);
        const int pix = n - b * HW;
        voffB = (unsigned)(b * p.Ctot * HW + pix + (lane / Q) * HW) * 4u;
        is = 0;
        return;
      }
    };
    auto issue = [&](int slot) {
      const unsigned st = lds0 + (unsigned)(slot * STAGE) * 4u;
      const unsigned sa = (unsigned)is * (BK * 4u);
      const unsigned sb = (unsigned)(is * BK * HW) * 4u;
#pragma unroll
      for (int j = 0; j < JA; ++j)
        dma16(wr, st + (unsigned)j * 1024u, (j & 1) ? voffA1 : voffA0, sa + (unsigned)((j & ~1) * 8 * K) * 4u);
#pragma unroll
      for (int j = 0; j < JB; ++j)
        dma16(xr, st + A_FLOATS * 4u + (unsigned)j * 1024u, voffB, sb + (unsigned)(j * RB * HW) * 4u);
      ++is;
    };
    int issued = 0;   // stages issued so far (slot of the next one = issued % NS)
    int islot = 0;
    auto issue_next = [&]() {
      if (more && is == nst)
        next_tile();
      if (!more)
        return;
      issue(islot);
      islot = (islot + 1 == NS) ? 0 : islot + 1;
      ++issued;
    };
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue_next();
    // stage g of this block's sequence: wait until it has landed, meet the consumers, refill the
    // slot they have just left (stage g - 1's)
    for (int g = 0; g < issued; ++g)
    {
      if (issued - (g + 1) >= NS - 2)
        wait_vmcnt<LW*(NS - 2)>();
      else
        wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");
      issue_next();
    }
    return;
  }

  // ------------------------------ consumer waves -------------------------------------
  const int wm = wave / NWN, wn = wave % NWN;
  const int l31 = lane & 31, lh = lane >> 5;
  const int sw = (l31 >> 1) & 7;
  const int a_lane = (wm * WM + l31) * BK;
  const int b_lane = lh * BN + wn * WN + l31;
  int slot = 0;
  for (int v = bid; v < nwork; v += grid)
  {
    int g_unused, tile_m, tile_n;
    if (!conv_block_tile_of(p, v, nwork, g_unused, tile_m, tile_n))
      continue;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int s = 0; s < nst; ++s)
    {
      asm volatile("s_barrier" ::: "memory");   // the loader's stage has landed (its vmcnt wait precedes its barrier)
      const float* As = lds + slot * STAGE + a_lane;
      const float* Bs = lds + slot * STAGE + A_FLOATS + b_lane;
      // MFMA step contracts k = 2t (lanes 0-31) and 2t+1 (lanes 32-63), t ascending; the
      // fragments of granule g+1 (4 k) are read while the MFMAs of granule g run
      float4 a4[2][TM];
      float b4[2][2][TN];
      auto fetch = [&](int g, float4(&a)[TM], float(&b)[2][TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const float4*)(As + i * 32 * BK + ((g ^ sw) << 2));
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int j = 0; j < TN; ++j) b[h][j] = Bs[(4 * g + 2 * h) * BN + j * 32];
      };
      fetch(0, a4[0], b4[0]);
#pragma unroll
      for (int g = 0; g < BK / 4; ++g)
      {
        if (g + 1 < BK / 4)
          fetch(g + 1, a4[(g + 1) & 1], b4[(g + 1) & 1]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
        {
          float a[TM];
#pragma unroll
          for (int i = 0; i < TM; ++i)
          {
            const float4 q = a4[g & 1][i];
            a[i] = (h == 0) ? (lh ? q.y : q.x) : (lh ? q.w : q.z);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b4[g & 1][h][j], acc[i][j], 0, 0, 0);
        }
      }
      slot = (slot + 1 == NS) ? 0 : slot + 1;
    }
    conv_epilogue<BM, BN, WM, WN, TM, TN>(p, acc, tile_m * BM, tile_n * BN, 0, wm, wn, l31, lh);
  }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
namespace
{
struct DmaCfg
{
  int bm, bn, ns;
  const char* name;
  const char* kname;
  void (*kernel)(const ConvArgs);
};

#define DK_MCFG(BM, BN, NWM, NS)                                                         \
  {                                                                                      \
    BM, BN, NS, "dma1x1_" #BM "x" #BN "_s" #NS, "conv1x1_dma_f32<" #BM ", " #BN ", " #NWM ", " #NS ">", \
        conv1x1_dma_f32<BM, BN, NWM, NS>                                                 \
  }

const DmaCfg g_mcfgs[] = {
    DK_MCFG(64, 64, 2, 3),
    DK_MCFG(64, 64, 2, 4),
    DK_MCFG(64, 128, 2, 3),
    DK_MCFG(64, 128, 2, 4),
    DK_MCFG(128, 64, 2, 3),
    DK_MCFG(128, 128, 2, 3),
};
const int g_nmcfg = sizeof(g_mcfgs) / sizeof(g_mcfgs[0]);
}  // namespace

int dk_conv_dma1x1_num_configs() { return g_nmcfg; }

const char* dk_conv_dma1x1_config_name(int c) { return (c >= 0 && c < g_nmcfg) ? g_mcfgs[c].name : nullptr; }

const char* dk_conv_dma1x1_kernel_name(int c) { return (c >= 0 && c < g_nmcfg) ? g_mcfgs[c].kname : nullptr; }

int dk_conv_dma1x1_bm(int c) { return (c >= 0 && c < g_nmcfg) ? g_mcfgs[c].bm : 0; }

// x / weights may be NULL (shape-only query)
bool dk_conv_dma1x1_applicable(const DkConvDesc* d, const float* x, const float* weights, int c)
{
  if (c < 0 || c >= g_nmcfg)
    return false;
  if (d->size != 1 || d->stride_x != 1 || d->stride_y != 1 || d->pad != 0 || d->groups != 1)
    return false;
  if (d->c % 32 != 0 || (d->h * d->w) % 4 != 0)
    return false;
  if ((x && ((uintptr_t)x & 15)) || (weights && ((uintptr_t)weights & 15)))
    return false;
  return true;
}

// Launches one chunk; `a` was filled by dk_conv_forward_cfg.
void dk_conv_dma1x1_launch(ConvArgs a, int c, hipStream_t st)
{
  const DmaCfg& cf = g_mcfgs[c];
  a.tiles_m = (a.M + cf.bm - 1) / cf.bm;
  a.tiles_n = (a.N + cf.bn - 1) / cf.bn;
  a.groups = 1;
  a.mode = 0;
  const int bytes = cf.ns * (cf.bm * 32 + 32 * cf.bn) * (int)sizeof(float);
  dk_set_max_dynamic_lds((const void*)cf.kernel, bytes);
  conv_args_finish(a);
  const long long nblk = conv_pick_partition(a, (size_t)a.M * a.K * sizeof(float), cf.bm);
  a.nwork = (int)nblk;
  // persistent grid: `bpc` workgroups per CU (a multiple of 8 workgroups, so that a workgroup's
  // virtual ids v = bid + i*grid keep its XCD label v & 7), each walking nwork/grid tiles
  static int bpc = getenv("DK_PERS_BPC") ? atoi(getenv("DK_PERS_BPC")) : 2;
  long long grid = 256LL * (bpc > 0 ? bpc : 2);
  if (grid > nblk)
    grid = (nblk + 7) / 8 * 8;
  hipLaunchKernelGGL(cf.kernel, dim3((unsigned)grid), dim3(320), bytes, st, a);
}
