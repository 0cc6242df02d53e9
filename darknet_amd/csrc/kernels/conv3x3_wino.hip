// conv3x3_wino.hip -- 3x3 / stride 1 / pad 1 convolution by Winograd's minimal filtering
// F(2x2, 3x3), fused into ONE kernel on the fp32 MFMA pipe (gfx950).
//
// What it computes is the reference's convolutional forward for such a layer
// (src/convolutional_layer.cpp:1128-1305: im2col + gemm_nn + bias + activation, optionally the
// following linear [shortcut]); what differs is the arithmetic: Y = A^T [ (G g G^T) .* (B^T d B) ] A
// needs 16 multiplications per (filter, channel, 2x2 output tile) where the direct contraction
// needs 36, i.e. 2.25x fewer MFMA FLOPs -- the measured practical ceiling of the fp32 MFMA pipe on
// this part (137 TFLOP/s) is otherwise the limit of the direct kernel (conv3x3_direct.hip runs at
// 90 % of it).  The reference's own GPU build makes the same trade: cuDNN's `cudnn_fastest`
// search picks Winograd algorithms for these layers (src/convolutional_layer.cpp:216-290).
// The result is NOT bitwise equal to the k-ascending fmaf chain of the other kernels; it stays
// within the fp32 tolerance of tests/util.py (measured in tests/test_gpu_ops.py).
//
// Decomposition: the 16 positions xi = (i, j) of the 4x4 transformed tile are 16 independent GEMMs
//     Mx[xi][m][t] = sum_c U[xi][m][c] * V[xi][c][t]      m = filter, t = output tile (b, ty, tx)
// A workgroup (4 waves) owns 64 filters x 64 consecutive tiles for ALL 16 positions; wave (wm, wn)
// owns a 32 x 32 sub-block, 16 accumulators of v_mfma_f32_32x32x2_f32 = 256 registers: the C/D
// layout puts the 16 positions of one (m, t) in the SAME lane and register index, so the output
// transform A^T Mx A is pure per-lane register arithmetic -- no exchange through LDS.
//
// K loop, 8 input channels per stage (64 MFMAs per wave), two barriers per stage:
//   * U (filters transformed once per layer by dk_conv_wino_transform_weights) lies in HBM as one
//     contiguous 32 KB slab per (filter tile, stage) in exactly the LDS image, so staging is a
//     straight 16-byte copy;
//   * the input rows the strip of 64 tiles touches are staged ONCE per stage as a raw patch in
//     LDS with coalesced 16 / 8 / 4-byte loads (template VW = 4 / 2 / 1 by the row alignment W
//     allows): per channel, the 4 input rows of every tile row of the strip, pitch Pw.  (The first
//     version gathered each tile's 4x4 patch straight from global memory with 32 scalar loads per
//     thread and stage: the address path of those uncoalesced loads, not their bytes, cost 30 % of
//     the kernel -- ablation builds)
//   * V = B^T d B is produced from the raw patch by VALU work that is INTERLEAVED with the MFMAs
//     of the previous stage (one wave per SIMD: nothing else would overlap them), into the other
//     half of a double-buffered V image;
//   * operands are read back as ds_read_b128: lanes 0-31 get channels 0-3 of the stage, lanes
//     32-63 channels 4-7, so one read feeds four MFMAs (k pairs (j, 4+j)); the LDS image
//     [xi][sub-block][half][32 rows][4 floats] makes both the b128 reads and the transform's
//     b32 writes bank-conflict free (MI355X_MICROARCH.md, LDS table).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <mutex>
#include <unordered_map>

#include "conv_common.h"
#include "dark_hip.h"
#include "dk_device_math.h"
#include "dk_internal.h"

// diagnostic ablation builds (tools/build_ablate_wino.sh; results are garbage by construction):
// bit 0 no input DMA, 1 no filter DMA, 2 no input transform / V writes, 3 no wait for the DMA (exposed latency?)
#ifndef DK_WABL
#define DK_WABL 0
#endif
// scheduling fences around the MFMA rounds (bit 4 of DK_WABL compiles them out: measurement of their worth)
#if DK_WABL & 16
#define DK_WINO_SB
#else
#define DK_WINO_SB __builtin_amdgcn_sched_barrier(0)
#endif

// diagnostic build -DDK_WSTAMP=1 (tools/build_ablate_wino.sh): lane 0 of every wave of the first 64 workgroups
// stamps s_memtime at the phase boundaries of the SCHED-0 kernel into g_wstamp (read with dk_wino_stamps_read);
// never compiled into the product library
#ifndef DK_WSTAMP
#define DK_WSTAMP 0
#endif
#if DK_WSTAMP
__device__ long long g_wstamp[64 * 8 * 160];
#define WSTAMP(idx)                                                                              \
  do {                                                                                           \
    if (blockIdx.x < 64 && (threadIdx.x & 63) == 0 && (idx) < 160)                               \
      g_wstamp[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 160 + (idx)] = (long long)__builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define WSTAMP(idx)
#endif

namespace
{
constexpr int WBM = 64;              // filters per workgroup
constexpr int WBN = 64;              // output tiles (2x2 pixels each) per workgroup
constexpr int WCK = 4;               // input channels per stage
constexpr int W_STAGE = 16 * 64 * WCK;  // floats of one operand's stage image (16 KB)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c)
{
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// element offset of (xi, sub-block, row r32, channel c4) inside a stage image:
// [xi][sub-block][channel pair][32 rows][2 floats] -- a lane's MFMA operands of one position are one ds_read_b64
__host__ __device__ __forceinline__ int img_off(int xi, int sub, int r32, int c4)
{
  return xi * 256 + sub * 128 + (c4 >> 1) * 64 + r32 * 2 + (c4 & 1);
}

// LDS-DMA pieces per thread and stage of the raw patch the kernel is compiled for (host: wino_geometry)
constexpr int wino_kmax(int vw) { return vw == 4 ? 6 : 18; }

// One LDS-DMA wave instruction: 64 lanes x 16 (4) bytes from buffer `rsrc` at byte offset voff (per lane;
// out of range = dropped) to LDS bytes [lds_addr, lds_addr + 1024 (256)), lane-linear.  Inline asm on
// purpose (as in conv1x1_dma.hip): hipcc must not count these in its own vmcnt bookkeeping, the kernel's
// counted waits are the only ones.  M0 (the LDS destination) is written and restored inside the statement.
__device__ __forceinline__ void dma16(u32x4_t rsrc, unsigned lds_addr, unsigned voff)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
               "buffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rsrc)
               : "memory");
}
__device__ __forceinline__ void dma4(u32x4_t rsrc, unsigned lds_addr, unsigned voff)
{
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\t"
               "buffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "s"(lds_addr), "v"(voff), "s"(rsrc)
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
// s_waitcnt vmcnt(n) for a wave-uniform n (the immediate must be a constant)
__device__ __forceinline__ void wait_vmcnt_n(int n)
{
#define DK_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n)
  {
    DK_W(0) DK_W(1) DK_W(2) DK_W(3) DK_W(4) DK_W(5) DK_W(6) DK_W(7) DK_W(8) DK_W(9) DK_W(10) DK_W(11) DK_W(12)
    DK_W(13) DK_W(14) DK_W(15) DK_W(16) DK_W(17) DK_W(18) DK_W(19) DK_W(20) DK_W(21) DK_W(22) DK_W(23) DK_W(24)
    DK_W(25) DK_W(26) DK_W(27) DK_W(28) DK_W(29) DK_W(30) DK_W(31) DK_W(32) DK_W(33) DK_W(34) DK_W(35) DK_W(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef DK_W
}
// workgroup barrier that leaves LDS-DMA in flight (a __syncthreads() would drain vmcnt)
__device__ __forceinline__ void barrier_lds()
{
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- output transform A^T Mx A across the two waves that share a sub-block, bias / activation / residual, stores ---
// Rows 2 xh, 2 xh + 1 of the 4x4 position grid are in this wave: t_q[i] = row sums (q = 0: M0 + M1 + M2,
// q = 1: M1 - M2 - M3).  Y0q = (t_q[0] + t_q[1]) + t_q[2], Y1q = (t_q[1] - t_q[2]) - t_q[3]: wave xh = 0 holds the
// bracketed parts, wave xh = 1 holds t_q[2], t_q[3].  Each wave finishes HALF of the 16 C/D registers (xh = 0:
// r < 8, xh = 1: r >= 8) and hands the partner, through LDS, the four partial values per register of the other half.
// Shared by both kernels of this file (same wave -> sub-block / position-row ownership).
template <bool PAIR>
__device__ __forceinline__ void wino_epilogue(const ConvArgs& p, f32x16 (&acc)[8], float* lds, int m0, int n0, int wave,
    int lane)
{
  const int l31 = lane & 31, lh = lane >> 5;
  const int xh = wave >> 2, wq = wave & 3;
  const int wm = wq & 1, wn = wq >> 1;
  const int TW = p.tiles_w, THW = p.tiles_hw;
  // ---- epilogue addressing, and every global LOAD of the epilogue issued NOW: bias and residual values are in flight
  // while the row sums are formed and exchanged.  (Inside the store loop they were serialised: the compiler may not
  // move a load above the previous register's store -- the buffers could alias -- so each of the 8 C/D registers paid a
  // full global-load latency: 22 000 of a workgroup's 158 000 cycles on [128->128 76x76], stamped build.)
  const int n = n0 + wn * 32 + l31;
  const bool nv = n < p.N;
  const int nn = nv ? n : 0;
  const int b = fdiv(nn, THW, p.inv_tiles_hw);
  const int rr_ = nn - b * THW;
  const int ty = fdiv(rr_, TW, p.inv_tiles_w);
  const int tx = rr_ - ty * TW;
  const int oy = 2 * ty, ox = 2 * tx;
  const bool row1 = oy + 1 < p.OH, col1 = ox + 1 < p.OW;
  __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.y_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)p.residual, 0, p.residual ? p.y_bytes : 0u, 0x00020000);
  __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.w), 0,
      p.bias ? (unsigned)p.M * 4u : 0u, 0x00020000);
  const bool has_res = p.residual != nullptr;
  const bool quad = p.wino_quad != 0;
  const bool odd = lane & 1;
  const int act = p.act;
  const unsigned row_bytes = (unsigned)p.OHW * 4u;
  const unsigned pbase = nv ? (unsigned)(b * p.Mtot * p.OHW + oy * p.OW + ox) * 4u : 0u;
  const unsigned o00 = nv ? pbase : 0xFFFFFFF0u;
  const unsigned o01 = (nv && col1) ? pbase + 4u : 0xFFFFFFF0u;
  const unsigned o10 = (nv && row1) ? pbase + (unsigned)p.OW * 4u : 0xFFFFFFF0u;
  const unsigned o11 = (nv && row1 && col1) ? pbase + (unsigned)p.OW * 4u + 4u : 0xFFFFFFF0u;
  // per C/D register rr8 of this wave's half: byte offsets of its stores (quad: one 16-byte group; pair: two 8-byte
  // groups; else four pixels) and the values loaded from them
  const unsigned oq = nv ? (odd ? o10 - 8u : o00) : 0xFFFFFFF0u;
  const int mbase = m0 + wm * 32 + 4 * lh;
  float bvv[8];
  unsigned resv[8][4];
#pragma unroll
  for (int rr8 = 0; rr8 < 8; ++rr8)
  {
    const int r = xh * 8 + rr8;
    const int m = mbase + (r & 3) + 8 * (r >> 2);
    bvv[rr8] = ld_buf(br, (unsigned)m * 4u);   // no bias: zero-record descriptor, the load returns 0
    const unsigned mo = (unsigned)m * row_bytes;
    if (has_res)
    {
      if (PAIR && quad)
      {
        const u32x4_t rv = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)(oq == 0xFFFFFFF0u ? oq : oq + mo), 0, 0);
        resv[rr8][0] = rv.x; resv[rr8][1] = rv.y; resv[rr8][2] = rv.z; resv[rr8][3] = rv.w;
      }
      else if (PAIR)
      {
        const u32x2 ra = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)(nv ? o00 + mo : 0xFFFFFFF0u), 0, 0);
        const u32x2 rb = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)((nv && row1) ? o10 + mo : 0xFFFFFFF0u), 0, 0);
        resv[rr8][0] = ra.x; resv[rr8][1] = ra.y; resv[rr8][2] = rb.x; resv[rr8][3] = rb.y;
      }
      else
      {
        resv[rr8][0] = __float_as_uint(ld_buf(rr, o00 == 0xFFFFFFF0u ? o00 : o00 + mo));
        resv[rr8][1] = __float_as_uint(ld_buf(rr, o01 == 0xFFFFFFF0u ? o01 : o01 + mo));
        resv[rr8][2] = __float_as_uint(ld_buf(rr, o10 == 0xFFFFFFF0u ? o10 : o10 + mo));
        resv[rr8][3] = __float_as_uint(ld_buf(rr, o11 == 0xFFFFFFF0u ? o11 : o11 + mo));
      }
    }
    else
      resv[rr8][0] = resv[rr8][1] = resv[rr8][2] = resv[rr8][3] = 0u;
  }
  WSTAMP(5);
  barrier_lds();   // every MFMA operand read is done: the rings' LDS is free
  float* const Ex = lds;   // [8 waves][8 registers][4 values][64 lanes]
  float part[16][4];
#pragma unroll
  for (int r = 0; r < 16; ++r)
  {
    float t0[2], t1[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
    {
      t0[i] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
      t1[i] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
    }
    if (xh == 0)
    {
      part[r][0] = t0[0] + t0[1];   // Y00 without t0[2]
      part[r][1] = t1[0] + t1[1];   // Y01 without t1[2]
      part[r][2] = t0[1];           // Y10 before - t0[2] - t0[3]
      part[r][3] = t1[1];           // Y11 before - t1[2] - t1[3]
    }
    else
    {
      part[r][0] = t0[0];           // t0[2] of the 4x4 grid
      part[r][1] = t1[0];           // t1[2]
      part[r][2] = t0[1];           // t0[3]
      part[r][3] = t1[1];           // t1[3]
    }
  }
  {
    float* const mine = Ex + (size_t)(wave * 8) * 4 * 64 + lane;
    const int rbase = xh == 0 ? 8 : 0;   // the half the PARTNER finishes
#pragma unroll
    for (int rr = 0; rr < 8; ++rr)
#pragma unroll
      for (int k = 0; k < 4; ++k) mine[(rr * 4 + k) * 64] = xh == 0 ? part[8 + rr][k] : part[rr][k];
    (void)rbase;
  }
  WSTAMP(6);
  barrier_lds();
  const float* const theirs = Ex + (size_t)((wave ^ 4) * 8) * 4 * 64 + lane;

  // the activation is a launch constant: dispatched once into straight-line code
  auto emit = [&](auto actc) {
    constexpr int A = decltype(actc)::value;
#pragma unroll
    for (int rr8 = 0; rr8 < 8; ++rr8)
    {
      const int r = xh * 8 + rr8;     // xh is wave-uniform: both halves are compiled, one runs
      float q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) q[k] = theirs[(rr8 * 4 + k) * 64];
      float y00, y01, y10, y11;
      if (xh == 0)
      {
        y00 = part[rr8][0] + q[0];
        y01 = part[rr8][1] + q[1];
        y10 = part[rr8][2] - q[0] - q[2];
        y11 = part[rr8][3] - q[1] - q[3];
      }
      else
      {
        y00 = q[0] + part[8 + rr8][0];
        y01 = q[1] + part[8 + rr8][1];
        y10 = q[2] - part[8 + rr8][0] - part[8 + rr8][2];
        y11 = q[3] - part[8 + rr8][1] - part[8 + rr8][3];
      }
      const int m = mbase + (r & 3) + 8 * (r >> 2);
      const float bv = bvv[rr8];
      y00 = dk_activate(y00 + bv, A < 0 ? act : A);
      y01 = dk_activate(y01 + bv, A < 0 ? act : A);
      y10 = dk_activate(y10 + bv, A < 0 ? act : A);
      y11 = dk_activate(y11 + bv, A < 0 ? act : A);
      const unsigned mo = (unsigned)m * row_bytes;
      if (PAIR && quad)
      {
        // OW a multiple of 4, OH even: lanes 2j / 2j + 1 hold horizontally adjacent tiles (tile columns 2i, 2i + 1) whose
        // four pixels per row are one aligned 16-byte group.  The even lane takes row 0 of both tiles, the odd lane row 1
        // (one DPP swap of two registers), so every lane issues ONE 16-byte store per C/D register instead of two 8-byte
        // ones (cdna_hip_programming.md T21).
        const float s0 = odd ? y00 : y10, s1 = odd ? y01 : y11;
        const float r0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s0), 0xB1, 0xF, 0xF, true));
        const float r1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s1), 0xB1, 0xF, 0xF, true));
        float v0 = odd ? r0 : y00, v1 = odd ? r1 : y01, v2 = odd ? y10 : r0, v3 = odd ? y11 : r1;
        if (has_res)
        {
          v0 += __uint_as_float(resv[rr8][0]);
          v1 += __uint_as_float(resv[rr8][1]);
          v2 += __uint_as_float(resv[rr8][2]);
          v3 += __uint_as_float(resv[rr8][3]);
        }
        u32x4_t vv;
        vv.x = __float_as_uint(v0); vv.y = __float_as_uint(v1); vv.z = __float_as_uint(v2); vv.w = __float_as_uint(v3);
        __builtin_amdgcn_raw_buffer_store_b128(vv, yr, (int)(oq == 0xFFFFFFF0u ? oq : oq + mo), 0, 0);
      }
      else
      {
        if (has_res)
        {
          y00 += __uint_as_float(resv[rr8][0]);
          y01 += __uint_as_float(resv[rr8][1]);
          y10 += __uint_as_float(resv[rr8][2]);
          y11 += __uint_as_float(resv[rr8][3]);
        }
        if (PAIR)
        {
          // OW even: both pixels of a row exist and the pair is 8-byte aligned
          u32x2 v0, v1;
          v0.x = __float_as_uint(y00); v0.y = __float_as_uint(y01);
          v1.x = __float_as_uint(y10); v1.y = __float_as_uint(y11);
          __builtin_amdgcn_raw_buffer_store_b64(v0, yr, (int)(nv ? o00 + mo : 0xFFFFFFF0u), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b64(v1, yr, (int)((nv && row1) ? o10 + mo : 0xFFFFFFF0u), 0, 0);
        }
        else
        {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y00), yr, (int)(o00 == 0xFFFFFFF0u ? o00 : o00 + mo), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y01), yr, (int)(o01 == 0xFFFFFFF0u ? o01 : o01 + mo), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y10), yr, (int)(o10 == 0xFFFFFFF0u ? o10 : o10 + mo), 0, 0);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y11), yr, (int)(o11 == 0xFFFFFFF0u ? o11 : o11 + mo), 0, 0);
        }
      }
    }
  };
  if (act == DK_LINEAR)
    emit(std::integral_constant<int, DK_LINEAR>());
  else if (act == DK_LEAKY)
    emit(std::integral_constant<int, DK_LEAKY>());
  else if (act == (DK_MISH | DK_ACT_FAST))
    emit(std::integral_constant<int, (DK_MISH | DK_ACT_FAST)>());
  else
    emit(std::integral_constant<int, -1>());
}

// SCHED selects the K loop's schedule (both compute the same sums in the same order):
//   0  every wave reads the 16 operand fragments of a stage behind the stage's barrier, then issues its 16 MFMAs;
//      the LDS-DMA waves issue their pieces in one burst behind the barrier (round 2)
//   1  software-pipelined across the barrier: a stage's second half (positions 4-7 of the wave) is multiplied BEHIND
//      the next barrier, from fragments read before it, while the first half's fragments of the new stage are in
//      flight -- no wave waits for LDS right behind a barrier.  Within 1-3 % of schedule 0 on every shape; the tuner
//      picks it for the short-K layers (C = 32 / 64).
// Round 3 also measured, and dropped (`git log` for the code): the LDS-DMA pieces spread between the
// MFMAs; complementary phases for the two waves of a SIMD (transform first / multiply first); s_setprio for either
// half; a symmetric kernel in which every wave issues a share of the pieces and transforms half of V, the halves
// staggered by half a stage (10-18 % SLOWER); the filters read from L2 straight into the operand registers instead of
// through the LDS ring (11-18 % slower).  None beat schedule 0.
template <int VW, bool PAIR, int SCHED>
__global__ void __launch_bounds__(512) conv3x3_wino_f32(const ConvArgs p)
{
  constexpr int KMAX = wino_kmax(VW);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = p.wino_ring;             // ring depth (3 or 4: what fits the 160 KB), prefetch distance D - 1 stages
  float* const Us = lds;                       // [D][W_STAGE]  transformed filters, LDS-DMA ring
  float* const Vs = lds + D * W_STAGE;         // [2][W_STAGE]  transformed input, double-buffered
  float* const Rs = lds + (D + 2) * W_STAGE;   // [D][RAWF]     raw input rows, LDS-DMA ring: [4 ch][RS rows][Pw]

  int g, tile_m, tile_n;
  if (!conv_block_tile(p, g, tile_m, tile_n))
    return;
  WSTAMP(0);
  // 8 waves = TWO per SIMD (256 registers each): wave (xh, wn, wm) owns the 32 x 32 sub-block (wm, wn) for the
  // position rows 2 xh, 2 xh + 1 (8 of the 16 positions, 8 accumulators = 128 AGPRs).  The second wave on a SIMD
  // hides what one 512-register wave could not: LDS latency, the barrier, the transform slices.  Waves 0-3 also
  // produce V (input transform), waves 4-7 also issue the LDS-DMA pieces.
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int xh = wave >> 2, wq = wave & 3;
  const int wm = wq & 1, wn = wq >> 1;
  const bool xf_wave = wave < 4, dma_wave = wave >= 4;
  const int t256 = wq * 64 + lane;             // index inside the 256-thread half this wave belongs to
  const int m0 = tile_m * WBM, n0 = tile_n * WBN;
  const int nst = p.C / WCK;
  const int TW = p.tiles_w, THW = p.tiles_hw, TH = p.wino_th;
  const int GP = p.wino_gp, RS = p.wino_rs, NK = p.wino_nk;
  const int Pw = VW * GP;
  const int RAWF = NK * 256 * VW;

  // 16-byte pieces may straddle the end of a row whose length is not a multiple of 4 floats: the last piece of the
  // tensor then reads up to 12 bytes past it (masked in the transform; device arrays carry 64 bytes of slack,
  // dark_hip.cpp: cuda_make_array) -- the descriptor must not drop that piece as out of range
  const u32x4_t xr = make_rsrc(p.x, p.x_bytes + (VW == 4 ? 16u : 0u));
  const u32x4_t ur = make_rsrc(p.w, p.w_bytes);
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) float*)lds);
  const unsigned u_lds = lds0 + (unsigned)(wq * 64) * 16u;                               // + slot * 16 KB + j * 4 KB
  const unsigned r_lds = lds0 + (unsigned)((D + 2) * W_STAGE) * 4u + (unsigned)(wq * 64 * VW) * 4u;   // + slot * RAWF * 4 + k * 256 * VW * 4

  // The filters of the first two stages depend on nothing computed below: their LDS-DMA goes out FIRST, so its latency
  // runs under the index arithmetic, the piece offsets and the zeroing of the raw ring (round 3: the prologue was 9 400
  // of a [128->128 76x76] workgroup's 145 000 cycles, 2 200 of them waiting for exactly these pieces).  Issue order of
  // the prologue is then U(0) U(1) raw(0) raw(1) raw(2) instead of raw(0) {U(0) raw(1)} {U(1) raw(2)}; only the first
  // counted wait of the K loop differs.  (Ring depth 3 only; the 4-slot diagnostic variant keeps the old order.)
  const bool early_u = D == 3 && !(DK_WABL & 2);
  const unsigned ubase = (unsigned)(tile_m * nst) * (unsigned)(W_STAGE * 4) + (unsigned)t256 * 16u;
  if (early_u && dma_wave)
  {
#pragma unroll
    for (int s = 0; s < 2; ++s)
      if (s < nst)
      {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          dma16(ur, u_lds + (unsigned)(s * W_STAGE) * 4u + (unsigned)j * 4096u,
              ubase + (unsigned)s * (unsigned)(W_STAGE * 4) + (unsigned)j * 4096u);
      }
  }

  // ---- geometry of the strip: tile rows R0 .. Rlast (R = b * TH + ty), first tile column tx0 ----
  const int R0 = fdiv(n0, TW, p.inv_tiles_w);
  const int tx0 = n0 - R0 * TW;
  const int nlast = (n0 + WBN - 1 < p.N) ? n0 + WBN - 1 : p.N - 1;
  const int Rlast = fdiv(nlast, TW, p.inv_tiles_w);
  const bool wide = TW > WBN;   // then the strip spans at most two tile rows and row 0 starts at tx0
  // first loaded column group of local tile row r: floor((2 * txs - 1) / VW), txs = first tile column staged
  auto g0_of = [&](int r) { const int txs = (wide && r == 0) ? tx0 : 0; return (2 * txs - 1 + VW) / VW - 1; };

  // ---- raw patch pieces (DMA waves): element e = t256 + 256 * k is column group g of row (c, r, i) ---------
  // Padding and ragged edges get an out-of-range source offset: the DMA drops them and the zeros written
  // once below stay (the geometry of a piece does not depend on the stage).
  unsigned xoff[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k)
  {
    const int e = t256 + 256 * k;
    const int c = fdiv(e, RS * GP, p.inv_wino_rsg);
    const int rem = e - c * RS * GP;
    const int rowi = fdiv(rem, GP, p.inv_wino_gp);
    const int gg = rem - rowi * GP;
    const int r = rowi >> 2, i = rowi & 3;
    const int R = R0 + r;
    const int b = fdiv(R, TH, p.inv_wino_th);
    const int ty = R - b * TH;
    const int iy = 2 * ty - 1 + i;
    const int col0 = VW * (g0_of(r) + gg);
    // a piece is fetched when it INTERSECTS its row (its columns outside the row are masked by the transform)
    const bool ok = dma_wave && k < NK && c < WCK && R <= Rlast && iy >= 0 && iy < p.H && col0 > -VW && col0 < p.W;
    xoff[k] = ok ? (unsigned)(((b * p.Ctot + c) * p.H + iy) * p.W + col0) * 4u : OOB;
  }

  // ---- input-transform ownership (waves 0-3): one (channel, tile) pair per thread ----------------
  // lane -> channel (lane & 1) + 2 * (lane >> 5), tile (lane >> 1) & 15 of the wave's 16 tiles: the 32
  // lanes of a half-wave write 32 consecutive floats of every position (no bank conflicts).
  const int c4 = (lane & 1) + 2 * lh;
  int rsrc, vdst;
  unsigned cmask = 0;   // bit j: patch column 2 tx - 1 + j lies outside the row (left / right padding)
  {
    const int tl = wq * 16 + ((lane >> 1) & 15);
    const int n = (n0 + tl < p.N) ? n0 + tl : nlast;
    const int R = fdiv(n, TW, p.inv_tiles_w);
    const int tx = n - R * TW;
    const int r = R - R0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((unsigned)(2 * tx - 1 + j) >= (unsigned)p.W)
        cmask |= 1u << j;
    rsrc = (c4 * RS + 4 * r) * Pw + 2 * tx - 1 - VW * g0_of(r);
    vdst = img_off(0, tl >> 5, tl & 31, c4);
  }
  const unsigned stage_x_bytes = (unsigned)(WCK * p.H * p.W) * 4u;

  f32x16 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // zero the raw ring once (padding positions are never written again)
  for (int i = tid * 4; i < D * RAWF; i += 2048) *(float4*)(Rs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  WSTAMP(1);
  barrier_lds();
  WSTAMP(2);

  // LDS-DMA issue (waves 4-7): raw rows of stage s into ring slot s % D, filters of stage s into ring slot s % D
  auto issue_raw = [&](int s) {
    const unsigned xo = (unsigned)s * stage_x_bytes;
    const unsigned dst = r_lds + (unsigned)((s % D) * RAWF) * 4u;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < NK && !(DK_WABL & 1))
      {
        if constexpr (VW == 4)
          dma16(xr, dst + (unsigned)k * 4096u, xoff[k] + xo);
        else
          dma4(xr, dst + (unsigned)k * 1024u, xoff[k] + xo);
      }
  };
  auto issue_u = [&](int s) {
    const unsigned uo = ubase + (unsigned)s * (unsigned)(W_STAGE * 4);
    const unsigned dst = u_lds + (unsigned)((s % D) * W_STAGE) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (!(DK_WABL & 2))
        dma16(ur, dst + (unsigned)j * 4096u, uo + (unsigned)j * 4096u);
  };
  // bundle B(s) = {filters of stage s, raw rows of stage s + 1}: both are first needed in iteration s
  const int cnt_u = (DK_WABL & 2) ? 0 : 4, cnt_r = (DK_WABL & 1) ? 0 : NK;
  auto bundle_count = [&](int s) { return (s < nst ? cnt_u : 0) + (s + 1 < nst ? cnt_r : 0); };
  auto issue_bundle = [&](int s) {
    if (s < nst)
      issue_u(s);
    if (s + 1 < nst)
      issue_raw(s + 1);
  };

  // V = B^T d B of this thread's patch, cut into 8 slices: slices 0-3 request the four patch rows,
  // slice 4 applies (d B) to the rows, slices 4-7 finish one row of B^T (..) each and write its four positions.
  float dd[4][4], ww[4][4];
  auto tslice = [&](auto sc, const float* Rcur, float* Vnext) {
    constexpr int part = decltype(sc)::value;
    if (DK_WABL & 4)
      return;
    if constexpr (part < 4)
    {
      const float* const src = Rcur + rsrc + part * Pw;
      if constexpr (VW == 4)
      {
        // patch column 2 tx - 1 sits at an odd float: b32, aligned b64, b32
        const float d0 = src[0];
        const float2 mid = *(const float2*)(src + 1);
        const float d3 = src[3];
        // columns outside the row hold whatever the straddling piece brought along: they are padding
        dd[part][0] = (cmask & 1u) ? 0.f : d0;
        dd[part][1] = (cmask & 2u) ? 0.f : mid.x;
        dd[part][2] = (cmask & 4u) ? 0.f : mid.y;
        dd[part][3] = (cmask & 8u) ? 0.f : d3;
      }
      else
      {
        const float2 lo = *(const float2*)src, hi = *(const float2*)(src + 2);
        dd[part][0] = lo.x; dd[part][1] = lo.y; dd[part][2] = hi.x; dd[part][3] = hi.y;
      }
    }
    else
    {
      if constexpr (part == 4 || part == 8)   // (8: the column pass alone -- the prologue's second half of the waves)
      {
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
          ww[i][0] = dd[i][0] - dd[i][2];
          ww[i][1] = dd[i][1] + dd[i][2];
          ww[i][2] = dd[i][2] - dd[i][1];
          ww[i][3] = dd[i][1] - dd[i][3];
        }
      }
      if constexpr (part < 8)
      {
        constexpr int i = part - 4;
        float* const dst = Vnext + vdst + i * 4 * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const float v = i == 0 ? ww[0][j] - ww[2][j] : i == 1 ? ww[1][j] + ww[2][j] : i == 2 ? ww[2][j] - ww[1][j] : ww[1][j] - ww[3][j];
          dst[j * 256] = v;
        }
      }
    }
  };

  // ---- prologue: raw(0), B(0) .. B(D-2) in flight; V(0) from raw(0) ----------------------------
  auto raw_count = [&](int s) { return s < nst ? cnt_r : 0; };
  if (dma_wave)
  {
    if (early_u)
    {
      issue_raw(0);
      if (1 < nst) issue_raw(1);
      if (2 < nst) issue_raw(2);
      wait_vmcnt_n(raw_count(1) + raw_count(2));
    }
    else
    {
      issue_raw(0);
      issue_bundle(0);
      issue_bundle(1);
      if (D > 3)
        issue_bundle(2);
      wait_vmcnt_n(bundle_count(0) + bundle_count(1) + (D > 3 ? bundle_count(2) : 0));
    }
  }
  WSTAMP(3);
  barrier_lds();
  WSTAMP(4);
  // V(0): every wave holds the (channel, tile) mapping of its 256-thread half, so the halves share the work -- waves
  // 0-3 write V rows 0 and 1, waves 4-7 rows 2 and 3 (each reads the four raw rows and makes the column pass itself)
  tslice(std::integral_constant<int, 0>(), Rs, Vs);
  tslice(std::integral_constant<int, 1>(), Rs, Vs);
  tslice(std::integral_constant<int, 2>(), Rs, Vs);
  tslice(std::integral_constant<int, 3>(), Rs, Vs);
  if (xf_wave)
  {
    tslice(std::integral_constant<int, 4>(), Rs, Vs);
    tslice(std::integral_constant<int, 5>(), Rs, Vs);
  }
  else
  {
    tslice(std::integral_constant<int, 8>(), Rs, Vs);
    tslice(std::integral_constant<int, 6>(), Rs, Vs);
    tslice(std::integral_constant<int, 7>(), Rs, Vs);
  }

  if constexpr (SCHED == 0)
  {
    for (int t = 0; t < nst; ++t)
    {
      // B(t) has landed for the issuing wave; behind the barrier for every wave, and every wave has left
      // iteration t - 1 (V(t) complete; ring slots of U(t-1) and raw(t) free)
      WSTAMP(8 + 4 * t);      // arrived at the top of stage t (all MFMAs of t - 1 issued)
      if (dma_wave && !(DK_WABL & 8))
        wait_vmcnt_n((early_u && t == 0) ? raw_count(2) : bundle_count(t + 1) + (D > 3 ? bundle_count(t + 2) : 0));
      WSTAMP(9 + 4 * t);      // DMA of stage t landed (DMA waves)
      barrier_lds();
      WSTAMP(10 + 4 * t);     // released
      if (dma_wave)
        issue_bundle(t + D - 1);
      WSTAMP(11 + 4 * t);     // pieces issued
      const float2* const Ua = (const float2*)(Us + (t % D) * W_STAGE) + (xh * 8) * 128 + wm * 64 + lh * 32 + l31;
      const float2* const Va = (const float2*)(Vs + (t & 1) * W_STAGE) + (xh * 8) * 128 + wn * 64 + lh * 32 + l31;
      const float* const Rcur = Rs + ((t + 1) % D) * RAWF;
      float* const Vnext = Vs + ((t + 1) & 1) * W_STAGE;
      // ---- this wave's 8 positions x 2 k-pairs, four positions at a time with their MFMAs interleaved; the
      // fragments of the second group are requested before the MFMAs of the first are issued; behind each MFMA
      // round the transform waves place two slices of the NEXT stage's input transform (after the last stage
      // they work on stale rows into the unused V half: harmless).
      float2 fa[2][4], fb[2][4];
  #pragma unroll
      for (int gq = 0; gq < 2; ++gq)
  #pragma unroll
        for (int u = 0; u < 4; ++u)
        {
          fa[gq][u] = Ua[(gq * 4 + u) * 128];
          fb[gq][u] = Va[(gq * 4 + u) * 128];
        }
      auto group = [&](auto gc) {
        constexpr int grp = decltype(gc)::value;
        DK_WINO_SB;
  #pragma unroll
        for (int u = 0; u < 4; ++u) acc[grp * 4 + u] = mfma2(fa[grp][u].x, fb[grp][u].x, acc[grp * 4 + u]);
        if (xf_wave)
        {
          tslice(std::integral_constant<int, grp * 4 + 0>(), Rcur, Vnext);
          tslice(std::integral_constant<int, grp * 4 + 1>(), Rcur, Vnext);
        }
        DK_WINO_SB;
  #pragma unroll
        for (int u = 0; u < 4; ++u) acc[grp * 4 + u] = mfma2(fa[grp][u].y, fb[grp][u].y, acc[grp * 4 + u]);
        if (xf_wave)
        {
          tslice(std::integral_constant<int, grp * 4 + 2>(), Rcur, Vnext);
          tslice(std::integral_constant<int, grp * 4 + 3>(), Rcur, Vnext);
        }
        DK_WINO_SB;
      };
      group(std::integral_constant<int, 0>());
      group(std::integral_constant<int, 1>());
    }
  }
  else
  {
    // Pipelined schedule.  Register group 0 = positions 0-3 of the wave's 8, group 1 = positions 4-7.  Iteration t:
    //   barrier (stage t's operands complete; every read of stage t - 1 is done: its group-1 fragments were read
    //   before the barrier) -> request group 0 of stage t -> MFMAs of group 1 of stage t - 1 (no LDS wait: the
    //   fragments are in registers) -> request group 1 of stage t -> MFMAs of group 0 of stage t.
    float2 fa[2][4], fb[2][4];
    auto block = [&](auto gc, auto sc0, auto withx, const float* Rcur, float* Vnext) {
      constexpr int grp = decltype(gc)::value;      // register group multiplied here
      constexpr int s0 = decltype(sc0)::value;      // first of the four transform slices placed here
      constexpr bool XF = decltype(withx)::value;   // place transform slices (false: the drain block)
      DK_WINO_SB;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[grp * 4 + u] = mfma2(fa[grp][u].x, fb[grp][u].x, acc[grp * 4 + u]);
      if constexpr (XF)
        if (xf_wave)
        {
          tslice(std::integral_constant<int, s0 + 0>(), Rcur, Vnext);
          tslice(std::integral_constant<int, s0 + 1>(), Rcur, Vnext);
        }
      DK_WINO_SB;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[grp * 4 + u] = mfma2(fa[grp][u].y, fb[grp][u].y, acc[grp * 4 + u]);
      if constexpr (XF)
        if (xf_wave)
        {
          tslice(std::integral_constant<int, s0 + 2>(), Rcur, Vnext);
          tslice(std::integral_constant<int, s0 + 3>(), Rcur, Vnext);
        }
      DK_WINO_SB;
    };
    using std::integral_constant;
    for (int t = 0; t < nst; ++t)
    {
      WSTAMP(8 + 4 * t);
      if (dma_wave && !(DK_WABL & 8))
        wait_vmcnt_n((early_u && t == 0) ? raw_count(2) : bundle_count(t + 1) + (D > 3 ? bundle_count(t + 2) : 0));
      WSTAMP(9 + 4 * t);
      barrier_lds();
      WSTAMP(10 + 4 * t);
      if (dma_wave)
        issue_bundle(t + D - 1);
      const float2* const Ua = (const float2*)(Us + (t % D) * W_STAGE) + (xh * 8) * 128 + wm * 64 + lh * 32 + l31;
      const float2* const Va = (const float2*)(Vs + (t & 1) * W_STAGE) + (xh * 8) * 128 + wn * 64 + lh * 32 + l31;
      const float* const Rcur = Rs + ((t + 1) % D) * RAWF;
      float* const Vnext = Vs + ((t + 1) & 1) * W_STAGE;
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        fa[0][u] = Ua[u * 128];
        fb[0][u] = Va[u * 128];
      }
      if (t > 0)
        block(integral_constant<int, 1>(), integral_constant<int, 0>(), std::true_type(), Rcur, Vnext);
      else
      {
        // nothing to multiply yet: only the first half of the next stage's input transform
        if (xf_wave)
        {
          tslice(integral_constant<int, 0>(), Rcur, Vnext);
          tslice(integral_constant<int, 1>(), Rcur, Vnext);
          tslice(integral_constant<int, 2>(), Rcur, Vnext);
          tslice(integral_constant<int, 3>(), Rcur, Vnext);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        fa[1][u] = Ua[(4 + u) * 128];
        fb[1][u] = Va[(4 + u) * 128];
      }
      block(integral_constant<int, 0>(), integral_constant<int, 4>(), std::true_type(), Rcur, Vnext);
    }
    // drain: group 1 of the last stage
    block(integral_constant<int, 1>(), integral_constant<int, 0>(), std::false_type(), nullptr, nullptr);
  }

  wino_epilogue<PAIR>(p, acc, lds, m0, n0, wave, lane);
  WSTAMP(159);   // every store issued
#if DK_WSTAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  WSTAMP(7);
}

// U = G g G^T of one (filter m, channel c) with taps g[9], written in the kernel's slab order
// [filter tile][stage][xi][sub-block][half][32 rows][4 floats]; C = channels of the convolution
__device__ __forceinline__ void wino_transform_one(const float (&g)[9], float* __restrict__ U, int m, int c, int C)
{
  float gg[4][3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
  {
    const float g0 = g[j], g1 = g[3 + j], g2 = g[6 + j];
    gg[0][j] = g0;
    gg[1][j] = 0.5f * (g0 + g1 + g2);
    gg[2][j] = 0.5f * (g0 - g1 + g2);
    gg[3][j] = g2;
  }
  const int nst = C / WCK;
  float* const slab = U + ((size_t)(m / WBM) * nst + c / WCK) * W_STAGE;
  const int mm = m % WBM;
#pragma unroll
  for (int i = 0; i < 4; ++i)
  {
    const float u0 = gg[i][0];
    const float u1 = 0.5f * (gg[i][0] + gg[i][1] + gg[i][2]);
    const float u2 = 0.5f * (gg[i][0] - gg[i][1] + gg[i][2]);
    const float u3 = gg[i][2];
    slab[img_off(i * 4 + 0, mm >> 5, mm & 31, c % WCK)] = u0;
    slab[img_off(i * 4 + 1, mm >> 5, mm & 31, c % WCK)] = u1;
    slab[img_off(i * 4 + 2, mm >> 5, mm & 31, c % WCK)] = u2;
    slab[img_off(i * 4 + 3, mm >> 5, mm & 31, c % WCK)] = u3;
  }
}

__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int M, int C)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C)
    return;
  const int m = idx / C, c = idx - m * C;
  float g[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) g[t] = w[(size_t)idx * 9 + t];
  wino_transform_one(g, U, m, c, C);
}

// Every derived weight tensor of a training step in ONE launch (dk_train_prep_*): per task a block range;
// kinds 0-2 produce the data gradient's transposed weights (plain, rotated by 180 degrees, tap-major), kinds 3-4
// the Winograd filters of the forward convolution and of the data gradient (the latter straight from w: the
// convolution that computes the data gradient has filters w'[c][m][t] = w[m][c][8 - t]).
__global__ void __launch_bounds__(256) train_prep_kernel(const DkPrepTask* __restrict__ tasks, int ntasks)
{
  // the task whose block range holds this block (ranges ascend)
  int lo = 0, hi = ntasks - 1;
  while (lo < hi)
  {
    const int mid = (lo + hi + 1) >> 1;
    if (tasks[mid].first_block <= (int)blockIdx.x)
      lo = mid;
    else
      hi = mid - 1;
  }
  const DkPrepTask t = tasks[lo];
  const int lb = (int)blockIdx.x - t.first_block;
  const int M = t.M, C = t.C, ss = t.ss;
  if (t.kind <= 2)
  {
    const size_t total = (size_t)M * C * ss;
#pragma unroll
    for (int k = 0; k < 4; ++k)
    {
      const size_t i = ((size_t)lb * 4 + k) * 256 + threadIdx.x;
      if (i >= total)
        break;
      size_t src;
      if (t.kind == 2)
      {
        const int m = (int)(i % M);
        const size_t r = i / M;
        const int tap = (int)(r % ss), c = (int)(r / ss);
        src = ((size_t)m * C + c) * ss + tap;
      }
      else
      {
        const int tap = (int)(i % ss);
        const size_t r = i / ss;
        const int m = (int)(r % M), c = (int)(r / M);
        src = ((size_t)m * C + c) * ss + (t.kind == 1 ? ss - 1 - tap : tap);
      }
      t.dst[i] = t.w[src];
    }
    return;
  }
  const int idx = lb * 256 + (int)threadIdx.x;
  if (idx >= M * C)
    return;
  float g[9];
  if (t.kind == 3)
  {
    const int m = idx / C, c = idx - m * C;
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = t.w[(size_t)idx * 9 + k];
    wino_transform_one(g, t.dst, m, c, C);
  }
  else
  {
    // filter c of the data-gradient convolution, its channel m: idx = c * M + m
    const int c = idx / M, m = idx - c * M;
#pragma unroll
    for (int k = 0; k < 9; ++k) g[k] = t.w[((size_t)m * C + c) * 9 + (8 - k)];
    wino_transform_one(g, t.dst, c, m, M);
  }
}

bool shape_ok(const DkConvDesc* d)
{
  return d->size == 3 && d->stride_x == 1 && d->stride_y == 1 && d->dilation == 1 && d->pad == 1 &&
         d->groups == 1 && d->c % WCK == 0 && d->n % WBM == 0 && d->h >= 2 && d->w >= 2;
}

std::mutex g_reg_mu;
std::unordered_map<const float*, const float*> g_reg;  // layer weights (device) -> transformed filters
}  // namespace

// Raw-patch geometry of a launch (see the kernel): vector width of the row pieces, column groups
// per row (gp), rows per channel incl. bank padding (rs), DMA pieces per thread and stage (nk).
struct WinoGeo
{
  int vw, gp, rs, nk;
};

bool wino_geometry(int TW, int vw, WinoGeo& o)
{
  const int nr = 62 / TW + 2;                  // tile rows a strip of 64 consecutive tiles can touch
  const int nw = TW < WBN ? TW : WBN;          // tile columns staged per tile row
  int gp = 2 * nw / vw + 2;
  if (vw == 1 && (gp & 1))
    ++gp;                                      // even pitch: ds_read_b64 of the tile columns
  const int rows = 4 * nr;
  for (int pass = 0; pass < 2; ++pass)
  {
    int rs = rows;
    if (pass == 0)
    {
      // channel stride = 32 mod 64 floats: the two channels of a half-wave read disjoint banks
      rs = -1;
      for (int cand = rows; cand <= rows + 8; ++cand)
        if ((cand * vw * gp) % 64 == 32)
        {
          rs = cand;
          break;
        }
      if (rs < 0)
        continue;
    }
    const int nk = (WCK * rs * gp + 255) / 256;
    if (nk <= wino_kmax(vw))
    {
      o.vw = vw; o.gp = gp; o.rs = rs; o.nk = nk;
      return true;
    }
  }
  return false;
}

// 16-byte pieces whenever their geometry fits (rows of any length: a piece may straddle a row end, the transform
// masks the columns outside the row), else 4-byte pieces
bool wino_pick(int W, int TW, bool allow16, WinoGeo& o)
{
  (void)W;
  if (allow16 && wino_geometry(TW, 4, o))
    return true;
  return wino_geometry(TW, 1, o);
}

// configurations = K-loop schedules of the one 64 x 64 tile shape (template SCHED)
int dk_conv_wino_num_configs() { return 2; }
const char* dk_conv_wino_config_name(int c)
{
  static const char* names[2] = {"wino_64x64", "wino_64x64_pipe"};
  return (c >= 0 && c < 2) ? names[c] : nullptr;
}
const char* dk_conv_wino_kernel_name(int c, int variant)
{
  static const char* names[2][4] = {
      {"conv3x3_wino_f32<4, true, 0>", "conv3x3_wino_f32<4, false, 0>", "conv3x3_wino_f32<1, true, 0>", "conv3x3_wino_f32<1, false, 0>"},
      {"conv3x3_wino_f32<4, true, 1>", "conv3x3_wino_f32<4, false, 1>", "conv3x3_wino_f32<1, true, 0>", "conv3x3_wino_f32<1, false, 0>"}};
  return (c >= 0 && c < 2 && variant >= 0 && variant < 4) ? names[c][variant] : nullptr;
}
bool dk_conv_wino_applicable(const DkConvDesc* d, int c)
{
  WinoGeo o;
  return c >= 0 && c < 2 && shape_ok(d) && wino_pick(d->w, (d->w + 1) / 2, true, o);
}

const float* dk_conv_wino_lookup(const float* weights)
{
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.find(weights);
  return it == g_reg.end() ? nullptr : it->second;
}

// Launches one batch chunk; a.w must already point at the transformed filters.  Returns the variant
// (0 / 1: 16-byte row pieces with paired / single stores, 2 / 3: 4-byte pieces), or -1 when the geometry does not fit
// (caller falls back).
// the pipelined schedules exist for 16-byte pieces only (with 4-byte pieces -- up to 18 offsets per thread -- they
// spill); launches with 4-byte pieces take schedule 0 whatever the configuration says
template <int SCHED>
static void (*wino_kernel_of(int vw, bool pair))(const ConvArgs)
{
  return vw == 4 ? (pair ? conv3x3_wino_f32<4, true, SCHED> : conv3x3_wino_f32<4, false, SCHED>)
                 : (pair ? conv3x3_wino_f32<1, true, 0> : conv3x3_wino_f32<1, false, 0>);
}

int dk_conv_wino_launch(ConvArgs a, int c, hipStream_t st)
{
  const int TH = (a.OH + 1) / 2, TW = (a.OW + 1) / 2;
  const int nb = a.N / a.OHW;
  const bool pair = (a.OW % 2 == 0) && (((uintptr_t)a.y & 7) == 0) && (!a.residual || ((uintptr_t)a.residual & 7) == 0);
  WinoGeo o;
  static const bool allow16 = !(getenv("DK_WINO_VW1") && atoi(getenv("DK_WINO_VW1")));   // diagnostics: force 4-byte pieces
  // 16-byte pieces straddle row ends when W is not a multiple of 4 floats, and the last piece of the tensor then
  // reads up to 12 bytes past it.  Arrays from cuda_make_array carry 64 bytes of slack; a caller-owned tensor
  // (dk_conv_forward on foreign memory, a plugin slot's state.input) may end at the end of its allocation, so the
  // allocation is asked: without 16 readable bytes behind the tensor the 4-byte-piece variant runs instead.
  bool wide_ok = allow16 && ((uintptr_t)a.x & 3) == 0;
  if (wide_ok && (a.W & 3))
  {
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.x) != hipSuccess)
    {
      (void)hipGetLastError();
      wide_ok = false;
    }
    else
      wide_ok = (const char*)a.x + (size_t)a.x_bytes + 16 <= (const char*)base + size;
  }
  if (!wino_pick(a.W, TW, wide_ok, o))
    return -1;
  // 16-byte stores (see the epilogue): rows and planes 16-byte aligned, tile pairs never straddle a row or an image
  a.wino_quad = pair && (a.OW % 4 == 0) && (a.OH % 2 == 0) && (((uintptr_t)a.y & 15) == 0) &&
                (!a.residual || ((uintptr_t)a.residual & 15) == 0);
  a.tiles_w = TW;
  a.tiles_hw = TH * TW;
  a.inv_tiles_w = 1.0 / TW;
  a.inv_tiles_hw = 1.0 / (TH * TW);
  a.wino_th = TH;
  a.wino_gp = o.gp;
  a.wino_rs = o.rs;
  a.wino_nk = o.nk;
  a.inv_wino_th = 1.0 / TH;
  a.inv_wino_gp = 1.0 / o.gp;
  a.inv_wino_rsg = 1.0 / (o.rs * o.gp);
  a.N = nb * TH * TW;
  a.tiles_m = a.M / WBM;
  a.tiles_n = (a.N + WBN - 1) / WBN;
  a.groups = 1;
  const bool drop_w = a.w_bytes == 0;   // DK_DEBUG_DROP bit 2 (timing diagnostics): zero-record descriptor
  a.w_bytes = (unsigned)((size_t)16 * a.M * a.C * sizeof(float));
  conv_args_finish(a);
  const long long nblk = conv_pick_partition(a, (size_t)a.w_bytes, WBM);
  if (drop_w)
    a.w_bytes = 0;
  // ring depth 3 (prefetch distance 2 stages).  DK_WINO_RING=4 takes a 4th slot where it fits the 160 KB:
  // measured no gain (0.190 vs 0.184 ms on [128->128 76x76]) although skipping the DMA wait altogether is
  // worth 7 % -- the rings are limited by LDS write bandwidth shared with the fragment reads, not by latency
  const int raw_f = o.nk * 256 * o.vw;
  a.wino_ring = 3;
  {
    static const int force = getenv("DK_WINO_RING") ? atoi(getenv("DK_WINO_RING")) : 0;
    if (force == 4 && (6 * W_STAGE + 4 * raw_f) * (int)sizeof(float) <= 160 * 1024)
      a.wino_ring = 4;
  }
  const int bytes = ((a.wino_ring + 2) * W_STAGE + a.wino_ring * raw_f) * (int)sizeof(float);
  void (*k)(const ConvArgs) = c == 1 ? wino_kernel_of<1>(o.vw, pair) : wino_kernel_of<0>(o.vw, pair);
  dk_set_max_dynamic_lds((const void*)k, bytes);
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(512), bytes, st, a);
  return (o.vw == 4 ? 0 : 2) + (pair ? 0 : 1);
}

extern "C" size_t dk_conv_wino_weights_size(const DkConvDesc* d)
{
  return (d && shape_ok(d)) ? (size_t)16 * d->n * d->c : 0;
}

extern "C" int dk_conv_wino_transform_weights(const DkConvDesc* d, const float* weights, float* U, void* stream)
{
  if (!d || !weights || !U || !shape_ok(d))
  {
    fprintf(stderr, "dk_conv_wino_transform_weights: layer does not take the Winograd kernel\n");
    return 1;
  }
  const int total = d->n * d->c;
  hipLaunchKernelGGL(wino_weights_kernel, dim3((total + 255) / 256), dim3(256), 0,
      stream ? (hipStream_t)stream : get_cuda_stream(), weights, U, d->n, d->c);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

// The forward entry points take the layer's ORIGINAL weights pointer (same signature as every other
// tile configuration); the transformed copy is found through this registry.
extern "C" void dk_conv_wino_register(const float* weights, const float* U)
{
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (U)
    g_reg[weights] = U;
  else
    g_reg.erase(weights);
}

// ---- derived weights of a training step in one launch (host/train.cpp) ----------------------------------------
struct DkTrainPrep
{
  DkPrepTask* tasks = nullptr;
  int ntasks = 0, nblocks = 0;
};

void* dk_train_prep_create(int ntasks, DkPrepTask* host_tasks)
{
  if (ntasks <= 0)
    return nullptr;
  DkTrainPrep* p = new DkTrainPrep();
  int nb = 0;
  for (int i = 0; i < ntasks; ++i)
  {
    DkPrepTask& t = host_tasks[i];
    t.first_block = nb;
    const size_t items = t.kind <= 2 ? ((size_t)t.M * t.C * t.ss + 1023) / 1024 : ((size_t)t.M * t.C + 255) / 256;
    nb += (int)items;
  }
  p->ntasks = ntasks;
  p->nblocks = nb;
  CHECK_HIP(hipMalloc((void**)&p->tasks, sizeof(DkPrepTask) * ntasks));
  CHECK_HIP(hipMemcpy(p->tasks, host_tasks, sizeof(DkPrepTask) * ntasks, hipMemcpyHostToDevice));
  return p;
}

void dk_train_prep_destroy(void* plan)
{
  DkTrainPrep* p = (DkTrainPrep*)plan;
  if (!p)
    return;
  (void)hipFree(p->tasks);
  delete p;
}

int dk_train_prep_run(void* plan, void* stream)
{
  DkTrainPrep* p = (DkTrainPrep*)plan;
  if (!p || !p->nblocks)
    return 0;
  hipLaunchKernelGGL(train_prep_kernel, dim3((unsigned)p->nblocks), dim3(256), 0,
      stream ? (hipStream_t)stream : get_cuda_stream(), p->tasks, p->ntasks);
  CHECK_HIP(hipPeekAtLastError());
  return 0;
}

#if DK_WSTAMP
extern "C" __attribute__((visibility("default"))) int dk_wino_stamps_read(long long* dst, int n)
{
  const int have = 64 * 8 * 160;
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wstamp), sizeof(long long) * (n < have ? n : have)));
  return have;
}
#endif
