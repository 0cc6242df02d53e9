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
// A workgroup (4 waves) owns 64 filters x 64 tiles for ALL 16 positions; wave (wm, wn) owns a
// 32 x 32 sub-block, 16 accumulators of v_mfma_f32_32x32x2_f32 = 256 registers: the C/D layout puts
// the 16 positions of one (m, t) in the SAME lane and register index, so the output transform
// A^T Mx A is pure per-lane register arithmetic -- no exchange through LDS.
//
// K loop, 8 input channels per stage, LDS double-buffered, one barrier per stage (64 MFMAs/wave):
//   * U (filters transformed once per layer by dk_conv_wino_transform_weights) lies in HBM as one
//     contiguous 32 KB slab per (filter tile, stage) in exactly the LDS image, so staging is a
//     straight 16-byte copy;
//   * V is produced in the kernel: a thread owns (channel, tile) pairs, fetches the 4x4 input
//     patch (padding and ragged edges through the buffer descriptor's range check: masked
//     elements get an out-of-range offset and read as 0), applies B^T d B (32 additions) and
//     writes the 16 positions to LDS;
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

namespace
{
constexpr int WBM = 64;              // filters per workgroup
constexpr int WBN = 64;              // output tiles (2x2 pixels each) per workgroup
constexpr int WCK = 8;               // input channels per stage
constexpr int W_STAGE = 16 * 64 * WCK;  // floats of one operand's stage image (32 KB)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16 mfma2(float a, float b, f32x16 c)
{
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// element offset of (xi, sub-block, row r32, channel c8) inside a stage image
__host__ __device__ __forceinline__ int img_off(int xi, int sub, int r32, int c8)
{
  return xi * 512 + sub * 256 + (c8 >> 2) * 128 + r32 * 4 + (c8 & 3);
}

template <bool PAIR>
__global__ void __launch_bounds__(256) conv3x3_wino_f32(const ConvArgs p)
{
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Us = lds;                 // [2][W_STAGE]
  float* const Vs = lds + 2 * W_STAGE;   // [2][W_STAGE]

  int g, tile_m, tile_n;
  if (!conv_block_tile(p, g, tile_m, tile_n))
    return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = tile_m * WBM, n0 = tile_n * WBN;
  const int nst = p.C / WCK;
  const int TW = p.tiles_w, THW = p.tiles_hw;

  __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

  // ---- input-transform ownership: two (channel, tile) pairs per thread ------------------------
  // lane -> channel (lane & 3) + 4 * (lane >> 5), tile ((lane >> 2) & 7) of an 8-tile group;
  // wave w, pass q -> tile group 2w + q.  (The 32 lanes of a half-wave then write 32 distinct
  // LDS banks for every position.)
  const int c8 = (lane & 3) + 4 * lh;
  unsigned xoff[2][16];
  int vdst[2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
  {
    const int tl = (wave * 2 + q) * 8 + ((lane >> 2) & 7);
    const int n = n0 + tl;
    const bool nv = n < p.N;
    const int nn = nv ? n : 0;
    const int b = fdiv(nn, THW, p.inv_tiles_hw);
    const int r = nn - b * THW;
    const int ty = fdiv(r, TW, p.inv_tiles_w);
    const int tx = r - ty * TW;
    const int iy0 = 2 * ty - 1, ix0 = 2 * tx - 1;
    const int base = ((b * p.Ctot + c8) * p.H + iy0) * p.W + ix0;
#pragma unroll
    for (int e = 0; e < 16; ++e)
    {
      const int iy = iy0 + (e >> 2), ix = ix0 + (e & 3);
      const bool ok = nv && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      xoff[q][e] = ok ? (unsigned)(base + (e >> 2) * p.W + (e & 3)) * 4u : OOB;
    }
    vdst[q] = img_off(0, tl >> 5, tl & 31, c8);
  }
  const unsigned stage_x_bytes = (unsigned)(WCK * p.H * p.W) * 4u;
  const unsigned ubase = (unsigned)(tile_m * nst) * (unsigned)(W_STAGE * 4);

  f32x16 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  float raw[2][16];
  float4 ureg[8];
  auto load_stage = [&](int t) {
    const unsigned xo = (unsigned)t * stage_x_bytes;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) raw[q][e] = ld_buf(xr, xoff[q][e] + xo);
    const unsigned uo = ubase + (unsigned)t * (unsigned)(W_STAGE * 4) + (unsigned)tid * 16u;
#pragma unroll
    for (int j = 0; j < 8; ++j) ureg[j] = ld_buf4(ur, uo + (unsigned)j * 4096u);
  };

  load_stage(0);
  for (int t = 0; t < nst; ++t)
  {
    float* const Ub = Us + (t & 1) * W_STAGE;
    float* const Vb = Vs + (t & 1) * W_STAGE;
    // ---- V = B^T d B of this thread's two patches -> LDS -----------------------------------
#pragma unroll
    for (int q = 0; q < 2; ++q)
    {
      float tm[4][4];
#pragma unroll
      for (int s = 0; s < 4; ++s)
      {
        const float d0 = raw[q][s], d1 = raw[q][4 + s], d2 = raw[q][8 + s], d3 = raw[q][12 + s];
        tm[0][s] = d0 - d2;
        tm[1][s] = d1 + d2;
        tm[2][s] = d2 - d1;
        tm[3][s] = d1 - d3;
      }
      float* const dst = Vb + vdst[q];
#pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        dst[(i * 4 + 0) * 512] = tm[i][0] - tm[i][2];
        dst[(i * 4 + 1) * 512] = tm[i][1] + tm[i][2];
        dst[(i * 4 + 2) * 512] = tm[i][2] - tm[i][1];
        dst[(i * 4 + 3) * 512] = tm[i][1] - tm[i][3];
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) ((float4*)Ub)[tid + j * 256] = ureg[j];
    __syncthreads();
    if (t + 1 < nst)
      load_stage(t + 1);
    // ---- 16 positions x 4 k-pairs ----------------------------------------------------------
    const float4* const Ua = (const float4*)Ub + wm * 64 + lh * 32 + l31;
    const float4* const Va = (const float4*)Vb + wn * 64 + lh * 32 + l31;
    // fragments of position xi + 1 are requested before the MFMAs of position xi are issued: with one
    // wave per SIMD nothing else hides the LDS latency
    float4 fa[2], fb[2];
    fa[0] = Ua[0];
    fb[0] = Va[0];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
    {
      if (xi + 1 < 16)
      {
        fa[(xi + 1) & 1] = Ua[(xi + 1) * 128];
        fb[(xi + 1) & 1] = Va[(xi + 1) * 128];
      }
      __builtin_amdgcn_sched_barrier(0);
      const float4 a = fa[xi & 1], b = fb[xi & 1];
      acc[xi] = mfma2(a.x, b.x, acc[xi]);
      acc[xi] = mfma2(a.y, b.y, acc[xi]);
      acc[xi] = mfma2(a.z, b.z, acc[xi]);
      acc[xi] = mfma2(a.w, b.w, acc[xi]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- output transform A^T Mx A, bias, activation (+ residual), store ----------------------
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
  const bool has_res = p.residual != nullptr;
  const int act = p.act;
  const unsigned row_bytes = (unsigned)p.OHW * 4u;
  const unsigned pbase = nv ? (unsigned)(b * p.Mtot * p.OHW + oy * p.OW + ox) * 4u : 0u;
  const unsigned o00 = nv ? pbase : 0xFFFFFFF0u;
  const unsigned o01 = (nv && col1) ? pbase + 4u : 0xFFFFFFF0u;
  const unsigned o10 = (nv && row1) ? pbase + (unsigned)p.OW * 4u : 0xFFFFFFF0u;
  const unsigned o11 = (nv && row1 && col1) ? pbase + (unsigned)p.OW * 4u + 4u : 0xFFFFFFF0u;
  // the activation is a launch constant: dispatched once into straight-line code
  auto emit = [&](auto actc) {
    constexpr int A = decltype(actc)::value;
  #pragma unroll
    for (int r = 0; r < 16; ++r)
    {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      const float bv = p.bias ? p.bias[m] : 0.f;
      float t0[4], t1[4];
  #pragma unroll
      for (int i = 0; i < 4; ++i)
      {
        t0[i] = acc[i * 4 + 0][r] + acc[i * 4 + 1][r] + acc[i * 4 + 2][r];
        t1[i] = acc[i * 4 + 1][r] - acc[i * 4 + 2][r] - acc[i * 4 + 3][r];
      }
      float y00 = t0[0] + t0[1] + t0[2] + bv;
      float y01 = t1[0] + t1[1] + t1[2] + bv;
      float y10 = t0[1] - t0[2] - t0[3] + bv;
      float y11 = t1[1] - t1[2] - t1[3] + bv;
      y00 = dk_activate(y00, A < 0 ? act : A);
      y01 = dk_activate(y01, A < 0 ? act : A);
      y10 = dk_activate(y10, A < 0 ? act : A);
      y11 = dk_activate(y11, A < 0 ? act : A);
      const unsigned mo = (unsigned)m * row_bytes;
      if (PAIR)
      {
        // OW even: both pixels of a row exist and the pair is 8-byte aligned
        const unsigned a0 = nv ? o00 + mo : 0xFFFFFFF0u;
        const unsigned a1 = (nv && row1) ? o10 + mo : 0xFFFFFFF0u;
        if (has_res)
        {
          const u32x2 r0 = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)a0, 0, 0);
          const u32x2 r1 = __builtin_amdgcn_raw_buffer_load_b64(rr, (int)a1, 0, 0);
          y00 += __uint_as_float(r0.x);
          y01 += __uint_as_float(r0.y);
          y10 += __uint_as_float(r1.x);
          y11 += __uint_as_float(r1.y);
        }
        u32x2 v0, v1;
        v0.x = __float_as_uint(y00); v0.y = __float_as_uint(y01);
        v1.x = __float_as_uint(y10); v1.y = __float_as_uint(y11);
        __builtin_amdgcn_raw_buffer_store_b64(v0, yr, (int)a0, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64(v1, yr, (int)a1, 0, 0);
      }
      else
      {
        const unsigned a00 = o00 == 0xFFFFFFF0u ? o00 : o00 + mo;
        const unsigned a01 = o01 == 0xFFFFFFF0u ? o01 : o01 + mo;
        const unsigned a10 = o10 == 0xFFFFFFF0u ? o10 : o10 + mo;
        const unsigned a11 = o11 == 0xFFFFFFF0u ? o11 : o11 + mo;
        if (has_res)
        {
          y00 += ld_buf(rr, a00);
          y01 += ld_buf(rr, a01);
          y10 += ld_buf(rr, a10);
          y11 += ld_buf(rr, a11);
        }
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y00), yr, (int)a00, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y01), yr, (int)a01, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y10), yr, (int)a10, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y11), yr, (int)a11, 0, 0);
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

// U = G g G^T of every (filter, channel), written in the kernel's slab order
// [filter tile][stage][xi][sub-block][half][32 rows][4 floats]
__global__ void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ U, int M, int C)
{
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M * C)
    return;
  const int m = idx / C, c = idx - m * C;
  const float* g = w + (size_t)idx * 9;
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

bool shape_ok(const DkConvDesc* d)
{
  return d->size == 3 && d->stride_x == 1 && d->stride_y == 1 && d->dilation == 1 && d->pad == 1 &&
         d->groups == 1 && d->c % WCK == 0 && d->n % WBM == 0 && d->h >= 2 && d->w >= 2;
}

std::mutex g_reg_mu;
std::unordered_map<const float*, const float*> g_reg;  // layer weights (device) -> transformed filters
}  // namespace

int dk_conv_wino_num_configs() { return 1; }
const char* dk_conv_wino_config_name(int c) { return c == 0 ? "wino_64x64" : nullptr; }
const char* dk_conv_wino_kernel_name(int c, int variant)
{
  if (c != 0)
    return nullptr;
  return variant ? "conv3x3_wino_f32<true>" : "conv3x3_wino_f32<false>";
}
bool dk_conv_wino_applicable(const DkConvDesc* d, int c) { return c == 0 && shape_ok(d); }

const float* dk_conv_wino_lookup(const float* weights)
{
  std::lock_guard<std::mutex> lk(g_reg_mu);
  auto it = g_reg.find(weights);
  return it == g_reg.end() ? nullptr : it->second;
}

// Launches one batch chunk; a.w must already point at the transformed filters.  Returns the variant.
int dk_conv_wino_launch(ConvArgs a, int c, hipStream_t st)
{
  (void)c;
  const int TH = (a.OH + 1) / 2, TW = (a.OW + 1) / 2;
  const int nb = a.N / a.OHW;
  a.tiles_w = TW;
  a.tiles_hw = TH * TW;
  a.inv_tiles_w = 1.0 / TW;
  a.inv_tiles_hw = 1.0 / (TH * TW);
  a.N = nb * TH * TW;
  a.tiles_m = a.M / WBM;
  a.tiles_n = (a.N + WBN - 1) / WBN;
  a.groups = 1;
  a.w_bytes = (unsigned)((size_t)16 * a.M * a.C * sizeof(float));
  conv_args_finish(a);
  const long long nblk = conv_pick_partition(a, (size_t)a.w_bytes, WBM);
  const int bytes = 4 * W_STAGE * (int)sizeof(float);
  const bool pair = (a.OW % 2 == 0) && (((uintptr_t)a.y & 7) == 0) && (!a.residual || ((uintptr_t)a.residual & 7) == 0);
  auto k = pair ? conv3x3_wino_f32<true> : conv3x3_wino_f32<false>;
  dk_set_max_dynamic_lds((const void*)k, bytes);
  hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), bytes, st, a);
  return pair ? 1 : 0;
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
