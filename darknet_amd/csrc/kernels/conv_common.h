// conv_common.h -- argument block and device helpers shared by the implicit-GEMM
// convolution kernels (fp32: conv_igemm.hip, fp16 operands: conv_igemm_f16.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
  int mode;          // 0 forward gather; 1 data-gradient gather (x = delta, H/W = delta dims, OH/OW = input dims)
};

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

// Bijective XCD remap (8 XCDs, blocks dealt round-robin): block `bid` of `nwg`
// gets a logical id such that ids handled by one XCD are contiguous.
__device__ __forceinline__ int xcd_remap(int bid, int nwg)
{
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

constexpr unsigned OOB = 0x80000000u;  // ORed into a byte offset: always outside the buffer


// host side (conv_igemm.hip): per-shape tap table, created on first use
const int2* dk_conv_ktab(const DkConvDesc* d, int K, int C, int mode);
