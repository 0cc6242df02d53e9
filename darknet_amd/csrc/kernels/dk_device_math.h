// dk_device_math.h -- device-side scalar maths shared by the kernels.
// Each function restates the reference's *CPU* scalar definition (the parity
// target), with the same float/double choices the reference's C++ build makes.
#pragma once
#include <hip/hip_runtime.h>

// ACTIVATION ids, src/yolo_core.h:69-92
enum
{
  DK_LOGISTIC = 0,
  DK_RELU = 1,
  DK_LINEAR = 4,
  DK_LEAKY = 8,
  DK_MISH = 17
};

// logistic_activate, src/activations.h:80-83
__device__ __forceinline__ float dk_logistic(float x) { return 1.f / (1.f + expf(-x)); }

// tanh_activate, src/activations.h:106-109
__device__ __forceinline__ float dk_tanh(float x) { return (2.f / (1.f + expf(-2.f * x)) - 1.f); }

// softplus_activate, src/activations.h:114-121
__device__ __forceinline__ float dk_softplus(float x, float threshold)
{
  if (x > threshold)
    return x;
  else if (x < -threshold)
    return expf(x);
  return logf(expf(x) + 1.f);
}

// mish, src/activations.c:185-197 (MISH_THRESHOLD = 20)
__device__ __forceinline__ float dk_mish(float x) { return x * dk_tanh(dk_softplus(x, 20.f)); }

// leaky as the reference's scalar CPU path computes it: `.1 * x` with a DOUBLE
// literal (src/gemm.c:2642), narrowed on store.
__device__ __forceinline__ float dk_leaky(float x) { return (x > 0.f) ? x : (float)(.1 * (double)x); }

// Fast mish (conv epilogue default; DK_FAST_MISH=0 selects dk_mish; act bit 0x400): algebraically identical,
// tanh(log(1+e)) = (e*e+2e)/(e*e+2e+2), one hardware exp and one hardware
// reciprocal (1 ulp), no cancellation.  Differs from the reference's own (cancellation-prone) formula
// by at most ~|x|*1.2e-7 absolute, the same size as glibc-vs-device libm noise.
__device__ __forceinline__ float dk_mish_fast(float x)
{
  // branch-free: for x > 20 the quotient may be inf/inf, the select discards it
  const float e = __expf(x);
  const float w = e * (e + 2.f);
  const float r = x * (w * __builtin_amdgcn_rcpf(w + 2.f));
  return (x > 20.f) ? x : r;
}

#define DK_ACT_FAST 0x400

__device__ __forceinline__ float dk_activate(float x, int a)
{
  if (a == (DK_MISH | DK_ACT_FAST))
    return dk_mish_fast(x);
  switch (a & 0xff)
  {
    case DK_LINEAR: return x;
    case DK_LEAKY: return dk_leaky(x);
    case DK_MISH: return dk_mish(x);
    case DK_LOGISTIC: return dk_logistic(x);
    case DK_RELU: return x * (x > 0.f);
    // the rarer kinds of activate() (src/activations.c:97-137, scalar definitions
    // src/activations.h:60-138; that file is C: unsuffixed literals are double)
    case 2: /* RELU6 */ return fminf(fmaxf(x, 0.f), 6.f);
    case 3: /* RELIE */ return (x > 0.f) ? x : .01f * x;
    case 5: /* RAMP */ return x * (x > 0.f) + .1f * x;
    case 6: /* TANH */ return dk_tanh(x);
    case 7: /* PLSE */ return (x < -4.f) ? .01f * (x + 4.f) : (x > 4.f) ? .01f * (x - 4.f) + 1.f : .125f * x + .5f;
    case 9: /* ELU */ return (x >= 0.f) * x + (x < 0.f) * (expf(x) - 1.f);
    case 10: /* LOGGY */ return 2.f / (1.f + expf(-x)) - 1.f;
    case 12: /* HARDTAN */ return (x < -1.f) ? -1.f : (x > 1.f) ? 1.f : x;
    case 13: /* LHTAN */ return (x < 0.f) ? .001f * x : (x > 1.f) ? .001f * (x - 1.f) + 1.f : x;
    case 14: /* SELU */ return (x >= 0.f) * 1.0507f * x + (x < 0.f) * 1.0507f * 1.6732f * (expf(x) - 1.f);
    case 15: /* GELU */
      return (float)(0.5 * (double)x * (1 + (double)tanhf((float)(0.797885 * (double)x + 0.035677 * (double)powf(x, 3.f)))));
    case 16: /* SWISH */ return x * dk_logistic(x);
    default: return x;
  }
}

// gradient(), src/activations.c:351-399 with the scalar definitions of src/activations.h:139-192
// (as the reference's C++ build evaluates them: int/float promotions, DOUBLE literals in
// gelu_gradient), evaluated on the layer OUTPUT y; mish (activations.c:426-452) and swish
// (:413-423) need the saved pre-activation `pre`.
__device__ __forceinline__ float dk_sech(float x) { return 2 / (expf(x) + expf(-x)); }

__device__ __forceinline__ float dk_act_gradient(float y, float pre, int a)
{
  switch (a & 0xff)
  {
    case DK_LINEAR: return 1;
    case DK_LEAKY: return (y > 0) ? 1 : .1f;
    case DK_LOGISTIC: return (1 - y) * y;
    case DK_RELU: return (y > 0);
    case DK_MISH:
    {
      const float sp = dk_softplus(pre, 20.f);
      const float grad_sp = 1 - expf(-sp);
      const float tsp = tanhf(sp);
      const float grad_tsp = (1 - tsp * tsp) * grad_sp;
      return pre * grad_tsp + tsp;
    }
    case 16: /* SWISH */ return y + dk_logistic(pre) * (1 - y);
    case 10: /* LOGGY */
    {
      const float h = (y + 1.f) / 2.f;
      return 2 * (1 - h) * h;
    }
    case 2: /* RELU6 */ return (y > 0 && y < 6);
    case 9: /* ELU */ return (y >= 0) + (y < 0) * (y + 1);
    case 14: /* SELU */ return (y >= 0) * 1.0507f + (y < 0) * (y + 1.0507f * 1.6732f);
    case 15: /* GELU */
    {
      const float x3 = powf(y, 3.f);
      return (float)(0.5 * (double)tanhf((float)(0.0356774 * (double)x3 + 0.797885 * (double)y)) +
                     (0.0535161 * (double)x3 + 0.398942 * (double)y) *
                         (double)powf(dk_sech((float)(0.0356774 * (double)x3 + 0.797885 * (double)y)), 2.f) +
                     0.5);
    }
    case 3: /* RELIE */ return (y > 0) ? 1 : .01f;
    case 5: /* RAMP */ return (y > 0) + .1f;
    case 6: /* TANH */ return 1 - y * y;
    case 7: /* PLSE */ return (y < 0 || y > 1) ? .01f : .125f;
    case 11: /* STAIR */ return (floorf(y) == y) ? 0 : 1.0f;
    case 12: /* HARDTAN */ return (y > -1 && y < 1) ? 1 : 0;
    case 13: /* LHTAN */ return (y > 0 && y < 1) ? 1 : .001f;
    default: return 0;
  }
}
