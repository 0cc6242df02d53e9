/*
 * orc_ops.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY: nothing in the product
 * path (darknet_amd/, include/) may call, link or import this file.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * there only as the checker / the timed CPU baseline ("port").
 *
 * What it is: a plain-C restatement of the reference's (Ravicmoon/darknet,
 * /root/reference) *CPU* convolutional path, function by function, each one
 * citing the reference file:line it follows.  It is written so that, compiled
 * with `gcc -O2 -ffp-contract=off` (oracle/Makefile), it reproduces the
 * reference's canonical scalar build (`g++ -O2 -ffp-contract=off`, no AVX)
 * BIT FOR BIT.  That claim is pinned, not assumed: tests/test_oracle_vs_ref.py
 * compares every function here with oracle/_ref/libref_canon.so (the real
 * reference compiled from its own sources by oracle/Makefile) whenever that
 * library is present, and tests/test_oracle_golden.py compares it with the
 * committed fixtures in tests/golden/ that tools/make_golden.py dumped from
 * the real reference.
 *
 * Note on libm calls: the reference is compiled as C++11 (CMakeLists.txt:183),
 * so `sqrt(float)`, `exp(float)`, `pow(float,float)` resolve to the *float*
 * overloads (verified in the disassembly of the reference build: sqrtf in
 * normalize_cpu / FuseConvBatchNorm, expf in GetYoloBox, powf in
 * variance_delta_cpu), while `pow(float,int)` and anything mixed with a double
 * literal is evaluated in double.  This file spells those choices out
 * explicitly (sqrtf/expf/powf vs sqrt/exp/pow) because in C they are not
 * implied by the argument types.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ACTIVATION enum values, src/yolo_core.h:69-92 */
enum
{
  ORC_LOGISTIC = 0,
  ORC_RELU = 1,
  ORC_LINEAR = 4,
  ORC_LEAKY = 8,
  ORC_MISH = 17
};

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* ------------------------------------------------------------------ im2col */

/* src/im2col.c:50-53 */
static inline int a_ge_zero_and_lt_b(int a, int b)
{
  return (unsigned)a < (unsigned)b;
}

/* im2col_cpu_ext, src/im2col.c:56-104.
 * col[((c*kh+i)*kw+j)][y*ow+x] = im[c][y*sh-ph+i*dh][x*sw-pw+j*dw] or 0. */
void orc_im2col_ext(const float* im, int channels, int height, int width,
    int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h,
    int stride_w, int dilation_h, int dilation_w, float* col)
{
  const int out_h =
      (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
  const int out_w =
      (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  const int chan = height * width;
  for (int c = 0; c < channels; ++c, im += chan)
    for (int kr = 0; kr < kernel_h; ++kr)
      for (int kc = 0; kc < kernel_w; ++kc)
      {
        int in_row = -pad_h + kr * dilation_h;
        for (int oy = 0; oy < out_h; ++oy, in_row += stride_h)
        {
          if (!a_ge_zero_and_lt_b(in_row, height))
          {
            for (int ox = 0; ox < out_w; ++ox) *col++ = 0;
            continue;
          }
          int in_col = -pad_w + kc * dilation_w;
          for (int ox = 0; ox < out_w; ++ox, in_col += stride_w)
            *col++ = a_ge_zero_and_lt_b(in_col, width)
                         ? im[in_row * width + in_col]
                         : 0;
        }
      }
}

/* col2im_cpu_ext, src/col2im.c:65-108: zero-fills the image first (:70), then
 * scatter-adds in the same traversal order as im2col. */
void orc_col2im_ext(const float* col, int channels, int height, int width,
    int kernel_h, int kernel_w, int pad_h, int pad_w, int stride_h,
    int stride_w, int dilation_h, int dilation_w, float* im)
{
  const int out_h =
      (height + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) / stride_h + 1;
  const int out_w =
      (width + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) / stride_w + 1;
  const int chan = height * width;
  for (int i = 0; i < chan * channels; ++i) im[i] = 0.0f;
  for (int c = 0; c < channels; ++c, im += chan)
    for (int kr = 0; kr < kernel_h; ++kr)
      for (int kc = 0; kc < kernel_w; ++kc)
      {
        int in_row = -pad_h + kr * dilation_h;
        for (int oy = 0; oy < out_h; ++oy, in_row += stride_h)
        {
          if (!a_ge_zero_and_lt_b(in_row, height))
          {
            col += out_w;
            continue;
          }
          int in_col = -pad_w + kc * dilation_w;
          for (int ox = 0; ox < out_w; ++ox, in_col += stride_w, ++col)
            if (a_ge_zero_and_lt_b(in_col, width))
              im[in_row * width + in_col] += *col;
        }
      }
}

/* -------------------------------------------------------------------- gemm */

/* One output row of each variant: scalar gemm_nn src/gemm.c:2223-2239,
 * gemm_nt :2914-2930, gemm_tn :2932-2947, gemm_tt :2949-2965 (called with M=1
 * per row from gemm_cpu :2992-3003).  Per output element the K products are
 * accumulated in ascending k, each product rounded before the add. */
static void row_nn(int N, int K, float ALPHA, const float* A, const float* B,
    int ldb, float* C)
{
  for (int k = 0; k < K; ++k)
  {
    const float a = ALPHA * A[k];
    const float* b = B + (size_t)k * ldb;
    for (int j = 0; j < N; ++j) C[j] += a * b[j];
  }
}

static void row_tn(int N, int K, float ALPHA, const float* A, int lda,
    const float* B, int ldb, float* C)
{
  for (int k = 0; k < K; ++k)
  {
    const float a = ALPHA * A[(size_t)k * lda];
    const float* b = B + (size_t)k * ldb;
    for (int j = 0; j < N; ++j) C[j] += a * b[j];
  }
}

static void row_nt(int N, int K, float ALPHA, const float* A, const float* B,
    int ldb, float* C)
{
  for (int j = 0; j < N; ++j)
  {
    float sum = 0;
    const float* b = B + (size_t)j * ldb;
    for (int k = 0; k < K; ++k) sum += ALPHA * A[k] * b[k];
    C[j] += sum;
  }
}

static void row_tt(int N, int K, float ALPHA, const float* A, int lda,
    const float* B, int ldb, float* C)
{
  for (int j = 0; j < N; ++j)
  {
    float sum = 0;
    for (int k = 0; k < K; ++k)
      sum += ALPHA * A[(size_t)k * lda] * B[k + (size_t)j * ldb];
    C[j] += sum;
  }
}

/* gemm -> gemm_cpu, src/gemm.c:97-101, 2967-3005 (non-AVX branch :2988-3003:
 * C *= BETA when BETA != 1, then an OpenMP loop over the M rows). */
void orc_gemm(int TA, int TB, int M, int N, int K, float ALPHA, const float* A,
    int lda, const float* B, int ldb, float BETA, float* C, int ldc)
{
  if (BETA != 1)
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) C[(size_t)i * ldc + j] *= BETA;
#pragma omp parallel for
  for (int t = 0; t < M; ++t)
  {
    float* c = C + (size_t)t * ldc;
    if (!TA && !TB)
      row_nn(N, K, ALPHA, A + (size_t)t * lda, B, ldb, c);
    else if (TA && !TB)
      row_tn(N, K, ALPHA, A + t, lda, B, ldb, c);
    else if (!TA && TB)
      row_nt(N, K, ALPHA, A + (size_t)t * lda, B, ldb, c);
    else
      row_tt(N, K, ALPHA, A + t, lda, B, ldb, c);
  }
}

/* ----------------------------------------------------------- bias / scale */

/* add_bias, src/convolutional_layer.cpp:916-929 */
void orc_add_bias(float* out, const float* biases, int batch, int n, int size)
{
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < size; ++j)
        out[((size_t)b * n + i) * size + j] += biases[i];
}

/* scale_bias, src/convolutional_layer.cpp:931-944 */
void orc_scale_bias(float* out, const float* scales, int batch, int n, int size)
{
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < size; ++j)
        out[((size_t)b * n + i) * size + j] *= scales[i];
}

/* backward_bias, src/convolutional_layer.cpp:946-957 with sum_array
 * (src/utils.cpp: float accumulator, ascending). */
void orc_backward_bias(
    float* bias_updates, const float* delta, int batch, int n, int size)
{
  for (int b = 0; b < batch; ++b)
    for (int i = 0; i < n; ++i)
    {
      const float* d = delta + (size_t)size * (i + (size_t)b * n);
      float sum = 0;
      for (int j = 0; j < size; ++j) sum += d[j];
      bias_updates[i] += sum;
    }
}

/* FuseConvBatchNorm, src/network.cpp:647-682 (per conv layer):
 * std = sqrtf(var + 1e-5f); bias -= scale*mean/std; W[f][:] *= scale/std. */
void orc_fuse_conv_bn(float* weights, float* biases, const float* scales,
    const float* rolling_mean, const float* rolling_variance, int n,
    int filter_size)
{
  for (int f = 0; f < n; ++f)
  {
    float std = sqrtf(rolling_variance[f] + 0.00001f);
    biases[f] -= scales[f] * rolling_mean[f] / std;
    for (int i = 0; i < filter_size; ++i)
      weights[(size_t)f * filter_size + i] *= scales[f] / std;
  }
}

/* ------------------------------------------------------------- activations */

/* logistic_activate src/activations.h:80-83 */
static inline float logistic_f(float x) { return 1.f / (1.f + expf(-x)); }
/* tanh_activate src/activations.h:106-109 (all float: int literals promote) */
static inline float tanh_f(float x) { return (2 / (1 + expf(-2 * x)) - 1); }
/* softplus_activate src/activations.h:114-121 */
static inline float softplus_f(float x, float threshold)
{
  if (x > threshold)
    return x;
  else if (x < -threshold)
    return expf(x);
  return logf(expf(x) + 1);
}

/* activate_array_cpu_custom, scalar build src/gemm.c:2632-2652:
 * LINEAR no-op; LEAKY uses the DOUBLE literal `.1 * x`; everything else goes
 * through activate() (src/activations.c:97-140).  Only the kinds the YOLOv4
 * family uses are restated (LOGISTIC, RELU, LINEAR, LEAKY). */
void orc_activate_array(float* x, int n, int a)
{
  if (a == ORC_LINEAR)
    return;
  if (a == ORC_LEAKY)
  {
    for (int i = 0; i < n; ++i) x[i] = (x[i] > 0) ? x[i] : .1 * x[i];
    return;
  }
  if (a == ORC_LOGISTIC)
  {
    for (int i = 0; i < n; ++i) x[i] = logistic_f(x[i]);
    return;
  }
  if (a == ORC_RELU)
  {
    for (int i = 0; i < n; ++i) x[i] = x[i] * (x[i] > 0);
    return;
  }
  /* the remaining kinds go through activate() (src/activations.c:97-137) with the scalar
   * definitions of src/activations.h:60-138 -- compiled as C: unsuffixed literals are double */
  for (int i = 0; i < n; ++i)
  {
    const float v = x[i];
    float r;
    switch (a)
    {
      case 2: /* RELU6 */ r = ((((v > 0) ? v : 0) < 6) ? ((v > 0) ? v : 0) : 6); break;
      case 3: /* RELIE */ r = (v > 0) ? v : .01f * v; break;
      case 5: /* RAMP */ r = v * (v > 0) + .1f * v; break;
      case 6: /* TANH */ r = (2 / (1 + expf(-2 * v)) - 1); break;
      case 7: /* PLSE */ r = (v < -4) ? .01f * (v + 4) : (v > 4) ? .01f * (v - 4) + 1 : .125f * v + .5f; break;
      case 9: /* ELU */ r = (v >= 0) * v + (v < 0) * (expf(v) - 1); break;
      case 10: /* LOGGY */ r = 2.f / (1.f + expf(-v)) - 1; break;
      case 12: /* HARDTAN */ r = (v < -1) ? -1 : (v > 1) ? 1 : v; break;
      case 13: /* LHTAN */ r = (v < 0) ? .001f * v : (v > 1) ? .001f * (v - 1) + 1 : v; break;
      case 14: /* SELU */ r = (v >= 0) * 1.0507f * v + (v < 0) * 1.0507f * 1.6732f * (expf(v) - 1); break;
      case 15: /* GELU */ r = (0.5 * v * (1 + tanhf(0.797885 * v + 0.035677 * powf(v, 3)))); break;
      case 16: /* SWISH: activate_array_swish, src/activations.c:170-182 */ r = v * logistic_f(v); break;
      default: abort(); /* unsupported activation in the oracle */
    }
    x[i] = r;
  }
}

/* activate_array (src/activations.c:142-170): LEAKY here is the float
 * leaky_activate `.1f*x` (activations.h:104); used by the yolo layer for
 * LOGISTIC only. */
void orc_activate_array_plain(float* x, int n, int a)
{
  if (a == ORC_LINEAR)
    return;
  if (a == ORC_LEAKY)
  {
    for (int i = 0; i < n; ++i) x[i] = (x[i] > 0) ? x[i] : .1f * x[i];
    return;
  }
  if (a == ORC_LOGISTIC)
  {
#pragma omp parallel for
    for (int i = 0; i < n; ++i) x[i] = logistic_f(x[i]);
    return;
  }
  abort();
}

/* activate_array_mish, src/activations.c:185-197: saves the pre-activation. */
void orc_activate_array_mish(
    const float* x, int n, float* activation_input, float* output)
{
  const float MISH_THRESHOLD = 20;
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
  {
    float x_val = x[i];
    if (activation_input)
      activation_input[i] = x_val;
    output[i] = x_val * tanh_f(softplus_f(x_val, MISH_THRESHOLD));
  }
}

/* the rarer kinds of gradient(), src/activations.c:351-399 with the scalar definitions of
 * src/activations.h:139-192 as the reference's C++ build evaluates them (int/float promotions,
 * DOUBLE literals in gelu_gradient); evaluated on the layer OUTPUT like every other kind */
static float orc_sech(float x) { return 2 / (expf(x) + expf(-x)); }
static float orc_gradient_rare(float x, int a)
{
  switch (a)
  {
    case 10: /* LOGGY */
    {
      float y = (x + 1.f) / 2.f;
      return 2 * (1 - y) * y;
    }
    case 2: /* RELU6 */ return (x > 0 && x < 6);
    case 9: /* ELU */ return (x >= 0) + (x < 0) * (x + 1);
    case 14: /* SELU */ return (x >= 0) * 1.0507f + (x < 0) * (x + 1.0507f * 1.6732f);
    case 15: /* GELU */
    {
      const float x3 = powf(x, 3);
      return 0.5 * tanhf(0.0356774 * x3 + 0.797885 * x) +
             (0.0535161 * x3 + 0.398942 * x) * powf(orc_sech(0.0356774 * x3 + 0.797885 * x), 2) + 0.5;
    }
    case 3: /* RELIE */ return (x > 0) ? 1 : .01f;
    case 5: /* RAMP */ return (x > 0) + .1f;
    case 6: /* TANH */ return 1 - x * x;
    case 7: /* PLSE */ return (x < 0 || x > 1) ? .01f : .125f;
    case 11: /* STAIR */ return (floorf(x) == x) ? 0 : 1.0f;
    case 12: /* HARDTAN */ return (x > -1 && x < 1) ? 1 : 0;
    case 13: /* LHTAN */ return (x > 0 && x < 1) ? 1 : .001f;
    default: return 0;
  }
}

/* gradient_array, src/activations.c:401-410 with gradient() :351-399:
 * delta *= f'(y) evaluated on the OUTPUT y.  leaky: (y>0)?1:.1f
 * (activations.h:177), logistic: (1-y)*y (:150), linear: 1, relu: (y>0). */
void orc_gradient_array(const float* y, int n, int a, float* delta)
{
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
  {
    float g;
    if (a == ORC_LINEAR)
      g = 1;
    else if (a == ORC_LEAKY)
      g = (y[i] > 0) ? 1 : .1f;
    else if (a == ORC_LOGISTIC)
      g = (1 - y[i]) * y[i];
    else if (a == ORC_RELU)
      g = (y[i] > 0);
    else
      g = orc_gradient_rare(y[i], a);
    delta[i] *= g;
  }
}

/* gradient_array_swish, src/activations.c:413-423: x = the swish output, sigmoid = the stored sigmoid */
void orc_gradient_array_swish(const float* x, int n, const float* sigmoid, float* delta)
{
  for (int i = 0; i < n; ++i)
  {
    const float swish = x[i];
    delta[i] *= swish + sigmoid[i] * (1 - swish);
  }
}

/* gradient_array_mish, src/activations.c:426-452 (uses the saved input; exp
 * and tanh resolve to the float overloads in the reference's C++ build). */
void orc_gradient_array_mish(int n, const float* activation_input, float* delta)
{
#pragma omp parallel for
  for (int i = 0; i < n; ++i)
  {
    const float MISH_THRESHOLD = 20.0f;
    float inp = activation_input[i];
    const float sp = softplus_f(inp, MISH_THRESHOLD);
    const float grad_sp = 1 - expf(-sp);
    const float tsp = tanhf(sp);
    const float grad_tsp = (1 - tsp * tsp) * grad_sp;
    const float grad = inp * grad_tsp + tsp;
    delta[i] *= grad;
  }
}

/* --------------------------------------------------------------- batchnorm */

/* mean_cpu, src/blas.c:164-181 */
/* Analysis switch (NOT the reference's arithmetic): when set, the batch statistics accumulate in
 * double and are rounded once.  The reference adds up to millions of fp32 terms sequentially into a
 * float, so ITS statistics carry a summation error that grows with the layer size; comparing both
 * variants separates that error from any error of the implementation under test
 * (tests/test_gpu_train.py, tools/make_golden.py train_big). */
static int g_bn_stats_f64 = 0;
void orc_set_bn_stats_f64(int on) { g_bn_stats_f64 = on; }

void orc_mean(const float* x, int batch, int filters, int spatial, float* mean)
{
  float scale = 1. / (batch * spatial);
  if (g_bn_stats_f64)
  {
#pragma omp parallel for
    for (int i = 0; i < filters; ++i)
    {
      double acc = 0;
      for (int j = 0; j < batch; ++j)
        for (int k = 0; k < spatial; ++k)
          acc += x[(size_t)j * filters * spatial + (size_t)i * spatial + k];
      mean[i] = (float)(acc / ((double)batch * spatial));
    }
    return;
  }
  for (int i = 0; i < filters; ++i)
  {
    mean[i] = 0;
    for (int j = 0; j < batch; ++j)
      for (int k = 0; k < spatial; ++k)
        mean[i] += x[(size_t)j * filters * spatial + (size_t)i * spatial + k];
    mean[i] *= scale;
  }
}

/* variance_cpu, src/blas.c:183-201: divides by N-1; pow(float,int) is the
 * double overload, and `float += double` adds in double then narrows. */
void orc_variance(const float* x, const float* mean, int batch, int filters,
    int spatial, float* variance)
{
  float scale = 1. / (batch * spatial - 1);
  if (g_bn_stats_f64)
  {
#pragma omp parallel for
    for (int i = 0; i < filters; ++i)
    {
      double acc = 0;
      for (int j = 0; j < batch; ++j)
        for (int k = 0; k < spatial; ++k)
        {
          const double d = (double)x[(size_t)j * filters * spatial + (size_t)i * spatial + k] - (double)mean[i];
          acc += d * d;
        }
      variance[i] = (float)(acc / ((double)batch * spatial - 1));
    }
    return;
  }
  for (int i = 0; i < filters; ++i)
  {
    variance[i] = 0;
    for (int j = 0; j < batch; ++j)
      for (int k = 0; k < spatial; ++k)
      {
        size_t index = (size_t)j * filters * spatial + (size_t)i * spatial + k;
        variance[i] += pow((double)(x[index] - mean[i]), 2);
      }
    variance[i] *= scale;
  }
}

/* normalize_cpu, src/blas.c:203-218: eps 1e-6 (.000001f), sqrtf. */
void orc_normalize(float* x, const float* mean, const float* variance,
    int batch, int filters, int spatial)
{
  for (int b = 0; b < batch; ++b)
    for (int f = 0; f < filters; ++f)
      for (int i = 0; i < spatial; ++i)
      {
        size_t index = (size_t)b * filters * spatial + (size_t)f * spatial + i;
        x[index] = (x[index] - mean[f]) / (sqrtf(variance[f] + .000001f));
      }
}

/* ForwardBatchnormLayer for the BN stage inside a conv, src/batchnorm_layer.cpp
 * :206-238.  train: batch stats, rolling = .9*rolling + .1*batch (:221-224,
 * scal_cpu then axpy_cpu with float ALPHA), x and x_norm saved; inference:
 * rolling stats.  Then scale_bias and add_bias. */
void orc_batchnorm_forward(float* output, int batch, int c, int spatial,
    float* scales, float* biases, float* rolling_mean, float* rolling_variance,
    float* mean, float* variance, float* x_save, float* x_norm_save, int train)
{
  size_t total = (size_t)batch * c * spatial;
  if (train)
  {
    orc_mean(output, batch, c, spatial, mean);
    orc_variance(output, mean, batch, c, spatial, variance);
    const float a9 = .9, a1 = .1;
    for (int i = 0; i < c; ++i) rolling_mean[i] *= a9;
    for (int i = 0; i < c; ++i) rolling_mean[i] += a1 * mean[i];
    for (int i = 0; i < c; ++i) rolling_variance[i] *= a9;
    for (int i = 0; i < c; ++i) rolling_variance[i] += a1 * variance[i];
    if (x_save)
      memcpy(x_save, output, total * sizeof(float));
    orc_normalize(output, mean, variance, batch, c, spatial);
    if (x_norm_save)
      memcpy(x_norm_save, output, total * sizeof(float));
  }
  else
  {
    orc_normalize(output, rolling_mean, rolling_variance, batch, c, spatial);
  }
  orc_scale_bias(output, scales, batch, c, spatial);
  orc_add_bias(output, biases, batch, c, spatial);
}

/* backward_scale_cpu, src/batchnorm_layer.cpp:92-109 */
void orc_backward_scale(const float* x_norm, const float* delta, int batch,
    int n, int size, float* scale_updates)
{
  for (int f = 0; f < n; ++f)
  {
    float sum = 0;
    for (int b = 0; b < batch; ++b)
      for (int i = 0; i < size; ++i)
      {
        size_t index = i + (size_t)size * (f + (size_t)n * b);
        sum += delta[index] * x_norm[index];
      }
    scale_updates[f] += sum;
  }
}

/* mean_delta_cpu, src/batchnorm_layer.cpp:111-127 (eps 1e-5; -1. is double) */
void orc_mean_delta(const float* delta, const float* variance, int batch,
    int filters, int spatial, float* mean_delta)
{
  for (int i = 0; i < filters; ++i)
  {
    mean_delta[i] = 0;
    for (int j = 0; j < batch; ++j)
      for (int k = 0; k < spatial; ++k)
        mean_delta[i] +=
            delta[(size_t)j * filters * spatial + (size_t)i * spatial + k];
    mean_delta[i] *= (-1. / sqrtf(variance[i] + .00001f));
  }
}

/* variance_delta_cpu, src/batchnorm_layer.cpp:129-145 (powf; -.5 double) */
void orc_variance_delta(const float* x, const float* delta, const float* mean,
    const float* variance, int batch, int filters, int spatial,
    float* variance_delta)
{
  for (int i = 0; i < filters; ++i)
  {
    variance_delta[i] = 0;
    for (int j = 0; j < batch; ++j)
      for (int k = 0; k < spatial; ++k)
      {
        size_t index = (size_t)j * filters * spatial + (size_t)i * spatial + k;
        variance_delta[i] += delta[index] * (x[index] - mean[i]);
      }
    variance_delta[i] *= -.5 * powf(variance[i] + .00001f, (float)(-3. / 2.));
  }
}

/* normalize_delta_cpu, src/batchnorm_layer.cpp:147-165: mixed double/float
 * expression evaluated left to right exactly as written there. */
void orc_normalize_delta(const float* x, const float* mean,
    const float* variance, const float* mean_delta,
    const float* variance_delta, int batch, int filters, int spatial,
    float* delta)
{
  for (int j = 0; j < batch; ++j)
    for (int f = 0; f < filters; ++f)
      for (int k = 0; k < spatial; ++k)
      {
        size_t index = (size_t)j * filters * spatial + (size_t)f * spatial + k;
        delta[index] =
            delta[index] * 1. / (sqrtf(variance[f]) + .00001f) +
            variance_delta[f] * 2. * (x[index] - mean[f]) / (spatial * batch) +
            mean_delta[f] / (spatial * batch);
      }
}

/* BackwardBatchnormLayer (conv-embedded), src/batchnorm_layer.cpp:240-255.
 * NOTE: the CPU reference never computes bias_updates here. */
void orc_batchnorm_backward(float* delta, int batch, int c, int spatial,
    const float* scales, const float* x, const float* x_norm, const float* mean,
    const float* variance, float* mean_delta, float* variance_delta,
    float* scale_updates)
{
  orc_backward_scale(x_norm, delta, batch, c, spatial, scale_updates);
  orc_scale_bias(delta, scales, batch, c, spatial);
  orc_mean_delta(delta, variance, batch, c, spatial, mean_delta);
  orc_variance_delta(
      x, delta, mean, variance, batch, c, spatial, variance_delta);
  orc_normalize_delta(
      x, mean, variance, mean_delta, variance_delta, batch, c, spatial, delta);
}

/* --------------------------------------------------------- convolutional */

static int conv_out(int in, int pad, int size, int stride)
{
  /* ConvOutHeight/Width, src/convolutional_layer.cpp:87-95 */
  return (in + 2 * pad - size) / stride + 1;
}

/* GEMM stage of ForwardConvolutionalLayer, src/convolutional_layer.cpp:1128-
 * 1262: zero the output, then per image and per group B = im2col(x) (skipped
 * for size==1: B = x, :1243-1246) and C += W*B.  No bias / BN / activation. */
void orc_conv_gemm_forward(const float* input, const float* weights,
    float* output, float* workspace, int batch, int c, int h, int w, int n,
    int groups, int size, int stride_x, int stride_y, int dilation, int pad)
{
  const int out_h = conv_out(h, pad, size, stride_y);
  const int out_w = conv_out(w, pad, size, stride_x);
  const int m = n / groups;
  const int k = size * size * c / groups;
  const int nn = out_h * out_w;
  const int nweights = (c / groups) * n * size * size;
  memset(output, 0, (size_t)batch * n * nn * sizeof(float));
  for (int i = 0; i < batch; ++i)
    for (int j = 0; j < groups; ++j)
    {
      const float* a = weights + (size_t)j * nweights / groups;
      float* cc = output + ((size_t)i * groups + j) * nn * m;
      const float* im =
          input + ((size_t)i * groups + j) * (c / groups) * h * w;
      const float* b;
      if (size == 1)
        b = im;
      else
      {
        orc_im2col_ext(im, c / groups, h, w, size, size, pad * dilation,
            pad * dilation, stride_y, stride_x, dilation, dilation, workspace);
        b = workspace;
      }
      orc_gemm(0, 0, m, nn, k, 1, a, k, b, nn, 1, cc, nn);
    }
}

/* Inference-mode conv layer after LoadNetwork(train=false), i.e. BN already
 * folded by FuseConvBatchNorm (src/parser.cpp:1866): GEMM (above) + add_bias
 * (:1271) + activation (:1274-1293).  activation_input may be NULL. */
void orc_conv_forward_fused(const float* input, const float* weights,
    const float* biases, float* output, float* workspace, float* activation_input,
    int batch, int c, int h, int w, int n, int groups, int size, int stride_x,
    int stride_y, int dilation, int pad, int activation)
{
  const int out_h = conv_out(h, pad, size, stride_y);
  const int out_w = conv_out(w, pad, size, stride_x);
  orc_conv_gemm_forward(input, weights, output, workspace, batch, c, h, w, n,
      groups, size, stride_x, stride_y, dilation, pad);
  orc_add_bias(output, biases, batch, n, out_h * out_w);
  int total = batch * n * out_h * out_w;
  if (activation == ORC_MISH)
    orc_activate_array_mish(output, total, activation_input, output);
  else
    orc_activate_array(output, total, activation);
}

/* BackwardConvolutionalLayer GEMM stage, src/convolutional_layer.cpp:1335-
 * 1379 (after the activation gradient and BN / bias backward): per image and
 * group, B = im2col(x) (always, even for size 1), wgrad dW += delta*B^T
 * (gemm(0,1,...,beta=1)); if prev_delta: col = W^T*delta (gemm(1,0,...,beta=0))
 * and prev_delta_b = col2im(col), which OVERWRITES. */
void orc_conv_backward(const float* input, const float* weights,
    const float* delta, float* weight_updates, float* prev_delta,
    float* workspace, int batch, int c, int h, int w, int n, int groups,
    int size, int stride_x, int stride_y, int dilation, int pad)
{
  const int out_h = conv_out(h, pad, size, stride_y);
  const int out_w = conv_out(w, pad, size, stride_x);
  const int m = n / groups;
  const int nn = size * size * c / groups;
  const int k = out_w * out_h;
  const int nweights = (c / groups) * n * size * size;
  for (int i = 0; i < batch; ++i)
    for (int j = 0; j < groups; ++j)
    {
      const float* a = delta + ((size_t)i * groups + j) * m * k;
      float* b = workspace;
      float* cc = weight_updates + (size_t)j * nweights / groups;
      const float* im =
          input + ((size_t)i * groups + j) * (c / groups) * h * w;
      orc_im2col_ext(im, c / groups, h, w, size, size, pad * dilation,
          pad * dilation, stride_y, stride_x, dilation, dilation, b);
      orc_gemm(0, 1, m, nn, k, 1, a, k, b, k, 1, cc, nn);
      if (prev_delta)
      {
        const float* wa = weights + (size_t)j * nweights / groups;
        const float* db = delta + ((size_t)i * groups + j) * m * k;
        orc_gemm(1, 0, nn, k, m, 1, wa, nn, db, k, 0, workspace, k);
        orc_col2im_ext(workspace, c / groups, h, w, size, size,
            pad * dilation, pad * dilation, stride_y, stride_x, dilation,
            dilation,
            prev_delta + ((size_t)i * groups + j) * (c / groups) * h * w);
      }
    }
}

/* UpdateConvolutionalLayer, src/convolutional_layer.cpp:1382-1399 with
 * axpy_cpu / scal_cpu (src/blas.c:238-250; ALPHA is a float parameter). */
static void axpy_f(int n, float alpha, const float* x, float* y)
{
  for (int i = 0; i < n; ++i) y[i] += alpha * x[i];
}
static void scal_f(int n, float alpha, float* x)
{
  for (int i = 0; i < n; ++i) x[i] *= alpha;
}
void orc_conv_update(float* weights, float* weight_updates, int nweights,
    float* biases, float* bias_updates, float* scales, float* scale_updates,
    int n, int batch, float learning_rate, float momentum, float decay)
{
  axpy_f(nweights, -decay * batch, weights, weight_updates);
  axpy_f(nweights, learning_rate / batch, weight_updates, weights);
  scal_f(nweights, momentum, weight_updates);
  axpy_f(n, learning_rate / batch, bias_updates, biases);
  scal_f(n, momentum, bias_updates);
  if (scales)
  {
    axpy_f(n, learning_rate / batch, scale_updates, scales);
    scal_f(n, momentum, scale_updates);
  }
}

/* constrain_cpu, src/blas.c:408-415 (GPU twin constrain_kernel blas_kernels.cu:450): the `clip=` clamp that
 * UpdateConvolutionalLayerGpu applies to the weights after the update (convolutional_kernels.cu:919-920).  The
 * reference's CPU UpdateConvolutionalLayer never calls it: the oracle applies it where the GPU path does. */
void orc_constrain(int size, float alpha, float* x)
{
  for (int i = 0; i < size; ++i) x[i] = fminf(alpha, fmaxf(-alpha, x[i]));
}

/* ----------------------------------------------------------------- maxpool */

/* ForwardMaxpoolLayer generic loop, src/maxpool_layer.cpp:255-297 (same
 * definition as the scalar forward_maxpool_layer_avx src/gemm.c:2725-2765):
 * window origin (i*stride - pad/2, j*stride - pad/2), out of range = -FLT_MAX,
 * strict '>' so the first maximum wins; indexes = flat input index. */
void orc_maxpool_forward(const float* input, float* output, int* indexes,
    int batch, int c, int h, int w, int size, int stride_x, int stride_y,
    int pad)
{
  const int out_w = (w + pad - size) / stride_x + 1; /* maxpool_layer.cpp:62-63 */
  const int out_h = (h + pad - size) / stride_y + 1;
  const int w_offset = -pad / 2;
  const int h_offset = -pad / 2;
  for (int b = 0; b < batch; ++b)
  {
#pragma omp parallel for
    for (int k = 0; k < c; ++k)
      for (int i = 0; i < out_h; ++i)
        for (int j = 0; j < out_w; ++j)
        {
          int out_index = j + out_w * (i + out_h * (k + c * b));
          float max = -FLT_MAX;
          int max_i = -1;
          for (int n = 0; n < size; ++n)
            for (int m = 0; m < size; ++m)
            {
              int cur_h = h_offset + i * stride_y + n;
              int cur_w = w_offset + j * stride_x + m;
              int index = cur_w + w * (cur_h + h * (k + b * c));
              int valid = (cur_h >= 0 && cur_h < h && cur_w >= 0 && cur_w < w);
              float val = (valid != 0) ? input[index] : -FLT_MAX;
              max_i = (val > max) ? index : max_i;
              max = (val > max) ? val : max;
            }
          output[out_index] = max;
          if (indexes)
            indexes[out_index] = max_i;
        }
  }
}

/* BackwardMaxpoolLayer, src/maxpool_layer.cpp:312-324 */
void orc_maxpool_backward(
    const float* delta, const int* indexes, int total, float* prev_delta)
{
  for (int i = 0; i < total; ++i) prev_delta[indexes[i]] += delta[i];
}

/* ------------------------------------------------ route/shortcut/upsample */

/* ForwardRouteLayer, src/route_layer.c:87-104: one source at a time. Call
 * once per source with the running channel `offset` (in floats per image). */
void orc_route_copy(const float* input, int input_size, int groups,
    int group_id, int batch, float* output, int outputs, int offset)
{
  int part = input_size / groups;
  for (int j = 0; j < batch; ++j)
    memcpy(output + offset + (size_t)j * outputs,
        input + (size_t)j * input_size + (size_t)part * group_id,
        (size_t)part * sizeof(float));
}

/* BackwardRouteLayer, src/route_layer.c:106-122 (axpy alpha=1 into source) */
void orc_route_backward(const float* delta, int outputs, int offset,
    int input_size, int groups, int group_id, int batch, float* src_delta)
{
  int part = input_size / groups;
  for (int j = 0; j < batch; ++j)
  {
    const float* d = delta + offset + (size_t)j * outputs;
    float* s = src_delta + (size_t)j * input_size + (size_t)part * group_id;
    for (int i = 0; i < part; ++i) s[i] += 1 * d[i];
  }
}

/* ForwardShortcutLayer same-shape branch, src/shortcut_layer.c:145-174
 * (activation applied by the caller through orc_activate_array). */
void orc_shortcut_forward(
    const float* input, const float* from, float* output, int total)
{
#pragma omp parallel for
  for (int i = 0; i < total; ++i) output[i] = input[i] + from[i];
}

/* BackwardShortcutCpu, src/blas.c:101-129 with n=1: prev_delta += delta and
 * from_delta += delta. */
void orc_shortcut_backward(
    const float* delta, int total, float* prev_delta, float* from_delta)
{
  for (int i = 0; i < total; ++i)
  {
    prev_delta[i] += delta[i];
    from_delta[i] += delta[i];
  }
}

/* upsample_cpu forward, src/blas.c:382-406 via ForwardUpsampleLayer
 * src/upsample_layer.c:76-89 (output zero-filled, then out = scale*in). */
void orc_upsample_forward(const float* in, int w, int h, int c, int batch,
    int stride, float scale, float* out)
{
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < c; ++k)
      for (int j = 0; j < h * stride; ++j)
        for (int i = 0; i < w * stride; ++i)
        {
          size_t in_index = (size_t)b * w * h * c + (size_t)k * w * h +
                            (size_t)(j / stride) * w + i / stride;
          size_t out_index = (size_t)b * w * h * c * stride * stride +
                             (size_t)k * w * h * stride * stride +
                             (size_t)j * w * stride + i;
          out[out_index] = scale * in[in_index];
        }
}

/* upsample_cpu backward (forward=0): in += scale*out */
void orc_upsample_backward(const float* delta, int w, int h, int c, int batch,
    int stride, float scale, float* prev_delta)
{
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < c; ++k)
      for (int j = 0; j < h * stride; ++j)
        for (int i = 0; i < w * stride; ++i)
        {
          size_t in_index = (size_t)b * w * h * c + (size_t)k * w * h +
                            (size_t)(j / stride) * w + i / stride;
          size_t out_index = (size_t)b * w * h * c * stride * stride +
                             (size_t)k * w * h * stride * stride +
                             (size_t)j * w * stride + i;
          prev_delta[in_index] += scale * delta[out_index];
        }
}

/* -------------------------------------------------------------------- yolo */

/* EntryIndex, src/yolo_layer.cpp:380-386 */
static int entry_index(int lw, int lh, int classes, int outputs, int batch,
    int location, int entry)
{
  int n = location / (lw * lh);
  int loc = location % (lw * lh);
  return batch * outputs + n * lw * lh * (4 + classes + 1) + entry * lw * lh +
         loc;
}

/* ForwardYoloLayer inference part, src/yolo_layer.cpp:388-407 (the CPU build's
 * `#ifndef GPU` block): copy, logistic on x,y then x = x*s - 0.5*(s-1)
 * (scal_add_cpu src/blas.c:252-256; BETA narrowed to float at the call),
 * logistic on obj+classes; w,h raw. */
void orc_yolo_forward(const float* input, float* output, int batch, int lw,
    int lh, int n_anchors, int classes, float scale_x_y)
{
  const int outputs = lh * lw * n_anchors * (classes + 4 + 1);
  memcpy(output, input, (size_t)outputs * batch * sizeof(float));
  const float beta = -0.5 * (scale_x_y - 1);
  for (int b = 0; b < batch; ++b)
    for (int n = 0; n < n_anchors; ++n)
    {
      int index = entry_index(lw, lh, classes, outputs, b, n * lw * lh, 0);
      orc_activate_array_plain(output + index, 2 * lw * lh, ORC_LOGISTIC);
      for (int i = 0; i < 2 * lw * lh; ++i)
        output[index + i] = output[index + i] * scale_x_y + beta;
      index = entry_index(lw, lh, classes, outputs, b, n * lw * lh, 4);
      orc_activate_array_plain(
          output + index, (1 + classes) * lw * lh, ORC_LOGISTIC);
    }
}

/* YoloNumDetections, src/yolo_layer.cpp:779-792, generalised to batch item b
 * (the reference reads item 0 only). */
int orc_yolo_num_detections(const float* output, int b, int lw, int lh,
    int n_anchors, int classes, float thresh)
{
  const int outputs = lh * lw * n_anchors * (classes + 4 + 1);
  int count = 0;
  for (int n = 0; n < n_anchors; ++n)
    for (int i = 0; i < lw * lh; ++i)
    {
      int obj = entry_index(lw, lh, classes, outputs, b, n * lw * lh + i, 4);
      if (output[obj] > thresh)
        ++count;
    }
  return count;
}

/* GetYoloDetections + GetYoloBox, src/yolo_layer.cpp:794-832, :139-148.
 * dets: per detection [x,y,w,h,objectness, prob[classes]]; ids: per detection
 * [anchor n, row, col] (the "box indices" the parity bar pins).  expf: the
 * reference's exp(float) is the float overload. */
int orc_yolo_detections(const float* output, int b, int lw, int lh,
    int n_anchors, int classes, const float* biases, const int* mask,
    int net_w, int net_h, float thresh, float* dets, int* ids)
{
  const int outputs = lh * lw * n_anchors * (classes + 4 + 1);
  const int stride = lw * lh;
  const int rec = 5 + classes;
  int count = 0;
  for (int n = 0; n < n_anchors; ++n)
    for (int i = 0; i < lw * lh; ++i)
    {
      int loc = n * lw * lh + i;
      int obj_idx = entry_index(lw, lh, classes, outputs, b, loc, 4);
      float objectness = output[obj_idx];
      if (objectness <= thresh)
        continue;
      int box_idx = entry_index(lw, lh, classes, outputs, b, loc, 0);
      int col = i % lw;
      int row = i / lw;
      float* d = dets + (size_t)count * rec;
      int a = mask[n];
      d[0] = (col + output[box_idx + 0 * stride]) / lw;
      d[1] = (row + output[box_idx + 1 * stride]) / lh;
      d[2] = expf(output[box_idx + 2 * stride]) * biases[2 * a] / net_w;
      d[3] = expf(output[box_idx + 3 * stride]) * biases[2 * a + 1] / net_h;
      d[4] = objectness;
      for (int j = 0; j < classes; ++j)
      {
        int cls = entry_index(lw, lh, classes, outputs, b, loc, 4 + 1 + j);
        float prob = objectness * output[cls];
        d[5 + j] = (prob > thresh) ? prob : 0;
      }
      if (ids)
      {
        ids[3 * count + 0] = n;
        ids[3 * count + 1] = row;
        ids[3 * count + 2] = col;
      }
      ++count;
    }
  return count;
}

/* ---- sibling-cfg layer kinds -------------------------------------- */

/* ForwardAvgpoolLayer, src/avgpool_layer.cpp:40-56: sequential fp32 sum / (h*w). */
void orc_avgpool_forward(const float* in, float* out, int batch, int c, int h, int w)
{
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < c; ++k)
    {
      const int out_index = k + b * c;
      out[out_index] = 0;
      for (int i = 0; i < h * w; ++i)
      {
        const size_t in_index = (size_t)i + (size_t)h * w * (k + b * c);
        out[out_index] += in[in_index];
      }
      out[out_index] /= h * w;
    }
}

/* BackwardAvgpoolLayer, src/avgpool_layer.cpp:58-72 */
void orc_avgpool_backward(const float* delta, float* prev_delta, int batch, int c, int h, int w)
{
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < c; ++k)
    {
      const int out_index = k + b * c;
      for (int i = 0; i < h * w; ++i)
      {
        const size_t in_index = (size_t)i + (size_t)h * w * (k + b * c);
        prev_delta[in_index] += delta[out_index] / (h * w);
      }
    }
}

/* ForwardScaleChannelsLayer, src/scale_channels_layer.c:70-95 (activation by the caller) */
void orc_scale_channels_forward(const float* in, const float* from, float* out, int batch,
    int out_c, int out_h, int out_w, int scale_wh)
{
  const int size = batch * out_c * out_w * out_h;
  const int channel_size = out_w * out_h;
  const int batch_size = out_c * out_w * out_h;
  if (scale_wh)
    for (int i = 0; i < size; ++i)
    {
      const int input_index = i % channel_size + (i / batch_size) * channel_size;
      out[i] = in[input_index] * from[i];
    }
  else
    for (int i = 0; i < size; ++i) out[i] = in[i / channel_size] * from[i];
}

/* BackwardScaleChannelsLayer, src/scale_channels_layer.c:97-127 (gradient_array by the caller) */
void orc_scale_channels_backward(const float* delta, const float* in, const float* from,
    float* from_delta, float* in_delta, int batch, int out_c, int out_h, int out_w, int scale_wh)
{
  const int size = batch * out_c * out_w * out_h;
  const int channel_size = out_w * out_h;
  const int batch_size = out_c * out_w * out_h;
  for (int i = 0; i < size; ++i)
  {
    const int si = scale_wh ? (i % channel_size + (i / batch_size) * channel_size) : i / channel_size;
    in_delta[si] += delta[i] * from[i];
    from_delta[i] += in[si] * delta[i];
  }
}

/* UpdateBatchnormLayer, src/batchnorm_layer.cpp:257-266 */
void orc_batchnorm_update(float* biases, float* bias_updates, float* scales, float* scale_updates,
    int c, int batch, float learning_rate, float momentum)
{
  for (int i = 0; i < c; ++i) biases[i] += (learning_rate / batch) * bias_updates[i];
  for (int i = 0; i < c; ++i) bias_updates[i] *= momentum;
  for (int i = 0; i < c; ++i) scales[i] += (learning_rate / batch) * scale_updates[i];
  for (int i = 0; i < c; ++i) scale_updates[i] *= momentum;
}


/* ---- [Gaussian_yolo] head, inference ------------------------------------------
 * EntryGaussianIndex src/gaussian_yolo_layer.cpp:477-484: 8 box entries (mu/sigma of x, y, w, h) +
 * objectness + classes per anchor. */
static int gaussian_entry_index(int lw, int lh, int classes, int outputs, int batch, int location, int entry)
{
  int n = location / (lw * lh);
  int loc = location % (lw * lh);
  return batch * outputs + n * lw * lh * (8 + classes + 1) + entry * lw * lh + loc;
}

/* ForwardGaussianYoloLayer, inference part (src/gaussian_yolo_layer.cpp:486-518; the GPU twin :934-966
 * applies the same activations): logistic on entries 0-3 (x, y: mu and sigma), 5, 7 (sigma of w, h) and
 * 8.. (objectness, classes); scale_x_y on the two mu planes; entries 4 and 6 (mu of w, h) stay raw. */
void orc_gaussian_yolo_forward(const float* input, float* output, int batch, int lw, int lh, int n_anchors,
    int classes, float scale_x_y)
{
  const int outputs = lh * lw * n_anchors * (classes + 8 + 1);
  const int wh = lw * lh;
  memcpy(output, input, (size_t)outputs * batch * sizeof(float));
  const float beta = -0.5 * (scale_x_y - 1);
  for (int b = 0; b < batch; ++b)
    for (int n = 0; n < n_anchors; ++n)
    {
      for (int e = 0; e <= 2; e += 2)
      {
        int index = gaussian_entry_index(lw, lh, classes, outputs, b, n * wh, e);
        orc_activate_array_plain(output + index, 2 * wh, ORC_LOGISTIC);
        for (int i = 0; i < wh; ++i) output[index + i] = output[index + i] * scale_x_y + beta;
      }
      orc_activate_array_plain(output + gaussian_entry_index(lw, lh, classes, outputs, b, n * wh, 5), wh, ORC_LOGISTIC);
      orc_activate_array_plain(output + gaussian_entry_index(lw, lh, classes, outputs, b, n * wh, 7), wh, ORC_LOGISTIC);
      orc_activate_array_plain(output + gaussian_entry_index(lw, lh, classes, outputs, b, n * wh, 8), (1 + classes) * wh,
          ORC_LOGISTIC);
    }
}

/* GaussianYoloNumDetections :859-874, generalised to batch item b */
int orc_gaussian_yolo_num_detections(const float* output, int b, int lw, int lh, int n_anchors, int classes, float thresh)
{
  const int outputs = lh * lw * n_anchors * (classes + 8 + 1);
  int count = 0;
  for (int i = 0; i < lw * lh; ++i)
    for (int n = 0; n < n_anchors; ++n)
      if (output[gaussian_entry_index(lw, lh, classes, outputs, b, n * lw * lh + i, 8)] > thresh)
        ++count;
  return count;
}

/* GetGaussianYoloDetections :876-930 + GetGaussianYoloBox :151-177 (yolo_point = center).
 * dets: per detection [x, y, w, h, objectness, prob[classes], uc[4]]; ids: [anchor, row, col]. */
int orc_gaussian_yolo_detections(const float* output, int b, int lw, int lh, int n_anchors, int classes,
    const float* biases, const int* mask, int net_w, int net_h, float thresh, float* dets, int* ids)
{
  const int outputs = lh * lw * n_anchors * (classes + 8 + 1);
  const int stride = lw * lh;
  const int rec = 5 + classes + 4;
  int count = 0;
  for (int n = 0; n < n_anchors; ++n)
    for (int i = 0; i < lw * lh; ++i)
    {
      int loc = n * lw * lh + i;
      float objectness = output[gaussian_entry_index(lw, lh, classes, outputs, b, loc, 8)];
      if (objectness <= thresh)
        continue;
      int box = gaussian_entry_index(lw, lh, classes, outputs, b, loc, 0);
      int col = i % lw, row = i / lw;
      int a = mask[n];
      float* d = dets + (size_t)count * rec;
      d[2] = expf(output[box + 4 * stride]) * biases[2 * a] / net_w;
      d[3] = expf(output[box + 6 * stride]) * biases[2 * a + 1] / net_h;
      d[0] = (col + output[box + 0 * stride]) / lw;
      d[1] = (row + output[box + 2 * stride]) / lh;
      d[4] = objectness;
      float uc[4];
      for (int k = 0; k < 4; ++k) uc[k] = output[gaussian_entry_index(lw, lh, classes, outputs, b, loc, 2 * k + 1)];
      for (int j = 0; j < classes; ++j)
      {
        float cls = output[gaussian_entry_index(lw, lh, classes, outputs, b, loc, 9 + j)];
        float uc_avg = (uc[0] + uc[1] + uc[2] + uc[3]) / 4.0;
        float prob = objectness * cls * (1.0 - uc_avg);
        d[5 + j] = (prob > thresh) ? prob : 0;
      }
      for (int k = 0; k < 4; ++k) d[5 + classes + k] = uc[k];
      if (ids)
      {
        ids[3 * count + 0] = n;
        ids[3 * count + 1] = row;
        ids[3 * count + 2] = col;
      }
      ++count;
    }
  return count;
}


/* adam_update_gpu + adam_kernel, src/blas_kernels.cu:99-134 (the reference has NO CPU adam: this restates its
 * GPU launch sequence one launch at a time; the CUDA build contracts a*b+c into FMAs, here every product is
 * rounded -- "parity unpinned" for this function, the HIP kernel is checked against it as written). */
void orc_adam_update(float* w, float* d, float* m, float* v, float B1, float B2, float eps, float decay, float rate,
    int n, int batch, int t)
{
  for (int i = 0; i < n; ++i) m[i] *= B1;                       /* scal_ongpu(n, B1, m) */
  for (int i = 0; i < n; ++i) v[i] *= B2;                       /* scal_ongpu(n, B2, v) */
  const float db = -decay * batch;
  for (int i = 0; i < n; ++i) d[i] += db * w[i];                /* axpy_ongpu(n, -decay*batch, w, d) */
  const float ob1 = 1 - B1, ob2 = 1 - B2;
  for (int i = 0; i < n; ++i) m[i] += ob1 * d[i];               /* axpy_ongpu(n, 1-B1, d, m) */
  for (int i = 0; i < n; ++i) d[i] *= d[i];                     /* mul_ongpu(n, d, d) */
  for (int i = 0; i < n; ++i) v[i] += ob2 * d[i];               /* axpy_ongpu(n, 1-B2, d, v) */
  for (int i = 0; i < n; ++i)                                   /* adam_kernel */
  {
    float mhat = m[i] / (1.f - powf(B1, t));
    float vhat = v[i] / (1.f - powf(B2, t));
    w[i] = w[i] + rate * mhat / (sqrtf(vhat) + eps);
  }
  for (int i = 0; i < n; ++i) d[i] = 0;                         /* fill_ongpu(n, 0, d) */
}
