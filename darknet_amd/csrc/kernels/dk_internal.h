// dk_internal.h -- library-internal C++ entry points (not exported).
#pragma once
#include "dk_kernels.h"

#include <stddef.h>

// dk_conv_forward with an explicit tile configuration (cfg < 0: heuristic).
// out_ctot > 0: `y` is a channel slice of a wider tensor with out_ctot channels
// (zero-copy concatenation: batch stride = out_ctot*oh*ow); no residual then.
// dual != nullptr: ONE launch computes two convolutions of the same input (d->n = n1 + n2 filters,
// weights/biases concatenated): filters [0, m_split) go to y, filters [m_split, d->n) to dual->y2
// (each optionally a channel slice, out_ctot / out_ctot2).
struct DkConvDual
{
  float* y2;
  int m_split, out_ctot2;
};
int dk_conv_forward_cfg(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream,
    int cfg, int out_ctot = 0, const DkConvDual* dual = nullptr, const float* wino_filters = nullptr);
// wino_filters != nullptr: the transformed filters (dk_conv_wino_transform_weights) to use when `cfg` is the
// Winograd configuration, instead of the registry entry of `weights` (training: the filters change every step)
// dk_conv_backward_data with an explicit gather tile configuration (cfg < 0 or not a gather shape: heuristic)
// wt_tapmajor != 0: `wt` comes from dk_transpose_weights_tapmajor and the layer takes the parity-class form
// (dk_conv_dgrad_tapmajor): stride-2 layers visit only the taps whose parity matches the pixel class
int dk_conv_backward_data_cfg(const DkConvDesc* d, const float* delta, const float* wt, float* prev_delta,
    void* stream, int cfg, int wt_tapmajor);
bool dk_conv_dgrad_tapmajor(const DkConvDesc* d);
// number of gather (implicit-GEMM) tile configurations: indices [0, n) of the configuration table
int dk_conv_num_gather_configs();
// dk_conv_backward_weights with an explicit tile shape (0..3: 128x128, 64x128, 128x64, 64x64 rows x taps; < 0: heuristic)
// 1 when configuration 4 of dk_conv_backward_weights_cfg (conv_wgrad3_f32) takes the layer
bool dk_wgrad3_applicable(const DkConvDesc* d);
int dk_conv_backward_weights_cfg(const DkConvDesc* d, const float* x, const float* delta, float* weight_updates,
    void* stream, int cfg);
int dk_conv_forward_half_strided(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream,
    int out_ctot);
// maxpool / upsample writing a channel slice of a wider tensor (out_batch_stride floats
// between batch items; 0 = dense)
int dk_maxpool_forward_strided(const float* x, float* y, int* indexes, int batch, int c, int h,
    int w, int size, int stride_x, int stride_y, int pad, size_t out_batch_stride, void* stream);
int dk_upsample_forward_strided(const float* in, int w, int h, int c, int batch, int stride,
    float scale, float* out, size_t out_batch_stride, void* stream);
// Makes sure the per-shape tap table exists (must happen outside stream capture).
void dk_conv_prepare(const DkConvDesc* d);
int dk_conv_num_configs();
// true when tile configuration `cfg` can run this layer (the direct 3x3 configurations,
// indices >= the number of gather configurations, only take 3x3/s1/p1 layers)
bool dk_conv_config_applicable(const DkConvDesc* d, int cfg);
// true for the Winograd 3x3 configurations (they need transformed filters registered for the layer)
bool dk_conv_config_is_wino(int cfg);
// DK_FAST_MISH (default on): closed-form mish / mish gradient instead of the libm chains
bool dk_fast_mish_enabled();
// conv3x3_direct_f16.hip: fp16-operand patch-in-LDS kernel with weights pre-packed per layer
int dk_conv_forward_half_direct(const DkConvDesc* d, const float* x, const void* packed_weights,
    const float* biases, float* y, const float* residual, void* stream, int out_ctot);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size): function
// attributes are per device, and several host threads (one per GPU) may launch concurrently
void dk_set_max_dynamic_lds(const void* kernel, int bytes);

// ---- launch profiling shared by every conv-family launcher (bench roofline lines) ----------
// When profiling is on (dk_profile_enable), a launch is bracketed by HIP events on ITS stream and
// booked under a slot: slots < 256 are the forward conv configurations (cfg * 4 + variant); kernels
// outside that table (fp16 operands, weight gradient ...) register a slot by their rocprofv3 name.
struct DkProfScope
{
  void* e0 = nullptr;
};
bool dk_prof_on();
int dk_prof_named_slot(const char* kernel_name);
void dk_prof_begin(DkProfScope& s, void* stream);
void dk_prof_end(DkProfScope& s, void* stream, int slot, double gflop);

// multi-tensor SGD (train_ops.hip): one launch for every tensor of UpdateNetworkGpu
void* dk_sgd_plan_create(int ntensors, float* const* weights, float* const* updates, const size_t* counts,
    const float* lr_scales, const int* use_decay);
void dk_sgd_plan_destroy(void* plan);
int dk_sgd_update_multi(void* plan, int batch, float learning_rate, float momentum, float decay, void* stream);

// ---- derived weight tensors of a training step, all in one launch (conv3x3_wino.hip) -------------------------
// kind 0: dst[c][m][t] = w[m][c][t]; 1: dst[c][m][t] = w[m][c][ss-1-t]; 2: dst[c][t*M + m] = w[m][c][t];
// 3: dst = Winograd filters of w (M filters, C channels); 4: dst = Winograd filters of the data-gradient
// convolution (C filters, M channels, taps rotated).  M / C are the LAYER's filters / channels per group.
struct DkPrepTask
{
  const float* w;
  float* dst;
  int M, C, ss, kind;
  int first_block;   // set by dk_train_prep_create
};
void* dk_train_prep_create(int ntasks, DkPrepTask* host_tasks);
void dk_train_prep_destroy(void* plan);
int dk_train_prep_run(void* plan, void* stream);

// deterministic reductions requested (dk_set_deterministic / DK_DETERMINISTIC): see conv_wgrad.hip
bool dk_deterministic();
