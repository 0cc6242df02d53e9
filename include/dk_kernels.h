/*
 * dk_kernels.h -- flat C-ABI over the hand-written gfx950 (CDNA4) kernels of the
 * Darknet convolutional path.  Plain device pointers and sizes only; every
 * call enqueues on `stream` (a hipStream_t passed as void*; NULL = the
 * per-device compute stream of dark_hip.h) and does not synchronise.
 *
 * Each entry point names the reference GPU function it replaces
 * (Ravicmoon/darknet, file:line under src/) -- the layer-level drop-ins
 * (ForwardConvolutionalLayerGpu & co., include/yolo_core_hip.h) are thin
 * wrappers over these.  Activation ids are the reference's ACTIVATION enum
 * values (src/yolo_core.h:69-92): LOGISTIC 0, RELU 1, LINEAR 4, LEAKY 8,
 * MISH 17.  All tensors are fp32 NCHW, batch-major.
 *
 * Return value: 0 on success, non-zero (after printing to stderr) on invalid
 * arguments.  Device errors go through check_error() and exit(), like the
 * reference (src/dark_cuda.c:85-106).
 */
#ifndef DK_KERNELS_H
#define DK_KERNELS_H

#include <stddef.h>

#ifndef DK_API
#define DK_API __attribute__((visibility("default")))
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DkConvDesc
{
  int batch;            /* images */
  int c, h, w;          /* input: total channels, height, width */
  int n;                /* filters (total output channels) */
  int groups;
  int size;             /* square kernel */
  int stride_x, stride_y;
  int dilation;
  int pad;              /* l->pad; effective padding = pad * dilation */
  int activation;       /* ACTIVATION id applied in the epilogue */
} DkConvDesc;

/*
 * Implicit-GEMM forward convolution on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces, in one launch, the reference's fill_ongpu + per-image
 * { im2col_gpu_ext ; gemm_ongpu(cublasSgemm) } loop + add_bias_gpu +
 * activate_array_ongpu (src/convolutional_kernels.cu:471-532).
 *   y[b][f][oy][ox] = act( bias[f] + sum_k W[f][k] * im2col(x_b)[k][oy*ow+ox] )
 *                     (+ residual[b][f][oy][ox] if residual != NULL)
 * biases may be NULL (treated as 0; used by the training path where BN
 * follows).  activation_input (may be NULL) receives the pre-activation
 * value, as activate_array_mish_ongpu stores it (src/activation_kernels.cu:290).
 * No im2col buffer is materialised and no workspace is needed.
 * Memory contract: every tensor is read and written strictly inside
 * [ptr, ptr + its size): x, weights, y, residual may be any device pointers,
 * no slack behind them is assumed.  (The Winograd kernel's 16-byte row pieces
 * would read up to 12 bytes past x on feature maps whose width is not a
 * multiple of 4; it asks the allocation -- hipMemGetAddressRange -- and takes
 * its 4-byte-piece variant when fewer than 16 readable bytes follow x.)
 */
DK_API int dk_conv_forward(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual,
    float* activation_input, void* stream);

/* fp16-operand variant (BASELINE config C5; reference: the CUDNN_HALF branch,
 * src/convolutional_kernels.cu:357-456): x and weights are rounded to fp16
 * (round-to-nearest-even) while being staged, products accumulate in fp32 on
 * v_mfma_f32_32x32x16_f16, bias/activation/output stay fp32.  Only for layers
 * the reference's rule admits (dk_conv_half_eligible: size > 1, c % 8 == 0,
 * n % 8 == 0, groups == 1, not the network's first layer). */
DK_API int dk_conv_forward_half(const DkConvDesc* d, const float* x, const float* weights,
    const float* biases, float* y, const float* residual, float* activation_input, void* stream);
DK_API int dk_conv_half_eligible(const DkConvDesc* d, int layer_index);

/* Tile-configuration control for tuning: cfg >= 0 forces one of the compiled
 * tile shapes for every subsequent dk_conv_forward (-1 = heuristic).  Returns
 * the number of compiled configurations. */
DK_API int dk_conv_force_config(int cfg);
/* Which configuration the heuristic (or the forced value) picks for d. */
DK_API int dk_conv_pick_config(const DkConvDesc* d);
/* Human-readable name of configuration cfg, or NULL. */
DK_API const char* dk_conv_config_name(int cfg);

/* forward_maxpool_layer_kernel, src/maxpool_layer_kernels.cu:58-101: window
 * origin (i*stride - pad/2), out-of-range = -INF, first max wins; indexes (may
 * be NULL) = flat input index of the max. */
DK_API int dk_maxpool_forward(const float* x, float* y, int* indexes, int batch, int c,
    int h, int w, int size, int stride_x, int stride_y, int pad, void* stream);

/* One source of ForwardRouteLayerGpu, src/route_layer.c:124-142: for every batch
 * item copy part = input_size/groups floats from src + j*input_size +
 * part*group_id to dst + offset + j*outputs (one launch for all batch items). */
DK_API int dk_route_copy(const float* src, int input_size, int groups, int group_id,
    int batch, float* dst, int outputs, int offset, void* stream);

/* shortcut_singlelayer_simple_kernel, src/blas_kernels.cu:941-978 (n = 1, same
 * shape) + activation: out = act(in + from). */
DK_API int dk_shortcut_forward(const float* in, const float* from, float* out,
    size_t total, int activation, void* stream);

/* upsample_kernel forward, src/blas_kernels.cu:1121-1146: nearest x stride,
 * out = scale * in (written once; no zero-fill + accumulate). */
DK_API int dk_upsample_forward(const float* in, int w, int h, int c, int batch,
    int stride, float scale, float* out, void* stream);

/* fp16-operand path for 3x3/s1/p1 layers with c % 16 == 0 (the cuDNN half branch,
 * src/convolutional_kernels.cu:357-456): number of halves of the re-laid-out weights
 * [n][c/16][tap][16] (0: the layer does not take this kernel), and the one-time packing
 * (fp32 -> fp16 round to nearest even).  The network plan packs once per layer at load. */
DK_API size_t dk_conv_half_direct_weights_size(const DkConvDesc* d);
DK_API int dk_conv_half_pack_weights(const DkConvDesc* d, const float* weights, void* packed, void* stream);
/* Winograd F(2x2,3x3) path of dk_conv_forward (3x3/s1/p1, groups 1, c % 8 == 0, n % 64 == 0): the filters are
 * transformed once (U = G g G^T, dk_conv_wino_weights_size floats) and registered under the layer's weights
 * pointer; dk_conv_forward_cfg with the Winograd configuration index then finds them.  Re-transform after
 * every change of the weights; register U = NULL before freeing. */
DK_API size_t dk_conv_wino_weights_size(const DkConvDesc* d);
DK_API int dk_conv_wino_transform_weights(const DkConvDesc* d, const float* weights, float* U, void* stream);
DK_API void dk_conv_wino_register(const float* weights, const float* U);
/* y = act(conv(fp16(x), packed fp16 weights) + bias) (+ residual), fp32 accumulate */
DK_API int dk_conv_forward_half_packed(const DkConvDesc* d, const float* x, const void* packed_weights,
    const float* biases, float* y, const float* residual, void* stream);

/* Device half of GetYoloDetections (src/yolo_layer.cpp:794-834): appends one record
 * {int tag, int image, int loc = n*w*h + i, x, y, w, h, objectness, classes...} (3 + 5 + classes
 * floats, ints stored bitwise) per predictor with objectness > thresh to records[], counting in
 * *counter (may exceed cap: the caller then falls back to the full head). */
DK_API int dk_yolo_compact(const float* decoded, int batch, int lw, int lh, int n_anchors,
    int classes, float thresh, int tag, float* records, int* counter, int cap, void* stream);

/* NmsSort on the device (src/box.cpp:393-419 on the boxes of src/yolo_layer.cpp:139-148, :794-834):
 * `records` are dk_yolo_compact's candidates [tag, image, loc, x, y, w, h, obj, cls...]; on return
 * x..h hold the box (centre-normalised) and cls[j] the class probability after thresholding
 * (obj * cls[j] > thresh, else 0) and per-class suppression (nms_kind 0 = greedy IoU, 1 = DIoU with
 * beta).  heads_dev[tag] describes yolo layer `tag` (device memory).  *overflow_dev is incremented
 * when one (image, class) has more than 4096 live candidates (the caller then falls back). */
typedef struct DkYoloHead
{
  int lw, lh;
  float anchor_w[8], anchor_h[8];   /* biases[2 * mask[n]], biases[2 * mask[n] + 1] */
} DkYoloHead;
DK_API int dk_nms_records(float* records, int count, int classes, const DkYoloHead* heads_dev, int net_w,
    int net_h, int batch, float thresh, float nms_thresh, int nms_kind, float beta, int* overflow_dev, void* stream);

/* Mat2Image (src/visualize.cpp:26-55) for `batch` interleaved u8 images of h rows of row_step bytes
 * already in device memory: chw[b][k][y][x] = hwc[b][y*row_step + x*c + k] / 255.0f. */
DK_API int dk_image_u8_to_chw(const unsigned char* hwc, float* chw, int batch, int w, int h, int c,
    size_t row_step, void* stream);
/* cv::resize (INTER_LINEAR, 8-bit, OpenCV's fixed-point arithmetic) + optional RGB<->BGR swap + Mat2Image in one
 * pass: the whole input step of the reference's ProcImage (src/yolo_core.cpp:104-112).  src: batch frames of
 * src_h rows of src_row_step bytes (interleaved, c channels); chw: [batch][c][h][w] floats in [0, 1]. */
DK_API int dk_image_resize_u8_to_chw(const unsigned char* src_hwc, int src_w, int src_h, size_t src_row_step,
    float* chw, int batch, int w, int h, int c, int swap_rb, void* stream);

/* ForwardYoloLayerGpu decode, src/yolo_layer.cpp:836-853, fused into one
 * launch: copy; logistic on x,y then v*scale_x_y - 0.5*(scale_x_y-1);
 * logistic on objectness and classes; w,h raw. */
DK_API int dk_yolo_forward(const float* in, float* out, int batch, int lw, int lh,
    int n_anchors, int classes, float scale_x_y, void* stream);
/* [Gaussian_yolo] decode (src/gaussian_yolo_layer.cpp:934-966): (8 + 1 + classes) entries per anchor */
DK_API int dk_gaussian_yolo_forward(const float* in, float* out, int batch, int lw, int lh, int n_anchors,
    int classes, float scale_x_y, void* stream);

/* activate_array_ongpu, src/activation_kernels.cu:505-560 (in place). */
DK_API int dk_activate_array(float* x, size_t n, int activation, void* stream);
/* activate_array_mish_ongpu, src/activation_kernels.cu:290-309, :577 */
DK_API int dk_activate_array_mish(const float* x, size_t n, float* activation_input,
    float* out, void* stream);
/* add_bias_gpu / scale_bias_gpu, src/convolutional_kernels.cu:15-35,
 * src/blas_kernels.cu:45-65 */
DK_API int dk_add_bias(float* out, const float* biases, int batch, int n, int size, void* stream);
DK_API int dk_scale_bias(float* out, const float* scales, int batch, int n, int size, void* stream);

/* fill_ongpu / simple_copy_ongpu / axpy_ongpu / scal_ongpu,
 * src/blas_kernels.cu:427-532 (unit strides only). */
DK_API int dk_fill(size_t n, float alpha, float* x, void* stream);
DK_API int dk_copy(size_t n, const float* x, float* y, void* stream);
DK_API int dk_axpy(size_t n, float alpha, const float* x, float* y, void* stream);
DK_API int dk_scal(size_t n, float alpha, float* x, void* stream);
/* constrain_ongpu (blas_kernels.cu:450, :874): x = min(alpha, max(-alpha, x)) -- the `clip=` key of a layer */
DK_API int dk_constrain(size_t n, float alpha, float* x, void* stream);

/* ---- training path ------------------------------------------------------- */

/* Conv GEMM stage + batch-norm with batch statistics (ForwardBatchnormLayerGpu,
 * src/batchnorm_layer.cpp:268-322, with the CPU path's numerics: variance / (N-1),
 * eps 1e-6, rolling = .9*rolling + .1*batch): `raw` is the convolution output
 * without bias (dk_conv_forward with biases = NULL, LINEAR); when train != 0 the
 * batch mean/variance are computed, the rolling statistics updated, x (= raw) and
 * x_norm saved; then out = act(x_norm*scale + bias), activation_input (may be
 * NULL) receives the pre-activation.  train == 0 normalises with the rolling
 * statistics. */
DK_API int dk_bn_forward_train(const float* raw, float* x_save, float* x_norm,
    float* activation_input, float* out, float* mean, float* variance, float* rolling_mean,
    float* rolling_variance, const float* scales, const float* biases, int batch, int filters,
    int spatial, int activation, int train, void* stream);
/* gradient_array_ongpu / gradient_array_mish_ongpu, src/activation_kernels.cu:383-503:
 * delta *= f'(.), evaluated on the output y (mish: on the saved pre-activation). */
DK_API int dk_gradient_array(const float* y, const float* activation_input, float* delta,
    size_t n, int activation, void* stream);
/* backward_bias_gpu, src/convolutional_kernels.cu:37-67: bias_updates[f] += sum delta */
DK_API int dk_backward_bias(float* bias_updates, const float* delta, int batch, int n, int size,
    void* stream);
/* BackwardBatchnormLayerGpu, src/batchnorm_layer.cpp:324-374 with the CPU eps values
 * (1e-5): scale_updates += sum(delta*x_norm), bias_updates += sum(delta) (may be NULL),
 * delta <- normalised-delta.  mean_delta / variance_delta are scratch [filters]. */
DK_API int dk_bn_backward(float* delta, const float* x, const float* x_norm, const float* mean,
    const float* variance, const float* scales, float* mean_delta, float* variance_delta,
    float* scale_updates, float* bias_updates, int batch, int filters, int spatial, void* stream);
/* gradient_array + backward_batchnorm fused (activations.c:401-452 + batchnorm_layer.cpp:240-255):
 * x_norm, the pre-activation value and the activation gradient are recomputed from x with the
 * forward's own float operations instead of being stored; delta: d/d(output) in, d/d(x) out. */
DK_API int dk_bn_act_backward(float* delta, const float* x, const float* mean, const float* variance,
    const float* scales, const float* biases, float* mean_delta, float* variance_delta,
    float* scale_updates, float* bias_updates, int batch, int filters, int spatial, int activation,
    void* stream);
/* Weight gradient: weight_updates[m][k] += sum_n delta[m][n]*im2col(x)[k][n] (wgrad half of
 * BackwardConvolutionalLayerGpu, src/convolutional_kernels.cu:757-781). */
DK_API int dk_conv_backward_weights(const DkConvDesc* d, const float* x, const float* delta,
    float* weight_updates, void* stream);
/* Test / diagnostics hook: force a variant of the weight-gradient kernel in this process.  knob 0: tile shape 0..3
 * (128x128, 64x128, 128x64, 64x64 rows x taps) of the gather kernel, 4 / 5 the row-staged kernel of the 3x3 layers with
 * ~2 / ~1 workgroups per CU (where it applies; it is also the default there), knob 1: tap-major tiles (0 never, 1 where
 * applicable), knob 2: 16 / 8-byte loads (0 never); value < 0 restores the default.  Returns the previous value (-1 = default), -3 for an unknown
 * knob.  Together with dk_conv_force_config (forward / data-gradient kernels) and dk_set_deterministic this reaches every
 * kernel the training step's first-step timing can choose. */
DK_API int dk_train_force(int knob, int value);
/* Data gradient: prev_delta = col2im(W^T * delta), overwriting (dgrad half, :784-812).
 * wt = dk_transpose_weights(weights). */
DK_API int dk_conv_backward_data(const DkConvDesc* d, const float* delta, const float* wt,
    float* prev_delta, void* stream);
/* wt[g][c][(m,kh,kw)] = w[g][m][c][kh][kw]; call per group with M = n/groups, C = c/groups */
DK_API int dk_transpose_weights(const float* w, float* wt, int M, int C, int size, void* stream);
/* wt[c][m][t] = w[m][c][size*size-1-t]: the filters of the convolution that IS the data gradient of a stride-1 "same"
 * convolution (taps rotated by 180 degrees); dk_conv_forward on delta with these filters (c <-> n swapped in the
 * descriptor, no bias, linear) then overwrites prev_delta */
DK_API int dk_transpose_weights_flip(const float* w, float* wt, int M, int C, int size, void* stream);
/* 1 when tile / kernel configuration cfg (dk_conv_config_name) can run the layer d */
DK_API int dk_conv_config_can_run(const DkConvDesc* d, int cfg);
/* 1: weight gradients and BN channel sums through ordered workspaces instead of atomics (two runs of a training step
 * with the same kernel choices are then bitwise equal); 0: atomics (default, faster); -1: follow DK_DETERMINISTIC */
DK_API void dk_set_deterministic(int on);
/* Stride-2 layers (both directions, even input dimensions, one group, filters a multiple of 32, size 2 or 3): the
 * same data gradient with the pixels enumerated parity class by parity class and the contraction index tap-major
 * (wt = dk_transpose_weights_tapmajor(weights): wt[c][t * n + m] = w[m][c][t]), so that only the taps whose parity
 * matches a pixel are visited -- 9 instead of 36 tap-pixel pairs for a 3x3 layer.  Returns 1 when the layer does
 * not take this form. */
DK_API int dk_conv_backward_data_tapmajor(const DkConvDesc* d, const float* delta, const float* wt_tapmajor,
    float* prev_delta, void* stream);
DK_API int dk_transpose_weights_tapmajor(const float* w, float* wt, int M, int C, int size, void* stream);
/* backward_maxpool_layer_kernel, src/maxpool_layer_kernels.cu:103-143 (scatter-add by index) */
DK_API int dk_maxpool_backward(const float* delta, const int* indexes, size_t n,
    float* prev_delta, void* stream);
/* The same gradient without atomics (deterministic mode): every input element visits the windows that contain it in
 * row-major order.  Needs the layer geometry (`pad` as in dk_maxpool_forward: the window starts at -pad/2). */
DK_API int dk_maxpool_backward_gather(const float* delta, const int* indexes, int batch, int c, int h, int w,
    int out_h, int out_w, int size, int stride_x, int stride_y, int pad, float* prev_delta, void* stream);
/* one source of BackwardRouteLayerGpu, src/route_layer.c:144-160 */
DK_API int dk_route_backward(const float* delta, int outputs, int offset, int input_size,
    int groups, int group_id, int batch, float* src_delta, void* stream);
/* backward_shortcut_multilayer_kernel with n = 1, src/blas_kernels.cu:980-1034 */
DK_API int dk_shortcut_backward(const float* delta, size_t n, float* prev_delta,
    float* from_delta, void* stream);
/* upsample_kernel forward = 0, src/blas_kernels.cu:1121-1146 */
DK_API int dk_upsample_backward(const float* delta, int w, int h, int c, int batch, int stride,
    float scale, float* prev_delta, void* stream);
/* UpdateConvolutionalLayerGpu's SGD branch (src/convolutional_kernels.cu:895-915) for one
 * tensor, fused: wu += -decay*batch*w (if use_decay); w += (lr/batch)*wu; wu *= momentum. */
DK_API int dk_sgd_update(float* weights, float* weight_updates, size_t n, int batch,
    float learning_rate, float momentum, float decay, int use_decay, void* stream);
/* adam_update_gpu (src/blas_kernels.cu:120-134) fused: moments m, v updated in place, d zeroed; t = iteration */
DK_API int dk_adam_update(float* w, float* d, float* m, float* v, float B1, float B2, float eps, float decay,
    float rate, size_t n, int batch, int t, void* stream);

/* ---- sibling-cfg layer kinds ----------------------------------- */
/* ForwardAvgpoolLayerGpu / BackwardAvgpoolLayerGpu, src/avgpool_layer_kernels.cu:9-62
 * (CPU: src/avgpool_layer.cpp:40-72): out[b][k] = mean over the h*w plane; backward ADDS
 * delta[b][k] / (h*w) to every input pixel. */
DK_API int dk_avgpool_forward(const float* in, float* out, int batch, int c, int h, int w, void* stream);
DK_API int dk_avgpool_backward(const float* delta, float* prev_delta, int batch, int c, int h, int w, void* stream);
/* scale_channels_gpu / backward_scale_channels_gpu (src/scale_channels_layer.c:70-160):
 * out = act(scale * from) with one scale per (image, channel) (scale_wh = 0, in: [batch][c])
 * or per (image, pixel) (scale_wh = 1, in: [batch][h*w]); backward: delta already holds
 * d/d(out) * act'; from_delta += scale * delta, in_delta += sum(delta * from).  Either
 * gradient target may be NULL. */
DK_API int dk_scale_channels_forward(const float* in, const float* from, float* out, int batch, int out_c,
    int out_h, int out_w, int scale_wh, int activation, void* stream);
DK_API int dk_scale_channels_backward(const float* delta, const float* in, const float* from, float* from_delta,
    float* in_delta, int batch, int out_c, int out_h, int out_w, int scale_wh, void* stream);
/* [dropout] in train mode, in place (dropout_layer_kernels.cu): rnd[i] uniform in [0,1) from a counter-based hash of
 * (seed, i); x[i] = rnd[i] < probability ? 0 : x[i] * scale; the backward applies the same mask to the delta */
DK_API int dk_dropout_forward(float* x, float* rnd, size_t n, float probability, float scale, unsigned long long seed,
    void* stream);
DK_API int dk_dropout_backward(float* delta, const float* rnd, size_t n, float probability, float scale, void* stream);
/* x[0..n) = uniform draws in [0, 1): the generator behind `cuda_random` (src/dark_cuda.c:464-477) */
DK_API int dk_random_uniform(float* x, size_t n, unsigned long long seed, void* stream);

/* Profiling hooks used by bench.py (measurement only): when enabled every
 * dk_conv_forward is bracketed by HIP events on its stream; dk_profile_read
 * synchronises and returns per tile-configuration totals.
 * (slot = tile configuration * 4 + AVEC + 2*BVEC, i.e. one slot per kernel symbol):
 *   out[slot*3+0] = launches, out[slot*3+1] = algorithmic GFLOP
 *   (2*nweights*oh*ow*batch/1e9, the reference's counter
 *   src/convolutional_layer.cpp:714), out[slot*3+2] = milliseconds.
 * Returns the number of slots; dk_conv_kernel_name(slot) is the kernel's name
 * exactly as rocprofv3 prints it. */
DK_API void dk_profile_enable(int on);
DK_API int dk_profile_read(double* out, int max_slots);
DK_API const char* dk_conv_kernel_name(int slot);

#ifdef __cplusplus
}
#endif
#endif /* DK_KERNELS_H */
