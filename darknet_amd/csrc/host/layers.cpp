// layers.cpp -- host side of the hot-path layer kinds: construction (the
// reference's Fill*Layer) and the *_gpu plugin slots, which enqueue the
// hand-written HIP kernels of dk_kernels.h on the per-device stream.
//
// Reference twins (Ravicmoon/darknet, src/):
//   FillConvLayer convolutional_layer.cpp:401-802, ForwardConvolutionalLayerGpu
//   convolutional_kernels.cu:252-553, Push/PullConvolutionalLayer :817-863;
//   FillMaxpoolLayer maxpool_layer.cpp:19-120, ForwardMaxpoolLayerGpu
//   maxpool_layer_kernels.cu:145-200; FillRouteLayer route_layer.c:9-44,
//   ForwardRouteLayerGpu :124-142; FillShortcutLayer shortcut_layer.c:11-98,
//   ForwardShortcutLayerGpu :190-204; FillUpsampleLayer upsample_layer.c:9-47,
//   ForwardUpsampleLayerGpu :106-119; FillYoloLayer yolo_layer.cpp:15-86,
//   ForwardYoloLayerGpu :836-882, YoloNumDetections :779, GetYoloDetections :794.
//
// There is no CPU compute path in this library: the `forward` (CPU) slots point
// at a function that fails loudly.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"
#include "dk_internal.h"

static int verbose()
{
  static int v = -1;
  if (v < 0)
  {
    const char* e = getenv("DK_VERBOSE");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v;
}

bool dk_gpu_enabled() { return cuda_get_device() >= 0; }

static void no_cpu_path(layer* l, NetworkState)
{
  fprintf(stderr,
      "darknet_amd: layer %d has no CPU compute path; this library is the HIP path only "
      "(no fallback). Run on a device via the *_gpu slots.\n",
      l->index);
  exit(EXIT_FAILURE);
}

static float* dev_array(float* host, size_t n)
{
  return dk_gpu_enabled() ? cuda_make_array(host, n) : nullptr;
}

// ---------------------------------------------------------------- convolution
static int conv_out_h(const layer* l) { return (l->h + 2 * l->pad - l->size) / l->stride_y + 1; }
static int conv_out_w(const layer* l) { return (l->w + 2 * l->pad - l->size) / l->stride_x + 1; }

static float rand_uniform(float lo, float hi)
{
  return lo + (hi - lo) * ((float)rand() / (float)RAND_MAX);
}

void FillConvLayer(layer* l, int batch, int h, int w, int c, int n, int groups, int size,
    int stride_x, int stride_y, int dilation, int padding, ACTIVATION activation,
    int batch_normalize, int index, int train)
{
  l->type = CONVOLUTIONAL;
  l->train = train;
  if (groups < 1)
    groups = 1;
  l->index = index;
  l->h = h; l->w = w; l->c = c;
  l->groups = groups;
  l->n = n;
  l->batch = batch;
  l->steps = 1;
  l->stride = stride_x; l->stride_x = stride_x; l->stride_y = stride_y;
  l->dilation = dilation;
  l->size = size;
  l->pad = padding;
  l->batch_normalize = batch_normalize;
  l->learning_rate_scale = 1;
  l->nweights = (c / groups) * n * size * size;
  l->nbiases = n;
  l->activation = activation;
  l->fuse_residual_from = -1;
  l->conv_cfg = -1;

  l->weights = (float*)xcalloc(l->nweights, sizeof(float));
  l->biases = (float*)xcalloc(n, sizeof(float));
  if (train)
  {
    l->weight_updates = (float*)xcalloc(l->nweights, sizeof(float));
    l->bias_updates = (float*)xcalloc(n, sizeof(float));
  }
  // convolutional_layer.cpp:476-493: He-style init from rand()
  const float scale = sqrt(2. / (size * size * c / groups));
  for (int i = 0; i < l->nweights; ++i) l->weights[i] = scale * rand_uniform(-1, 1);

  l->out_h = conv_out_h(l);
  l->out_w = conv_out_w(l);
  l->out_c = n;
  l->outputs = l->out_h * l->out_w * l->out_c;
  l->inputs = l->w * l->h * l->c;
  const size_t total = (size_t)batch * l->outputs;

  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardConvolutionalLayerGpu;
  l->backward_gpu = BackwardConvolutionalLayerGpu;
  l->update_gpu = UpdateConvolutionalLayerGpu;

  if (batch_normalize)
  {
    l->scales = (float*)xcalloc(n, sizeof(float));
    for (int i = 0; i < n; ++i) l->scales[i] = 1;
    if (train)
    {
      l->scale_updates = (float*)xcalloc(n, sizeof(float));
      l->mean = (float*)xcalloc(n, sizeof(float));
      l->variance = (float*)xcalloc(n, sizeof(float));
      l->mean_delta = (float*)xcalloc(n, sizeof(float));
      l->variance_delta = (float*)xcalloc(n, sizeof(float));
    }
    l->rolling_mean = (float*)xcalloc(n, sizeof(float));
    l->rolling_variance = (float*)xcalloc(n, sizeof(float));
  }

  if (dk_gpu_enabled())
  {
    l->weights_gpu = cuda_make_array(l->weights, l->nweights);
    l->biases_gpu = cuda_make_array(l->biases, n);
    l->output_gpu = cuda_make_array(nullptr, total);
    if (train)
    {
      l->weight_updates_gpu = cuda_make_array(l->weight_updates, l->nweights);
      l->bias_updates_gpu = cuda_make_array(l->bias_updates, n);
      l->delta_gpu = cuda_make_array(nullptr, total);
      CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total * sizeof(float), get_cuda_stream()));
      if (activation == MISH || activation == SWISH)
        l->activation_input_gpu = cuda_make_array(nullptr, total);
    }
    if (batch_normalize)
    {
      l->scales_gpu = cuda_make_array(l->scales, n);
      l->rolling_mean_gpu = cuda_make_array(l->rolling_mean, n);
      l->rolling_variance_gpu = cuda_make_array(l->rolling_variance, n);
      if (train)
      {
        l->scale_updates_gpu = cuda_make_array(l->scale_updates, n);
        l->mean_gpu = cuda_make_array(l->mean, n);
        l->variance_gpu = cuda_make_array(l->variance, n);
        l->mean_delta_gpu = cuda_make_array(l->mean_delta, n);
        l->variance_delta_gpu = cuda_make_array(l->variance_delta, n);
        l->x_gpu = cuda_make_array(nullptr, total);
        l->x_norm_gpu = cuda_make_array(nullptr, total);
      }
    }
  }
  // GetWorkspaceSize32, convolutional_layer.cpp:131-132 (im2col buffer of one
  // image).  The forward kernel here is an implicit GEMM and needs none; the
  // value is kept because it is part of the plugin contract (state.workspace).
  l->workspace_size =
      (size_t)l->out_h * l->out_w * l->size * l->size * (l->c / l->groups) * sizeof(float);
  l->bflops = (2.0 * l->nweights * l->out_h * l->out_w) / 1000000000.;
  if (verbose())
    fprintf(stderr, "conv  %5d %2d x%2d/%2d   %4d x%4d x%4d -> %4d x%4d x%4d %5.3f BF\n", n, size,
        size, stride_x, w, h, c, l->out_w, l->out_h, l->out_c, l->bflops);
}

static DkConvDesc conv_desc(const layer* l)
{
  DkConvDesc d;
  d.batch = l->batch; d.c = l->c; d.h = l->h; d.w = l->w; d.n = l->n; d.groups = l->groups;
  d.size = l->size; d.stride_x = l->stride_x; d.stride_y = l->stride_y;
  d.dilation = l->dilation; d.pad = l->pad; d.activation = (int)l->activation;
  return d;
}

void DkConvPrepare(layer* l)
{
  DkConvDesc d = conv_desc(l);
  dk_conv_prepare(&d);
}

// Inference (BN folded or no BN): one fused launch.  Train mode with BN is
// handled in train.cpp (raw GEMM, then batch statistics, then activation).
void ForwardConvTrainGpu(layer* l, NetworkState state);  // train.cpp
void DkDropWinograd(layer* l);                            // network.cpp
void DkYoloTrainDelta(layer* l, NetworkState state);      // train.cpp
void DkFreeLossTask(layer* l);                            // train.cpp

void ForwardConvolutionalLayerGpu(layer* l, NetworkState state)
{
  if (l->batch_normalize)
  {
    ForwardConvTrainGpu(l, state);
    return;
  }
  if (l->dual_slave && !state.train)
    return;  // written by the dual launch of layer l->index - 2
  DkConvDesc d = conv_desc(l);
  float* out = l->output_gpu;
  const float* residual = nullptr;
  if (l->dual_with > 0 && !state.train)
  {
    layer* l2 = &state.net->layers[l->dual_with];
    DkConvDual dual;
    dual.y2 = l2->out_view ? l2->out_view : l2->output_gpu;
    dual.m_split = l->n;
    dual.out_ctot2 = l2->out_view ? l2->out_view_ctot : 0;
    d.n = l->n + l2->n;
    if (dk_conv_forward_cfg(&d, state.input, l->dual_weights_gpu, l->dual_biases_gpu,
            l->out_view ? l->out_view : l->output_gpu, nullptr, nullptr, get_cuda_stream(), l->conv_cfg,
            l->out_view ? l->out_view_ctot : 0, &dual))
      error("ForwardConvolutionalLayerGpu (dual output) failed");
    return;
  }
  if (l->fuse_residual_from >= 0)
  {
    // shortcut folded into this conv's epilogue: write the sum straight into the
    // shortcut layer's buffer (the next layer).
    layer* sc = &state.net->layers[l->index + 1];
    out = sc->output_gpu;
    residual = DkLayerOut(&state.net->layers[l->fuse_residual_from]);
  }
  float* act_in = (state.train && l->activation_input_gpu) ? l->activation_input_gpu : nullptr;
  int out_ctot = 0;
  if (l->out_view && !state.train)
  {
    out = l->out_view;  // channel slice of the consuming route's buffer
    out_ctot = l->out_view_ctot;
  }
  if (state.net->cudnn_half && !state.train && l->weights_half_gpu && !act_in)
  {
    if (dk_conv_forward_half_direct(&d, state.input, l->weights_half_gpu, l->biases_gpu, out, residual,
            get_cuda_stream(), out_ctot))
      error("ForwardConvolutionalLayerGpu (fp16 operands, direct) failed");
    return;
  }
  if (state.net->cudnn_half && !state.train && dk_conv_half_eligible(&d, l->index))
  {
    if (dk_conv_forward_half_strided(&d, state.input, l->weights_gpu, l->biases_gpu, out, residual,
            act_in, get_cuda_stream(), out_ctot))
      error("ForwardConvolutionalLayerGpu (fp16 operands) failed");
    return;
  }
  // (the Winograd copy of the filters follows Push/Load, not the optimizer: train passes keep the direct kernels)
  const int cfg = (state.train && dk_conv_config_is_wino(l->conv_cfg)) ? -1 : l->conv_cfg;
  if (dk_conv_forward_cfg(&d, state.input, l->weights_gpu, l->biases_gpu, out, residual, act_in,
          get_cuda_stream(), cfg, out_ctot))
    error("ForwardConvolutionalLayerGpu failed");
}

void add_bias_gpu(float* output, float* biases, int batch, int n, int size)
{
  dk_add_bias(output, biases, batch, n, size, nullptr);
}

void PushConvolutionalLayer(layer* l)
{
  if (!dk_gpu_enabled())
    return;
  cuda_push_array(l->weights_gpu, l->weights, l->nweights);
  cuda_push_array(l->biases_gpu, l->biases, l->n);
  if (l->train)
  {
    if (l->weight_updates_gpu)
      cuda_push_array(l->weight_updates_gpu, l->weight_updates, l->nweights);
    if (l->bias_updates_gpu)
      cuda_push_array(l->bias_updates_gpu, l->bias_updates, l->n);
  }
  if (l->scales_gpu && l->scales)
  {
    cuda_push_array(l->scales_gpu, l->scales, l->n);
    cuda_push_array(l->rolling_mean_gpu, l->rolling_mean, l->n);
    cuda_push_array(l->rolling_variance_gpu, l->rolling_variance, l->n);
  }
  // copies the inference plan derived from the weights follow them (same allocations, so a
  // captured graph stays valid): the concatenated filters of a dual launch, the packed fp16 set
  hipStream_t st = get_cuda_stream();
  if (l->dual_peer)
  {
    layer* a = l->dual_slave ? l->dual_peer : l;
    layer* b = a->dual_peer;
    if (a->dual_weights_gpu && b)
    {
      CHECK_HIP(hipMemcpyAsync(a->dual_weights_gpu, a->weights_gpu, (size_t)a->nweights * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_weights_gpu + a->nweights, b->weights_gpu, (size_t)b->nweights * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_biases_gpu, a->biases_gpu, (size_t)a->n * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_biases_gpu + a->n, b->biases_gpu, (size_t)b->n * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  }
  if (l->weights_half_gpu)
  {
    DkConvDesc d = conv_desc(l);
    if (dk_conv_half_pack_weights(&d, l->weights_gpu, l->weights_half_gpu, st))
      error("PushConvolutionalLayer: re-packing the fp16 weights failed");
  }
  if (l->weights_wino_gpu)
  {
    DkConvDesc d = conv_desc(l);
    if (dk_conv_wino_transform_weights(&d, l->weights_gpu, l->weights_wino_gpu, st))
      error("PushConvolutionalLayer: re-transforming the Winograd filters failed");
  }
  CHECK_HIP(hipStreamSynchronize(st));
}

void PullConvolutionalLayer(layer* l)
{
  if (!dk_gpu_enabled())
    return;
  cuda_pull_array(l->weights_gpu, l->weights, l->nweights);
  cuda_pull_array(l->biases_gpu, l->biases, l->n);
  if (l->train)
  {
    if (l->weight_updates_gpu)
      cuda_pull_array(l->weight_updates_gpu, l->weight_updates, l->nweights);
    if (l->bias_updates_gpu)
      cuda_pull_array(l->bias_updates_gpu, l->bias_updates, l->n);
  }
  if (l->batch_normalize && l->scales_gpu)
  {
    cuda_pull_array(l->scales_gpu, l->scales, l->n);
    cuda_pull_array(l->rolling_mean_gpu, l->rolling_mean, l->n);
    cuda_pull_array(l->rolling_variance_gpu, l->rolling_variance, l->n);
  }
}

// -------------------------------------------------------------------- maxpool
void FillMaxpoolLayer(layer* l, int batch, int h, int w, int c, int size, int stride_x,
    int stride_y, int padding, int train)
{
  l->type = MAXPOOL;
  l->train = train;
  l->batch = batch;
  l->h = h; l->w = w; l->c = c;
  l->pad = padding;
  l->out_w = (w + padding - size) / stride_x + 1;  // maxpool_layer.cpp:62-63
  l->out_h = (h + padding - size) / stride_y + 1;
  l->out_c = c;
  l->outputs = l->out_h * l->out_w * l->out_c;
  l->inputs = h * w * c;
  l->size = size;
  l->stride = stride_x; l->stride_x = stride_x; l->stride_y = stride_y;
  const size_t output_size = (size_t)l->outputs * batch;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardMaxpoolLayerGpu;
  l->backward_gpu = BackwardMaxpoolLayerGpu;
  if (dk_gpu_enabled())
  {
    if (train)
    {
      l->indexes_gpu = cuda_make_int_array(output_size);
      l->delta_gpu = cuda_make_array(nullptr, output_size);
      CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, output_size * sizeof(float), get_cuda_stream()));
    }
    l->output_gpu = cuda_make_array(nullptr, output_size);
  }
  l->bflops = (l->size * l->size * l->c * l->out_h * l->out_w) / 1000000000.;
  if (verbose())
    fprintf(stderr, "max   %2dx%2d/%2d   %4d x%4d x%4d -> %4d x%4d x%4d %5.3f BF\n", size, size,
        stride_x, w, h, c, l->out_w, l->out_h, l->out_c, l->bflops);
}

void ForwardMaxpoolLayerGpu(layer* l, NetworkState state)
{
  float* out = l->output_gpu;
  size_t bstride = 0;
  if (l->out_view && !state.train)
  {
    out = l->out_view;
    bstride = (size_t)l->out_view_ctot * l->out_h * l->out_w;
  }
  if (dk_maxpool_forward_strided(state.input, out, state.train ? l->indexes_gpu : nullptr, l->batch,
          l->c, l->h, l->w, l->size, l->stride_x, l->stride_y, l->pad, bstride, get_cuda_stream()))
    error("ForwardMaxpoolLayerGpu failed");
}

// ---------------------------------------------------------------------- route
void FillRouteLayer(layer* l, int batch, int n, int* input_layers, int* input_sizes, int groups,
    int group_id)
{
  l->type = ROUTE;
  l->batch = batch;
  l->n = n;
  l->input_layers = input_layers;
  l->input_sizes = input_sizes;
  l->groups = groups;
  l->group_id = group_id;
  int outputs = 0;
  for (int i = 0; i < n; ++i) outputs += input_sizes[i];
  outputs = outputs / groups;
  l->outputs = outputs;
  l->inputs = outputs;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardRouteLayerGpu;
  l->backward_gpu = BackwardRouteLayerGpu;
  if (dk_gpu_enabled())
  {
    l->output_gpu = cuda_make_array(nullptr, (size_t)outputs * batch);
    l->delta_gpu = cuda_make_array(nullptr, (size_t)outputs * batch);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, (size_t)outputs * batch * sizeof(float), get_cuda_stream()));
  }
}

void ForwardRouteLayerGpu(layer* l, NetworkState state)
{
  if (l->out_alias && !state.train)
    return;  // single input: readers use the source's buffer (DkLayerOut)
  int offset = 0;
  for (int i = 0; i < l->n; ++i)
  {
    const int index = l->input_layers[i];
    const float* input = DkLayerOut(&state.net->layers[index]);
    const int input_size = l->input_sizes[i];
    const int part = input_size / l->groups;
    if (l->input_inplace && l->input_inplace[i] && !state.train)
    {
      offset += part;  // the producer wrote this slice itself
      continue;
    }
    if (dk_route_copy(input, input_size, l->groups, l->group_id, l->batch, l->output_gpu,
            l->outputs, offset, get_cuda_stream()))
      error("ForwardRouteLayerGpu failed");
    offset += part;
  }
}

// ------------------------------------------------------------------- shortcut
void FillShortcutLayer(layer* l, int batch, int index, int w, int h, int c, int from_outputs,
    ACTIVATION activation, int train)
{
  l->type = SHORTCUT;
  l->train = train;
  l->activation = activation;
  l->batch = batch;
  l->n = 1;
  l->index = index;  // the layer added to the input (shortcut_layer.c:34)
  l->w = l->out_w = w;
  l->h = l->out_h = h;
  l->c = l->out_c = c;
  l->outputs = w * h * c;
  l->inputs = l->outputs;
  l->input_sizes = (int*)xcalloc(1, sizeof(int));
  l->input_sizes[0] = from_outputs;
  l->input_layers = (int*)xcalloc(1, sizeof(int));
  l->input_layers[0] = index;
  l->nweights = 0;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardShortcutLayerGpu;
  l->backward_gpu = BackwardShortcutLayerGpu;
  if (dk_gpu_enabled())
  {
    l->output_gpu = cuda_make_array(nullptr, (size_t)l->outputs * batch);
    if (train)
    {
      l->delta_gpu = cuda_make_array(nullptr, (size_t)l->outputs * batch);
      CHECK_HIP(hipMemsetAsync(
          l->delta_gpu, 0, (size_t)l->outputs * batch * sizeof(float), get_cuda_stream()));
    }
  }
  l->bflops = l->out_w * l->out_h * l->out_c * l->n / 1000000000.;
}

void ForwardShortcutLayerGpu(layer* l, NetworkState state)
{
  if (l->fused_into_prev)
    return;  // the previous conv's epilogue already wrote in + from into output_gpu
  layer* from = &state.net->layers[l->index];
  if (from->out_w != l->w || from->out_h != l->h || from->out_c != l->c)
  {
    printf("something went wrong\n");  // shortcut_layer.c:162
    return;
  }
  if (l->activation != LINEAR && l->activation != LEAKY && l->activation != LOGISTIC &&
      l->activation != RELU && l->activation != MISH)
    error("ForwardShortcutLayerGpu: unsupported activation");
  if (dk_shortcut_forward(state.input, DkLayerOut(from), l->output_gpu,
          (size_t)l->outputs * l->batch, (int)l->activation, get_cuda_stream()))
    error("ForwardShortcutLayerGpu failed");
}

// ------------------------------------------------------------------- upsample
void FillUpsampleLayer(layer* l, int batch, int w, int h, int c, int stride)
{
  l->type = UPSAMPLE;
  l->batch = batch;
  l->w = w; l->h = h; l->c = c;
  l->out_w = w * stride;
  l->out_h = h * stride;
  l->out_c = c;
  if (stride < 0)
    error("[upsample] negative stride (downsample) is outside the supported hot path");
  l->stride = stride;
  l->outputs = l->out_w * l->out_h * l->out_c;
  l->inputs = l->w * l->h * l->c;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardUpsampleLayerGpu;
  l->backward_gpu = BackwardUpsampleLayerGpu;
  if (dk_gpu_enabled())
  {
    l->output_gpu = cuda_make_array(nullptr, (size_t)l->outputs * batch);
    l->delta_gpu = cuda_make_array(nullptr, (size_t)l->outputs * batch);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, (size_t)l->outputs * batch * sizeof(float), get_cuda_stream()));
  }
}

void ForwardUpsampleLayerGpu(layer* l, NetworkState state)
{
  float* out = l->output_gpu;
  size_t bstride = 0;
  if (l->out_view && !state.train)
  {
    out = l->out_view;
    bstride = (size_t)l->out_view_ctot * l->out_h * l->out_w;
  }
  if (dk_upsample_forward_strided(state.input, l->w, l->h, l->c, l->batch, l->stride, l->scale, out,
          bstride, get_cuda_stream()))
    error("ForwardUpsampleLayerGpu failed");
}

// ----------------------------------------------------------------------- yolo
void FillYoloLayer(layer* l, int batch, int w, int h, int n, int total, int* mask, int classes,
    int max_boxes)
{
  l->type = YOLO;
  l->n = n;
  l->total = total;
  l->batch = batch;
  l->h = h; l->w = w;
  l->c = n * (classes + 4 + 1);
  l->out_w = l->w; l->out_h = l->h; l->out_c = l->c;
  l->classes = classes;
  l->cost = (float*)xcalloc(1, sizeof(float));
  l->biases = (float*)xcalloc(total * 2, sizeof(float));
  l->nbiases = total * 2;
  if (mask)
    l->mask = mask;
  else
  {
    l->mask = (int*)xcalloc(n, sizeof(int));
    for (int i = 0; i < n; ++i) l->mask[i] = i;
  }
  l->bias_updates = (float*)xcalloc(n * 2, sizeof(float));
  l->outputs = h * w * n * (classes + 4 + 1);
  l->inputs = l->outputs;
  l->max_boxes = max_boxes;
  l->truths = l->max_boxes * (4 + 1);
  for (int i = 0; i < total * 2; ++i) l->biases[i] = .5;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardYoloLayerGpu;
  l->backward_gpu = BackwardYoloLayerGpu;
  const size_t total_out = (size_t)batch * l->outputs;
  if (dk_gpu_enabled())
  {
    l->output_gpu = cuda_make_array(nullptr, total_out);
    l->delta_gpu = cuda_make_array(nullptr, total_out);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total_out * sizeof(float), get_cuda_stream()));
    // pinned host mirror, as the reference does (yolo_layer.cpp:63-72)
    l->output = cuda_make_array_pinned(nullptr, total_out);
    l->output_pinned = 1;
    memset(l->output, 0, total_out * sizeof(float));
  }
  else
    l->output = (float*)xcalloc(total_out, sizeof(float));
}

void ForwardYoloLayerGpu(layer* l, NetworkState state)
{
  if (dk_yolo_forward(state.input, l->output_gpu, l->batch, l->w, l->h, l->n, l->classes,
          l->scale_x_y, get_cuda_stream()))
    error("ForwardYoloLayerGpu failed");
  // D2H of the decoded head is issued by the graph engine (network.cpp) so that
  // it can run on the copy stream / outside a captured graph.
  if (state.train && !l->onlyforward)
    DkYoloTrainDelta(l, state);
}

// EntryIndex, yolo_layer.cpp:380-386
static int entry_index(const layer* l, int batch, int location, int entry)
{
  const int n = location / (l->w * l->h);
  const int loc = location % (l->w * l->h);
  return batch * l->outputs + n * l->w * l->h * (4 + l->classes + 1) + entry * l->w * l->h + loc;
}

int DkYoloNumDetectionsBatch(layer const* l, int b, float thresh)
{
  int count = 0;
  for (int n = 0; n < l->n; ++n)
    for (int i = 0; i < l->w * l->h; ++i)
      if (l->output[entry_index(l, b, n * l->w * l->h + i, 4)] > thresh)
        ++count;
  return count;
}

int YoloNumDetections(layer const* l, float thresh) { return DkYoloNumDetectionsBatch(l, 0, thresh); }

int DkGetYoloDetectionsBatch(
    layer const* l, int b, int net_w, int net_h, float thresh, Detection* dets, int* ids)
{
  float const* pred = l->output;
  const int stride = l->w * l->h;
  int count = 0;
  for (int n = 0; n < l->n; ++n)
    for (int i = 0; i < l->w * l->h; ++i)
    {
      const int loc = n * l->w * l->h + i;
      const float objectness = pred[entry_index(l, b, loc, 4)];
      if (objectness <= thresh)
        continue;
      const int box_idx = entry_index(l, b, loc, 0);
      const int col = i % l->w, row = i / l->w;
      const int a = l->mask[n];
      // GetYoloBox, yolo_layer.cpp:139-148 (exp on a float resolves to expf there)
      Box bx;
      bx.x = (col + pred[box_idx + 0 * stride]) / l->w;
      bx.y = (row + pred[box_idx + 1 * stride]) / l->h;
      bx.w = expf(pred[box_idx + 2 * stride]) * l->biases[2 * a] / net_w;
      bx.h = expf(pred[box_idx + 3 * stride]) * l->biases[2 * a + 1] / net_h;
      dets[count].bbox = bx;
      dets[count].objectness = objectness;
      dets[count].classes = l->classes;
      for (int j = 0; j < l->classes; ++j)
      {
        const float prob = objectness * pred[entry_index(l, b, loc, 4 + 1 + j)];
        dets[count].prob[j] = (prob > thresh) ? prob : 0;
      }
      if (ids)
      {
        ids[4 * count + 0] = -1;  // layer index filled by the caller
        ids[4 * count + 1] = n;
        ids[4 * count + 2] = row;
        ids[4 * count + 3] = col;
      }
      ++count;
    }
  return count;
}

int GetYoloDetections(layer const* l, int net_w, int net_h, float thresh, Detection* dets)
{
  return DkGetYoloDetectionsBatch(l, 0, net_w, net_h, thresh, dets, nullptr);
}

// ------------------------------------------------------------------ free_layer
void free_layer(layer* l, bool)
{
  // layer.cpp:14-255: a layer owns its tensors
  if (l->output_pinned && l->output)
    cuda_free_host(l->output);
  else
    free(l->output);
  l->output = nullptr;
  free(l->mask); free(l->cost); free(l->indexes);
  free(l->input_layers); free(l->input_sizes); free(l->input_inplace);
  free(l->biases); free(l->bias_updates); free(l->scales); free(l->scale_updates);
  free(l->weights); free(l->weight_updates);
  DkFreeLossTask(l);
  if (l->delta_pinned && l->delta)
    cuda_free_host(l->delta);
  else
    free(l->delta);
  free(l->activation_input);
  free(l->mean); free(l->variance); free(l->mean_delta); free(l->variance_delta);
  free(l->rolling_mean); free(l->rolling_variance);
  free(l->x); free(l->x_norm);
  free(l->classes_multipliers); free(l->map);
  if (dk_gpu_enabled())
  {
    cuda_free((float*)l->indexes_gpu);
    cuda_free(l->mean_gpu); cuda_free(l->variance_gpu);
    cuda_free(l->rolling_mean_gpu); cuda_free(l->rolling_variance_gpu);
    cuda_free(l->variance_delta_gpu); cuda_free(l->mean_delta_gpu);
    cuda_free(l->x_gpu); cuda_free(l->x_norm_gpu);
    DkDropWinograd(l);
    cuda_free(l->weights_gpu); cuda_free(l->weight_updates_gpu);
    cuda_free((float*)l->weights_half_gpu);
    cuda_free(l->dual_weights_gpu); cuda_free(l->dual_biases_gpu);
    cuda_free(l->biases_gpu); cuda_free(l->bias_updates_gpu);
    cuda_free(l->scales_gpu); cuda_free(l->scale_updates_gpu);
    cuda_free(l->rand_gpu);
    cuda_free(l->m_gpu); cuda_free(l->v_gpu); cuda_free(l->bias_m_gpu); cuda_free(l->bias_v_gpu);
    cuda_free(l->scale_m_gpu); cuda_free(l->scale_v_gpu);
    cuda_free(l->activation_input_gpu);
    if (!l->buffers_aliased)
    {
      cuda_free(l->output_gpu);
      if (!l->delta_in_arena)
        cuda_free(l->delta_gpu);
    }
  }
  memset(l, 0, sizeof(*l));
}
