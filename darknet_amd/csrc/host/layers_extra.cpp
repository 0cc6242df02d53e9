// layers_extra.cpp -- layer kinds used by sibling cfgs of the YOLOv4 family and
// the remaining plugin symbols of the boundary: standalone [batchnorm], global
// [avgpool], [scale_channels] (squeeze-and-excitation), [dropout] at inference, backward_bias_gpu,
// the upstream-darknet aliases and init_cpu.
//
// Reference twins (Ravicmoon/darknet src/): FillBatchnormLayer batchnorm_layer.cpp:9-88,
// Forward/Backward/UpdateBatchnormLayerGpu :268-409 (numerics of the CPU twins :206-266: variance
// over N-1, eps 1e-6 forward / 1e-5 backward, rolling .9/.1); FillAvgpoolLayer avgpool_layer.cpp:6-40,
// ForwardAvgpoolLayerGpu avgpool_layer_kernels.cu:46-62; FillScaleChannelsLayer
// scale_channels_layer.c:9-48, Forward/BackwardScaleChannelsLayerGpu :129-160; ParseDropout
// parser.cpp:672-712 + the aliasing of its buffers :1232-1242; backward_bias_gpu
// convolutional_kernels.cu:37-67 (declared convolutional_layer.h:14); init_cpu utils.cpp.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"
#include "dk_internal.h"

extern "C" {
int dk_bn_forward_train(const float*, float*, float*, float*, float*, float*, float*, float*, float*,
    const float*, const float*, int, int, int, int, int, void*);
}

static void no_cpu_path(layer* l, NetworkState)
{
  fprintf(stderr, "darknet_amd: layer %d has no CPU compute path (HIP path only, no fallback)\n", l->index);
  exit(EXIT_FAILURE);
}

// ---------------------------------------------------------------- [batchnorm]
void FillBatchnormLayer(layer* l, int batch, int w, int h, int c, int train)
{
  l->type = BATCHNORM;
  l->batch = batch;
  l->train = train;
  l->h = l->out_h = h;
  l->w = l->out_w = w;
  l->c = l->out_c = c;
  l->n = c;
  l->inputs = w * h * c;
  l->outputs = l->inputs;
  l->activation = LINEAR;
  l->biases = (float*)xcalloc(c, sizeof(float));
  l->bias_updates = (float*)xcalloc(c, sizeof(float));
  l->scales = (float*)xcalloc(c, sizeof(float));
  l->scale_updates = (float*)xcalloc(c, sizeof(float));
  for (int i = 0; i < c; ++i) l->scales[i] = 1;
  l->mean = (float*)xcalloc(c, sizeof(float));
  l->variance = (float*)xcalloc(c, sizeof(float));
  l->rolling_mean = (float*)xcalloc(c, sizeof(float));
  l->rolling_variance = (float*)xcalloc(c, sizeof(float));
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardBatchnormLayerGpu;
  l->backward_gpu = BackwardBatchnormLayerGpu;
  l->update_gpu = UpdateBatchnormLayerGpu;
  if (dk_gpu_enabled())
  {
    const size_t total = (size_t)batch * l->outputs;
    l->output_gpu = cuda_make_array(nullptr, total);
    l->biases_gpu = cuda_make_array(l->biases, c);
    l->scales_gpu = cuda_make_array(l->scales, c);
    l->mean_gpu = cuda_make_array(l->mean, c);
    l->variance_gpu = cuda_make_array(l->variance, c);
    l->rolling_mean_gpu = cuda_make_array(l->mean, c);
    l->rolling_variance_gpu = cuda_make_array(l->variance, c);
    if (train)
    {
      l->delta_gpu = cuda_make_array(nullptr, total);
      CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total * sizeof(float), get_cuda_stream()));
      l->bias_updates_gpu = cuda_make_array(l->bias_updates, c);
      l->scale_updates_gpu = cuda_make_array(l->scale_updates, c);
      l->mean_delta_gpu = cuda_make_array(l->mean, c);
      l->variance_delta_gpu = cuda_make_array(l->variance, c);
      l->x_gpu = cuda_make_array(nullptr, total);
      l->x_norm_gpu = cuda_make_array(nullptr, total);
    }
  }
}

void ForwardBatchnormLayerGpu(layer* l, NetworkState state)
{
  // one pass: statistics (train) -> normalise, scale, bias; x / x_norm saved for the backward pass
  const int train = state.train && l->x_gpu && l->x_norm_gpu;
  if (dk_bn_forward_train(state.input, l->x_gpu, l->x_norm_gpu, nullptr, l->output_gpu, l->mean_gpu,
          l->variance_gpu, l->rolling_mean_gpu, l->rolling_variance_gpu, l->scales_gpu, l->biases_gpu,
          l->batch, l->out_c, l->out_h * l->out_w, (int)LINEAR, train, get_cuda_stream()))
    error("ForwardBatchnormLayerGpu failed");
}

void BackwardBatchnormLayerGpu(layer* l, NetworkState state)
{
  hipStream_t st = get_cuda_stream();
  if (!state.train)
  {
    // batchnorm_layer.cpp:326-335: statistics of an inference-mode backward = the rolling ones
    dk_copy(l->out_c, l->rolling_mean_gpu, l->mean_gpu, st);
    dk_copy(l->out_c, l->rolling_variance_gpu, l->variance_gpu, st);
  }
  if (dk_bn_backward(l->delta_gpu, l->x_gpu, l->x_norm_gpu, l->mean_gpu, l->variance_gpu, l->scales_gpu,
          l->mean_delta_gpu, l->variance_delta_gpu, l->scale_updates_gpu, l->bias_updates_gpu, l->batch,
          l->out_c, l->out_h * l->out_w, st))
    error("BackwardBatchnormLayerGpu failed");
  if (l->type == BATCHNORM && state.delta)
    dk_copy((size_t)l->outputs * l->batch, l->delta_gpu, state.delta, st);
}

void UpdateBatchnormLayerGpu(layer* l, int batch, float learning_rate_init, float momentum, float decay,
    float loss_scale)
{
  (void)decay;
  hipStream_t st = get_cuda_stream();
  const float lr = learning_rate_init * l->learning_rate_scale / loss_scale;
  dk_sgd_update(l->biases_gpu, l->bias_updates_gpu, l->c, batch, lr, momentum, 0.f, 0, st);
  dk_sgd_update(l->scales_gpu, l->scale_updates_gpu, l->c, batch, lr, momentum, 0.f, 0, st);
}

void PushBatchnormLayer(layer* l)
{
  if (!dk_gpu_enabled())
    return;
  cuda_push_array(l->biases_gpu, l->biases, l->out_c);
  cuda_push_array(l->scales_gpu, l->scales, l->out_c);
  cuda_push_array(l->rolling_mean_gpu, l->rolling_mean, l->out_c);
  cuda_push_array(l->rolling_variance_gpu, l->rolling_variance, l->out_c);
  CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
}

void PullBatchnormLayer(layer* l)
{
  if (!dk_gpu_enabled())
    return;
  cuda_pull_array(l->biases_gpu, l->biases, l->out_c);
  cuda_pull_array(l->scales_gpu, l->scales, l->out_c);
  cuda_pull_array(l->rolling_mean_gpu, l->rolling_mean, l->out_c);
  cuda_pull_array(l->rolling_variance_gpu, l->rolling_variance, l->out_c);
}

// ------------------------------------------------------------------ [avgpool]
void FillAvgpoolLayer(layer* l, int batch, int w, int h, int c)
{
  l->type = AVGPOOL;
  l->batch = batch;
  l->h = h; l->w = w; l->c = c;
  l->out_w = 1; l->out_h = 1; l->out_c = c;
  l->outputs = l->out_c;
  l->inputs = h * w * c;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardAvgpoolLayerGpu;
  l->backward_gpu = BackwardAvgpoolLayerGpu;
  if (dk_gpu_enabled())
  {
    const size_t total = (size_t)l->outputs * batch;
    l->output_gpu = cuda_make_array(nullptr, total);
    l->delta_gpu = cuda_make_array(nullptr, total);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total * sizeof(float), get_cuda_stream()));
  }
}

void ForwardAvgpoolLayerGpu(layer* l, NetworkState state)
{
  if (dk_avgpool_forward(state.input, l->output_gpu, l->batch, l->c, l->h, l->w, get_cuda_stream()))
    error("ForwardAvgpoolLayerGpu failed");
}

void BackwardAvgpoolLayerGpu(layer* l, NetworkState state)
{
  if (!state.delta)
    return;
  if (dk_avgpool_backward(l->delta_gpu, state.delta, l->batch, l->c, l->h, l->w, get_cuda_stream()))
    error("BackwardAvgpoolLayerGpu failed");
}

// ----------------------------------------------------------- [scale_channels]
void FillScaleChannelsLayer(layer* l, int batch, int index, int w, int h, int c, int w2, int h2, int c2,
    int scale_wh)
{
  l->type = SCALE_CHANNELS;
  l->batch = batch;
  l->scale_wh = scale_wh;
  l->w = w; l->h = h; l->c = c;
  if (!scale_wh && !(w == 1 && h == 1))
    error("[scale_channels]: the scale tensor must be 1 x 1 x c (put an [avgpool] before it)");
  if (scale_wh && c != 1)
    error("[scale_channels] scale_wh=1: the scale tensor must have one channel");
  l->out_w = w2; l->out_h = h2; l->out_c = c2;
  if (!scale_wh && l->out_c != l->c)
    error("[scale_channels]: channel counts differ");
  if (scale_wh && !(l->out_w == l->w && l->out_h == l->h))
    error("[scale_channels] scale_wh=1: spatial sizes differ");
  l->outputs = l->out_w * l->out_h * l->out_c;
  l->inputs = l->outputs;
  l->index = index;
  l->activation = LINEAR;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardScaleChannelsLayerGpu;
  l->backward_gpu = BackwardScaleChannelsLayerGpu;
  if (dk_gpu_enabled())
  {
    const size_t total = (size_t)l->outputs * batch;
    l->output_gpu = cuda_make_array(nullptr, total);
    l->delta_gpu = cuda_make_array(nullptr, total);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total * sizeof(float), get_cuda_stream()));
  }
}

void ForwardScaleChannelsLayerGpu(layer* l, NetworkState state)
{
  const layer* from = &state.net->layers[l->index];
  if (dk_scale_channels_forward(state.input, DkLayerOut(from), l->output_gpu, l->batch, l->out_c, l->out_h,
          l->out_w, l->scale_wh, (int)l->activation, get_cuda_stream()))
    error("ForwardScaleChannelsLayerGpu failed");
}

void BackwardScaleChannelsLayerGpu(layer* l, NetworkState state)
{
  hipStream_t st = get_cuda_stream();
  dk_gradient_array(l->output_gpu, nullptr, l->delta_gpu, (size_t)l->outputs * l->batch, (int)l->activation, st);
  layer* from = &state.net->layers[l->index];
  if (dk_scale_channels_backward(l->delta_gpu, state.input, from->output_gpu, from->delta_gpu, state.delta,
          l->batch, l->out_c, l->out_h, l->out_w, l->scale_wh, st))
    error("BackwardScaleChannelsLayerGpu failed");
}

// ------------------------------------------------------------------ [dropout]
// Inference: identity; the layer aliases its predecessor's buffers exactly as the reference does
// (parser.cpp:1232-1242).  Training (dropout_layer_kernels.cu / dropout_layer.c:90-120): uniform draws, zero below
// `probability`, the rest scaled by 1/(1-p), in place; backward applies the same mask to the incoming delta.  The
// reference's cuRAND / rand() streams cannot be reproduced: the draws are a counter-based hash seeded by
// (iteration, layer index, call count) -- deterministic per run, statistically the same layer ("parity unpinned").
static void forward_dropout_gpu(layer* l, NetworkState state)
{
  if (!state.train)
    return;
  const size_t n = (size_t)l->inputs * l->batch;
  if (!l->rand_gpu)
    l->rand_gpu = cuda_make_array(nullptr, n);
  const unsigned long long seed = ((unsigned long long)(unsigned)state.net->curr_iter << 40) ^
                                  ((unsigned long long)(unsigned)l->index << 24) ^ (unsigned long long)(l->t++);
  if (dk_dropout_forward(state.input, l->rand_gpu, n, l->probability, l->scale, seed, get_cuda_stream()))
    error("forward_dropout_gpu failed");
}
static void backward_dropout_gpu(layer* l, NetworkState state)
{
  if (!state.delta)
    return;
  if (!l->rand_gpu)
    error("[dropout] backward without a train-mode forward");
  if (dk_dropout_backward(state.delta, l->rand_gpu, (size_t)l->inputs * l->batch, l->probability, l->scale, get_cuda_stream()))
    error("backward_dropout_gpu failed");
}

void FillDropoutLayer(layer* l, int batch, int inputs, float probability, int w, int h, int c)
{
  l->type = DROPOUT;
  l->probability = probability;
  l->inputs = inputs;
  l->outputs = inputs;
  l->batch = batch;
  l->out_w = l->w = w;
  l->out_h = l->h = h;
  l->out_c = l->c = c;
  l->scale = 1. / (1. - probability);
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = forward_dropout_gpu;
  l->backward_gpu = backward_dropout_gpu;
  l->buffers_aliased = 1;  // output_gpu / delta_gpu belong to the previous layer (set by the parser)
}

// ---------------------------------------------------------------- boundary odds and ends
void backward_bias_gpu(float* bias_updates, float* delta, int batch, int n, int size)
{
  dk_backward_bias(bias_updates, delta, batch, n, size, nullptr);
}

// upstream darknet spellings of the conv slots (BASELINE north_star names them)
void forward_convolutional_layer_gpu(layer* l, NetworkState state) { ForwardConvolutionalLayerGpu(l, state); }
void backward_convolutional_layer_gpu(layer* l, NetworkState state) { BackwardConvolutionalLayerGpu(l, state); }
void update_convolutional_layer_gpu(layer* l, int batch, float learning_rate, float momentum, float decay,
    float loss_scale)
{
  UpdateConvolutionalLayerGpu(l, batch, learning_rate, momentum, decay, loss_scale);
}

// The reference's init_cpu probes AVX/FMA for its CPU GEMM (utils.cpp / gemm.c); this library has no
// CPU compute path, so there is nothing to initialise.
void init_cpu(void) {}


// ------------------------------------------------------------------ [Gaussian_yolo]
// The reference's Gaussian YOLOv3 head (src/gaussian_yolo_layer.cpp): decode on the device; in train mode the
// head is pulled, the host loss (DkGaussianYoloLossHost, yolo_loss.cpp -- host code in the reference too, :968-995)
// fills the delta and the cost, and the delta is pushed back.  Synchronous, like the reference.
extern "C" float DkGaussianYoloLossHost(const layer* l, int net_w, int net_h, float* out, const float* truth, float* delta);

void ForwardGaussianYoloLayerGpu(layer* l, NetworkState state)
{
  if (dk_gaussian_yolo_forward(state.input, l->output_gpu, l->batch, l->w, l->h, l->n, l->classes, l->scale_x_y,
          get_cuda_stream()))
    error("ForwardGaussianYoloLayerGpu failed");
  if (!state.train || l->onlyforward)
    return;
  const size_t total = (size_t)l->batch * l->outputs;
  if (l->injected_delta)
  {
    cuda_push_array(l->delta_gpu, l->injected_delta, total);
    return;
  }
  if (!state.net->truth)
    error("[Gaussian_yolo] loss: no truth supplied (TrainNetworkDatum(net, x, y) with y != NULL)");
  if (!l->delta)
  {
    l->delta = cuda_make_array_pinned(nullptr, total);
    l->delta_pinned = 1;
  }
  cuda_pull_array(l->output_gpu, l->output, total);   // synchronises the stream
  *(l->cost) = DkGaussianYoloLossHost(l, state.net->w, state.net->h, l->output, state.net->truth, l->delta);
  cuda_push_array(l->delta_gpu, l->delta, total);
}

// BackwardGaussianYoloLayerGpu, src/gaussian_yolo_layer.cpp:997-1000
void BackwardGaussianYoloLayerGpu(layer* l, NetworkState state)
{
  if (state.delta)
    dk_axpy((size_t)l->batch * l->inputs, 1.f, l->delta_gpu, state.delta, get_cuda_stream());
}

// FillGaussianYoloLayer, src/gaussian_yolo_layer.cpp:26-100
void FillGaussianYoloLayer(layer* l, int batch, int w, int h, int n, int total, int* mask, int classes, int max_boxes)
{
  l->type = GAUSSIAN_YOLO;
  l->n = n;
  l->total = total;
  l->batch = batch;
  l->h = h; l->w = w;
  l->c = n * (classes + 8 + 1);
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
  l->outputs = h * w * n * (classes + 8 + 1);
  l->inputs = l->outputs;
  l->max_boxes = max_boxes;
  l->truths = l->max_boxes * (4 + 1);
  for (int i = 0; i < total * 2; ++i) l->biases[i] = .5;
  l->forward = no_cpu_path;
  l->backward = no_cpu_path;
  l->forward_gpu = ForwardGaussianYoloLayerGpu;
  l->backward_gpu = BackwardGaussianYoloLayerGpu;
  const size_t total_out = (size_t)batch * l->outputs;
  if (dk_gpu_enabled())
  {
    l->output_gpu = cuda_make_array(nullptr, total_out);
    l->delta_gpu = cuda_make_array(nullptr, total_out);
    CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, total_out * sizeof(float), get_cuda_stream()));
    l->output = cuda_make_array_pinned(nullptr, total_out);   // pinned host mirror, as for [yolo]
    l->output_pinned = 1;
    memset(l->output, 0, total_out * sizeof(float));
  }
  else
    l->output = (float*)xcalloc(total_out, sizeof(float));
}

// EntryGaussianIndex, src/gaussian_yolo_layer.cpp:477-484
static int gaussian_entry_index(const layer* l, int batch, int location, int entry)
{
  const int n = location / (l->w * l->h);
  const int loc = location % (l->w * l->h);
  return batch * l->outputs + n * l->w * l->h * (8 + l->classes + 1) + entry * l->w * l->h + loc;
}

// GaussianYoloNumDetections :859-874, for batch item b
int DkGaussianYoloNumDetectionsBatch(layer const* l, int b, float thresh)
{
  int count = 0;
  for (int i = 0; i < l->w * l->h; ++i)
    for (int n = 0; n < l->n; ++n)
      if (l->output[gaussian_entry_index(l, b, n * l->w * l->h + i, 8)] > thresh)
        ++count;
  return count;
}

// GetGaussianYoloDetections :876-930 + GetGaussianYoloBox :151-177, for batch item b.  dets[k].uc (4 floats)
// must be allocated by the caller (MakeNetworkBoxes does when the last layer is a Gaussian head).
int DkGetGaussianYoloDetectionsBatch(layer const* l, int b, int net_w, int net_h, float thresh, Detection* dets, int* ids)
{
  float const* pred = l->output;
  const int stride = l->w * l->h;
  int count = 0;
  for (int n = 0; n < l->n; ++n)
    for (int i = 0; i < l->w * l->h; ++i)
    {
      const int loc = n * l->w * l->h + i;
      const float objectness = pred[gaussian_entry_index(l, b, loc, 8)];
      if (objectness <= thresh)
        continue;
      const int box = gaussian_entry_index(l, b, loc, 0);
      const int col = i % l->w, row = i / l->w;
      const int a = l->mask[n];
      Box bx;
      bx.w = expf(pred[box + 4 * stride]) * l->biases[2 * a] / net_w;
      bx.h = expf(pred[box + 6 * stride]) * l->biases[2 * a + 1] / net_h;
      bx.x = (col + pred[box + 0 * stride]) / l->w;
      bx.y = (row + pred[box + 2 * stride]) / l->h;
      if (l->yolo_point == YOLO_LEFT_TOP)
      {
        bx.x = (col + pred[box + 0 * stride]) / l->w + bx.w / 2;
        bx.y = (row + pred[box + 2 * stride]) / l->h + bx.h / 2;
      }
      else if (l->yolo_point == YOLO_RIGHT_BOTTOM)
      {
        bx.x = (col + pred[box + 0 * stride]) / l->w - bx.w / 2;
        bx.y = (row + pred[box + 2 * stride]) / l->h - bx.h / 2;
      }
      dets[count].bbox = bx;
      dets[count].objectness = objectness;
      dets[count].classes = l->classes;
      float uc[4];
      for (int k = 0; k < 4; ++k) uc[k] = pred[gaussian_entry_index(l, b, loc, 2 * k + 1)];
      if (dets[count].uc)
        for (int k = 0; k < 4; ++k) dets[count].uc[k] = uc[k];
      dets[count].points = l->yolo_point;
      for (int j = 0; j < l->classes; ++j)
      {
        const float uc_avg = (uc[0] + uc[1] + uc[2] + uc[3]) / 4.0;
        const float prob = objectness * pred[gaussian_entry_index(l, b, loc, 9 + j)] * (1.0 - uc_avg);
        dets[count].prob[j] = (prob > thresh) ? prob : 0;
      }
      if (ids)
      {
        ids[4 * count + 0] = -1;
        ids[4 * count + 1] = n;
        ids[4 * count + 2] = row;
        ids[4 * count + 3] = col;
      }
      ++count;
    }
  return count;
}
