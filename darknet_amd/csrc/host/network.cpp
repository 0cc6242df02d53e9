// network.cpp -- graph engine of the HIP path: layer loop, predict entry points,
// BN folding, detection collection, plus the MI355X-native planning pass
// (conv+shortcut epilogue fusion, per-layer tile autotune, hipGraph replay).
//
// Reference twins (Ravicmoon/darknet src/): ForwardNetworkGpu network_kernels.cu
// :45-114, NetworkPredictGpu :502-522, GetNetworkOutputGpu :486-500,
// NetworkPredict network.cpp:412-430, NumDetections/MakeNetworkBoxes/
// FillNetworkBoxes/GetNetworkBoxes/FreeDetections :432-516, FuseConvBatchNorm
// :647-682, GetCurrLr :32-84, FreeNetwork :600-645.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>
#include <thread>

#include "dk_host.h"
#include "dk_internal.h"

int g_dk_fusion = 1;
int g_dk_graph = 1;
int g_dk_autotune = 1;
int g_dk_pull_heads = 1;

int DkYoloNumDetectionsBatch(layer const* l, int b, float thresh);
int DkGetYoloDetectionsBatch(
    layer const* l, int b, int net_w, int net_h, float thresh, Detection* dets, int* ids);
void DkConvPrepare(layer* l);
void DkDropWinograd(layer* l);

void DkSetFusion(int on) { g_dk_fusion = on; }
void DkSetGraph(int on) { g_dk_graph = on; }
void DkSetAutotune(int on) { g_dk_autotune = on; }
int g_dk_half = -1;
void DkSetHalf(int on) { g_dk_half = on; }
int g_dk_winograd = -1;
void DkSetWinograd(int on) { g_dk_winograd = on; }
extern "C" LIB_API void DkSetPullHeads(int on) { g_dk_pull_heads = on; }
// per-network overrides (stored +1; 0 = follow the process-wide setting)
void DkNetSetGraph(Network* net, int on) { net->opt_graph = on < 0 ? 0 : (on ? 2 : 1); }
void DkNetSetPullHeads(Network* net, int on) { net->opt_pull_heads = on < 0 ? 0 : (on ? 2 : 1); }
static bool net_graph(const Network* net) { return net->opt_graph ? net->opt_graph == 2 : g_dk_graph != 0; }
static bool net_pull_heads(const Network* net) { return net->opt_pull_heads ? net->opt_pull_heads == 2 : g_dk_pull_heads != 0; }
void DkFreeDpState(Network* net);
void DkInvalidateSgdPlan(Network* net);

// ---------------------------------------------------------------------------
int GetNetworkInputSize(Network* net) { return net->layers[0].inputs; }

int GetNetworkOutputSize(Network* net)
{
  int i;
  for (i = net->n - 1; i > 0; --i)
    if (net->layers[i].type != COST)
      break;
  return net->layers[i].outputs;
}

float GetCurrLr(Network* net)
{
  // network.cpp:32-84 (pow on float arguments is the float overload there)
  const int64_t iter = net->curr_iter;
  if (iter < net->burn_in)
    return net->lr * powf((float)iter / net->burn_in, net->power);
  switch (net->policy)
  {
    case CONSTANT: return net->lr;
    // pow(float, integer) promotes to double and the product with lr stays double until the return narrows it
    case STEP: return (float)(net->lr * pow((double)net->scale, (double)(iter / net->step)));
    case STEPS:
    {
      float lr = net->lr;
      for (int i = 0; i < net->num_steps; ++i)
      {
        if (net->max_iter * net->steps[i] > iter)
          return lr;
        lr *= net->scales[i];
      }
      return lr;
    }
    case EXP: return (float)(net->lr * pow((double)net->gamma, (double)iter));
    case POLY: return net->lr * powf(1 - (float)iter / net->max_iter, net->power);
    case SIG: return net->lr * (1. / (1. + exp(net->gamma * (iter - net->step))));
    case SGDR:
    {
      // network.cpp:66-78: warm restarts; cos() on a float product promoted to double by M_PI
      int last_iter = 0;
      int cycle = net->sgdr_cycle;
      while (last_iter + cycle < iter)
      {
        last_iter += cycle;
        cycle *= net->sgdr_mult;
      }
      return net->lr_min + 0.5f * (net->lr - net->lr_min) * (1.0f + cos((float)(iter - last_iter) * M_PI / cycle));
    }
    default: break;
  }
  fprintf(stderr, "GetCurrLr: policy %d is outside the supported hot path\n", (int)net->policy);
  return net->lr;
}

// FuseConvBatchNorm, network.cpp:647-682 (host arithmetic, then re-push)
void FuseConvBatchNorm(Network* net)
{
  for (int j = 0; j < net->n; ++j)
  {
    layer* l = &net->layers[j];
    if (l->type != CONVOLUTIONAL || !l->batch_normalize)
      continue;
    const int filter_size = l->size * l->size * l->c / l->groups;
    for (int f = 0; f < l->n; ++f)
    {
      const float std = sqrtf(l->rolling_variance[f] + 0.00001f);
      l->biases[f] -= l->scales[f] * l->rolling_mean[f] / std;
      for (int i = 0; i < filter_size; ++i)
        l->weights[(size_t)f * filter_size + i] *= l->scales[f] / std;
    }
    // free_convolutional_batchnorm: the BN tensors are released
    free(l->scales); l->scales = nullptr;
    free(l->rolling_mean); l->rolling_mean = nullptr;
    free(l->rolling_variance); l->rolling_variance = nullptr;
    if (dk_gpu_enabled())
    {
      cuda_free(l->scales_gpu); l->scales_gpu = nullptr;
      cuda_free(l->rolling_mean_gpu); l->rolling_mean_gpu = nullptr;
      cuda_free(l->rolling_variance_gpu); l->rolling_variance_gpu = nullptr;
    }
    l->batch_normalize = 0;
    PushConvolutionalLayer(l);
  }
}

// ---------------------------------------------------------------------------
// planning pass for inference loads
// ---------------------------------------------------------------------------
static bool output_referenced_elsewhere(Network* net, int idx, int except_layer)
{
  for (int i = 0; i < net->n; ++i)
  {
    if (i == except_layer)
      continue;
    layer* l = &net->layers[i];
    if (l->type == ROUTE)
      for (int k = 0; k < l->n; ++k)
        if (l->input_layers[k] == idx)
          return true;
    if ((l->type == SHORTCUT || l->type == SCALE_CHANNELS) && l->index == idx)
      return true;
  }
  return false;
}

static void run_layers_eager(Network* net, NetworkState state)
{
  for (int i = 0; i < net->n; ++i)
  {
    state.index = i;
    layer* l = &net->layers[i];
    if (l->forward_gpu)
      l->forward_gpu(l, state);
    if (l->output_gpu)
      state.input = DkLayerOut(l);
  }
}

void DkInvalidateGraph(Network* net)
{
  if (net->graph_exec)
  {
    (void)hipGraphExecDestroy((hipGraphExec_t)net->graph_exec);
    net->graph_exec = nullptr;
  }
}

void DkDropWinograd(layer* l)
{
  if (!l->weights_wino_gpu)
    return;
  dk_conv_wino_register(l->weights_gpu, nullptr);
  cuda_free(l->weights_wino_gpu);
  l->weights_wino_gpu = nullptr;
}

static void autotune_convs(Network* net)
{
  const int ncfg = dk_conv_num_configs();
  hipStream_t st = get_cuda_stream();
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  // sane values in every buffer first
  const size_t in_n = (size_t)GetNetworkInputSize(net) * net->batch;
  dk_fill(in_n, 0.5f, net->input_state_gpu, st);
  NetworkState state;
  memset(&state, 0, sizeof(state));
  state.net = net;
  state.input = net->input_state_gpu;
  state.workspace = net->workspace;
  run_layers_eager(net, state);
  CHECK_HIP(hipStreamSynchronize(st));

  std::map<std::vector<int>, int> cache;
  float* in = net->input_state_gpu;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    DkConvDesc hd = {l->batch, l->c, l->h, l->w, l->n, l->groups, l->size, l->stride_x, l->stride_y, l->dilation, l->pad, (int)l->activation};
    const bool half_layer = net->cudnn_half && l->type == CONVOLUTIONAL && dk_conv_half_eligible(&hd, i);
    if (l->type == CONVOLUTIONAL && !l->batch_normalize && !half_layer && !l->dual_slave)
    {
      std::vector<int> key = {l->batch, l->c, l->h, l->w, l->n, l->groups, l->size, l->stride_x,
          l->stride_y, l->dilation, l->pad, (int)l->activation, l->fuse_residual_from >= 0,
          l->dual_with > 0 ? net->layers[l->dual_with].n : 0};
      auto it = cache.find(key);
      if (it != cache.end())
        l->conv_cfg = it->second;
      else
      {
        NetworkState s = state;
        s.input = in;
        s.index = i;
        int best = -1;
        float best_ms = 1e30f;
        std::vector<float> times(ncfg, 0.f);
        for (int c = 0; c < ncfg; ++c)
        {
          if (!dk_conv_config_applicable(&hd, c) || (dk_conv_config_is_wino(c) && !l->weights_wino_gpu))
            continue;
          l->conv_cfg = c;
          l->forward_gpu(l, s);  // warm
          // minimum of three timings of two launches: a single noisy sample must not
          // decide between shapes that are a few per cent apart
          float ms = 1e30f;
          for (int rep = 0; rep < 3; ++rep)
          {
            CHECK_HIP(hipEventRecord(e0, st));
            l->forward_gpu(l, s);
            l->forward_gpu(l, s);
            CHECK_HIP(hipEventRecord(e1, st));
            CHECK_HIP(hipEventSynchronize(e1));
            float t = 0;
            CHECK_HIP(hipEventElapsedTime(&t, e0, e1));
            if (t < ms)
              ms = t;
          }
          times[c] = ms;
          if (ms < best_ms)
          {
            best_ms = ms;
            best = c;
          }
        }
        // hysteresis in favour of the heuristic's shape: shapes within 3 % of each other are
        // separated by timing noise, not by merit (observed: whole groups of layers flipping
        // between runs), and the heuristic is right on average
        const int heur = dk_conv_pick_config(&hd);
        static const float hyst = getenv("DK_TUNE_HYST") ? (float)atof(getenv("DK_TUNE_HYST")) : 3.0f;  // per cent
        if (heur >= 0 && heur < ncfg && times[heur] > 0 && best != heur && times[heur] <= best_ms * (1.f + hyst / 100.f))
          best = heur;
        l->conv_cfg = best;
        cache[key] = best;
      }
    }
    if (l->output_gpu)
      in = DkLayerOut(l);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
}

void DkPlanInference(Network* net)
{
  {
    const char* eh = getenv("DK_HALF");
    net->cudnn_half = g_dk_half >= 0 ? g_dk_half : (eh ? atoi(eh) != 0 : 0);
  }
  // 1. conv + shortcut(linear) epilogue fusion
  for (int i = 1; i < net->n; ++i)
  {
    layer* sc = &net->layers[i];
    layer* cv = &net->layers[i - 1];
    sc->fused_into_prev = 0;
    if (cv->type == CONVOLUTIONAL)
      cv->fuse_residual_from = -1;
    if (!g_dk_fusion)
      continue;
    if (sc->type != SHORTCUT || sc->activation != LINEAR || cv->type != CONVOLUTIONAL ||
        cv->batch_normalize)
      continue;
    layer* from = &net->layers[sc->index];
    if (from->out_w != sc->w || from->out_h != sc->h || from->out_c != sc->c)
      continue;
    if (sc->index == i - 1 || output_referenced_elsewhere(net, i - 1, i))
      continue;
    cv->fuse_residual_from = sc->index;
    sc->fused_into_prev = 1;
  }
  // 1b. zero-copy concatenation: a producer read by nobody but one multi-input [route]
  // writes its channel slice of the route's buffer itself; a one-input [route] aliases
  // its source.  (route_layer.c:124-142 copies every input; here most copies vanish.)
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    l->out_view = nullptr;
    l->out_view_ctot = 0;
    l->out_alias = nullptr;
    free(l->input_inplace);
    l->input_inplace = nullptr;
  }
  if (g_dk_fusion)
  {
    std::vector<int> readers(net->n, 0);
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type == ROUTE)
        for (int k = 0; k < l->n; ++k) readers[l->input_layers[k]]++;
      else
      {
        if (i > 0)
          readers[i - 1]++;  // every other layer kind reads its predecessor
        if (l->type == SHORTCUT || l->type == SCALE_CHANNELS)
          readers[l->index]++;
      }
    }
    readers[net->n - 1]++;  // the network's output
    for (int i = 0; i < net->n; ++i)
    {
      layer* r = &net->layers[i];
      if (r->type != ROUTE || r->groups != 1)
        continue;
      if (r->n == 1)
        continue;
      r->input_inplace = (int*)xcalloc(r->n, sizeof(int));
      size_t offset = 0;
      for (int k = 0; k < r->n; ++k)
      {
        layer* pl = &net->layers[r->input_layers[k]];
        const bool kind_ok = (pl->type == CONVOLUTIONAL && !pl->batch_normalize && pl->fuse_residual_from < 0) ||
                             pl->type == MAXPOOL || pl->type == UPSAMPLE;
        const size_t hw = (size_t)pl->out_h * pl->out_w;
        if (kind_ok && readers[r->input_layers[k]] == 1 && !pl->out_view && hw > 0 &&
            (size_t)r->outputs % hw == 0 && pl->outputs == r->input_sizes[k])
        {
          pl->out_view = r->output_gpu + offset;
          pl->out_view_ctot = (int)((size_t)r->outputs / hw);
          r->input_inplace[k] = 1;
        }
        offset += r->input_sizes[k];
      }
    }
    for (int i = 0; i < net->n; ++i)
    {
      layer* r = &net->layers[i];
      if (r->type != ROUTE || r->groups != 1 || r->n != 1)
        continue;
      layer* src = &net->layers[r->input_layers[0]];
      if (!src->out_view && src->outputs == r->outputs)
        r->out_alias = DkLayerOut(src);
    }
  }
  // 1b'. the two 1x1 branches of a CSP stage read the same tensor (conv, [route -2], conv):
  // one launch with concatenated filters computes both and reads the tensor once
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    l->dual_with = 0;
    l->dual_slave = 0;
    l->dual_peer = nullptr;
    if (l->type == CONVOLUTIONAL)
    {
      cuda_free(l->dual_weights_gpu); l->dual_weights_gpu = nullptr;
      cuda_free(l->dual_biases_gpu); l->dual_biases_gpu = nullptr;
    }
  }
  {
    const char* ed = getenv("DK_DUAL");
    const bool enable = g_dk_fusion && !(ed && !atoi(ed));
    for (int i = 1; enable && i + 2 < net->n; ++i)
    {
      layer* a = &net->layers[i];
      layer* r = &net->layers[i + 1];
      layer* b = &net->layers[i + 2];
      auto plain1x1 = [](const layer* c) {
        return c->type == CONVOLUTIONAL && !c->batch_normalize && c->size == 1 && c->stride_x == 1 &&
               c->stride_y == 1 && c->groups == 1 && c->pad == 0 && c->fuse_residual_from < 0 && !c->dual_slave &&
               c->dual_with == 0;
      };
      if (!plain1x1(a) || !plain1x1(b) || r->type != ROUTE || !r->out_alias)
        continue;
      if (r->out_alias != DkLayerOut(&net->layers[i - 1]))
        continue;  // b must read the tensor a reads
      if (a->c != b->c || a->h != b->h || a->w != b->w || a->activation != b->activation || a->n % 64 ||
          a->batch != b->batch)
        continue;
      const size_t k = (size_t)a->c;
      a->dual_weights_gpu = cuda_make_array(nullptr, (size_t)(a->n + b->n) * k);
      a->dual_biases_gpu = cuda_make_array(nullptr, (size_t)(a->n + b->n));
      hipStream_t st = get_cuda_stream();
      CHECK_HIP(hipMemcpyAsync(a->dual_weights_gpu, a->weights_gpu, (size_t)a->n * k * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_weights_gpu + (size_t)a->n * k, b->weights_gpu, (size_t)b->n * k * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_biases_gpu, a->biases_gpu, (size_t)a->n * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipMemcpyAsync(a->dual_biases_gpu + a->n, b->biases_gpu, (size_t)b->n * sizeof(float), hipMemcpyDeviceToDevice, st));
      CHECK_HIP(hipStreamSynchronize(st));
      a->dual_with = i + 2;
      b->dual_slave = 1;
      a->dual_peer = b;
      b->dual_peer = a;
    }
  }
  // 1c. fp16 path: pack the weights of the layers that take the direct fp16 kernel
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type != CONVOLUTIONAL)
      continue;
    cuda_free((float*)l->weights_half_gpu);
    l->weights_half_gpu = nullptr;
    const char* hd = getenv("DK_HALF_DIRECT");
    if (!net->cudnn_half || l->batch_normalize || (hd && !atoi(hd)))
      continue;
    DkConvDesc hd2 = {l->batch, l->c, l->h, l->w, l->n, l->groups, l->size, l->stride_x, l->stride_y, l->dilation, l->pad, (int)l->activation};
    const size_t halves = dk_conv_half_eligible(&hd2, i) ? dk_conv_half_direct_weights_size(&hd2) : 0;
    if (!halves)
      continue;
    l->weights_half_gpu = cuda_make_array(nullptr, (halves + 1) / 2);
    if (dk_conv_half_pack_weights(&hd2, l->weights_gpu, l->weights_half_gpu, get_cuda_stream()))
      error("dk_conv_half_pack_weights failed");
  }
  // 1d. Winograd candidates: transformed filters for every layer the kernel can take; the autotune
  // below decides per layer, and the copies of the layers that keep another kernel are dropped again
  {
    const char* ew = getenv("DK_WINOGRAD");
    const bool wino = (g_dk_winograd >= 0 ? g_dk_winograd : (ew ? atoi(ew) : 1)) != 0;
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type != CONVOLUTIONAL)
        continue;
      DkDropWinograd(l);
      DkConvDesc wd = {l->batch, l->c, l->h, l->w, l->n, l->groups, l->size, l->stride_x, l->stride_y, l->dilation, l->pad, (int)l->activation};
      const size_t nu = dk_conv_wino_weights_size(&wd);
      // DK_WINO_MAXC: optional cap on the input channels of Winograd candidates (diagnostics).  At K = 9 x 512 the
      // kernel sits 1.5e-5 x rms from the CPU oracle on single elements (tests/test_gpu_ops.py); measured over whole
      // networks with EVERY eligible layer on it (tools/wino_margin.py, yolov4 / csp b = 16): worst per-layer sample at
      // 0.16 of the parity bound (0.11 without Winograd), so no cap by default.
      static const int wino_maxc = getenv("DK_WINO_MAXC") ? atoi(getenv("DK_WINO_MAXC")) : (1 << 30);
      if (!wino || !nu || l->c > wino_maxc || l->batch_normalize || l->weights_half_gpu || l->dual_with > 0 || l->dual_slave)
        continue;
      l->weights_wino_gpu = cuda_make_array(nullptr, nu);
      if (dk_conv_wino_transform_weights(&wd, l->weights_gpu, l->weights_wino_gpu, get_cuda_stream()))
        error("dk_conv_wino_transform_weights failed");
      dk_conv_wino_register(l->weights_gpu, l->weights_wino_gpu);
    }
  }
  // 2. tap tables (must exist before any stream capture)
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == CONVOLUTIONAL)
      DkConvPrepare(&net->layers[i]);
  // 3. per-layer tile choice
  const char* e = getenv("DK_AUTOTUNE");
  const bool tune = e ? atoi(e) != 0 : g_dk_autotune != 0;
  // DK_TUNE_FILE: reuse a previous run's per-layer choice (profiling runs then
  // contain no tuning launches); written when absent.
  const char* tf = getenv("DK_TUNE_FILE");
  bool loaded = false;
  // the cache is keyed on every conv descriptor, the plan flags and half mode (FNV-1a)
  unsigned long long tkey = 1469598103934665603ULL;
  auto mix = [&tkey](long long v) {
    for (int k = 0; k < 8; ++k) { tkey ^= (unsigned long long)((v >> (8 * k)) & 0xff); tkey *= 1099511628211ULL; }
  };
  mix(net->n); mix(net->batch); mix(net->cudnn_half); mix(dk_conv_num_configs());
  for (int i = 0; i < net->n; ++i) mix(net->layers[i].weights_wino_gpu != nullptr);
  for (int i = 0; i < net->n; ++i)
  {
    const layer* l = &net->layers[i];
    if (l->type != CONVOLUTIONAL)
      continue;
    for (long long v : {(long long)i, (long long)l->c, (long long)l->h, (long long)l->w, (long long)l->n, (long long)l->groups,
             (long long)l->size, (long long)l->stride_x, (long long)l->stride_y, (long long)l->dilation, (long long)l->pad,
             (long long)l->activation, (long long)(l->fuse_residual_from >= 0), (long long)l->dual_with, (long long)l->dual_slave})
      mix(v);
  }
  if (tune && tf)
  {
    if (FILE* f = fopen(tf, "r"))
    {
      int n = 0, b = 0;
      unsigned long long k = 0;
      if (fscanf(f, "%d %d %llu", &n, &b, &k) == 3 && n == net->n && b == net->batch && k == tkey)
      {
        loaded = true;
        for (int i = 0; i < net->n; ++i)
        {
          int c = -1;
          if (fscanf(f, "%d", &c) != 1)
          {
            loaded = false;
            break;
          }
          if (net->layers[i].type == CONVOLUTIONAL)
            net->layers[i].conv_cfg = c;
        }
      }
      fclose(f);
    }
  }
  if (tune && !loaded)
  {
    autotune_convs(net);
    if (tf)
      if (FILE* f = fopen(tf, "w"))
      {
        fprintf(f, "%d %d %llu\n", net->n, net->batch, tkey);
        for (int i = 0; i < net->n; ++i) fprintf(f, "%d\n", net->layers[i].conv_cfg);
        fclose(f);
      }
  }
  // transformed filters of the layers that did not choose the Winograd kernel are not needed
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL && l->weights_wino_gpu && !dk_conv_config_is_wino(l->conv_cfg))
      DkDropWinograd(l);
  }
  DkInvalidateGraph(net);
  net->planned = 1;
}

// ---------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------
void ForwardNetworkGpu(Network* net, NetworkState state)
{
  state.workspace = net->workspace;
  hipStream_t st = get_cuda_stream();
  int pulled = 0;
  if (state.train)
    DkTrainPrepRun(net);
  if (state.train && net->delta_arena_gpu && net->delta_arena_zero)
    CHECK_HIP(hipMemsetAsync(net->delta_arena_gpu, 0, net->delta_arena_zero * sizeof(float), st));   // see DkBuildDeltaArena
  for (int i = 0; i < net->n; ++i)
  {
    state.index = i;
    layer* l = &net->layers[i];
    if (l->delta_gpu && state.train && !l->delta_in_arena)
      CHECK_HIP(hipMemsetAsync(l->delta_gpu, 0, (size_t)l->outputs * l->batch * sizeof(float), st));
    std::chrono::steady_clock::time_point t0;
    if (net->benchmark_layers)
    {
      CHECK_HIP(hipStreamSynchronize(st));
      t0 = std::chrono::steady_clock::now();
    }
    if (l->forward_gpu)
      l->forward_gpu(l, state);
    if (net->benchmark_layers)
    {
      CHECK_HIP(hipStreamSynchronize(st));
      const double ms =
          std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      printf("\n fw-layer %d - type: %d - %lf ms \n", i, (int)l->type, ms);
    }
    if (net->wait_stream)
      CHECK_HIP(hipStreamSynchronize(st));
    // yolo heads -> pinned host memory on the copy stream AS SOON AS the head exists (the reference pulls them
    // asynchronously inside ForwardYoloLayerGpu, yolo_layer.cpp:854-858): yolov4's 76x76 head is 94 of the 123 MB a batch
    // of 16 sends back and is final a fifth of the forward before its end, so most of the PCIe time runs under the
    // remaining layers.  (Never inside a stream capture: NetworkPredictDevice sets the flag only for plain launches.)
    if (net->pull_in_forward && !state.train && (l->type == YOLO || l->type == GAUSSIAN_YOLO) && pulled < 8)
    {
      hipStream_t cs = get_cuda_memcpy_stream();
      hipEvent_t ev = (hipEvent_t)net->head_ev[pulled++];
      CHECK_HIP(hipEventRecord(ev, st));
      CHECK_HIP(hipStreamWaitEvent(cs, ev, 0));
      CHECK_HIP(hipMemcpyAsync(l->output, l->output_gpu, (size_t)l->batch * l->outputs * sizeof(float), hipMemcpyDeviceToHost, cs));
    }
    if (l->output_gpu)
      state.input = DkLayerOut(l);
  }
  if (pulled)
  {
    // join: the compute stream is finished when the heads are on the host (and may rewrite them afterwards)
    hipStream_t cs = get_cuda_memcpy_stream();
    CHECK_HIP(hipEventRecord((hipEvent_t)net->copy_done_ev, cs));
    CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)net->copy_done_ev, 0));
  }
}

// the events live in the Network (one thread per network may run predictions concurrently)
static void ensure_events(Network* net)
{
  if (!net->fwd_done_ev)
  {
    hipEvent_t a, b;
    CHECK_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    net->fwd_done_ev = a;
    net->copy_done_ev = b;
    for (int k = 0; k < 8; ++k)
    {
      hipEvent_t e;
      CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      net->head_ev[k] = e;
    }
  }
}

extern "C" int dk_profile_is_on();

void NetworkPredictDevice(Network* net, float* input_gpu)
{
  if (net->gpu_index < 0)
    error("NetworkPredictDevice: network was loaded without a HIP device (no CPU fallback)");
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  ensure_events(net);
  hipStream_t st = get_cuda_stream();
  net->predict_seq++;
  NetworkState state;
  memset(&state, 0, sizeof(state));
  state.net = net;
  state.input = input_gpu ? input_gpu : net->input_state_gpu;
  state.train = 0;

  // (the head copies of the previous call were joined into the compute stream by that call's forward)
  net->copy_pending = 0;
  const bool pull = net_pull_heads(net);
  int yolo_heads = 0;
  size_t head_bytes = 0;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == YOLO || net->layers[i].type == GAUSSIAN_YOLO)
    {
      ++yolo_heads;
      head_bytes += (size_t)net->layers[i].batch * net->layers[i].outputs * sizeof(float);
    }

  // Large heads go out under PLAIN launches, each as soon as its layer has run (ForwardNetworkGpu): captured into a graph
  // the copy branch runs after the kernels (measured, yolov4 608 b16, frames in / heads out: 809 images/s replayed, 919
  // launched, 930 without the heads), and at the batch sizes where the heads are tens of MB the launches hide behind the
  // kernels anyway.  Small heads (batch 1-2) keep the replay and are copied after it -- the capture never touches the
  // process-wide copy stream, so other host threads may use that stream meanwhile.
  const bool big_heads = pull && head_bytes > ((size_t)8 << 20);
  const bool can_graph = net_graph(net) && !net->benchmark_layers && !net->wait_stream &&
                         !dk_profile_is_on() && state.input == net->input_state_gpu && !big_heads;
  net->pull_in_forward = (pull && !can_graph && yolo_heads <= 8) ? 1 : 0;
  const bool late_pull = pull && !net->pull_in_forward;
  if (can_graph)
  {
    if (!net->graph_exec)
    {
      hipGraph_t graph = nullptr;
      CHECK_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      ForwardNetworkGpu(net, state);
      CHECK_HIP(hipStreamEndCapture(st, &graph));
      hipGraphExec_t exec = nullptr;
      CHECK_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      CHECK_HIP(hipGraphDestroy(graph));
      net->graph_exec = exec;
    }
    CHECK_HIP(hipGraphLaunch((hipGraphExec_t)net->graph_exec, st));
  }
  else
    ForwardNetworkGpu(net, state);
  net->pull_in_forward = 0;

  if (late_pull)
  {
    hipStream_t cs = get_cuda_memcpy_stream();
    CHECK_HIP(hipEventRecord((hipEvent_t)net->fwd_done_ev, st));
    CHECK_HIP(hipStreamWaitEvent(cs, (hipEvent_t)net->fwd_done_ev, 0));
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type == YOLO || l->type == GAUSSIAN_YOLO)
        CHECK_HIP(hipMemcpyAsync(l->output, l->output_gpu,
            (size_t)l->batch * l->outputs * sizeof(float), hipMemcpyDeviceToHost, cs));
    }
    CHECK_HIP(hipEventRecord((hipEvent_t)net->copy_done_ev, cs));
    CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)net->copy_done_ev, 0));
  }
}

void NetworkSync(Network* net)
{
  if (net->gpu_index < 0)
    return;
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
  CHECK_HIP(hipStreamSynchronize(get_cuda_memcpy_stream()));
}

float* GetNetworkOutputGpu(Network* net)
{
  int i;
  for (i = net->n - 1; i > 0; --i)
    if (net->layers[i].type != COST)
      break;
  layer* l = &net->layers[i];
  NetworkSync(net);
  if ((l->type != YOLO && l->type != GAUSSIAN_YOLO) || !net_pull_heads(net))
    cuda_pull_array(l->output_gpu, l->output, (size_t)l->outputs * l->batch);
  return l->output;
}

// Host frames (float) -> input_state_gpu.  A batch of 608x608 frames is 71 MB: one memcpy into pinned memory and
// one H2D behind it cost ~8 ms in front of a 17 ms forward.  The copy is cut into pieces handled by a few host
// threads; each piece crosses PCIe on the copy stream as soon as it is in pinned memory, so the host copy (several
// cores) and the DMA overlap, and the forward waits for one event.  DK_STAGE_THREADS (default 4; 1 = the serial form).
static void stage_float_input(Network* net, const float* input, size_t size)
{
  static const int want = getenv("DK_STAGE_THREADS") ? atoi(getenv("DK_STAGE_THREADS")) : 4;
  const size_t piece = (size_t)1 << 20;   // floats (4 MB)
  const int npieces = (int)((size + piece - 1) / piece);
  int nthreads = want < 1 ? 1 : want;
  if (nthreads > npieces) nthreads = npieces;
  if (nthreads <= 1)
  {
    memcpy(net->input_pinned_cpu, input, size * sizeof(float));
    cuda_push_array(net->input_state_gpu, net->input_pinned_cpu, size);
    return;
  }
  ensure_events(net);
  hipStream_t st = get_cuda_stream(), cs = get_cuda_memcpy_stream();
  // whatever still reads the input buffer on the compute stream goes first
  CHECK_HIP(hipEventRecord((hipEvent_t)net->fwd_done_ev, st));
  CHECK_HIP(hipStreamWaitEvent(cs, (hipEvent_t)net->fwd_done_ev, 0));
  const int dev = cuda_get_device();
  std::atomic<int> next(0);
  auto work = [&]() {
    (void)hipSetDevice(dev);
    for (;;)
    {
      const int k = next.fetch_add(1);
      if (k >= npieces)
        return;
      const size_t off = (size_t)k * piece, n = (off + piece <= size) ? piece : size - off;
      memcpy(net->input_pinned_cpu + off, input + off, n * sizeof(float));
      CHECK_HIP(hipMemcpyAsync(net->input_state_gpu + off, net->input_pinned_cpu + off, n * sizeof(float),
          hipMemcpyHostToDevice, cs));
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  CHECK_HIP(hipEventRecord((hipEvent_t)net->copy_done_ev, cs));
  CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)net->copy_done_ev, 0));
}

float* NetworkPredictGpu(Network* net, float* input)
{
  if (net->gpu_index < 0)
    error("NetworkPredict: no HIP device (this library has no CPU fallback)");
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  const size_t size = (size_t)GetNetworkInputSize(net) * net->batch;
  stage_float_input(net, input, size);
  NetworkPredictDevice(net, nullptr);
  return GetNetworkOutputGpu(net);
}

float* NetworkPredict(Network* net, float* input) { return NetworkPredictGpu(net, input); }

// Staged inputs cross PCIe on a stream of their own: on the shared copy stream they would queue behind the D2H of the
// previous batch's heads (123 MB for yolov4 b=16) and start only when that forward has finished.
static hipStream_t stage_stream_of(Network* net)
{
  if (!net->stage_stream)
  {
    hipStream_t s;
    CHECK_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    net->stage_stream = s;
  }
  return (hipStream_t)net->stage_stream;
}

static void free_float_stage(Network* net)
{
  if (net->f32_stage_gpu) (void)hipFree(net->f32_stage_gpu);
  if (net->f32_stage_pinned) (void)hipHostFree(net->f32_stage_pinned);
  net->f32_stage_gpu = net->f32_stage_pinned = nullptr;
  net->f32_stage_floats = 0;
  net->f32_copy_pending = net->f32_staged = 0;
}

void DkNetworkStageFloat(Network* net, const float* input)
{
  if (net->gpu_index < 0)
    error("DkNetworkStageFloat: no HIP device (this library has no CPU fallback)");
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  const size_t size = (size_t)GetNetworkInputSize(net) * net->batch;
  if (net->f32_stage_floats != size)
  {
    NetworkSync(net);
    free_float_stage(net);
    CHECK_HIP(hipMalloc((void**)&net->f32_stage_gpu, size * sizeof(float)));
    CHECK_HIP(hipHostMalloc((void**)&net->f32_stage_pinned, size * sizeof(float), hipHostMallocDefault));
    net->f32_stage_floats = size;
  }
  if (!net->f32_h2d_ev)
  {
    hipEvent_t a, b;
    CHECK_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
    net->f32_h2d_ev = a;
    net->f32_copy_ev = b;
  }
  // the device copy that read the staging buffer (previous DkNetworkPredictStaged) must be done before it is refilled;
  // it waited for the previous H2D, so the pinned buffer is free as well
  if (net->f32_copy_pending)
  {
    CHECK_HIP(hipEventSynchronize((hipEvent_t)net->f32_copy_ev));
    net->f32_copy_pending = 0;
  }
  else if (net->f32_staged)
    CHECK_HIP(hipEventSynchronize((hipEvent_t)net->f32_h2d_ev));   // re-staging without a forward in between
  static const int want = getenv("DK_STAGE_THREADS") ? atoi(getenv("DK_STAGE_THREADS")) : 4;
  const size_t piece = (size_t)1 << 20;   // floats (4 MB)
  const int npieces = (int)((size + piece - 1) / piece);
  int nthreads = want < 1 ? 1 : want;
  if (nthreads > npieces) nthreads = npieces;
  hipStream_t ss = stage_stream_of(net);
  const int dev = cuda_get_device();
  std::atomic<int> next(0);
  auto work = [&]() {
    (void)hipSetDevice(dev);
    for (;;)
    {
      const int k = next.fetch_add(1);
      if (k >= npieces)
        return;
      const size_t off = (size_t)k * piece, n = (off + piece <= size) ? piece : size - off;
      memcpy(net->f32_stage_pinned + off, input + off, n * sizeof(float));
      CHECK_HIP(hipMemcpyAsync(net->f32_stage_gpu + off, net->f32_stage_pinned + off, n * sizeof(float), hipMemcpyHostToDevice, ss));
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  CHECK_HIP(hipEventRecord((hipEvent_t)net->f32_h2d_ev, ss));
  net->f32_staged = 1;
}

// the input step before the path.  The frames cross PCIe as bytes (4x fewer
// than floats) and Mat2Image's arithmetic runs on the device.
static void stage_frames(Network* net, const unsigned char* frames_hwc, int src_w, int src_h, size_t row_step, int swap_rb);

void DkNetworkStageU8(Network* net, const unsigned char* frames_hwc, size_t row_step)
{
  stage_frames(net, frames_hwc, 0, 0, row_step, 0);
}

void DkNetworkStageFrames(Network* net, const unsigned char* frames_hwc, int src_w, int src_h, size_t row_step, int swap_rb)
{
  if (src_w < 1 || src_h < 1)
    error("DkNetworkStageFrames: invalid frame size");
  stage_frames(net, frames_hwc, src_w, src_h, row_step, swap_rb);
}

static void stage_frames(Network* net, const unsigned char* frames_hwc, int src_w, int src_h, size_t row_step, int swap_rb)
{
  if (net->gpu_index < 0)
    error("DkNetworkStageU8: no HIP device (this library has no CPU fallback)");
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  if (row_step < (size_t)(src_w ? src_w : net->w) * net->c)
    error("DkNetworkStageU8: row_step smaller than one row");
  const size_t bytes = row_step * (src_h ? src_h : net->h) * net->batch;
  if (net->u8_bytes < bytes)
  {
    NetworkSync(net);
    if (net->u8_gpu) CHECK_HIP(hipFree(net->u8_gpu));
    if (net->u8_pinned) CHECK_HIP(hipHostFree(net->u8_pinned));
    if (net->u8_gpu2) CHECK_HIP(hipFree(net->u8_gpu2));
    if (net->u8_pinned2) CHECK_HIP(hipHostFree(net->u8_pinned2));
    CHECK_HIP(hipMalloc((void**)&net->u8_gpu, bytes));
    CHECK_HIP(hipHostMalloc((void**)&net->u8_pinned, bytes, hipHostMallocDefault));
    CHECK_HIP(hipMalloc((void**)&net->u8_gpu2, bytes));
    CHECK_HIP(hipHostMalloc((void**)&net->u8_pinned2, bytes, hipHostMallocDefault));
    net->u8_bytes = bytes;
    net->u8_conv_pending[0] = net->u8_conv_pending[1] = 0;
  }
  for (int i = 0; i < 2; ++i)
    if (!net->u8_h2d_ev[i])
    {
      hipEvent_t a, b;
      CHECK_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      CHECK_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      net->u8_h2d_ev[i] = a;
      net->u8_conv_ev[i] = b;
    }
  const int slot = net->u8_next & 1;
  // the conversion that last read this slot (two stage calls ago) must be done before its buffers are reused
  if (net->u8_conv_pending[slot])
  {
    CHECK_HIP(hipEventSynchronize((hipEvent_t)net->u8_conv_ev[slot]));
    net->u8_conv_pending[slot] = 0;
  }
  unsigned char* pinned = slot ? net->u8_pinned2 : net->u8_pinned;
  unsigned char* dev = slot ? net->u8_gpu2 : net->u8_gpu;
  memcpy(pinned, frames_hwc, bytes);
  hipStream_t cs = stage_stream_of(net);
  CHECK_HIP(hipMemcpyAsync(dev, pinned, bytes, hipMemcpyHostToDevice, cs));
  CHECK_HIP(hipEventRecord((hipEvent_t)net->u8_h2d_ev[slot], cs));
  net->u8_staged = slot + 1;
  net->u8_row_step = row_step;
  net->u8_src_w = src_w;
  net->u8_src_h = src_h;
  net->u8_swap_rb = swap_rb;
  net->u8_next = slot ^ 1;
}

void DkNetworkPredictStaged(Network* net)
{
  if (net->gpu_index < 0)
    error("DkNetworkPredictStaged: no HIP device (this library has no CPU fallback)");
  if (net->gpu_index != cuda_get_device())
    cuda_set_device(net->gpu_index);
  if (net->f32_staged)
  {
    hipStream_t st = get_cuda_stream();
    CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)net->f32_h2d_ev, 0));
    // (stream order puts the copy behind the previous forward's reads of the input tensor)
    CHECK_HIP(hipMemcpyAsync(net->input_state_gpu, net->f32_stage_gpu, net->f32_stage_floats * sizeof(float),
        hipMemcpyDeviceToDevice, st));
    CHECK_HIP(hipEventRecord((hipEvent_t)net->f32_copy_ev, st));
    net->f32_copy_pending = 1;
    net->f32_staged = 0;
    NetworkPredictDevice(net, nullptr);
    return;
  }
  if (net->u8_staged < 1)
    error("DkNetworkPredictStaged: nothing staged (call DkNetworkStageU8 / DkNetworkStageFrames / DkNetworkStageFloat first)");
  const int slot = net->u8_staged - 1;
  hipStream_t st = get_cuda_stream();
  // stream order already puts the conversion behind the previous forward's reads of the input tensor
  CHECK_HIP(hipStreamWaitEvent(st, (hipEvent_t)net->u8_h2d_ev[slot], 0));
  const unsigned char* const dev = slot ? net->u8_gpu2 : net->u8_gpu;
  if (net->u8_src_w > 0)
  {
    if (dk_image_resize_u8_to_chw(dev, net->u8_src_w, net->u8_src_h, net->u8_row_step, net->input_state_gpu, net->batch,
            net->w, net->h, net->c, net->u8_swap_rb, st))
      error("dk_image_resize_u8_to_chw failed");
  }
  else if (dk_image_u8_to_chw(dev, net->input_state_gpu, net->batch, net->w, net->h, net->c, net->u8_row_step, st))
    error("dk_image_u8_to_chw failed");
  CHECK_HIP(hipEventRecord((hipEvent_t)net->u8_conv_ev[slot], st));
  net->u8_conv_pending[slot] = 1;
  net->u8_staged = 0;
  NetworkPredictDevice(net, nullptr);
}

void DkNetworkPredictU8(Network* net, const unsigned char* frames_hwc, size_t row_step)
{
  DkNetworkStageU8(net, frames_hwc, row_step);
  DkNetworkPredictStaged(net);
}

// ---------------------------------------------------------------------------
// detections (host C++, as in the reference)
// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// device-side candidate extraction: with DkSetPullHeads(0) the decoded
// heads stay in HBM; GetNetworkBoxes* compacts the predictors above the threshold on the
// device (dk_yolo_compact), copies the few records out, orders them as the reference's
// scan does (layer, anchor, cell) and applies the reference's box arithmetic on the host:
// same Detection arrays, ~100 KB instead of the whole head across PCIe.
// ---------------------------------------------------------------------------
// nms <= 0: raw candidates (class scores as decoded); nms > 0: NmsSort applied on the device
// (dk_nms_records), the records then hold boxes and post-NMS probabilities
static void ensure_candidates(Network* net, float thresh, float nms = 0.f)
{
  if (net->cand_valid && net->cand_seq == net->predict_seq && net->cand_thresh == thresh && net->cand_nms == nms)
    return;
  net->cand_nms = nms;
  net->cand_nms_done = 0;
  int classes = -1;
  bool uniform = true;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == YOLO)
    {
      if (classes < 0)
        classes = net->layers[i].classes;
      else if (classes != net->layers[i].classes)
        uniform = false;
    }
  hipStream_t st = get_cuda_stream();
  net->cand_fallback = 0;
  net->cand_count = 0;
  if (classes < 0)
  {
    net->cand_valid = 1; net->cand_seq = net->predict_seq; net->cand_thresh = thresh;
    return;
  }
  const int rec = 3 + 5 + classes;
  if (!net->cand_gpu || net->cand_rec != rec)
  {
    net->cand_cap = 65536;
    net->cand_rec = rec;
    cuda_free(net->cand_gpu);
    if (net->cand_host) cuda_free_host(net->cand_host);
    free(net->cand_order);
    net->cand_gpu = cuda_make_array(nullptr, (size_t)net->cand_cap * rec);
    net->cand_host = cuda_make_array_pinned(nullptr, (size_t)net->cand_cap * rec);
    net->cand_order = (int*)xcalloc(net->cand_cap, sizeof(int));
    if (!net->cand_counter_gpu)
      net->cand_counter_gpu = cuda_make_int_array(1);
  }
  int count = 0;
  if (uniform)
  {
    CHECK_HIP(hipMemsetAsync(net->cand_counter_gpu, 0, sizeof(int), st));
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type == YOLO)
        if (dk_yolo_compact(l->output_gpu, l->batch, l->w, l->h, l->n, l->classes, thresh, i, net->cand_gpu,
                net->cand_counter_gpu, net->cand_cap, st))
          error("dk_yolo_compact failed");
    }
    CHECK_HIP(hipMemcpyAsync(&count, net->cand_counter_gpu, sizeof(int), hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    if (nms > 0 && count > 0 && count <= net->cand_cap)
    {
      // per-layer decode parameters, once
      if (!net->nms_heads_gpu)
      {
        std::vector<DkYoloHead> hh(net->n);
        memset(hh.data(), 0, sizeof(DkYoloHead) * net->n);
        for (int i = 0; i < net->n; ++i)
        {
          const layer* l = &net->layers[i];
          if (l->type != YOLO)
            continue;
          if (l->n > 8)
            error("device NMS: more than 8 anchors per yolo layer");
          hh[i].lw = l->w; hh[i].lh = l->h;
          for (int n = 0; n < l->n; ++n)
          {
            hh[i].anchor_w[n] = l->biases[2 * l->mask[n]];
            hh[i].anchor_h[n] = l->biases[2 * l->mask[n] + 1];
          }
        }
        CHECK_HIP(hipMalloc(&net->nms_heads_gpu, sizeof(DkYoloHead) * net->n + sizeof(int)));
        CHECK_HIP(hipMemcpy(net->nms_heads_gpu, hh.data(), sizeof(DkYoloHead) * net->n, hipMemcpyHostToDevice));
      }
      int* overflow_gpu = (int*)((char*)net->nms_heads_gpu + sizeof(DkYoloHead) * net->n);
      CHECK_HIP(hipMemsetAsync(overflow_gpu, 0, sizeof(int), st));
      const layer* last = &net->layers[net->n - 1];
      if (dk_nms_records(net->cand_gpu, count, classes, (const DkYoloHead*)net->nms_heads_gpu, net->w, net->h, net->batch,
              thresh, nms, (int)last->nms_kind, last->beta_nms, overflow_gpu, st))
        error("dk_nms_records failed");
      int overflow = 0;
      CHECK_HIP(hipMemcpyAsync(&overflow, overflow_gpu, sizeof(int), hipMemcpyDeviceToHost, st));
      CHECK_HIP(hipStreamSynchronize(st));
      if (overflow)
        error("device NMS: more than 4096 live candidates for one (image, class); use GetNetworkBoxes + NmsSort");
      net->cand_nms_done = 1;
    }
  }
  if (!uniform || count > net->cand_cap)
  {
    // too many candidates (or mixed heads): fall back to pulling the whole heads
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type == YOLO)
        cuda_pull_array(l->output_gpu, l->output, (size_t)l->batch * l->outputs);
    }
    net->cand_fallback = 1;
  }
  else if (count > 0)
  {
    CHECK_HIP(hipMemcpyAsync(net->cand_host, net->cand_gpu, (size_t)count * rec * sizeof(float),
        hipMemcpyDeviceToHost, st));
    CHECK_HIP(hipStreamSynchronize(st));
    for (int k = 0; k < count; ++k) net->cand_order[k] = k;
    const float* h = net->cand_host;
    auto key = [h, rec](int k, int f) { int v; memcpy(&v, h + (size_t)k * rec + f, sizeof(int)); return v; };
    std::sort(net->cand_order, net->cand_order + count, [&](int a, int b) {
      if (key(a, 1) != key(b, 1)) return key(a, 1) < key(b, 1);   // image
      if (key(a, 0) != key(b, 0)) return key(a, 0) < key(b, 0);   // layer
      return key(a, 2) < key(b, 2);                               // anchor-major location
    });
  }
  net->cand_count = (net->cand_fallback) ? 0 : count;
  net->cand_valid = 1;
  net->cand_seq = net->predict_seq;
  net->cand_thresh = thresh;
}

// Detections of image b from the candidate records; dets may be NULL (count only).
static int candidates_to_dets(Network* net, int b, float thresh, Detection* dets, int* ids, int max_dets)
{
  const int rec = net->cand_rec;
  int out = 0;
  for (int k = 0; k < net->cand_count; ++k)
  {
    const float* r = net->cand_host + (size_t)net->cand_order[k] * rec;
    int tag, img, loc;
    memcpy(&tag, r + 0, sizeof(int));
    memcpy(&img, r + 1, sizeof(int));
    memcpy(&loc, r + 2, sizeof(int));
    if (img != b)
      continue;
    if (dets && out < max_dets)
    {
      const layer* l = &net->layers[tag];
      const float* v = r + 3;
      const int wh = l->w * l->h;
      const int n = loc / wh, i = loc - n * wh;
      const int col = i % l->w, row = i / l->w;
      const int a = l->mask[n];
      const float objectness = v[4];
      Box bx;
      if (net->cand_nms_done)
      {
        // the device already decoded the box and applied threshold + NMS to the probabilities
        bx.x = v[0]; bx.y = v[1]; bx.w = v[2]; bx.h = v[3];
        for (int j = 0; j < l->classes; ++j) dets[out].prob[j] = v[5 + j];
      }
      else
      {
        // GetYoloBox, yolo_layer.cpp:139-148 -- the same float operations as DkGetYoloDetectionsBatch
        bx.x = (col + v[0]) / l->w;
        bx.y = (row + v[1]) / l->h;
        bx.w = expf(v[2]) * l->biases[2 * a] / net->w;
        bx.h = expf(v[3]) * l->biases[2 * a + 1] / net->h;
        for (int j = 0; j < l->classes; ++j)
        {
          const float prob = objectness * v[5 + j];
          dets[out].prob[j] = (prob > thresh) ? prob : 0;
        }
      }
      dets[out].bbox = bx;
      dets[out].objectness = objectness;
      dets[out].classes = l->classes;
      if (ids)
      {
        ids[4 * out + 0] = tag;
        ids[4 * out + 1] = n;
        ids[4 * out + 2] = row;
        ids[4 * out + 3] = col;
      }
    }
    ++out;
  }
  return out;
}

// (the device-side candidate compaction knows the [yolo] record layout only: nets with [Gaussian_yolo]
// heads always pull their heads)
static bool has_gaussian_heads(const Network* net)
{
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == GAUSSIAN_YOLO)
      return true;
  return false;
}
static bool heads_on_device(Network* net) { return !net_pull_heads(net) && net->gpu_index >= 0 && !has_gaussian_heads(net); }

static int num_detections(Network* net, int b, float thresh)
{
  if (heads_on_device(net))
  {
    ensure_candidates(net, thresh);
    if (!net->cand_fallback)
      return candidates_to_dets(net, b, thresh, nullptr, nullptr, 0);
  }
  int s = 0;
  for (int i = 0; i < net->n; ++i)
  {
    if (net->layers[i].type == YOLO)
      s += DkYoloNumDetectionsBatch(&net->layers[i], b, thresh);
    else if (net->layers[i].type == GAUSSIAN_YOLO)
      s += DkGaussianYoloNumDetectionsBatch(&net->layers[i], b, thresh);
  }
  return s;
}

static Detection* make_boxes(Network* net, int b, float thresh, int* num)
{
  layer* l = &net->layers[net->n - 1];
  const int num_boxes = num_detections(net, b, thresh);
  if (num)
    *num = num_boxes;
  Detection* dets = (Detection*)xcalloc(num_boxes, sizeof(Detection));
  for (int i = 0; i < num_boxes; ++i)
  {
    dets[i].prob = (float*)xcalloc(l->classes, sizeof(float));
    if (l->type == GAUSSIAN_YOLO)   // tx, ty, tw, th uncertainty (network.cpp:449-452)
      dets[i].uc = (float*)xcalloc(4, sizeof(float));
  }
  return dets;
}

Detection* MakeNetworkBoxes(Network* net, float thresh, int* num)
{
  return make_boxes(net, 0, thresh, num);
}

Detection* GetNetworkBoxesBatch(Network* net, int b, float thresh, int* num)
{
  if (b < 0 || b >= net->batch)
    error("GetNetworkBoxesBatch: batch index out of range");
  Detection* dets = make_boxes(net, b, thresh, num);
  if (heads_on_device(net) && !net->cand_fallback)
  {
    candidates_to_dets(net, b, thresh, dets, nullptr, num ? *num : 0x7fffffff);
    return dets;
  }
  Detection* d = dets;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == YOLO)
      d += DkGetYoloDetectionsBatch(l, b, net->w, net->h, thresh, d, nullptr);
    else if (l->type == GAUSSIAN_YOLO)
      d += DkGetGaussianYoloDetectionsBatch(l, b, net->w, net->h, thresh, d, nullptr);
  }
  return dets;
}

Detection* GetNetworkBoxes(Network* net, float thresh, int* num)
{
  return GetNetworkBoxesBatch(net, 0, thresh, num);
}

// GetNetworkBoxes + NmsSort in one call with the suppression done on the device;
// needs the heads in HBM (DkSetPullHeads(0) / DkNetSetPullHeads(net, 0)).  The array holds every
// candidate of image b in scan order with its post-NMS probabilities (NmsSort's final re-ordering by
// the last class is not reproduced).
Detection* DkGetNetworkBoxesNms(Network* net, int b, float thresh, float nms, int* num)
{
  if (b < 0 || b >= net->batch)
    error("DkGetNetworkBoxesNms: batch index out of range");
  if (!heads_on_device(net))
    error("DkGetNetworkBoxesNms: the heads must stay on the device (DkSetPullHeads(0))");
  ensure_candidates(net, thresh, nms);
  if (net->cand_fallback)
  {
    Detection* dets = GetNetworkBoxesBatch(net, b, thresh, num);
    const layer* last = &net->layers[net->n - 1];
    NmsSort(dets, *num, last->classes, nms, last->nms_kind, last->beta_nms);
    return dets;
  }
  const int n = candidates_to_dets(net, b, thresh, nullptr, nullptr, 0);
  if (num)
    *num = n;
  const int classes = net->cand_rec - 8;
  Detection* dets = (Detection*)xcalloc(n > 0 ? n : 1, sizeof(Detection));
  for (int i = 0; i < n; ++i) dets[i].prob = (float*)xcalloc(classes, sizeof(float));
  candidates_to_dets(net, b, thresh, dets, nullptr, n);
  return dets;
}

// flat form: out[k] = [x, y, w, h, objectness, prob[classes]] post-NMS, ids[k] = [layer, anchor, row, col]
int DkGetBoxesBatchNms(Network* net, int b, float thresh, float nms, float* out, int* ids, int max_dets)
{
  if (b < 0 || b >= net->batch || !heads_on_device(net))
    return -1;
  ensure_candidates(net, thresh, nms);
  if (net->cand_fallback)
    return -2;
  const int num = candidates_to_dets(net, b, thresh, nullptr, nullptr, 0);
  if (num == 0)
    return 0;
  const int classes = net->cand_rec - 8;
  Detection* dets = (Detection*)xcalloc(num, sizeof(Detection));
  for (int k = 0; k < num; ++k) dets[k].prob = (float*)xcalloc(classes, sizeof(float));
  int* lid = (int*)xcalloc((size_t)num * 4, sizeof(int));
  candidates_to_dets(net, b, thresh, dets, lid, num);
  const int rec = 5 + classes;
  for (int k = 0; k < num && k < max_dets; ++k)
  {
    float* o = out + (size_t)k * rec;
    o[0] = dets[k].bbox.x; o[1] = dets[k].bbox.y; o[2] = dets[k].bbox.w; o[3] = dets[k].bbox.h;
    o[4] = dets[k].objectness;
    memcpy(o + 5, dets[k].prob, classes * sizeof(float));
    if (ids)
      memcpy(ids + (size_t)k * 4, lid + (size_t)k * 4, 4 * sizeof(int));
  }
  FreeDetections(dets, num);
  free(lid);
  return num;
}

void FreeDetections(Detection* dets, int n)
{
  for (int i = 0; i < n; ++i)
  {
    free(dets[i].prob);
    free(dets[i].uc);
    free(dets[i].mask);
  }
  free(dets);
}

// Detection2Json (src/network.cpp:518-592; yolo_core.h:635): the JSON record the reference's server / file output
// writes for one frame.  Same text byte for byte (printf %f fields, separators, the fixed 0.005 threshold, classes
// whose name starts with "dont_show" skipped); built in a std::string, returned as a malloc()ed C string the
// caller frees, like the reference's buffer.
char* Detection2Json(Detection* dets, int nboxes, int classes, char** names, long long int frame_id, char const* filename)
{
  const float thresh = 0.005f;
  std::string out;
  char head[64];
  snprintf(head, sizeof(head), "{\n \"frame_id\":%lld, \n", frame_id);
  out += head;
  if (filename)
  {
    out += " \"filename\":\"";
    out += filename;
    out += "\", \n";
  }
  out += " \"objects\": [ \n";
  bool first = true;
  for (int i = 0; i < nboxes; ++i)
    for (int j = 0; j < classes; ++j)
    {
      if (!(dets[i].prob[j] > thresh) || strncmp(names[j], "dont_show", 9) == 0)
        continue;
      if (!first)
        out += ", \n";
      first = false;
      char num[400];
      snprintf(num, sizeof(num), "  {\"class_id\":%d, \"name\":\"", j);
      out += num;
      out += names[j];
      snprintf(num, sizeof(num),
          "\", \"relative_coordinates\":{\"center_x\":%f, \"center_y\":%f, \"width\":%f, \"height\":%f}, \"confidence\":%f}",
          dets[i].bbox.x, dets[i].bbox.y, dets[i].bbox.w, dets[i].bbox.h, dets[i].prob[j]);
      out += num;
    }
  out += "\n ] \n}";
  char* buf = (char*)malloc(out.size() + 1);
  if (!buf)
    return nullptr;
  memcpy(buf, out.c_str(), out.size() + 1);
  return buf;
}

int DkGetBoxesBatch(Network* net, int b, float thresh, float* out, int* ids, int max_dets)
{
  if (b < 0 || b >= net->batch)
    return -1;
  if (heads_on_device(net))
  {
    ensure_candidates(net, thresh);
    if (!net->cand_fallback)
    {
      const int num = candidates_to_dets(net, b, thresh, nullptr, nullptr, 0);
      if (num == 0)
        return 0;
      const int classes = net->cand_rec - 8;
      Detection* dets = (Detection*)xcalloc(num, sizeof(Detection));
      for (int k = 0; k < num; ++k) dets[k].prob = (float*)xcalloc(classes, sizeof(float));
      int* lid = (int*)xcalloc((size_t)num * 4, sizeof(int));
      candidates_to_dets(net, b, thresh, dets, lid, num);
      const int rec = 5 + classes;
      for (int k = 0; k < num && k < max_dets; ++k)
      {
        float* o = out + (size_t)k * rec;
        o[0] = dets[k].bbox.x; o[1] = dets[k].bbox.y; o[2] = dets[k].bbox.w; o[3] = dets[k].bbox.h;
        o[4] = dets[k].objectness;
        memcpy(o + 5, dets[k].prob, classes * sizeof(float));
        if (ids)
          memcpy(ids + (size_t)k * 4, lid + (size_t)k * 4, 4 * sizeof(int));
      }
      FreeDetections(dets, num);
      free(lid);
      return num;
    }
  }
  int total = 0;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type != YOLO && l->type != GAUSSIAN_YOLO)
      continue;
    const bool gauss = l->type == GAUSSIAN_YOLO;
    const int num = gauss ? DkGaussianYoloNumDetectionsBatch(l, b, thresh) : DkYoloNumDetectionsBatch(l, b, thresh);
    if (num == 0)
      continue;
    Detection* dets = (Detection*)xcalloc(num, sizeof(Detection));
    for (int k = 0; k < num; ++k) dets[k].prob = (float*)xcalloc(l->classes, sizeof(float));
    int* lid = (int*)xcalloc((size_t)num * 4, sizeof(int));
    const int got = gauss ? DkGetGaussianYoloDetectionsBatch(l, b, net->w, net->h, thresh, dets, lid)
                          : DkGetYoloDetectionsBatch(l, b, net->w, net->h, thresh, dets, lid);
    const int rec = 5 + l->classes;
    for (int k = 0; k < got; ++k)
    {
      if (total + k >= max_dets)
        break;
      float* o = out + (size_t)(total + k) * rec;
      o[0] = dets[k].bbox.x; o[1] = dets[k].bbox.y; o[2] = dets[k].bbox.w; o[3] = dets[k].bbox.h;
      o[4] = dets[k].objectness;
      memcpy(o + 5, dets[k].prob, l->classes * sizeof(float));
      if (ids)
      {
        int* q = ids + (size_t)(total + k) * 4;
        q[0] = i; q[1] = lid[4 * k + 1]; q[2] = lid[4 * k + 2]; q[3] = lid[4 * k + 3];
      }
    }
    total += got;
    FreeDetections(dets, num);
    free(lid);
  }
  return total;
}

// ---------------------------------------------------------------------------
// lifetime + FFI helpers
// ---------------------------------------------------------------------------
void FreeNetwork(Network* net)
{
  if (!net || !net->layers)
    return;
  if (net->gpu_index >= 0)
  {
    cuda_set_device(net->gpu_index);
    NetworkSync(net);
  }
  DkInvalidateGraph(net);
  float* last_out = net->layers[net->n - 1].output;
  (void)last_out;
  if (net->grad_bucket)  // gradients live in caller-owned memory
    for (int i = 0; i < net->n; ++i)
    {
      layer* l = &net->layers[i];
      if (l->type == CONVOLUTIONAL || l->type == BATCHNORM)
        l->weight_updates_gpu = l->bias_updates_gpu = l->scale_updates_gpu = nullptr;
    }
  for (int i = 0; i < net->n; ++i) free_layer(&net->layers[i], false);
  if (net->gpu_index >= 0)
  {
    DkInvalidateSgdPlan(net);
    DkFreeDpState(net);
    if (net->fwd_done_ev) (void)hipEventDestroy((hipEvent_t)net->fwd_done_ev);
    if (net->copy_done_ev) (void)hipEventDestroy((hipEvent_t)net->copy_done_ev);
    for (int k = 0; k < 8; ++k)
      if (net->head_ev[k]) (void)hipEventDestroy((hipEvent_t)net->head_ev[k]);
  }
  free(net->layers);
  free(net->steps);
  free(net->scales);
  free(net->input_gpu);
  free(net->truth_gpu);
  if (net->gpu_index >= 0)
  {
    cuda_free(net->input_state_gpu);
    if (net->input_pinned_cpu_flag)
      cuda_free_host(net->input_pinned_cpu);
    cuda_free(net->workspace);
    cuda_free(net->wt_scratch_gpu);
    cuda_free(net->wino_scratch_gpu);
    DkFreeTrainPrep(net);
    DkFreeWgradStream(net);
    cuda_free(net->delta_arena_gpu);
    cuda_free(net->cand_gpu);
    cuda_free((float*)net->cand_counter_gpu);
    if (net->nms_heads_gpu) (void)hipFree(net->nms_heads_gpu);
    if (net->cand_host) cuda_free_host(net->cand_host);
    free_float_stage(net);
    if (net->f32_h2d_ev) (void)hipEventDestroy((hipEvent_t)net->f32_h2d_ev);
    if (net->f32_copy_ev) (void)hipEventDestroy((hipEvent_t)net->f32_copy_ev);
    if (net->stage_stream) (void)hipStreamDestroy((hipStream_t)net->stage_stream);
    if (net->u8_gpu) (void)hipFree(net->u8_gpu);
    if (net->u8_pinned) (void)hipHostFree(net->u8_pinned);
    if (net->u8_gpu2) (void)hipFree(net->u8_gpu2);
    if (net->u8_pinned2) (void)hipHostFree(net->u8_pinned2);
    for (int i = 0; i < 2; ++i)
    {
      if (net->u8_h2d_ev[i]) (void)hipEventDestroy((hipEvent_t)net->u8_h2d_ev[i]);
      if (net->u8_conv_ev[i]) (void)hipEventDestroy((hipEvent_t)net->u8_conv_ev[i]);
    }
  }
  free(net->cand_order);
  memset(net, 0, sizeof(*net));
}

Network* DkNetworkCreate(void) { return (Network*)xcalloc(1, sizeof(Network)); }

void DkNetworkDestroy(Network* net)
{
  if (!net)
    return;
  FreeNetwork(net);
  free(net);
}

float* DkNetworkInputGpu(Network* net) { return net->input_state_gpu; }

void DkNetworkInfo(Network* net, int* o)
{
  o[0] = net->n; o[1] = net->batch; o[2] = net->w; o[3] = net->h; o[4] = net->c;
  o[5] = net->inputs; o[6] = net->outputs; o[7] = net->gpu_index;
}

void DkLayerInfo(Network* net, int i, int* o)
{
  layer* l = &net->layers[i];
  o[0] = l->type; o[1] = l->batch; o[2] = l->outputs; o[3] = l->out_c; o[4] = l->out_h;
  o[5] = l->out_w; o[6] = l->n; o[7] = l->size; o[8] = l->stride; o[9] = l->pad; o[10] = l->c;
  o[11] = l->h; o[12] = l->w; o[13] = l->activation; o[14] = l->batch_normalize;
  o[15] = l->nweights; o[16] = l->groups; o[17] = l->inputs; o[18] = l->classes; o[19] = l->total;
  o[20] = l->index; o[21] = l->dilation; o[22] = l->stride_x; o[23] = l->stride_y;
}

float DkLayerBflops(Network* net, int i) { return net->layers[i].bflops; }

float* DkLayerOutputGpu(Network* net, int i) { return net->layers[i].output_gpu; }

int DkLayerOutput(Network* net, int i, float* dst, size_t n)
{
  layer* l = &net->layers[i];
  const size_t total = (size_t)l->batch * l->outputs;
  if (!l->output_gpu || n < total)
    return 1;
  NetworkSync(net);
  if (l->out_view)
  {
    // written in place as a channel slice of the consuming route's buffer
    const size_t hw = (size_t)l->out_h * l->out_w;
    CHECK_HIP(hipMemcpy2D(dst, (size_t)l->outputs * sizeof(float), l->out_view,
        (size_t)l->out_view_ctot * hw * sizeof(float), (size_t)l->outputs * sizeof(float), l->batch,
        hipMemcpyDeviceToHost));
    return 0;
  }
  cuda_pull_array(DkLayerOut(l), dst, total);
  return 0;
}

float* DkLayerHostPtr(Network* net, int i, int which)
{
  layer* l = &net->layers[i];
  switch (which)
  {
    case 1: return l->weights;
    case 2: return l->biases;
    case 3: return l->scales;
    case 4: return l->rolling_mean;
    case 5: return l->rolling_variance;
  }
  return nullptr;
}

extern "C" LIB_API int DkLayerConvCfg(Network* net, int i) { return net->layers[i].conv_cfg; }
extern "C" LIB_API int DkLayerFused(Network* net, int i)
{
  layer* l = &net->layers[i];
  return l->type == SHORTCUT ? l->fused_into_prev : (l->fuse_residual_from >= 0);
}

extern "C" LIB_API layer* DkLayerPtr(Network* net, int i) { return &net->layers[i]; }
