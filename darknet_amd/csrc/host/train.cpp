// train.cpp -- training-mode graph engine: conv forward with batch statistics,
// the backward sweep, the SGD update.  Reference twins (Ravicmoon/darknet src/):
// ForwardConvolutionalLayerGpu + ForwardBatchnormLayerGpu convolutional_kernels.cu
// :471-532 / batchnorm_layer.cpp:268-322; BackwardConvolutionalLayerGpu :555-815;
// UpdateConvolutionalLayerGpu :865-921; BackwardNetworkGpu / UpdateNetworkGpu /
// ForwardBackwardNetworkGpu / TrainNetworkDatumGpu network_kernels.cu:116-293;
// Backward{Maxpool,Route,Shortcut,Upsample,Yolo}LayerGpu.
// Numerics follow the reference's CPU path:
// BN eps / rolling momentum / N-1 variance as on the CPU; the data gradient
// OVERWRITES the previous layer's delta; for BN layers bias_updates is the true
// sum of delta (the CPU reference leaves it 0 -- quirk 3).
#include <chrono>
#include <future>
#include <map>
#include <mutex>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"
#include "dk_internal.h"

extern "C" {
int dk_bn_forward_train(const float*, float*, float*, float*, float*, float*, float*, float*, float*,
    const float*, const float*, int, int, int, int, int, void*);
}

static DkConvDesc conv_desc_of(const layer* l, int activation)
{
  DkConvDesc d;
  d.batch = l->batch; d.c = l->c; d.h = l->h; d.w = l->w; d.n = l->n; d.groups = l->groups;
  d.size = l->size; d.stride_x = l->stride_x; d.stride_y = l->stride_y;
  d.dilation = l->dilation; d.pad = l->pad; d.activation = activation;
  return d;
}

// ---- first-step kernel selection for the three GEMMs of a training step ----------------------
// The inference plan times every tile configuration per layer at load (network.cpp); a train-mode
// network does the same lazily, on the layer's OWN tensors the first time the layer runs: forward
// (raw convolution), data gradient, weight gradient.  Results are kept per (device, shape) for the
// whole process -- replicas of one model must run the same kernels, or their sums stop being bitwise
// comparable -- and the choice per layer (l->train_plan[k] = configuration + 2).  DK_AUTOTUNE=0 / DkSetAutotune(0) or
// DK_TRAIN_TUNE=0 keep the heuristics; DK_TRAIN_WINO=0 keeps the Winograd kernel out of training.
namespace
{
struct TrainTune
{
  std::mutex mu;
  std::map<std::vector<int>, int> best;
  double seconds = 0;
};
TrainTune g_tune;

bool train_tune_on()
{
  static const int env = getenv("DK_TRAIN_TUNE") ? atoi(getenv("DK_TRAIN_TUNE")) : -1;
  if (env >= 0)
    return env != 0;
  const char* e = getenv("DK_AUTOTUNE");
  return e ? atoi(e) != 0 : g_dk_autotune != 0;
}
bool train_wino_on()
{
  static const int env = getenv("DK_TRAIN_WINO") ? atoi(getenv("DK_TRAIN_WINO")) : 1;
  return env != 0;
}

std::vector<int> tune_key(int kind, const DkConvDesc& d)
{
  return {cuda_get_device(), kind, d.batch, d.c, d.h, d.w, d.n, d.groups, d.size, d.stride_x, d.stride_y, d.dilation,
      d.pad};
}

// times launch(cfg) for every candidate (minimum of three timings of two launches, as the inference
// tuner) and returns the fastest; `heur` wins when it is within 3 % of the fastest (timing noise)
template <class Launch>
int time_candidates(const std::vector<int>& cands, int heur, Launch launch, hipStream_t st)
{
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  int best = heur;
  float best_ms = 1e30f, heur_ms = -1.f;
  for (int c : cands)
  {
    launch(c);
    float ms = 1e30f;
    for (int rep = 0; rep < 3; ++rep)
    {
      CHECK_HIP(hipEventRecord(e0, st));
      launch(c);
      launch(c);
      CHECK_HIP(hipEventRecord(e1, st));
      CHECK_HIP(hipEventSynchronize(e1));
      float t = 0;
      CHECK_HIP(hipEventElapsedTime(&t, e0, e1));
      if (t < ms)
        ms = t;
    }
    if (c == heur)
      heur_ms = ms;
    if (ms < best_ms)
    {
      best_ms = ms;
      best = c;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (heur_ms > 0 && best != heur && heur_ms <= best_ms * 1.03f)
    best = heur;
  return best;
}

// the choice for (kind, shape): cached, else timed now through `launch`
template <class Launch>
int train_choice(layer* l, int kind, const DkConvDesc& d, const std::vector<int>& cands, int heur,
    Launch launch, hipStream_t st)
{
  if (l->train_plan[kind])
    return l->train_plan[kind] - 2;
  int choice = heur;
  if (train_tune_on() && cands.size() > 1)
  {
    const std::vector<int> key = tune_key(kind, d);
    bool known = false;
    {
      std::lock_guard<std::mutex> lk(g_tune.mu);
      auto it = g_tune.best.find(key);
      if (it != g_tune.best.end())
      {
        choice = it->second;
        known = true;
      }
    }
    if (!known)
    {
      const auto t0 = std::chrono::steady_clock::now();
      choice = time_candidates(cands, heur, launch, st);
      std::lock_guard<std::mutex> lk(g_tune.mu);
      g_tune.best[key] = choice;
      g_tune.seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
  }
  l->train_plan[kind] = choice + 2;
  return choice;
}

// every forward configuration that can run convolution `d` (Winograd only with a filter scratch)
std::vector<int> forward_candidates(const DkConvDesc& d, bool wino_ok)
{
  std::vector<int> v;
  const int n = dk_conv_num_configs();
  for (int c = 0; c < n; ++c)
    if (dk_conv_config_applicable(&d, c) && (!dk_conv_config_is_wino(c) || wino_ok))
      v.push_back(c);
  return v;
}

// raw convolution y = w * x through configuration cfg; Winograd takes freshly transformed filters
// from the network's scratch (stream-ordered: one scratch serves every layer)
void train_conv(Network* net, const DkConvDesc& d, const float* x, const float* w, float* y, int cfg, hipStream_t st,
    const char* what, const float* prepared_u = nullptr)
{
  const float* u = prepared_u;
  if (!u && cfg >= 0 && dk_conv_config_is_wino(cfg))
  {
    if (dk_conv_wino_transform_weights(&d, w, net->wino_scratch_gpu, st))
      error(what);
    u = net->wino_scratch_gpu;
  }
  if (dk_conv_forward_cfg(&d, x, w, nullptr, y, nullptr, nullptr, st, cfg, 0, nullptr, u))
    error(what);
}

// DK_TRAIN_LAYERS=1 (diagnostics, tools/train_layers.py): every GEMM of the step is bracketed by events and
// DkTrainLayerReport() returns "layer kind cfg gflop ms" lines for the steps since the last call
struct LayerRec
{
  int layer, kind, cfg;
  double gflop;
  hipEvent_t e0, e1;
};
std::vector<LayerRec> g_layer_recs;
std::mutex g_layer_mu;
bool train_layers_on()
{
  static const int v = getenv("DK_TRAIN_LAYERS") ? atoi(getenv("DK_TRAIN_LAYERS")) : 0;
  return v != 0;
}
struct LayerScope
{
  LayerRec r;
  hipStream_t st;
  bool on;
  LayerScope(const layer* l, int index, int kind, int cfg, hipStream_t s) : st(s), on(train_layers_on())
  {
    if (!on)
      return;
    r.layer = index; r.kind = kind; r.cfg = cfg;
    r.gflop = 2.0 * l->n * (double)(l->c / l->groups) * l->size * l->size * l->out_h * l->out_w * l->batch / 1e9;
    CHECK_HIP(hipEventCreate(&r.e0));
    CHECK_HIP(hipEventCreate(&r.e1));
    CHECK_HIP(hipEventRecord(r.e0, st));
  }
  ~LayerScope()
  {
    if (!on)
      return;
    CHECK_HIP(hipEventRecord(r.e1, st));
    std::lock_guard<std::mutex> lk(g_layer_mu);
    g_layer_recs.push_back(r);
  }
};
}  // namespace

extern "C" LIB_API double DkTrainTuneSeconds()
{
  std::lock_guard<std::mutex> lk(g_tune.mu);
  return g_tune.seconds;
}
// text report of the GEMM timings recorded since the last call (needs DK_TRAIN_LAYERS=1); returns the number of
// bytes written (0: nothing recorded).  Synchronises the stream.
extern "C" LIB_API int DkTrainLayerReport(char* out, int cap)
{
  CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
  std::lock_guard<std::mutex> lk(g_layer_mu);
  int n = 0;
  for (LayerRec& r : g_layer_recs)
  {
    float ms = 0;
    CHECK_HIP(hipEventSynchronize(r.e1));
    CHECK_HIP(hipEventElapsedTime(&ms, r.e0, r.e1));
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    if (out && n < cap - 96)
      n += snprintf(out + n, cap - n, "%d %d %d %.4f %.5f\n", r.layer, r.kind, r.cfg, r.gflop, ms);
  }
  g_layer_recs.clear();
  return n;
}
extern "C" LIB_API int DkLayerTrainCfg(Network* net, int i, int kind)
{
  return (i >= 0 && i < net->n && kind >= 0 && kind < 3) ? net->layers[i].train_plan[kind] - 2 : -3;
}

// ---- derived weight tensors, one launch per step -------------------------------------------------------------
// Once the first step has chosen every layer's kernels, the transposed weights of the data gradients and the
// Winograd filters of both convolution passes are known tensors of the step: they are produced by ONE launch at
// the start of the forward pass (weights only change in the update) into per-layer buffers, instead of one or two
// small launches in front of every layer's kernels (~170 per yolov4 step).  DK_TRAIN_PREP=0 keeps the per-layer form.
static bool same3_layer(const layer* l)
{
  return l->size == 3 && l->stride_x == 1 && l->stride_y == 1 && l->pad == 1 && l->dilation == 1 && l->groups == 1;
}
static bool plain1_layer(const layer* l)
{
  return l->size == 1 && l->stride_x == 1 && l->stride_y == 1 && l->pad == 0 && l->groups == 1;
}

void DkFreeTrainPrep(Network* net)
{
  if (net->train_prep)
    dk_train_prep_destroy(net->train_prep);
  net->train_prep = nullptr;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type != CONVOLUTIONAL)
      continue;
    cuda_free(l->train_wt_gpu);
    cuda_free(l->train_u_fwd_gpu);
    cuda_free(l->train_u_dgrad_gpu);
    l->train_wt_gpu = l->train_u_fwd_gpu = l->train_u_dgrad_gpu = nullptr;
  }
}

static void build_train_prep(Network* net)
{
  std::vector<DkPrepTask> tasks;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type != CONVOLUTIONAL)
      continue;
    const int Cg = l->c / l->groups, Mg = l->n / l->groups, ss = l->size * l->size;
    if (l->train_plan[0] && dk_conv_config_is_wino(l->train_plan[0] - 2))
    {
      l->train_u_fwd_gpu = cuda_make_array(nullptr, (size_t)16 * l->n * l->c);
      tasks.push_back({l->weights_gpu, l->train_u_fwd_gpu, l->n, l->c, 9, 3, 0});
    }
    if (i == 0 || !l->train_plan[1])
      continue;   // no data gradient (first layer), or the layer never ran one (below a stopbackward)
    const int cfg = l->train_plan[1] - 2;
    if (same3_layer(l) && cfg >= 0 && dk_conv_config_is_wino(cfg))
    {
      l->train_u_dgrad_gpu = cuda_make_array(nullptr, (size_t)16 * l->n * l->c);
      tasks.push_back({l->weights_gpu, l->train_u_dgrad_gpu, l->n, l->c, 9, 4, 0});
      continue;
    }
    l->train_wt_gpu = cuda_make_array(nullptr, (size_t)l->nweights);
    if (same3_layer(l))
      tasks.push_back({l->weights_gpu, l->train_wt_gpu, Mg, Cg, ss, 1, 0});
    else if (plain1_layer(l))
      tasks.push_back({l->weights_gpu, l->train_wt_gpu, Mg, Cg, ss, 0, 0});
    else
    {
      DkConvDesc d = conv_desc_of(l, (int)LINEAR);
      if (dk_conv_dgrad_tapmajor(&d))
        tasks.push_back({l->weights_gpu, l->train_wt_gpu, Mg, Cg, ss, 2, 0});
      else
        for (int g = 0; g < l->groups; ++g)
          tasks.push_back({l->weights_gpu + (size_t)g * l->nweights / l->groups, l->train_wt_gpu + (size_t)g * l->nweights / l->groups,
              Mg, Cg, ss, 0, 0});
    }
  }
  net->train_prep = dk_train_prep_create((int)tasks.size(), tasks.data());
}

static int g_train_prep = -1;   // -1: DK_TRAIN_PREP (default on)
extern "C" LIB_API void DkSetTrainPrep(int on) { g_train_prep = on; }

void DkTrainPrepRun(Network* net)
{
  static const bool env_on = !(getenv("DK_TRAIN_PREP") && !atoi(getenv("DK_TRAIN_PREP")));
  const bool on = g_train_prep >= 0 ? g_train_prep != 0 : env_on;
  if (!net->train)
    return;
  if (!on)
  {
    if (net->train_prep)
      DkFreeTrainPrep(net);   // switched off: back to the per-layer launches
    ++net->train_steps;
    return;
  }
  // the first step chooses the kernels (which tensors are needed follows from the choice)
  if (!net->train_prep && net->train_steps >= 1 && train_tune_on())
    build_train_prep(net);
  ++net->train_steps;
  if (net->train_prep && dk_train_prep_run(net->train_prep, get_cuda_stream()))
    error("derived-weights launch failed");
}

// ---- second stream for the weight gradients (DK_TRAIN_STREAMS=0: everything on the main stream) ----------------
struct WgradStream
{
  hipStream_t s = nullptr;
  hipEvent_t ready = nullptr, done = nullptr;   // main -> second (inputs final), second -> main (join)
  bool pending = false;
};

static int g_train_streams = -1;   // -1: DK_TRAIN_STREAMS (default on)
extern "C" LIB_API void DkSetTrainStreams(int on) { g_train_streams = on; }
extern "C" LIB_API void DkSetDeterministic(int on) { dk_set_deterministic(on); }

static bool train_streams_on()
{
  static const bool env_on = !(getenv("DK_TRAIN_STREAMS") && !atoi(getenv("DK_TRAIN_STREAMS")));
  return g_train_streams >= 0 ? g_train_streams != 0 : env_on;
}

// the stream this layer's weight gradient goes to, ordered behind everything enqueued on `st` so far
static hipStream_t wgrad_stream_of(Network* net, bool tuned_before, hipStream_t st)
{
  if (!train_streams_on() || !tuned_before)
    return st;
  WgradStream* w = (WgradStream*)net->wgrad_stream;
  if (!w)
  {
    w = new WgradStream();
    // lowest priority: when both streams have workgroups ready, the critical path (main stream) goes first
    int prio_least = 0, prio_greatest = 0;
    CHECK_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    static const bool low = !(getenv("DK_WGRAD_PRIO") && !atoi(getenv("DK_WGRAD_PRIO")));
    CHECK_HIP(hipStreamCreateWithPriority(&w->s, hipStreamNonBlocking, low ? prio_least : prio_greatest));
    CHECK_HIP(hipEventCreateWithFlags(&w->ready, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&w->done, hipEventDisableTiming));
    net->wgrad_stream = w;
  }
  CHECK_HIP(hipEventRecord(w->ready, st));
  CHECK_HIP(hipStreamWaitEvent(w->s, w->ready, 0));
  w->pending = true;
  return w->s;
}

// the main stream waits for every weight gradient enqueued so far
void DkJoinWgradStream(Network* net)
{
  WgradStream* w = (WgradStream*)net->wgrad_stream;
  if (!w || !w->pending)
    return;
  CHECK_HIP(hipEventRecord(w->done, w->s));
  CHECK_HIP(hipStreamWaitEvent(get_cuda_stream(), w->done, 0));
  w->pending = false;
}

void DkFreeWgradStream(Network* net)
{
  WgradStream* w = (WgradStream*)net->wgrad_stream;
  if (!w)
    return;
  (void)hipStreamSynchronize(w->s);
  (void)hipEventDestroy(w->ready);
  (void)hipEventDestroy(w->done);
  (void)hipStreamDestroy(w->s);
  delete w;
  net->wgrad_stream = nullptr;
}

// conv with un-folded batch norm: raw GEMM -> x_gpu, statistics, normalise+scale+bias+act
void ForwardConvTrainGpu(layer* l, NetworkState state)
{
  hipStream_t st = get_cuda_stream();
  const int spatial = l->out_h * l->out_w;
  const size_t total = (size_t)l->batch * l->outputs;
  float* raw = l->x_gpu;
  bool temp = false;
  if (!raw)
  {
    // inference on a train-mode parse without train buffers: borrow the output buffer
    raw = l->output_gpu;
    temp = true;
  }
  DkConvDesc d = conv_desc_of(l, (int)LINEAR);
  const int train = state.train && !temp;
  int cfg = -1;
  if (train)
  {
    Network* net = state.net;
    const bool wino_ok = train_wino_on() && net->wino_scratch_gpu != nullptr;
    const float* x = state.input;
    cfg = train_choice(l, 0, d, forward_candidates(d, wino_ok), dk_conv_pick_config(&d),
        [&](int c) { train_conv(net, d, x, l->weights_gpu, raw, c, st, "ForwardConvolutionalLayerGpu (train, timing) failed"); }, st);
    LayerScope ls(l, state.index, 0, cfg, st);
    train_conv(net, d, x, l->weights_gpu, raw, cfg, st, "ForwardConvolutionalLayerGpu (train) failed",
        (net->train_prep && cfg >= 0 && dk_conv_config_is_wino(cfg)) ? l->train_u_fwd_gpu : nullptr);
  }
  else if (dk_conv_forward_cfg(&d, state.input, l->weights_gpu, nullptr, raw, nullptr, nullptr, st, -1))
    error("ForwardConvolutionalLayerGpu (train) failed");
  // x_norm / pre-activation are not stored: the fused backward recomputes them from x
  if (dk_bn_forward_train(raw, nullptr, nullptr, nullptr, l->output_gpu, l->mean_gpu,
          l->variance_gpu, l->rolling_mean_gpu, l->rolling_variance_gpu, l->scales_gpu,
          l->biases_gpu, l->batch, l->n, spatial, (int)l->activation, train, st))
    error("batch-norm forward failed");
  (void)total;
}

void BackwardConvolutionalLayerGpu(layer* l, NetworkState state)
{
  hipStream_t st = get_cuda_stream();
  Network* net = state.net;
  const int spatial = l->out_h * l->out_w;
  const size_t total = (size_t)l->batch * l->outputs;
  if (l->batch_normalize)
  {
    if (dk_bn_act_backward(l->delta_gpu, l->x_gpu, l->mean_gpu, l->variance_gpu, l->scales_gpu,
            l->biases_gpu, l->mean_delta_gpu, l->variance_delta_gpu, l->scale_updates_gpu,
            l->bias_updates_gpu, l->batch, l->n, spatial, (int)l->activation, st))
      error("activation + batch-norm backward failed");
  }
  else
  {
    if (dk_gradient_array(l->output_gpu, l->activation_input_gpu, l->delta_gpu, total,
            (int)l->activation, st))
      error("activation gradient failed");
    dk_backward_bias(l->bias_updates_gpu, l->delta_gpu, l->batch, l->n, spatial, st);
  }

  DkConvDesc d = conv_desc_of(l, (int)LINEAR);
  {
    // weight gradient: tile shapes 0..3 (conv_wgrad.hip) or its own heuristic (-1); the timing runs
    // accumulate into the transpose scratch (nweights floats fit: it is sized for the largest layer)
    float* scratch_dw = net->wt_scratch_gpu;
    // (kernel timing of this layer -- first step -- happens on the main stream with the second one drained)
    const bool tuned_before = l->train_plan[2] != 0 && (!state.delta || l->train_plan[1] != 0);
    if (!tuned_before)
      DkJoinWgradStream(net);
    // (-1 is the row-staged 3x3 kernel where it applies, 5 its one-workgroup-per-CU split)
    std::vector<int> wcands = {-1, 0, 1, 2, 3};
    if (dk_wgrad3_applicable(&d))
      wcands.push_back(5);
    const int wcfg = train_choice(l, 2, d, scratch_dw ? wcands : std::vector<int>{-1}, -1,
        [&](int c) {
          if (dk_conv_backward_weights_cfg(&d, state.input, l->delta_gpu, scratch_dw, st, c))
            error("weight gradient (timing) failed");
        }, st);
    // Nothing in the backward sweep reads weight_updates, so the weight gradient leaves the critical path: it is
    // launched on the network's second stream behind an event (its inputs -- this layer's delta, the previous
    // layer's output -- are final), and the data gradient, the next layers' batch-norm passes (bandwidth-bound)
    // and their small-grid GEMMs share the chip with it.  backward_range joins the stream when it returns.
    hipStream_t sw = wgrad_stream_of(net, tuned_before, st);
    LayerScope ls(l, state.index, 2, wcfg, sw);
    if (dk_conv_backward_weights_cfg(&d, state.input, l->delta_gpu, l->weight_updates_gpu, sw, wcfg))
      error("weight gradient failed");
  }
  if (state.delta)
  {
    const int Cg = l->c / l->groups, Mg = l->n / l->groups;
    float* wt = net->wt_scratch_gpu;
    const bool same3 = same3_layer(l), plain1 = plain1_layer(l);
    // derived tensors already made by this step's DkTrainPrepRun?
    const bool prepared = net->train_prep && l->train_plan[1] && (l->train_wt_gpu || l->train_u_dgrad_gpu);
    if (prepared)
      wt = l->train_wt_gpu;
    if (same3 || plain1)
    {
      // stride-1 "same" 3x3: the data gradient IS a 3x3/s1/p1 convolution of delta with the transposed,
      // 180-degree-rotated filters; 1x1/s1: a 1x1 convolution with the transposed matrix -> the forward
      // kernels (patch-in-LDS / Winograd / LDS-DMA GEMM where they apply).  Overwrites prev_delta like the
      // gather path.
      if (prepared)
        ;
      else if (same3)
        dk_transpose_weights_flip(l->weights_gpu, wt, Mg, Cg, 3, st);
      else
        dk_transpose_weights(l->weights_gpu, wt, Mg, Cg, 1, st);
      DkConvDesc dd = d;
      dd.c = l->n; dd.h = l->out_h; dd.w = l->out_w; dd.n = l->c;
      const bool wino_ok = train_wino_on() && net->wino_scratch_gpu != nullptr;
      const int cfg = train_choice(l, 1, dd, forward_candidates(dd, wino_ok), dk_conv_pick_config(&dd),
          [&](int c) { train_conv(net, dd, l->delta_gpu, wt, state.delta, c, st, "data gradient (as convolution, timing) failed"); }, st);
      LayerScope ls(l, state.index, 1, cfg, st);
      // (with prepared Winograd filters `wt` is NULL and never read: the kernel takes the filters alone)
      train_conv(net, dd, l->delta_gpu, prepared && !wt ? l->weights_gpu : wt, state.delta, cfg, st,
          "data gradient (as convolution) failed", prepared ? l->train_u_dgrad_gpu : nullptr);
      return;
    }
    // stride-2 layers: parity-class form (tap-major contraction index, only the matching taps are visited)
    const int tapmajor = dk_conv_dgrad_tapmajor(&d) ? 1 : 0;
    if (prepared)
      ;
    else if (tapmajor)
      dk_transpose_weights_tapmajor(l->weights_gpu, wt, Mg, Cg, l->size, st);
    else
      for (int g = 0; g < l->groups; ++g)
        dk_transpose_weights(l->weights_gpu + (size_t)g * l->nweights / l->groups,
            wt + (size_t)g * l->nweights / l->groups, Mg, Cg, l->size, st);
    std::vector<int> cands = {-1};
    for (int c = 0; c < dk_conv_num_gather_configs(); ++c) cands.push_back(c);
    const int cfg = train_choice(l, 1, d, cands, -1,
        [&](int c) {
          if (dk_conv_backward_data_cfg(&d, l->delta_gpu, wt, state.delta, st, c, tapmajor))
            error("data gradient (timing) failed");
        }, st);
    LayerScope ls(l, state.index, 1, cfg, st);
    if (dk_conv_backward_data_cfg(&d, l->delta_gpu, wt, state.delta, st, cfg, tapmajor))
      error("data gradient failed");
  }
}

void UpdateConvolutionalLayerGpu(layer* l, int batch, float learning_rate_init, float momentum,
    float decay, float loss_scale)
{
  hipStream_t st = get_cuda_stream();
  const float lr = learning_rate_init * l->learning_rate_scale;
  if (loss_scale != 1.0f)
  {
    dk_scal(l->nweights, 1.0f / loss_scale, l->weight_updates_gpu, st);
    dk_scal(l->n, 1.0f / loss_scale, l->bias_updates_gpu, st);
    if (l->scale_updates_gpu)
      dk_scal(l->n, 1.0f / loss_scale, l->scale_updates_gpu, st);
  }
  if (l->adam)
  {
    // convolutional_kernels.cu:884-898: the same adam_update_gpu for weights, biases and scales (decay included)
    if (dk_adam_update(l->weights_gpu, l->weight_updates_gpu, l->m_gpu, l->v_gpu, l->B1, l->B2, l->eps, decay, lr,
            l->nweights, batch, l->t, st) ||
        dk_adam_update(l->biases_gpu, l->bias_updates_gpu, l->bias_m_gpu, l->bias_v_gpu, l->B1, l->B2, l->eps, decay,
            lr, l->n, batch, l->t, st) ||
        (l->scales_gpu && dk_adam_update(l->scales_gpu, l->scale_updates_gpu, l->scale_m_gpu, l->scale_v_gpu, l->B1,
                              l->B2, l->eps, decay, lr, l->n, batch, l->t, st)))
      error("dk_adam_update failed");
  }
  else
  {
    dk_sgd_update(l->weights_gpu, l->weight_updates_gpu, l->nweights, batch, lr, momentum, decay, 1, st);
    dk_sgd_update(l->biases_gpu, l->bias_updates_gpu, l->n, batch, lr, momentum, decay, 0, st);
    if (l->scales_gpu)
      dk_sgd_update(l->scales_gpu, l->scale_updates_gpu, l->n, batch, lr, momentum, decay, 0, st);
  }
  // convolutional_kernels.cu:919-920 (`clip=`): weights clamped to [-clip, clip] after either optimizer.  (The
  // reference's reset_nan_and_inf / fix_nan_and_inf scrubs at :881-882 are GPU-only there and absent from the CPU path
  // this library is pinned to; they are not reproduced: a NaN gradient stays visible instead of being zeroed.)
  if (l->clip)
    dk_constrain(l->nweights, l->clip, l->weights_gpu, st);
}

void BackwardMaxpoolLayerGpu(layer* l, NetworkState state)
{
  if (!state.delta)
    return;
  if (dk_deterministic())
  {
    // (float atomics over the overlapping windows of the stride-1 SPP pools would make the step irreproducible)
    if (dk_maxpool_backward_gather(l->delta_gpu, l->indexes_gpu, l->batch, l->c, l->h, l->w, l->out_h, l->out_w, l->size,
            l->stride_x, l->stride_y, l->pad, state.delta, get_cuda_stream()))
      error("maxpool backward failed");
    return;
  }
  dk_maxpool_backward(l->delta_gpu, l->indexes_gpu, (size_t)l->batch * l->outputs, state.delta,
      get_cuda_stream());
}

void BackwardRouteLayerGpu(layer* l, NetworkState state)
{
  int offset = 0;
  for (int i = 0; i < l->n; ++i)
  {
    layer* src = &state.net->layers[l->input_layers[i]];
    const int input_size = l->input_sizes[i];
    if (src->delta_gpu)
      dk_route_backward(l->delta_gpu, l->outputs, offset, input_size, l->groups, l->group_id,
          l->batch, src->delta_gpu, get_cuda_stream());
    offset += input_size / l->groups;
  }
}

void BackwardShortcutLayerGpu(layer* l, NetworkState state)
{
  hipStream_t st = get_cuda_stream();
  const size_t total = (size_t)l->batch * l->outputs;
  dk_gradient_array(l->output_gpu, nullptr, l->delta_gpu, total, (int)l->activation, st);
  layer* from = &state.net->layers[l->index];
  dk_shortcut_backward(l->delta_gpu, total, state.delta, from->delta_gpu, st);
}

void BackwardUpsampleLayerGpu(layer* l, NetworkState state)
{
  if (!state.delta)
    return;
  dk_upsample_backward(l->delta_gpu, l->w, l->h, l->c, l->batch, l->stride, l->scale, state.delta,
      get_cuda_stream());
}

static void DkYoloLossJoin(layer* l);

void BackwardYoloLayerGpu(layer* l, NetworkState state)
{
  DkYoloLossJoin(l);
  // axpy_ongpu(batch*inputs, loss_scale, delta_gpu, state.delta), yolo_layer.cpp:884-888
  dk_axpy((size_t)l->batch * l->inputs, state.net->loss_scale, l->delta_gpu, state.delta,
      get_cuda_stream());
}

void DkSetYoloDelta(Network* net, int i, float* host_delta)
{
  if (i < 0 || i >= net->n || net->layers[i].type != YOLO)
    error("DkSetYoloDelta: not a yolo layer");
  net->layers[i].injected_delta = host_delta;
}

void DkSetMaxIter(Network* net, int max_iter) { net->max_iter = max_iter; }

extern "C" float DkYoloLossHost(const layer* l, int net_w, int net_h, float* out,
    const float* truth, float* delta);

// The yolo loss lives on the host, as in the reference (src/yolo_layer.cpp:861-881:
// pull the decoded output, compute delta and cost, push the delta).  The reference
// does this synchronously inside the forward pass; here the head is copied out on
// the copy stream and the loss runs on host threads WHILE the GPU computes the
// remaining layers; the delta is pushed when the backward pass reaches the layer
// (DkYoloLossJoin).  Same values, no GPU idle time.
static bool train_timing()
{
  static int v = -1;
  if (v < 0)
  {
    const char* e = getenv("DK_TRAIN_TIMING");
    v = (e && atoi(e)) ? 1 : 0;
  }
  return v == 1;
}
static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct LossTask
{
  std::future<float> fut;
  hipEvent_t decoded = nullptr, copied = nullptr, pushed = nullptr;   // head decoded / on the host / delta on the device
  bool running = false;
};

static LossTask* loss_task_of(layer* l)
{
  if (!l->loss_task)
  {
    LossTask* t = new LossTask();
    CHECK_HIP(hipEventCreateWithFlags(&t->decoded, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&t->copied, hipEventDisableTiming));
    CHECK_HIP(hipEventCreateWithFlags(&t->pushed, hipEventDisableTiming));
    l->loss_task = t;
  }
  return (LossTask*)l->loss_task;
}

void DkFreeLossTask(layer* l)
{
  LossTask* t = (LossTask*)l->loss_task;
  if (!t)
    return;
  if (t->running)
    t->fut.wait();
  (void)hipEventDestroy(t->decoded);
  (void)hipEventDestroy(t->copied);
  (void)hipEventDestroy(t->pushed);
  delete t;
  l->loss_task = nullptr;
}

// called by ForwardYoloLayerGpu in train mode (layers.cpp), after the decode kernel
void DkYoloTrainDelta(layer* l, NetworkState state)
{
  const size_t total = (size_t)l->batch * l->outputs;
  if (l->injected_delta)
  {
    cuda_push_array(l->delta_gpu, l->injected_delta, total);
    return;
  }
  if (!state.net->truth)
    error("yolo loss: no truth supplied (TrainNetworkDatum(net, x, y) with y != NULL)");
  if (!l->delta)
  {
    l->delta = cuda_make_array_pinned(nullptr, total);
    l->delta_pinned = 1;
  }
  LossTask* t = loss_task_of(l);
  if (t->running)
    t->fut.wait();
  hipStream_t st = get_cuda_stream(), cs = get_cuda_memcpy_stream();
  CHECK_HIP(hipEventRecord(t->decoded, st));
  CHECK_HIP(hipStreamWaitEvent(cs, t->decoded, 0));
  CHECK_HIP(hipMemcpyAsync(l->output, l->output_gpu, total * sizeof(float), hipMemcpyDeviceToHost, cs));
  CHECK_HIP(hipEventRecord(t->copied, cs));
  const int dev = cuda_get_device();
  const int net_w = state.net->w, net_h = state.net->h;
  const float* truth = state.net->truth;
  hipEvent_t copied = t->copied, pushed = t->pushed;
  t->fut = std::async(std::launch::async, [l, dev, net_w, net_h, truth, copied, pushed, cs, total]() {
    (void)hipSetDevice(dev);
    CHECK_HIP(hipEventSynchronize(copied));
    const float cost = DkYoloLossHost(l, net_w, net_h, l->output, truth, l->delta);
    // the delta goes up on the copy stream as soon as it exists (the big head's loss finishes while the GPU is
    // still in the forward pass), so the backward sweep only waits for an event instead of 62 MB of PCIe traffic;
    // delta_gpu has no other writer between the step's arena clear (long done: the decode came after it) and
    // the layer's backward
    CHECK_HIP(hipMemcpyAsync(l->delta_gpu, l->delta, total * sizeof(float), hipMemcpyHostToDevice, cs));
    CHECK_HIP(hipEventRecord(pushed, cs));
    return cost;
  });
  t->running = true;
}

// waits for the layer's host loss, stores the cost, enqueues the delta upload
static void DkYoloLossJoin(layer* l)
{
  LossTask* t = (LossTask*)l->loss_task;
  if (!t || !t->running)
    return;
  *(l->cost) = t->fut.get();
  t->running = false;
  CHECK_HIP(hipStreamWaitEvent(get_cuda_stream(), t->pushed, 0));   // the upload was enqueued by the loss task
}

static void backward_range(Network* net, NetworkState state, int hi, int lo)
{
  state.workspace = net->workspace;
  float* original_input = state.input;
  float* original_delta = state.delta;
  // network_kernels.cu:140-143: a stopbackward layer ends the sweep FOR THE STEP; the flag makes
  // that hold across the DkBackwardRange segments of the overlapped trainer too
  if (net->backward_stopped)
    return;
  for (int i = hi - 1; i >= lo; --i)
  {
    state.index = i;
    layer* l = &net->layers[i];
    if (l->stopbackward == 1 || l->stopbackward > net->curr_iter)
    {
      net->backward_stopped = 1;
      break;
    }
    if (i == 0)
    {
      state.input = original_input;
      state.delta = original_delta;
    }
    else
    {
      layer* prev = &net->layers[i - 1];
      state.input = prev->output_gpu;
      state.delta = prev->delta_gpu;
    }
    if (l->onlyforward)
      continue;
    if (l->backward_gpu)
      l->backward_gpu(l, state);
  }
  DkJoinWgradStream(net);   // the caller's next step (all-reduce of this segment, update) reads weight_updates
}

void BackwardNetworkGpu(Network* net, NetworkState state)
{
  net->backward_stopped = 0;
  backward_range(net, state, net->n, 0);
}

extern "C" LIB_API size_t DkGradBucketSize(Network* net);

// One launch for every conv / batchnorm tensor of the update (the per-layer slots remain for
// callers of the plugin API).  Used when no layer asks for a per-iteration exception (burnin_update,
// train_only_bn, dont_update) and loss_scale is 1; rebuilt when gradient pointers move.
static bool sgd_plan_usable(Network* net)
{
  if (net->loss_scale != 1.0f || net->adam)
    return false;
  for (int i = 0; i < net->n; ++i)
  {
    const layer* l = &net->layers[i];
    if (l->burnin_update || l->train_only_bn || l->dont_update || l->clip)
      return false;   // (clip: the per-layer path clamps after the update)
    if (l->update_gpu && l->update_gpu != UpdateConvolutionalLayerGpu && l->update_gpu != UpdateBatchnormLayerGpu)
      return false;
  }
  return true;
}

static void build_sgd_plan(Network* net)
{
  std::vector<float*> w, wu;
  std::vector<size_t> cnt;
  std::vector<float> sc;
  std::vector<int> dec;
  auto add = [&](float* a, float* b, size_t n, float s, int d) {
    if (a && b && n)
    {
      w.push_back(a); wu.push_back(b); cnt.push_back(n); sc.push_back(s); dec.push_back(d);
    }
  };
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL && l->update_gpu)
    {
      add(l->weights_gpu, l->weight_updates_gpu, l->nweights, l->learning_rate_scale, 1);
      add(l->biases_gpu, l->bias_updates_gpu, l->n, l->learning_rate_scale, 0);
      add(l->scales_gpu, l->scale_updates_gpu, l->n, l->learning_rate_scale, 0);
    }
    else if (l->type == BATCHNORM && l->update_gpu)
    {
      add(l->biases_gpu, l->bias_updates_gpu, l->c, l->learning_rate_scale, 0);
      add(l->scales_gpu, l->scale_updates_gpu, l->c, l->learning_rate_scale, 0);
    }
  }
  dk_sgd_plan_destroy(net->sgd_plan);
  net->sgd_plan = dk_sgd_plan_create((int)w.size(), w.data(), wu.data(), cnt.data(), sc.data(), dec.data());
}

void DkInvalidateSgdPlan(Network* net)
{
  dk_sgd_plan_destroy(net->sgd_plan);
  net->sgd_plan = nullptr;
}

void UpdateNetworkGpu(Network* net)
{
  cuda_set_device(net->gpu_index);
  // B = images behind the accumulated gradients: sub-batches of this replica x data-parallel replicas
  const int actual_batch = net->batch * net->subdiv * (net->grad_replicas > 1 ? net->grad_replicas : 1);
  const int iter = net->curr_iter;
  const float lr = GetCurrLr(net);
  static const bool multi = !(getenv("DK_SGD_MULTI") && !atoi(getenv("DK_SGD_MULTI")));
  if (multi && sgd_plan_usable(net))
  {
    if (!net->sgd_plan)
      build_sgd_plan(net);
    dk_sgd_update_multi(net->sgd_plan, actual_batch, lr, net->momentum, net->decay, get_cuda_stream());
    if (net->grad_replicas > 1 && net->grad_bucket)
      dk_scal(DkGradBucketSize(net), 1.0f / net->grad_replicas, net->grad_bucket, get_cuda_stream());
    return;
  }
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    l->t = iter;   // network_kernels.cu:229 (adam's bias-correction exponent)
    if (l->burnin_update && (l->burnin_update * net->burn_in > iter))
      continue;
    if (l->train_only_bn)
      continue;
    if (l->update_gpu && l->dont_update < iter)
      l->update_gpu(l, actual_batch, lr, net->momentum, net->decay, net->loss_scale);
  }
  // data parallel: the bucket now holds momentum * (summed gradients), identical on every
  // replica; keep 1/R of it so that the next all-reduce (a sum over R replicas) restores it
  // exactly once -- the single-process accumulation over subdivisions does the same implicitly
  if (net->grad_replicas > 1 && net->grad_bucket)
    dk_scal(DkGradBucketSize(net), 1.0f / net->grad_replicas, net->grad_bucket, get_cuda_stream());
}

void UpdateNetwork(Network* net) { UpdateNetworkGpu(net); }

void ForwardBackwardNetworkGpu(Network* net, float* x, float* y)
{
  net->truth = y;  // consumed by the (host) yolo loss
  if (net->gpu_index < 0)
    error("TrainNetworkDatum: no HIP device (this library has no CPU fallback)");
  if (!net->train)
    error("TrainNetworkDatum: the network was loaded for inference (LoadNetwork(train = true))");
  NetworkState state;
  memset(&state, 0, sizeof(state));
  state.net = net;
  const size_t x_size = (size_t)GetNetworkInputSize(net) * net->batch;
  const double t0 = now_ms();
  memcpy(net->input_pinned_cpu, x, x_size * sizeof(float));
  cuda_push_array(net->input_state_gpu, net->input_pinned_cpu, x_size);
  state.input = net->input_state_gpu;
  state.delta = 0;
  state.truth = 0;
  state.train = 1;
  const double t1 = now_ms();
  ForwardNetworkGpu(net, state);
  const double t2 = now_ms();
  BackwardNetworkGpu(net, state);
  if (train_timing())
  {
    const double t3 = now_ms();
    CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
    fprintf(stderr, "[train timing] input %.2f ms, forward issue %.2f, backward issue (+ loss joins) %.2f, drain %.2f\n",
        t1 - t0, t2 - t1, t3 - t2, now_ms() - t3);
  }
}

// ---- split train step (overlapped all-reduce, darknet_amd/train_dist.py) -----------------
static NetworkState train_state(Network* net)
{
  NetworkState state;
  memset(&state, 0, sizeof(state));
  state.net = net;
  state.input = net->input_state_gpu;
  state.train = 1;
  return state;
}

void DkTrainForward(Network* net, float* x, float* y)
{
  net->truth = y;
  if (net->gpu_index < 0 || !net->train)
    error("DkTrainForward: needs a train-mode network on a HIP device");
  net->seen += net->batch;
  net->backward_stopped = 0;
  const size_t x_size = (size_t)GetNetworkInputSize(net) * net->batch;
  memcpy(net->input_pinned_cpu, x, x_size * sizeof(float));
  cuda_push_array(net->input_state_gpu, net->input_pinned_cpu, x_size);
  ForwardNetworkGpu(net, train_state(net));
}

void DkBackwardRange(Network* net, int hi, int lo)
{
  if (hi > net->n) hi = net->n;
  if (lo < 0) lo = 0;
  backward_range(net, train_state(net), hi, lo);
}

float DkTrainFinish(Network* net)
{
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == YOLO)
      DkYoloLossJoin(&net->layers[i]);
  float sum = 0;
  int count = 0;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].cost)
    {
      sum += net->layers[i].cost[0];
      ++count;
    }
  return count ? sum / count : 0;
}

float TrainNetworkDatumGpu(Network* net, float* x, float* y)
{
  net->seen += net->batch;
  ForwardBackwardNetworkGpu(net, x, y);
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].type == YOLO)
      DkYoloLossJoin(&net->layers[i]);  // layers the backward pass did not reach
  CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
  // GetNetworkCost (network.cpp:145-158): mean of the layers' cost[0]
  float sum = 0;
  int count = 0;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].cost)
    {
      sum += net->layers[i].cost[0];
      ++count;
    }
  return count ? sum / count : 0;
}

float TrainNetworkDatum(Network* net, float* x, float* y) { return TrainNetworkDatumGpu(net, x, y); }

long DkLayerPull(Network* net, int i, int which, float* dst, size_t n)
{
  if (i < 0 || i >= net->n || net->gpu_index < 0)
    return -1;
  layer* l = &net->layers[i];
  float* p = nullptr;
  size_t cnt = 0;
  switch (which)
  {
    case 1: p = l->weights_gpu; cnt = l->nweights; break;
    case 2: p = l->biases_gpu; cnt = l->n; break;
    case 3: p = l->scales_gpu; cnt = l->n; break;
    case 4: p = l->rolling_mean_gpu; cnt = l->n; break;
    case 5: p = l->rolling_variance_gpu; cnt = l->n; break;
    case 6: p = l->delta_gpu; cnt = (size_t)l->batch * l->outputs; break;
    case 7: p = l->weight_updates_gpu; cnt = l->nweights; break;
    case 8: p = l->bias_updates_gpu; cnt = l->n; break;
    case 9: p = l->scale_updates_gpu; cnt = l->n; break;
    case 10: p = l->mean_gpu; cnt = l->n; break;
    case 11: p = l->variance_gpu; cnt = l->n; break;
  }
  if (!p || n < cnt)
    return -1;
  NetworkSync(net);
  cuda_pull_array(p, dst, cnt);
  return (long)cnt;
}


// ---------------------------------------------------------------------------
// Data-parallel training support: every replica keeps its
// gradients (weight_updates, bias_updates, scale_updates of every conv, in layer
// order) in ONE contiguous fp32 bucket so that a single RCCL all-reduce(sum) over
// xGMI replaces the reference's host-mediated weight averaging
// (SyncNetworks, src/network_kernels.cu:366-427).  The bucket is caller-owned
// device memory (bench/train drivers allocate it through torch so that
// torch.distributed can reduce it in place).
// ---------------------------------------------------------------------------
extern "C" LIB_API size_t DkGradBucketSize(Network* net)
{
  size_t n = 0;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL && (l->weight_updates_gpu || l->weight_updates))
      n += (size_t)l->nweights + l->n + ((l->scale_updates_gpu || l->scale_updates) ? l->n : 0);
    if (l->type == BATCHNORM && (l->scale_updates_gpu || (l->train && l->scale_updates)))
      n += 2 * (size_t)l->c;
  }
  return n;
}

extern "C" LIB_API void DkAttachGradBucket(Network* net, float* bucket)
{
  if (net->gpu_index < 0 || !net->train)
    error("DkAttachGradBucket: needs a train-mode network on a HIP device");
  hipStream_t st = get_cuda_stream();
  size_t off = 0;
  auto move = [&](float** p, size_t n) {
    CHECK_HIP(hipMemcpyAsync(bucket + off, *p, n * sizeof(float), hipMemcpyDeviceToDevice, st));
    CHECK_HIP(hipStreamSynchronize(st));
    cuda_free(*p);
    *p = bucket + off;
    off += n;
  };
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == BATCHNORM && l->scale_updates_gpu)
    {
      move(&l->bias_updates_gpu, l->c);
      move(&l->scale_updates_gpu, l->c);
      continue;
    }
    if (l->type != CONVOLUTIONAL || !l->weight_updates_gpu)
      continue;
    move(&l->weight_updates_gpu, l->nweights);
    move(&l->bias_updates_gpu, l->n);
    if (l->scale_updates_gpu)
      move(&l->scale_updates_gpu, l->n);
  }
  net->grad_bucket = bucket;
  DkInvalidateSgdPlan(net);   // the gradient tensors moved
}

// B = batch * subdivisions * replicas in the update (each replica contributes its sub-batches,
// exactly the reference's accumulation over subdivisions).
extern "C" LIB_API void DkSetSubdivisions(Network* net, int subdiv) { net->subdiv = subdiv; }
extern "C" LIB_API void DkSetReplicas(Network* net, int replicas) { net->grad_replicas = replicas; }
extern "C" LIB_API size_t DkGradBucketOffset(Network* net, int upto)
{
  size_t n = 0;
  for (int i = 0; i < net->n && i < upto; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL && (l->weight_updates_gpu || l->weight_updates))
      n += (size_t)l->nweights + l->n + ((l->scale_updates_gpu || l->scale_updates) ? l->n : 0);
    if (l->type == BATCHNORM && (l->scale_updates_gpu || (l->train && l->scale_updates)))
      n += 2 * (size_t)l->c;
  }
  return n;
}
extern "C" LIB_API void DkAdvanceIteration(Network* net) { net->curr_iter++; }
extern "C" LIB_API void DkSetCurrIter(Network* net, long long iter) { net->curr_iter = iter; }
