// parser.cpp -- .cfg -> layers and .weights reader/writer: the drop-in loader.
// Follows the reference's loader contract (Ravicmoon/darknet src/parser.cpp):
// ParseNetOptions :921-1055, ParseConv :179-242, ParseYolo :312-415,
// ParseMaxpool :640-659, ParseShortcut :720-779, ParseUpsample :820-826,
// ParseRoute :828-893, ParseNetworkCfg :1076-1519, SaveWeightsUpto :1590-1643,
// LoadConvolutionalWeights :1695-1759, LoadWeightsUpTo :1778-1844,
// LoadNetwork :1852-1876.  Only the layer kinds of the YOLOv4 family are built;
// any other section type prints "Type is not recognized" and leaves an EMPTY
// layer, like the reference does for unknown types (:1255-1258).
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dk_host.h"

// ---- [net] -----------------------------------------------------------------
static LearningRatePolicy GetPolicy(const char* s)
{
  if (strcmp(s, "random") == 0) return RANDOM;
  if (strcmp(s, "poly") == 0) return POLY;
  if (strcmp(s, "constant") == 0) return CONSTANT;
  if (strcmp(s, "step") == 0) return STEP;
  if (strcmp(s, "exp") == 0) return EXP;
  if (strcmp(s, "sigmoid") == 0) return SIG;
  if (strcmp(s, "steps") == 0) return STEPS;
  if (strcmp(s, "sgdr") == 0) return SGDR;
  fprintf(stderr, "Couldn't find policy %s, going with constant\n", s);
  return CONSTANT;
}

static int count_commas(const char* s)
{
  int n = 1;
  for (; *s; ++s)
    if (*s == ',')
      ++n;
  return n;
}

static void ParseNetOptions(Section& o, Network* net)
{
  net->seen = 0;
  net->curr_iter = 0;
  net->max_epoch = FindOptionIntQuiet(o, "max_epoch", 0);
  net->batch = FindOptionIntQuiet(o, "batch", 1);
  net->subdiv = FindOptionIntQuiet(o, "subdivisions", 1);
  if (net->subdiv < 1)
    net->subdiv = 1;
  net->batch /= net->subdiv;

  net->h = FindOptionIntQuiet(o, "height", 0);
  net->w = FindOptionIntQuiet(o, "width", 0);
  net->c = FindOptionIntQuiet(o, "channels", 0);
  if (!net->h || !net->w || !net->c)
    error("No input parameters supplied");
  net->inputs = net->h * net->w * net->c;

  net->lr = FindOptionFloatQuiet(o, "learning_rate", .001);
  net->lr_min = FindOptionFloatQuiet(o, "learning_rate_min", .00001);
  net->momentum = FindOptionFloatQuiet(o, "momentum", .9);
  net->decay = FindOptionFloatQuiet(o, "decay", .0001);
  net->policy = GetPolicy(FindOptionStrQuiet(o, "policy", "constant"));
  net->burn_in = FindOptionIntQuiet(o, "burn_in", 0);
  if (net->policy == STEP)
  {
    net->step = FindOptionIntQuiet(o, "step", 1);
    net->scale = FindOptionFloatQuiet(o, "scale", 1);
  }
  if (net->policy == STEPS || net->policy == SGDR)
  {
    net->sgdr_cycle = FindOptionIntQuiet(o, "sgdr_cycle", net->max_iter);
    net->sgdr_mult = FindOptionIntQuiet(o, "sgdr_mult", 2);
    const char* l = FindOption(o, "steps");
    const char* p = FindOption(o, "scales");
    if (net->policy == STEPS && (!l || !p))
      error("STEPS policy must have steps and scales in cfg file");
    if (l && p)
    {
      int n = count_commas(l);
      net->steps = (float*)xcalloc(n, sizeof(float));
      net->scales = (float*)xcalloc(n, sizeof(float));
      net->num_steps = n;
      for (int i = 0; i < n; ++i)
      {
        net->steps[i] = (float)atof(l);
        const char* nl = strchr(l, ',');
        l = nl ? nl + 1 : l + strlen(l);
        net->scales[i] = (float)atof(p);
        const char* np = strchr(p, ',');
        p = np ? np + 1 : p + strlen(p);
      }
    }
  }
  if (net->policy == EXP)
    net->gamma = FindOptionFloatQuiet(o, "gamma", 1);
  if (net->policy == SIG)
  {
    net->gamma = FindOptionFloatQuiet(o, "gamma", 1);
    net->step = FindOptionIntQuiet(o, "step", 1);
  }
  net->adam = FindOptionIntQuiet(o, "adam", 0);
  if (net->adam)
  {
    // parser.cpp:995-1000
    net->B1 = FindOptionFloat(o, "B1", .9f);
    net->B2 = FindOptionFloat(o, "B2", .999f);
    net->eps = FindOptionFloat(o, "eps", .000001f);
  }
  net->loss_scale = FindOptionFloatQuiet(o, "loss_scale", 1);
  net->power = FindOptionFloatQuiet(o, "power", 4);
  net->workspace_size_limit =
      (size_t)1024 * 1024 * FindOptionFloatQuiet(o, "workspace_size_limit_MB", 1024);
  // data-augmentation keys belong to the (out of scope) data loader: mark them
  // as known so they do not show up as "Unused field".
  static const char* aug[] = {"max_crop", "min_crop", "flip", "blur", "gaussian_noise", "cutmix",
      "mosaic", "label_smooth_eps", "resize_step", "angle", "aspect", "saturation", "exposure",
      "hue", "optimized_memory", "show_receptive_field"};
  for (const char* k : aug) (void)FindOption(o, k);
}

// ---- per-section parsers ----------------------------------------------------
static void ParseConv(layer* l, Section& o, SizeParams params)
{
  int n = FindOptionInt(o, "filters", 1);
  int groups = FindOptionIntQuiet(o, "groups", 1);
  int size = FindOptionInt(o, "size", 1);
  int stride = -1;
  int stride_x = FindOptionIntQuiet(o, "stride_x", -1);
  int stride_y = FindOptionIntQuiet(o, "stride_y", -1);
  if (stride_x < 1 || stride_y < 1)
  {
    stride = FindOptionInt(o, "stride", 1);
    if (stride_x < 1) stride_x = stride;
    if (stride_y < 1) stride_y = stride;
  }
  else
    stride = FindOptionIntQuiet(o, "stride", 1);
  int dilation = FindOptionIntQuiet(o, "dilation", 1);
  if (size == 1)
    dilation = 1;
  int pad = FindOptionIntQuiet(o, "pad", 0);
  int padding = FindOptionIntQuiet(o, "padding", 0);
  if (pad)
    padding = size / 2;
  ACTIVATION activation = get_activation(FindOptionStr(o, "activation", "logistic"));
  if (!(params.h && params.w && params.c))
    error("Layer before convolutional layer must output image.");
  int batch_normalize = FindOptionIntQuiet(o, "batch_normalize", 0);
  if (FindOptionIntQuiet(o, "binary", 0) || FindOptionIntQuiet(o, "xnor", 0))
    error("binary/xnor convolutions are outside the supported hot path");
  if (FindOptionIntQuiet(o, "antialiasing", 0))
    error("antialiasing is outside the supported hot path");
  if (FindOptionIntQuiet(o, "share_index", -1000000000) != -1000000000)
    error("share_index is outside the supported hot path");
  FillConvLayer(l, params.batch, params.h, params.w, params.c, n, groups, size, stride_x, stride_y,
      dilation, padding, activation, batch_normalize, params.index, params.train);
  if (params.net->adam && params.train && dk_gpu_enabled())
  {
    // convolutional_layer.cpp:589-620, parser.cpp:236-241: first / second moments of every trained tensor, zeroed
    l->adam = 1;
    l->B1 = params.net->B1; l->B2 = params.net->B2; l->eps = params.net->eps;
    l->m_gpu = cuda_make_array(nullptr, l->nweights);
    l->v_gpu = cuda_make_array(nullptr, l->nweights);
    l->bias_m_gpu = cuda_make_array(nullptr, n);
    l->bias_v_gpu = cuda_make_array(nullptr, n);
    l->scale_m_gpu = cuda_make_array(nullptr, n);
    l->scale_v_gpu = cuda_make_array(nullptr, n);
    for (float* q : {l->m_gpu, l->v_gpu})
      CHECK_HIP(hipMemsetAsync(q, 0, (size_t)l->nweights * sizeof(float), get_cuda_stream()));
    for (float* q : {l->bias_m_gpu, l->bias_v_gpu, l->scale_m_gpu, l->scale_v_gpu})
      CHECK_HIP(hipMemsetAsync(q, 0, (size_t)n * sizeof(float), get_cuda_stream()));
  }
}

static int* parse_int_list(const char* a, int* num)
{
  if (!a)
    return nullptr;
  int n = count_commas(a);
  int* v = (int*)xcalloc(n, sizeof(int));
  for (int i = 0; i < n; ++i)
  {
    v[i] = atoi(a);
    const char* nx = strchr(a, ',');
    a = nx ? nx + 1 : a + strlen(a);
  }
  *num = n;
  return v;
}

static void ParseYolo(layer* l, Section& o, SizeParams params)
{
  int classes = FindOptionInt(o, "classes", 20);
  int total = FindOptionInt(o, "num", 1);
  int num = total;
  int* mask = parse_int_list(FindOptionStr(o, "mask", 0), &num);
  int max_boxes = FindOptionIntQuiet(o, "max", 90);
  FillYoloLayer(l, params.batch, params.w, params.h, num, total, mask, classes, max_boxes);
  if (l->outputs != params.inputs)
  {
    printf("Error: l->outputs == params.inputs \n");
    printf("filters= in the [convolutional]-layer doesn't correspond to classes= or mask= in "
           "[yolo]-layer \n");
    exit(EXIT_FAILURE);
  }
  l->label_smooth_eps = FindOptionFloatQuiet(o, "label_smooth_eps", 0.0f);
  l->scale_x_y = FindOptionFloatQuiet(o, "scale_x_y", 1);
  l->max_delta = FindOptionFloatQuiet(o, "max_delta", FLT_MAX);
  l->iou_normalizer = FindOptionFloatQuiet(o, "iou_normalizer", 0.75);
  l->cls_normalizer = FindOptionFloatQuiet(o, "cls_normalizer", 1);
  const char* iou_loss = FindOptionStrQuiet(o, "iou_loss", "mse");
  if (strcmp(iou_loss, "mse") == 0) l->iou_loss = MSE;
  else if (strcmp(iou_loss, "giou") == 0) l->iou_loss = GIOU;
  else if (strcmp(iou_loss, "diou") == 0) l->iou_loss = DIOU;
  else if (strcmp(iou_loss, "ciou") == 0) l->iou_loss = CIOU;
  else l->iou_loss = IOU;
  const char* itk = FindOptionStrQuiet(o, "iou_thresh_kind", "iou");
  if (strcmp(itk, "giou") == 0) l->iou_thresh_kind = GIOU;
  else if (strcmp(itk, "diou") == 0) l->iou_thresh_kind = DIOU;
  else if (strcmp(itk, "ciou") == 0) l->iou_thresh_kind = CIOU;
  else l->iou_thresh_kind = IOU;
  l->beta_nms = FindOptionFloatQuiet(o, "beta_nms", 0.6);
  const char* nms_kind = FindOptionStrQuiet(o, "nms_kind", "greedynms");
  l->nms_kind = (strcmp(nms_kind, "diounms") == 0) ? DIOU_NMS : GREEDY_NMS;
  l->jitter = FindOptionFloatQuiet(o, "jitter", .2);
  l->focal_loss = FindOptionIntQuiet(o, "focal_loss", 0);
  l->ignore_thresh = FindOptionFloatQuiet(o, "ignore_thresh", .5);
  l->truth_thresh = FindOptionFloatQuiet(o, "truth_thresh", 1);
  l->iou_thresh = FindOptionFloatQuiet(o, "iou_thresh", 1);
  l->random = FindOptionFloatQuiet(o, "random", 0);
  (void)FindOption(o, "counters_per_class");
  (void)FindOption(o, "map");
  const char* a = FindOptionStr(o, "anchors", 0);
  if (a)
  {
    int n = count_commas(a);
    for (int i = 0; i < n && i < total * 2; ++i)
    {
      l->biases[i] = (float)atof(a);
      const char* nx = strchr(a, ',');
      a = nx ? nx + 1 : a + strlen(a);
    }
  }
}

// ParseGaussianYolo, src/parser.cpp:443-552 (inference fields; the loss options are accepted and kept)
static void ParseGaussianYolo(layer* l, Section& o, SizeParams params)
{
  int classes = FindOptionInt(o, "classes", 20);
  int max_boxes = FindOptionIntQuiet(o, "max", 90);
  int total = FindOptionInt(o, "num", 1);
  int num = total;
  int* mask = parse_int_list(FindOptionStr(o, "mask", 0), &num);
  FillGaussianYoloLayer(l, params.batch, params.w, params.h, num, total, mask, classes, max_boxes);
  if (l->outputs != params.inputs)
  {
    printf("Error: l->outputs == params.inputs \n");
    printf("filters= in the [convolutional]-layer doesn't correspond to classes= or mask= in "
           "[Gaussian_yolo]-layer \n");
    exit(EXIT_FAILURE);
  }
  (void)FindOption(o, "counters_per_class");
  l->label_smooth_eps = FindOptionFloatQuiet(o, "label_smooth_eps", 0.0f);
  l->scale_x_y = FindOptionFloatQuiet(o, "scale_x_y", 1);
  l->max_delta = FindOptionFloatQuiet(o, "max_delta", FLT_MAX);
  l->uc_normalizer = FindOptionFloatQuiet(o, "uc_normalizer", 1.0f);
  l->iou_normalizer = FindOptionFloatQuiet(o, "iou_normalizer", 0.75);
  l->cls_normalizer = FindOptionFloatQuiet(o, "cls_normalizer", 1);
  const char* iou_loss = FindOptionStrQuiet(o, "iou_loss", "mse");
  if (strcmp(iou_loss, "mse") == 0) l->iou_loss = MSE;
  else if (strcmp(iou_loss, "giou") == 0) l->iou_loss = GIOU;
  else if (strcmp(iou_loss, "diou") == 0) l->iou_loss = DIOU;
  else if (strcmp(iou_loss, "ciou") == 0) l->iou_loss = CIOU;
  else l->iou_loss = IOU;
  const char* itk = FindOptionStrQuiet(o, "iou_thresh_kind", "iou");
  if (strcmp(itk, "giou") == 0) l->iou_thresh_kind = GIOU;
  else if (strcmp(itk, "diou") == 0) l->iou_thresh_kind = DIOU;
  else if (strcmp(itk, "ciou") == 0) l->iou_thresh_kind = CIOU;
  else l->iou_thresh_kind = IOU;
  l->beta_nms = FindOptionFloatQuiet(o, "beta_nms", 0.6);
  const char* nms_kind = FindOptionStrQuiet(o, "nms_kind", "greedynms");
  l->nms_kind = (strcmp(nms_kind, "diounms") == 0) ? DIOU_NMS : GREEDY_NMS;
  const char* yp = FindOptionStrQuiet(o, "yolo_point", "center");
  l->yolo_point = strcmp(yp, "left_top") == 0 ? YOLO_LEFT_TOP : (strcmp(yp, "right_bottom") == 0 ? YOLO_RIGHT_BOTTOM : YOLO_CENTER);
  l->jitter = FindOptionFloatQuiet(o, "jitter", .2);
  l->ignore_thresh = FindOptionFloatQuiet(o, "ignore_thresh", .5);
  l->truth_thresh = FindOptionFloatQuiet(o, "truth_thresh", 1);
  l->iou_thresh = FindOptionFloatQuiet(o, "iou_thresh", 1);
  l->random = FindOptionFloatQuiet(o, "random", 0);
  (void)FindOption(o, "map");
  const char* a = FindOptionStr(o, "anchors", 0);
  if (a)
  {
    int n = count_commas(a);
    for (int i = 0; i < n && i < total * 2; ++i)
    {
      l->biases[i] = (float)atof(a);
      const char* nx = strchr(a, ',');
      a = nx ? nx + 1 : a + strlen(a);
    }
  }
}

static void ParseMaxpool(layer* l, Section& o, SizeParams params)
{
  int stride = FindOptionInt(o, "stride", 1);
  int stride_x = FindOptionIntQuiet(o, "stride_x", stride);
  int stride_y = FindOptionIntQuiet(o, "stride_y", stride);
  int size = FindOptionInt(o, "size", stride);
  int padding = FindOptionIntQuiet(o, "padding", size - 1);
  if (FindOptionIntQuiet(o, "maxpool_depth", 0) || FindOptionIntQuiet(o, "antialiasing", 0))
    error("maxpool_depth / antialiasing are outside the supported hot path");
  (void)FindOption(o, "out_channels");
  if (!(params.h && params.w && params.c))
    error("Layer before [maxpool] layer must output image.");
  FillMaxpoolLayer(
      l, params.batch, params.h, params.w, params.c, size, stride_x, stride_y, padding, params.train);
}

static void ParseShortcut(layer* l, Section& o, SizeParams params, Network* net)
{
  ACTIVATION activation = get_activation(FindOptionStr(o, "activation", "linear"));
  const char* from = FindOption(o, "from");
  if (!from)
    error("Route Layer must specify input layers: from = ...");
  int idx = atoi(from);  // n is fixed as 1 (parser.cpp:729-730)
  if (idx < 0)
    idx = params.index + idx;
  if (idx < 0 || idx >= params.index)
    error("[shortcut] from= out of range");
  FillShortcutLayer(l, params.batch, idx, params.w, params.h, params.c, net->layers[idx].outputs,
      activation, params.train);
  layer* f = &net->layers[idx];
  if (params.w != f->out_w || params.h != f->out_h || params.c != f->out_c)
    fprintf(stderr, " (%4d x%4d x%4d) + (%4d x%4d x%4d) \n", params.w, params.h, params.c, f->out_w,
        f->out_h, f->out_c);
}

static void ParseUpsample(layer* l, Section& o, SizeParams params)
{
  int stride = FindOptionInt(o, "stride", 2);
  FillUpsampleLayer(l, params.batch, params.w, params.h, params.c, stride);
  l->scale = FindOptionFloatQuiet(o, "scale", 1);
}

static void ParseRoute(layer* l, Section& o, SizeParams params)
{
  const char* input_layers = FindOption(o, "layers");
  if (!input_layers)
    error("Route Layer must specify input layers");
  int n = 0;
  int* layers = parse_int_list(input_layers, &n);
  int* sizes = (int*)xcalloc(n, sizeof(int));
  for (int i = 0; i < n; ++i)
  {
    if (layers[i] < 0)
      layers[i] = params.index + layers[i];
    if (layers[i] < 0 || layers[i] >= params.index)
      error("[route] layers= out of range");
    sizes[i] = params.net->layers[layers[i]].outputs;
  }
  int groups = FindOptionIntQuiet(o, "groups", 1);
  int group_id = FindOptionIntQuiet(o, "group_id", 0);
  FillRouteLayer(l, params.batch, n, layers, sizes, groups, group_id);
  layer* first = &params.net->layers[layers[0]];
  l->out_w = first->out_w;
  l->out_h = first->out_h;
  l->out_c = first->out_c;
  for (int i = 1; i < n; ++i)
  {
    layer* next = &params.net->layers[layers[i]];
    if (next->out_w == first->out_w && next->out_h == first->out_h)
      l->out_c += next->out_c;
    else
    {
      fprintf(stderr, " The width and height of the input layers are different. \n");
      l->out_h = l->out_w = l->out_c = 0;
    }
  }
  l->out_c = l->out_c / l->groups;
  l->w = first->w;
  l->h = first->h;
  l->c = l->out_c;
}

static void ParseBatchnorm(layer* l, Section&, SizeParams params)
{
  FillBatchnormLayer(l, params.batch, params.w, params.h, params.c, params.train);
}

static void ParseAvgpool(layer* l, Section&, SizeParams params)
{
  if (!(params.h && params.w && params.c))
    error("Layer before avgpool layer must output image.");
  FillAvgpoolLayer(l, params.batch, params.w, params.h, params.c);
}

static void ParseScaleChannels(layer* l, Section& o, SizeParams params, Network* net)
{
  const char* from = FindOption(o, "from");
  if (!from)
    error("[scale_channels] must specify from = ...");
  int idx = atoi(from);
  if (idx < 0)
    idx = params.index + idx;
  if (idx < 0 || idx >= params.index)
    error("[scale_channels] from= out of range");
  const int scale_wh = FindOptionIntQuiet(o, "scale_wh", 0);
  layer* f = &net->layers[idx];
  FillScaleChannelsLayer(l, params.batch, idx, params.w, params.h, params.c, f->out_w, f->out_h, f->out_c, scale_wh);
  l->activation = get_activation(FindOptionStrQuiet(o, "activation", "linear"));
  if (l->activation == SWISH || l->activation == MISH)
    printf(" [scale_channels] layer doesn't support SWISH or MISH activations \n");
}

static void ParseDropout(layer* l, Section& o, SizeParams params, Network* net)
{
  const float probability = FindOptionFloat(o, "probability", .2);
  if (FindOptionIntQuiet(o, "dropblock", 0))
    error("[dropout] dropblock is outside the supported hot path");
  (void)FindOption(o, "dropblock_size_rel");
  (void)FindOption(o, "dropblock_size_abs");
  if (params.index < 1)
    error("[dropout] cannot be the first layer");
  FillDropoutLayer(l, params.batch, params.inputs, probability, params.w, params.h, params.c);
  // parser.cpp:1232-1242: the layer works in place on its predecessor's buffers
  layer* prev = &net->layers[params.index - 1];
  l->output_gpu = prev->output_gpu;
  l->delta_gpu = prev->delta_gpu;
}

// ---- ParseNetworkCfg ----------------------------------------------------------
// One arena for all delta tensors: the per-layer zero-fills of the forward pass
// (forward_network_gpu's fill_ongpu per layer, network_kernels.cu:79) become one memset.
// A layer's delta only has to start the step at zero when something ACCUMULATES into it before anything
// overwrites it.  The data gradient of a convolution overwrites its predecessor's delta (col2im zero-fills its
// target) after every later route / shortcut has added to it, so whatever the predecessor's
// delta held before is discarded: those layers -- two thirds of yolov4's 3.4 GB of deltas -- sit behind the
// zeroed part of the arena and are never cleared.
static bool delta_needs_zero(const Network* net, int i)
{
  if (i + 1 >= net->n)
    return true;
  const layer* nx = &net->layers[i + 1];
  return !(nx->type == CONVOLUTIONAL && !nx->onlyforward && !nx->stopbackward && !nx->buffers_aliased &&
           !net->layers[i].buffers_aliased);
}

void DkBuildDeltaArena(Network* net)
{
  size_t tot = 0, zero = 0;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].delta_gpu && !net->layers[i].buffers_aliased)
    {
      const size_t sz = (((size_t)net->layers[i].outputs * net->layers[i].batch + 63) / 64) * 64;
      tot += sz;
      if (delta_needs_zero(net, i))
        zero += sz;
    }
  if (!tot)
    return;
  net->delta_arena_gpu = cuda_make_array(0, tot);
  net->delta_arena_size = tot;
  net->delta_arena_zero = zero;
  size_t off_zero = 0, off_rest = zero;
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->buffers_aliased)
    {
      l->delta_gpu = net->layers[i - 1].delta_gpu;   // follows its predecessor into the arena
      continue;
    }
    if (!l->delta_gpu)
      continue;
    cuda_free(l->delta_gpu);
    size_t& off = delta_needs_zero(net, i) ? off_zero : off_rest;
    l->delta_gpu = net->delta_arena_gpu + off;
    l->delta_in_arena = 1;
    off += (((size_t)l->outputs * l->batch + 63) / 64) * 64;
  }
  CHECK_HIP(hipMemsetAsync(net->delta_arena_gpu, 0, tot * sizeof(float), get_cuda_stream()));
}

void DkConvPrepare(layer* l);

static bool parse_cfg_batch(Network* net, char const* filename, bool train, int force_batch)
{
  std::vector<Section> sections;
  if (!ReadSections(filename, sections))
    return false;
  if (sections.empty())
    error("Config file has no sections");
  net->n = (int)sections.size() - 1;
  net->layers = (layer*)xcalloc(net->n, sizeof(layer));
  net->gpu_index = cuda_get_device();
  net->input_gpu = (float**)xcalloc(1, sizeof(float*));
  net->truth_gpu = (float**)xcalloc(1, sizeof(float*));

  Section& ns = sections[0];
  if (!(ns.type == "[net]" || ns.type == "[network]"))
    error("First section must be [net] or [network]");
  ParseNetOptions(ns, net);

  SizeParams params;
  params.train = train;
  params.h = net->h;
  params.w = net->w;
  params.c = net->c;
  params.inputs = net->inputs;
  if (force_batch > 0)
    net->batch = force_batch;  // additive: LoadNetworkBatch
  else if (!train || net->batch < 1)
    net->batch = 1;            // parser.cpp:1114-1115
  params.batch = net->batch;
  params.net = net;
  net->train = train;

  float bflops = 0;
  size_t workspace_size = 0;
  for (int count = 0; count < net->n; ++count)
  {
    params.index = count;
    Section& s = sections[count + 1];
    layer* l = &net->layers[count];
    l->fuse_residual_from = -1;
    l->conv_cfg = -1;
    if (s.type == "[convolutional]" || s.type == "[conv]")
      ParseConv(l, s, params);
    else if (s.type == "[yolo]")
      ParseYolo(l, s, params);
    else if (s.type == "[Gaussian_yolo]")
      ParseGaussianYolo(l, s, params);
    else if (s.type == "[maxpool]" || s.type == "[max]")
      ParseMaxpool(l, s, params);
    else if (s.type == "[route]")
      ParseRoute(l, s, params);
    else if (s.type == "[shortcut]")
      ParseShortcut(l, s, params, net);
    else if (s.type == "[upsample]")
      ParseUpsample(l, s, params);
    else if (s.type == "[batchnorm]")
      ParseBatchnorm(l, s, params);
    else if (s.type == "[avgpool]" || s.type == "[avg]")
      ParseAvgpool(l, s, params);
    else if (s.type == "[scale_channels]")
      ParseScaleChannels(l, s, params, net);
    else if (s.type == "[dropout]")
      ParseDropout(l, s, params, net);
    else
    {
      fprintf(stderr, "Type is not recognized: %s\n", s.type.c_str());
      l->type = EMPTY;
      l->batch = params.batch;
      l->out_h = params.h; l->out_w = params.w; l->out_c = params.c;
      l->outputs = params.inputs;
    }
    // per-layer common keys, parser.cpp:1361-1369
    l->clip = FindOptionFloatQuiet(s, "clip", 0);
    l->onlyforward = FindOptionIntQuiet(s, "onlyforward", 0);
    l->dont_update = FindOptionIntQuiet(s, "dont_update", 0);
    l->burnin_update = FindOptionIntQuiet(s, "burnin_update", 0);
    l->stopbackward = FindOptionIntQuiet(s, "stopbackward", 0);
    l->train_only_bn = FindOptionIntQuiet(s, "train_only_bn", 0);
    l->dontload = FindOptionIntQuiet(s, "dontload", 0);
    l->dontloadscales = FindOptionIntQuiet(s, "dontloadscales", 0);
    l->learning_rate_scale = FindOptionFloatQuiet(s, "learning_rate", 1);
    UnusedOption(s);

    if (l->workspace_size > workspace_size)
      workspace_size = l->workspace_size;
    params.h = l->out_h;
    params.w = l->out_w;
    params.c = l->out_c;
    params.inputs = l->outputs;
    if (l->bflops > 0)
      bflops += l->bflops;
  }
  net->outputs = GetNetworkOutputSize(net);
  net->truths = 0;
  for (int i = 0; i < net->n; ++i)
    if (net->layers[i].truths)
      net->truths = net->layers[i].truths;
  if (getenv("DK_VERBOSE"))
    fprintf(stderr, "Total BFLOPS %5.3f \n", bflops);

  if (dk_gpu_enabled())
  {
    const size_t size = (size_t)GetNetworkInputSize(net) * net->batch;
    net->input_state_gpu = cuda_make_array(0, size);
    net->input_pinned_cpu = cuda_make_array_pinned(nullptr, size);
    net->input_pinned_cpu_flag = 1;
    if (workspace_size)
      net->workspace = cuda_make_array(0, workspace_size / sizeof(float) + 1);
    if (train)
    {
      size_t maxw = 1;
      for (int i = 0; i < net->n; ++i)
        if (net->layers[i].type == CONVOLUTIONAL && (size_t)net->layers[i].nweights > maxw)
          maxw = net->layers[i].nweights;
      net->wt_scratch_gpu = cuda_make_array(0, maxw);
      // Winograd filters of one 3x3 layer (16 values per (filter, channel) pair instead of 9)
      size_t maxu = 0;
      for (int i = 0; i < net->n; ++i)
      {
        const layer* l = &net->layers[i];
        if (l->type == CONVOLUTIONAL && l->size == 3 && (size_t)16 * l->n * (l->c / l->groups) > maxu)
          maxu = (size_t)16 * l->n * (l->c / l->groups);
      }
      if (maxu)
        net->wino_scratch_gpu = cuda_make_array(0, maxu);
      DkBuildDeltaArena(net);
    }
    // tap tables of every conv shape exist before anybody can capture a stream (train and
    // inference loads alike: NetworkPredict on a train-mode or hand-assembled net captures too)
    for (int i = 0; i < net->n; ++i)
      if (net->layers[i].type == CONVOLUTIONAL)
        DkConvPrepare(&net->layers[i]);
    CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
  }
  // host mirror of the last layer for NetworkPredict's return value
  layer* last = &net->layers[net->n - 1];
  if (!last->output)
    last->output = (float*)xcalloc((size_t)last->outputs * last->batch, sizeof(float));
  net->output = last->output;
  return true;
}

bool ParseNetworkCfg(Network* net, char const* filename, bool train)
{
  return parse_cfg_batch(net, filename, train, 0);
}

// ---- weights ------------------------------------------------------------------
static void SaveConvolutionalWeights(layer* l, FILE* fp)
{
  if (dk_gpu_enabled())
    PullConvolutionalLayer(l);
  fwrite(l->biases, sizeof(float), l->n, fp);
  if (l->batch_normalize)
  {
    fwrite(l->scales, sizeof(float), l->n, fp);
    fwrite(l->rolling_mean, sizeof(float), l->n, fp);
    fwrite(l->rolling_variance, sizeof(float), l->n, fp);
  }
  fwrite(l->weights, sizeof(float), l->nweights, fp);
}

// parser.cpp:1562-1572 / :1683-1693
static void SaveBatchnormWeights(layer* l, FILE* fp)
{
  if (dk_gpu_enabled())
    PullBatchnormLayer(l);
  fwrite(l->biases, sizeof(float), l->c, fp);
  fwrite(l->scales, sizeof(float), l->c, fp);
  fwrite(l->rolling_mean, sizeof(float), l->c, fp);
  fwrite(l->rolling_variance, sizeof(float), l->c, fp);
}

static void LoadBatchnormWeights(layer* l, FILE* fp)
{
  fread(l->biases, sizeof(float), l->c, fp);
  fread(l->scales, sizeof(float), l->c, fp);
  fread(l->rolling_mean, sizeof(float), l->c, fp);
  fread(l->rolling_variance, sizeof(float), l->c, fp);
  PushBatchnormLayer(l);
}

void SaveWeightsUpto(Network* net, char const* filename, int cutoff)
{
  if (net->gpu_index >= 0 && net->layers && dk_gpu_enabled())
    cuda_set_device(net->gpu_index);
  FILE* fp = fopen(filename, "wb");
  if (!fp)
    FileError(filename);
  int major = 0, minor = 2, revision = 5;  // src/version.h
  fwrite(&major, sizeof(int), 1, fp);
  fwrite(&minor, sizeof(int), 1, fp);
  fwrite(&revision, sizeof(int), 1, fp);
  fwrite(&net->seen, sizeof(uint64_t), 1, fp);
  for (int i = 0; i < net->n && i < cutoff; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL && l->share_layer == NULL)
      SaveConvolutionalWeights(l, fp);
    if (l->type == BATCHNORM)
      SaveBatchnormWeights(l, fp);
  }
  fclose(fp);
}

void SaveWeights(Network* net, char const* filename) { SaveWeightsUpto(net, filename, net->n); }

static void LoadConvolutionalWeights(layer* l, FILE* fp)
{
  size_t r = fread(l->biases, sizeof(float), l->n, fp);
  if (r > 0 && r < (size_t)l->n)
    printf("\n Warning: Unexpected end of wights-file! l->biases - l->index = %d \n", l->index);
  if (l->batch_normalize && (!l->dontloadscales))
  {
    r = fread(l->scales, sizeof(float), l->n, fp);
    if (r > 0 && r < (size_t)l->n)
      printf("\n Warning: Unexpected end of wights-file! l->scales - l->index = %d \n", l->index);
    r = fread(l->rolling_mean, sizeof(float), l->n, fp);
    if (r > 0 && r < (size_t)l->n)
      printf("\n Warning: Unexpected end of wights-file! l->rolling_mean - l->index = %d \n",
          l->index);
    r = fread(l->rolling_variance, sizeof(float), l->n, fp);
    if (r > 0 && r < (size_t)l->n)
      printf("\n Warning: Unexpected end of wights-file! l->rolling_variance - l->index = %d \n",
          l->index);
  }
  r = fread(l->weights, sizeof(float), l->nweights, fp);
  if (r > 0 && r < (size_t)l->n)
    printf("\n Warning: Unexpected end of wights-file! l->weights - l->index = %d \n", l->index);
  PushConvolutionalLayer(l);
}

bool LoadWeightsUpTo(Network* net, char const* filename, int cutoff)
{
  if (net->gpu_index >= 0 && net->layers && dk_gpu_enabled())
    cuda_set_device(net->gpu_index);
  FILE* fp = fopen(filename, "rb");
  if (fp == nullptr)
    return false;
  int major, minor, revision;
  if (fread(&major, sizeof(int), 1, fp) != 1 || fread(&minor, sizeof(int), 1, fp) != 1 ||
      fread(&revision, sizeof(int), 1, fp) != 1 || fread(&net->seen, sizeof(uint64_t), 1, fp) != 1)
  {
    fclose(fp);
    return false;
  }
  net->curr_iter = (int)(net->seen / ((uint64_t)net->batch * net->subdiv));
  int num_layer = net->n < cutoff ? net->n : cutoff;
  for (int i = 0; i < num_layer; ++i)
  {
    layer* l = &net->layers[i];
    if (l->dontload)
      continue;
    if (l->type == CONVOLUTIONAL && l->share_layer == NULL)
      LoadConvolutionalWeights(l, fp);
    if (l->type == BATCHNORM)
      LoadBatchnormWeights(l, fp);
    if (feof(fp))
      break;
  }
  fclose(fp);
  // PushConvolutionalLayer refreshed the derived copies (dual / packed fp16) layer by layer
  return true;
}

bool LoadWeights(Network* net, char const* filename)
{
  return LoadWeightsUpTo(net, filename, net->n);
}

size_t DkWeightsFileSize(Network* net)
{
  size_t n = 3 * sizeof(int) + sizeof(uint64_t);
  for (int i = 0; i < net->n; ++i)
  {
    layer* l = &net->layers[i];
    if (l->type == CONVOLUTIONAL)
      n += sizeof(float) * ((size_t)l->n + l->nweights + (l->batch_normalize ? 3 * (size_t)l->n : 0));
    if (l->type == BATCHNORM)
      n += sizeof(float) * 4 * (size_t)l->c;
  }
  return n;
}

// ---- LoadNetwork ----------------------------------------------------------------
static bool load_common(Network* net, char const* model_file, char const* weights_file, bool train,
    bool clear, int force_batch)
{
  bool ret = parse_cfg_batch(net, model_file, train, force_batch);
  if (!ret)
    return false;
  if (weights_file != nullptr && weights_file[0])
    ret = LoadWeights(net, weights_file);
  if (!train)
  {
    FuseConvBatchNorm(net);
    if (dk_gpu_enabled())
      DkPlanInference(net);
  }
  if (clear)
  {
    net->seen = 0;
    net->curr_iter = 0;
  }
  return ret;
}

bool LoadNetwork(
    Network* net, char const* model_file, char const* weights_file, bool train, bool clear)
{
  return load_common(net, model_file, weights_file, train, clear, 0);
}

bool LoadNetworkBatch(Network* net, char const* model_file, char const* weights_file, int batch)
{
  if (batch < 1)
    batch = 1;
  return load_common(net, model_file, weights_file, false, false, batch);
}
