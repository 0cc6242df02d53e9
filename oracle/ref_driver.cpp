// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Thin flat-C driver around the *real* reference (Ravicmoon/darknet) CPU build.
// It is compiled together with the reference's own sources, where they lie
// under /root/reference/src, by oracle/Makefile into oracle/_ref/*.so.  It
// copies nothing from the reference: it only includes its public headers at
// compile time and calls its public API (LoadNetwork / NetworkPredict /
// GetNetworkBoxes / TrainNetworkDatum / UpdateNetwork, src/yolo_core.h:624-667,
// src/network.h:20-31), so that tools/make_golden.py can dump golden vectors and
// tests can pin oracle/orc_ops.c against the real thing.
//
// It also supplies the two symbols the hot path's object files reference from
// the OpenCV-dependent data.cpp, which is deliberately not compiled
// (SURVEY.md section 8c): get_next_batch (data.h:81) and GetList (data.h:79).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "box.h"
#include "data.h"
#include "network.h"
#include "yolo_core.h"

// ---- stand-ins for data.cpp (never on the measured / checked path) --------
std::vector<std::string> GetList(std::string) { return {}; }

void get_next_batch(data d, int n, int offset, float* X, float* y)
{
  for (int j = 0; j < n; ++j)
  {
    int index = offset + j;
    memcpy(X + j * d.X.cols, d.X.vals[index], d.X.cols * sizeof(float));
    if (y)
      memcpy(y + j * d.y.cols, d.y.vals[index], d.y.cols * sizeof(float));
  }
}

extern "C" {

Network* ref_net_load(const char* cfg, const char* weights, int train)
{
  Network* net = (Network*)calloc(1, sizeof(Network));
  if (!LoadNetwork(net, cfg, (weights && weights[0]) ? weights : nullptr,
          train != 0, false))
  {
    if (weights && weights[0])
    {
      free(net);
      return nullptr;
    }
  }
  return net;
}

void ref_net_free(Network* net)
{
  FreeNetwork(net);
  free(net);
}

float* ref_net_predict(Network* net, float* input)
{
  return NetworkPredict(net, input);
}

int ref_net_n(Network* net) { return net->n; }
int ref_net_batch(Network* net) { return net->batch; }
void ref_net_dims(Network* net, int* out)
{
  out[0] = net->w; out[1] = net->h; out[2] = net->c; out[3] = net->batch;
  out[4] = net->subdiv; out[5] = net->inputs; out[6] = net->outputs;
}

// 24 ints describing layer i
void ref_layer_info(Network* net, int i, int* o)
{
  layer* l = &net->layers[i];
  o[0] = l->type; o[1] = l->batch; o[2] = l->outputs; o[3] = l->out_c;
  o[4] = l->out_h; o[5] = l->out_w; o[6] = l->n; o[7] = l->size;
  o[8] = l->stride; o[9] = l->pad; o[10] = l->c; o[11] = l->h; o[12] = l->w;
  o[13] = l->activation; o[14] = l->batch_normalize; o[15] = l->nweights;
  o[16] = l->groups; o[17] = l->inputs; o[18] = l->classes; o[19] = l->total;
  o[20] = l->index; o[21] = l->dilation; o[22] = l->stride_x; o[23] = l->stride_y;
}

float ref_layer_bflops(Network* net, int i) { return net->layers[i].bflops; }

// which: 0 output 1 weights 2 biases 3 scales 4 rolling_mean 5 rolling_variance
//        6 delta 7 weight_updates 8 bias_updates 9 scale_updates 10 mean
//        11 variance 12 x 13 x_norm 14 activation_input
float* ref_layer_ptr(Network* net, int i, int which)
{
  layer* l = &net->layers[i];
  switch (which)
  {
    case 0: return l->output;
    case 1: return l->weights;
    case 2: return l->biases;
    case 3: return l->scales;
    case 4: return l->rolling_mean;
    case 5: return l->rolling_variance;
    case 6: return l->delta;
    case 7: return l->weight_updates;
    case 8: return l->bias_updates;
    case 9: return l->scale_updates;
    case 10: return l->mean;
    case 11: return l->variance;
    case 12: return l->x;
    case 13: return l->x_norm;
    case 14: return l->activation_input;
  }
  return nullptr;
}

int* ref_layer_indexes(Network* net, int i) { return net->layers[i].indexes; }
float ref_layer_cost(Network* net, int i)
{
  return net->layers[i].cost ? net->layers[i].cost[0] : 0.f;
}

// Flattened detections: per det [x,y,w,h,objectness, prob[classes]].
// Returns number of detections; writes at most max_dets.
int ref_get_boxes(Network* net, float thresh, float* out, int max_dets,
    int* classes_out)
{
  int num = 0;
  Detection* dets = GetNetworkBoxes(net, thresh, &num);
  int classes = net->layers[net->n - 1].classes;
  if (classes_out)
    *classes_out = classes;
  int stride = 5 + classes;
  for (int i = 0; i < num && i < max_dets; ++i)
  {
    float* o = out + (size_t)i * stride;
    o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y;
    o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
    o[4] = dets[i].objectness;
    for (int j = 0; j < classes; ++j) o[5 + j] = dets[i].prob[j];
  }
  FreeDetections(dets, num);
  return num;
}

// NMS on a flattened detection array (in place); layout as ref_get_boxes.
void ref_nms_sort(float* buf, int num, int classes, float thresh, int nms_kind,
    float beta)
{
  int stride = 5 + classes;
  Detection* dets = (Detection*)calloc(num, sizeof(Detection));
  for (int i = 0; i < num; ++i)
  {
    float* o = buf + (size_t)i * stride;
    dets[i].bbox = Box(o[0], o[1], o[2], o[3]);
    dets[i].objectness = o[4];
    dets[i].classes = classes;
    dets[i].prob = o + 5;
  }
  NmsSort(dets, num, classes, thresh, (NMS_KIND)nms_kind, beta);
  // NmsSort reorders the Detection structs; write back in the new order.
  std::vector<float> tmp((size_t)num * stride);
  for (int i = 0; i < num; ++i)
  {
    float* o = tmp.data() + (size_t)i * stride;
    o[0] = dets[i].bbox.x; o[1] = dets[i].bbox.y;
    o[2] = dets[i].bbox.w; o[3] = dets[i].bbox.h;
    o[4] = dets[i].objectness;
    for (int j = 0; j < classes; ++j) o[5 + j] = dets[i].prob[j];
  }
  memcpy(buf, tmp.data(), tmp.size() * sizeof(float));
  free(dets);
}

float ref_train_datum(Network* net, float* x, float* y)
{
  return TrainNetworkDatum(net, x, y);
}

void ref_update(Network* net)
{
  net->curr_iter++;
  UpdateNetwork(net);
}

float ref_curr_lr(Network* net) { return GetCurrLr(net); }
void ref_set_max_iter(Network* net, int max_iter) { net->max_iter = max_iter; }
void ref_set_curr_iter(Network* net, long long iter) { net->curr_iter = iter; }
void ref_save_weights(Network* net, const char* path);

void ref_forward_train(Network* net, float* x, float* y)
{
  NetworkState state = {0};
  state.index = 0;
  state.net = net;
  state.input = x;
  state.delta = 0;
  state.truth = y;
  state.train = 1;
  ForwardNetwork(net, state);
}

}  // extern "C"

void SaveWeights(Network* net, char const* filename);
extern "C" void ref_save_weights(Network* net, const char* path)
{
  SaveWeights(net, path);
}
