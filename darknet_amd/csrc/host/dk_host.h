// dk_host.h -- internal declarations shared by the host-side sources.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "dark_hip.h"
#include "dk_kernels.h"
#include "yolo_core_hip.h"

// ---- utils (reference: src/utils.cpp:107-148) -----------------------------
void* xcalloc(size_t nmemb, size_t size);
void* xrealloc(void* p, size_t size);
[[noreturn]] void error(const char* s);
[[noreturn]] void FileError(const char* s);

// ---- cfg sections (reference: src/parser.cpp:59-100, src/option_list.cpp) --
struct Option
{
  std::string key, val;
  bool used = false;
};
struct Section
{
  std::string type;  // "[convolutional]"
  std::vector<Option> options;
};
bool ReadSections(const char* filename, std::vector<Section>& out);
const char* FindOption(Section& s, const char* key);
const char* FindOptionStr(Section& s, const char* key, const char* def);
const char* FindOptionStrQuiet(Section& s, const char* key, const char* def);
int FindOptionInt(Section& s, const char* key, int def);
int FindOptionIntQuiet(Section& s, const char* key, int def);
float FindOptionFloat(Section& s, const char* key, float def);
float FindOptionFloatQuiet(Section& s, const char* key, float def);
void UnusedOption(Section& s);

ACTIVATION get_activation(const char* s);

// ---- layer construction (reference: Fill*Layer) ----------------------------
struct SizeParams
{
  int batch, inputs, h, w, c, index, train;
  Network* net;
};
void FillConvLayer(layer* l, int batch, int h, int w, int c, int n, int groups, int size,
    int stride_x, int stride_y, int dilation, int padding, ACTIVATION activation,
    int batch_normalize, int index, int train);
void FillMaxpoolLayer(layer* l, int batch, int h, int w, int c, int size, int stride_x,
    int stride_y, int padding, int train);
void FillRouteLayer(layer* l, int batch, int n, int* input_layers, int* input_sizes, int groups,
    int group_id);
void FillShortcutLayer(layer* l, int batch, int index, int w, int h, int c, int from_outputs,
    ACTIVATION activation, int train);
void FillUpsampleLayer(layer* l, int batch, int w, int h, int c, int stride);
void FillDropoutLayer(layer* l, int batch, int inputs, float probability, int w, int h, int c);
void FillYoloLayer(layer* l, int batch, int w, int h, int n, int total, int* mask, int classes,
    int max_boxes);
void FillGaussianYoloLayer(layer* l, int batch, int w, int h, int n, int total, int* mask, int classes, int max_boxes);
int DkGaussianYoloNumDetectionsBatch(layer const* l, int b, float thresh);
int DkGetGaussianYoloDetectionsBatch(layer const* l, int b, int net_w, int net_h, float thresh, Detection* dets, int* ids);

// the buffer readers of layer l's output use (a single-input [route] may alias its source)
inline float* DkLayerOut(const layer* l) { return l->out_alias ? l->out_alias : l->output_gpu; }
bool dk_gpu_enabled();  // true when a HIP device is usable (cuda_get_device() >= 0)

// graph-level options (network.cpp)
extern int g_dk_fusion, g_dk_graph, g_dk_autotune, g_dk_pull_heads;
void DkPlanInference(Network* net);  // fusion pass + autotune + plan creation
void DkInvalidateGraph(Network* net);
void DkTrainPrepRun(Network* net);   // train.cpp: refresh the derived weight tensors (start of a train-mode forward)
void DkFreeTrainPrep(Network* net);
void DkJoinWgradStream(Network* net);   // train.cpp: main stream waits for the weight gradients on the second stream
void DkFreeWgradStream(Network* net);
