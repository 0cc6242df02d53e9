/*
 * yolo_core_hip.h -- public network API of the MI355X-native Darknet conv path.
 *
 * Mirrors the reference's library surface for this path (Ravicmoon/darknet
 * src/yolo_core.h:43-667, src/libapi.h, src/box.h, src/network.h, src/parser.h,
 * src/convolutional_layer.h:7-18 ...): same function names, argument meaning,
 * ownership and error behaviour (loaders return bool; everything else prints
 * and exit()s), so a caller of lib_yolo_core recompiles against this header and
 * relinks against libdarknet_amd.so.  Like the reference's header it is a C++
 * header whose functions have C linkage names.
 *
 * `struct layer` / `Network` keep the reference's FIELD NAMES for everything the
 * YOLOv4-family hot path touches (so `net->layers[i].out_w`, `l->output_gpu`,
 * `l->forward_gpu` ... compile unchanged); fields that only serve subsystems
 * outside the hot path (XNOR, cuDNN descriptors, dropout, local/connected
 * layers, adam, data augmentation) do not exist here.  Binary layout is NOT the
 * reference's: recompile, do not mix object files.  See INTEGRATION.md.
 *
 * Additive extensions (names starting with Dk or ending in Batch) exist because
 * the reference forces batch = 1 for inference (src/parser.cpp:1114) and reads
 * only batch item 0 when extracting detections (src/yolo_layer.cpp:786, :805).
 */
#ifndef YOLO_CORE_HIP_H
#define YOLO_CORE_HIP_H

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
#include <vector>
#endif

#ifndef LIB_API
#define LIB_API __attribute__((visibility("default")))
#endif

#define SECRET_NUM -1234

/* ---- enums (values identical to the reference) ------------------------- */
typedef enum
{ /* src/yolo_core.h:69-92 */
  LOGISTIC, RELU, RELU6, RELIE, LINEAR, RAMP, TANH, PLSE, LEAKY, ELU, LOGGY,
  STAIR, HARDTAN, LHTAN, SELU, GELU, SWISH, MISH, NORM_CHAN, NORM_CHAN_SOFTMAX,
  NORM_CHAN_SOFTMAX_MAXVAL
} ACTIVATION;

typedef enum
{ /* src/yolo_core.h:112-138 */
  CONVOLUTIONAL, CONNECTED, MAXPOOL, LOCAL_AVGPOOL, DETECTION, DROPOUT, CROP,
  ROUTE, COST, AVGPOOL, LOCAL, SHORTCUT, SCALE_CHANNELS, ACTIVE, BATCHNORM,
  NETWORK, XNOR, YOLO, GAUSSIAN_YOLO, REORG, REORG_OLD, UPSAMPLE, EMPTY, BLANK
} LAYER_TYPE;

typedef enum { IOU, GIOU, MSE, DIOU, CIOU } IOU_LOSS;      /* src/box.h:6-13 */
typedef enum { GREEDY_NMS, DIOU_NMS } NMS_KIND;            /* src/box.h:15-19 */

/* yolo_core.h:94-100 */
typedef enum
{
  YOLO_CENTER = 1 << 0,
  YOLO_LEFT_TOP = 1 << 1,
  YOLO_RIGHT_BOTTOM = 1 << 2
} YOLO_POINT;
typedef enum
{ /* src/yolo_core.h:440-450 */
  CONSTANT, STEP, EXP, POLY, STEPS, SIG, RANDOM, SGDR
} LearningRatePolicy;

/* ---- boxes / detections (src/box.h:33-97) ------------------------------ */
#ifdef __cplusplus
class LIB_API Box
{
 public:
  Box() : x(0), y(0), w(0), h(0) {}
  Box(float _x, float _y, float _w, float _h) : x(_x), y(_y), w(_w), h(_h) {}
  static float Overlap(float x1, float w1, float x2, float w2);
  static float Intersect(Box const& b1, Box const& b2);
  static float Union(Box const& b1, Box const& b2);
  static float Iou(Box const& b1, Box const& b2);
  static float Diou(Box const& b1, Box const& b2, float beta = 0.6f);
  float x, y, w, h;
};
#else
typedef struct Box { float x, y, w, h; } Box;
#endif

typedef struct Detection
{
  Box bbox;          /* centre-normalised x, y, w, h */
  int classes;
  float* prob;       /* [classes] */
  float* mask;
  float objectness;
  int sort_class;
  float* uc;
  int points;
} Detection;

typedef struct MostProbDet
{
  Box bbox;
  int cid;
  float prob;
} MostProbDet;

/* ---- training data (src/yolo_core.h:561-576) ----------------------------- */
typedef struct matrix
{
  int rows, cols;
  float** vals;
} matrix;

typedef struct data
{
  int w, h;
  matrix X;
  matrix y;
  int shallow;
  int* num_boxes;
  Box** boxes;
} data;

/* ---- layer / network (field names: src/yolo_core.h:149-558) ------------- */
struct layer;
typedef struct layer layer;
struct Network;
typedef struct Network Network;
struct NetworkState;
typedef struct NetworkState NetworkState;

typedef struct NetworkState
{
  float* truth;
  float* input;      /* device pointer on the *_gpu path */
  float* delta;
  float* workspace;
  int train;
  int index;
  Network* net;
} NetworkState;

struct layer
{
  LAYER_TYPE type;
  ACTIVATION activation;
  /* plugin slots, src/yolo_core.h:154-159 */
  void (*forward)(struct layer*, struct NetworkState);
  void (*backward)(struct layer*, struct NetworkState);
  void (*update)(struct layer*, int, float, float, float);
  void (*forward_gpu)(struct layer*, struct NetworkState);
  void (*backward_gpu)(struct layer*, struct NetworkState);
  void (*update_gpu)(struct layer*, int, float, float, float, float);

  layer* share_layer;
  int train;
  int batch_normalize;
  int batch;
  int steps;
  int inputs, outputs;
  int nweights, nbiases;
  int truths;
  int h, w, c;
  int out_h, out_w, out_c;
  int n;
  int max_boxes;
  int groups, group_id;
  int size;
  int stride, stride_x, stride_y;
  int dilation;
  int pad;
  int index;
  int reverse;
  float scale;
  float bflops;
  int scale_wh;          /* [scale_channels]: 1 = one scale per pixel instead of per channel */
  float probability;     /* [dropout] */

  /* yolo */
  int classes, total;
  int* mask;
  float scale_x_y, max_delta, iou_normalizer, cls_normalizer, ignore_thresh,
      truth_thresh, iou_thresh, jitter, random, label_smooth_eps, beta_nms;
  int focal_loss;
  float* classes_multipliers;
  IOU_LOSS iou_loss, iou_thresh_kind;
  NMS_KIND nms_kind;
  YOLO_POINT yolo_point; /* [Gaussian_yolo]: which point of the box the head predicts */
  float uc_normalizer;   /* [Gaussian_yolo]: weight of the uncertainty gradients */
  float* rand_gpu;       /* [dropout] train mode: the uniform draws of the last forward (mask of the backward) */
  int* map;

  /* per-layer common keys, src/parser.cpp:1361-1369 */
  float clip, learning_rate_scale;
  int onlyforward, stopbackward, train_only_bn, dont_update, burnin_update,
      dontload, dontloadscales;

  /* route / shortcut */
  int* input_layers;
  int* input_sizes;

  /* host tensors */
  int* indexes;
  float* cost;
  float *biases, *bias_updates;
  float *scales, *scale_updates;
  float *weights, *weight_updates;
  float *delta, *output, *activation_input;
  int delta_pinned, output_pinned;
  float *mean, *variance, *mean_delta, *variance_delta;
  float *rolling_mean, *rolling_variance;
  float *x, *x_norm;
  size_t workspace_size;

  /* device tensors */
  int* indexes_gpu;
  float *mean_gpu, *variance_gpu, *rolling_mean_gpu, *rolling_variance_gpu;
  float *variance_delta_gpu, *mean_delta_gpu;
  float *x_gpu, *x_norm_gpu;
  float *weights_gpu, *weight_updates_gpu;
  float *biases_gpu, *bias_updates_gpu;
  float *scales_gpu, *scale_updates_gpu;
  float *output_gpu, *activation_input_gpu, *delta_gpu;

  /* adam (convolutional_layer.cpp:589-620): per-tensor first / second moments, iteration counter */
  int adam, t;
  float B1, B2, eps;
  float *m_gpu, *v_gpu, *bias_m_gpu, *bias_v_gpu, *scale_m_gpu, *scale_v_gpu;

  /* MI355X-native additions */
  int fused_into_prev;   /* shortcut folded into the previous conv's epilogue */
  int fuse_residual_from; /* conv: layer index whose output is added in the epilogue, or -1 */
  int conv_cfg;          /* tile configuration chosen by the autotuner, or -1 */
  /* zero-copy concatenation (inference plan): a producer whose only reader is a multi-input
   * [route] writes straight into that route's buffer; a single-input [route] is an alias */
  /* two 1x1 convolutions of the same tensor (the two branches of a CSP stage) in one launch:
   * the first conv carries the concatenated weights/biases and writes both outputs */
  int dual_with;         /* index of the second conv, or 0 */
  int dual_slave;        /* 1: computed by an earlier layer's dual launch */
  float *dual_weights_gpu, *dual_biases_gpu;
  struct layer* dual_peer; /* the other conv of a dual launch (master <-> slave), or NULL */
  void* weights_half_gpu; /* fp16 weights packed for conv3x3_direct_f16 (inference plan with cudnn_half), or NULL */
  float* weights_wino_gpu; /* filters transformed for conv3x3_wino_f32 (inference plan, DkSetWinograd), or NULL */
  int train_plan[3];     /* train step: forward / data-gradient / weight-gradient tile configuration + 2, timed on the
                          * layer's own tensors at the first step (1 = the heuristic, 0 = not chosen yet) */
  /* train: derived weight tensors refreshed by ONE launch per step (DkTrainPrepRun) instead of a transpose /
   * filter-transform launch per layer and pass: the data gradient's transposed weights, the Winograd filters of
   * the forward convolution and of the data-gradient convolution; NULL = made on the fly */
  float *train_wt_gpu, *train_u_fwd_gpu, *train_u_dgrad_gpu;
  int delta_in_arena;    /* delta_gpu points into net->delta_arena_gpu (not freed per layer) */
  int buffers_aliased;   /* [dropout]: output_gpu / delta_gpu are the previous layer's (not freed here) */
  float* out_view;       /* producer: where the output really goes (a channel slice), or NULL */
  int out_view_ctot;     /* producer: channels of the tensor out_view is a slice of */
  float* out_alias;      /* route with one input: the source's buffer, or NULL */
  int* input_inplace;    /* route: per input, 1 = the producer already wrote it in place */
  float* injected_delta; /* yolo (tests): host delta used instead of the loss, see DkSetYoloDelta */
  void* loss_task;       /* yolo (train): host loss of the current step, running beside the GPU */
};

struct Network
{
  int max_epoch, max_iter;
  int n;
  int batch, subdiv;
  uint64_t seen;
  int curr_iter;
  float loss_scale;
  layer* layers;
  float* output;
  LearningRatePolicy policy;
  int benchmark_layers;
  float lr, lr_min;
  int sgdr_cycle, sgdr_mult;
  float momentum, decay, gamma, scale, power;
  int step;
  float *steps, *scales;
  int num_steps;
  int burn_in;
  int cudnn_half;
  int adam;
  float B1, B2, eps; /* adam */
  int inputs, outputs, truths;
  int h, w, c;
  int curr_subdiv;
  int gpu_index;
  float* input;
  float* truth;
  float* workspace;
  int train;
  float* cost;
  float* input_state_gpu;
  float* input_pinned_cpu;
  int input_pinned_cpu_flag;
  float** input_gpu;
  float** truth_gpu;
  int wait_stream;
  size_t workspace_size_limit;

  /* MI355X-native additions */
  void* graph_exec;      /* hipGraphExec_t of the captured forward, or NULL */
  int graph_batch;
  float* wt_scratch_gpu; /* transposed weights of the layer whose data gradient is running */
  float* wino_scratch_gpu; /* train: Winograd-transformed filters of the layer that is running (they change every step) */
  void* wgrad_stream;    /* train: second HIP stream + events the weight gradients run on (off the backward critical path) */
  void* train_prep;      /* train: plan of the per-step derived-weights launch (built after the first step's kernel choices) */
  int train_steps;       /* train: forward passes run in train mode */
  /* device-side detection extraction and u8 input staging (see DkSetPullHeads, DkNetworkPredictU8) */
  float* cand_gpu;       /* candidate records written by dk_yolo_compact */
  int* cand_counter_gpu;
  float* cand_host;      /* pinned mirror of the records */
  int* cand_order;       /* record indices in the reference's scan order (image, layer, anchor, cell) */
  int cand_cap, cand_count, cand_rec, cand_fallback, cand_valid;
  long cand_seq, predict_seq;
  float cand_thresh;
  float cand_nms;        /* NMS threshold the candidate records were suppressed with on the device (0: raw) */
  int cand_nms_done;
  void* nms_heads_gpu;   /* DkYoloHead[n] + overflow word (device NMS) */
  unsigned char* u8_gpu; /* interleaved u8 frames on the device (staging slot 0) */
  unsigned char* u8_pinned;
  size_t u8_bytes;
  /* second staging slot + events of the double-buffered input step (DkNetworkStageU8 / DkNetworkPredictStaged) */
  unsigned char* u8_gpu2;
  unsigned char* u8_pinned2;
  void* u8_h2d_ev[2];   /* H2D of slot i finished (copy stream) */
  void* u8_conv_ev[2];  /* conversion kernel that read slot i finished (compute stream) */
  int u8_conv_pending[2];
  int u8_next, u8_staged; /* slot the next stage call fills; slot staged and not yet consumed (-1: none) + 1 */
  size_t u8_row_step;
  int u8_src_w, u8_src_h, u8_swap_rb; /* staged frames at another resolution (DkNetworkStageFrames): 0 = network size */
  /* float frames staged ahead of the forward that consumes them (DkNetworkStageFloat) */
  float* f32_stage_gpu;
  float* f32_stage_pinned;
  size_t f32_stage_floats;
  void* f32_h2d_ev;      /* H2D of the staged batch finished (staging stream) */
  void* f32_copy_ev;     /* the device copy stage -> input tensor finished (compute stream) */
  int f32_copy_pending, f32_staged;
  void* stage_stream;    /* H2D of staged inputs: a stream of its own, not queued behind the heads' D2H */
  float* delta_arena_gpu; /* train: every layer's delta_gpu lives in this one allocation (one memset per step) */
  size_t delta_arena_size;
  size_t delta_arena_zero; /* leading floats of the arena that must start a step at zero (the rest is overwritten before it is read) */
  float* grad_bucket;    /* caller-owned contiguous gradient bucket (DkAttachGradBucket), or NULL */
  int grad_replicas;     /* data-parallel replicas whose buckets are summed before the update (DkSetReplicas) */
  void* dp;              /* data-parallel state of this replica (TrainNetworks: bucket, RCCL communicator, streams) */
  void* fwd_done_ev;     /* NetworkPredictDevice: forward finished (hipEvent_t) */
  void* copy_done_ev;    /* NetworkPredictDevice: head copies of the previous call finished */
  int copy_pending;
  int pull_in_forward;   /* NetworkPredictDevice -> ForwardNetworkGpu: copy every yolo head to the host as soon as its layer has run */
  void* head_ev[8];      /* one event per pulled head (hipEvent_t) */
  int opt_graph, opt_pull_heads; /* per-network overrides of DkSetGraph / DkSetPullHeads (0/1), -1... stored +1: 0 = follow the process-wide setting */
  void* sgd_plan;        /* multi-tensor SGD plan (one launch per update), or NULL */
  int planned;           /* DkPlanInference ran (fusion plan, packed/dual weight copies exist) */
  int backward_stopped;  /* train: a stopbackward layer ended this step's backward sweep (DkBackwardRange segments) */
};

#ifdef __cplusplus
extern "C" {
#endif

/* ---- loader: src/parser.cpp:1852-1876, :1076-1519, :1590-1643, :1778-1850 */
LIB_API bool LoadNetwork(Network* net, char const* model_file,
    char const* weights_file, bool train
#ifdef __cplusplus
    = false
#endif
    , bool clear
#ifdef __cplusplus
    = false
#endif
);
LIB_API void FreeNetwork(Network* net);
LIB_API bool ParseNetworkCfg(Network* net, char const* filename, bool train);
LIB_API bool LoadWeights(Network* net, char const* filename);
LIB_API bool LoadWeightsUpTo(Network* net, char const* filename, int cutoff);
LIB_API void SaveWeights(Network* net, char const* filename);
LIB_API void SaveWeightsUpto(Network* net, char const* filename, int cutoff);
LIB_API void free_layer(layer* l, bool keep_cudnn_desc
#ifdef __cplusplus
    = false
#endif
);

/* ---- graph engine: src/network.cpp:412-516, :647-682;
 *      src/network_kernels.cu:45-114, :486-522 ------------------------------ */
LIB_API float* NetworkPredict(Network* net, float* input);
LIB_API Detection* GetNetworkBoxes(Network* net, float thresh, int* num);
LIB_API Detection* MakeNetworkBoxes(Network* net, float thresh, int* num);
LIB_API void FreeDetections(Detection* dets, int n);
/* src/network.cpp:518-592: one frame's detections as the reference's JSON text (malloc()ed; the caller frees) */
LIB_API char* Detection2Json(Detection* dets, int nboxes, int classes, char** names, long long int frame_id,
    char const* filename);
LIB_API void FuseConvBatchNorm(Network* net);
/* src/network.cpp:255-410: new input resolution; every layer re-derives its geometry and
 * re-allocates its tensors, plans / tap tables / the captured graph are rebuilt */
LIB_API void ResizeNetwork(Network* net, int w, int h);
LIB_API void ForwardNetworkGpu(Network* net, NetworkState state);
LIB_API float* NetworkPredictGpu(Network* net, float* input);
LIB_API float* GetNetworkOutputGpu(Network* net);
LIB_API int GetNetworkInputSize(Network* net);
LIB_API int GetNetworkOutputSize(Network* net);
LIB_API float GetCurrLr(Network* net);

/* ---- training: src/network.cpp:116-239, src/network_kernels.cu:116-293,
 *      src/convolutional_kernels.cu:555-921 -------------------------------- */
LIB_API float TrainNetworkDatum(Network* net, float* x, float* y);
LIB_API float TrainNetworkDatumGpu(Network* net, float* x, float* y);
LIB_API void ForwardBackwardNetworkGpu(Network* net, float* x, float* y);
LIB_API void BackwardNetworkGpu(Network* net, NetworkState state);
LIB_API void UpdateNetworkGpu(Network* net);
LIB_API void UpdateNetwork(Network* net);
LIB_API void BackwardConvolutionalLayerGpu(layer* l, NetworkState state);
LIB_API void UpdateConvolutionalLayerGpu(layer* l, int batch, float learning_rate,
    float momentum, float decay, float loss_scale);
LIB_API void BackwardMaxpoolLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardRouteLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardShortcutLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardUpsampleLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardYoloLayerGpu(layer* l, NetworkState state);
/* multi-GPU data parallelism in C (src/network.cpp:210-239, src/network_kernels.cu:398-484): one host
 * thread + stream + RCCL communicator per GPU; the replicas' gradient buckets are all-reduced (sum)
 * every iteration, overlapped with the backward pass, instead of the reference's weight averaging
 * every `sync_interval` iterations (accepted, unused).  `nets` is an array of `num_gpus` Network
 * structs, each parsed with train = true after cuda_set_device(its GPU); d holds
 * batch x subdivisions x num_gpus rows. */
LIB_API float TrainNetwork(Network* net, data d);
LIB_API float TrainNetworks(Network* nets, int num_gpus, data d, int sync_interval);
LIB_API void SyncNetworks(Network* nets, int num_gpus);
LIB_API void get_next_batch(data d, int n, int offset, float* X, float* y);
LIB_API data GetPartialData(data d, int idx, int num_split);
/* flat helpers for FFI callers */
LIB_API Network* DkNetworkArrayCreate(int n);
LIB_API Network* DkNetworkArrayAt(Network* nets, int i);
LIB_API void DkNetworkArrayDestroy(Network* nets, int n);
LIB_API float DkTrainNetworksFlat(Network* nets, int num_gpus, float* X, int x_cols, float* y, int y_cols,
    int rows, int sync_interval);
/* backward-order slices of the gradient bucket used by the overlapped all-reduce:
 * out[4i..4i+3] = hi, lo, offset, count; returns the number of slices */
LIB_API int DkBucketSegments(Network* net, int nseg, long long* out, int max_segs);
/* per-network overrides of the process-wide DkSetGraph / DkSetPullHeads (thread-safe use:
 * one thread per network); value < 0 returns to the process-wide setting */
LIB_API void DkNetSetGraph(Network* net, int on);
LIB_API void DkNetSetPullHeads(Network* net, int on);
/* additive: drive the backward pass from a given yolo-layer delta (host array of
 * batch*outputs floats, kept by reference) instead of the yolo loss */
LIB_API void DkSetYoloDelta(Network* net, int layer_index, float* host_delta);
LIB_API void DkSetMaxIter(Network* net, int max_iter);
/* data-parallel training: all conv gradients in one caller-owned device bucket
 * (weight_updates, bias_updates, scale_updates per conv, layer order) that the
 * caller all-reduces (RCCL) between TrainNetworkDatum and UpdateNetworkGpu */
LIB_API size_t DkGradBucketSize(Network* net);
LIB_API void DkAttachGradBucket(Network* net, float* bucket);
LIB_API void DkSetSubdivisions(Network* net, int subdiv);
/* R data-parallel replicas (one process per GPU) whose gradient buckets are all-reduced (summed)
 * between the backward pass and UpdateNetworkGpu: B = batch x R in the update, and the momentum
 * carry-over that UpdateNetworkGpu leaves in the bucket is kept as 1/R per replica, so that the
 * next all-reduce restores it exactly once. */
LIB_API void DkSetReplicas(Network* net, int replicas);
/* Split train step for overlapping the bucket all-reduce with the backward pass:
 * DkTrainForward (input push, forward, host losses started), DkBackwardRange(net, hi, lo) =
 * backward of layers hi-1 ... lo, DkTrainFinish = joins, returns the cost.  DkGradBucketOffset(net, i)
 * = offset (floats) of layer i's first gradient in the bucket (or of the next conv above it). */
LIB_API void DkTrainForward(Network* net, float* x, float* y);
LIB_API void DkBackwardRange(Network* net, int hi, int lo);
LIB_API float DkTrainFinish(Network* net);
LIB_API size_t DkGradBucketOffset(Network* net, int upto_layer);
LIB_API void DkAdvanceIteration(Network* net);
LIB_API void DkSetDeterministic(int on); /* 1: ordered reductions instead of atomics in the train step (dk_set_deterministic) */
LIB_API void DkSetTrainStreams(int on); /* 1 (default): weight gradients on the network's second stream; 0: one stream */
LIB_API void DkSetTrainPrep(int on); /* 1 (default): derived weight tensors of a train step in one launch; 0: per layer */
LIB_API void DkSetCurrIter(Network* net, long long iter); /* net->curr_iter = iter (GetCurrLr / burn-in / stopbackward schedules) */
/* D2H copy of a layer tensor: which = 6 delta, 7 weight_updates, 8 bias_updates,
 * 9 scale_updates, 1 weights, 2 biases, 3 scales, 4 rolling_mean, 5 rolling_variance,
 * 10 mean, 11 variance; returns the element count or -1 */
LIB_API long DkLayerPull(Network* net, int i, int which, float* dst, size_t n);

/* ---- post-processing kept as host C++: src/box.cpp:372-447 --------------- */
LIB_API void NmsSort(Detection* dets, int total, int classes, float thresh,
    NMS_KIND nms_kind, float beta);

/* ---- layer plugin entry points (the *_gpu slots), reference twins:
 *      src/convolutional_kernels.cu:252-553, :817-863; src/maxpool_layer_kernels.cu
 *      :145-240; src/route_layer.c:124-160; src/shortcut_layer.c:190-204;
 *      src/upsample_layer.c:106-133; src/yolo_layer.cpp:836-888 ------------- */
LIB_API void ForwardConvolutionalLayerGpu(layer* l, NetworkState state);
LIB_API void PushConvolutionalLayer(layer* l);
LIB_API void PullConvolutionalLayer(layer* l);
LIB_API void add_bias_gpu(float* output, float* biases, int batch, int n, int size);
LIB_API void backward_bias_gpu(float* bias_updates, float* delta, int batch, int n, int size);
/* upstream-darknet spellings of the same slots */
LIB_API void forward_convolutional_layer_gpu(layer* l, NetworkState state);
LIB_API void backward_convolutional_layer_gpu(layer* l, NetworkState state);
LIB_API void update_convolutional_layer_gpu(layer* l, int batch, float learning_rate, float momentum,
    float decay, float loss_scale);
LIB_API void init_cpu(void);
/* sibling-cfg layer kinds: src/batchnorm_layer.cpp:9-88, :268-425; src/avgpool_layer.cpp:6-72;
 * src/scale_channels_layer.c:9-160 */
LIB_API void FillBatchnormLayer(layer* l, int batch, int w, int h, int c, int train);
LIB_API void ForwardBatchnormLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardBatchnormLayerGpu(layer* l, NetworkState state);
LIB_API void UpdateBatchnormLayerGpu(layer* l, int batch, float learning_rate, float momentum, float decay,
    float loss_scale);
LIB_API void PushBatchnormLayer(layer* l);
LIB_API void PullBatchnormLayer(layer* l);
LIB_API void FillAvgpoolLayer(layer* l, int batch, int w, int h, int c);
LIB_API void ForwardAvgpoolLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardAvgpoolLayerGpu(layer* l, NetworkState state);
LIB_API void FillScaleChannelsLayer(layer* l, int batch, int index, int w, int h, int c, int w2, int h2,
    int c2, int scale_wh);
LIB_API void ForwardScaleChannelsLayerGpu(layer* l, NetworkState state);
LIB_API void BackwardScaleChannelsLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardMaxpoolLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardRouteLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardShortcutLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardUpsampleLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardYoloLayerGpu(layer* l, NetworkState state);
LIB_API void ForwardGaussianYoloLayerGpu(layer* l, NetworkState state);   /* inference; src/gaussian_yolo_layer.cpp:934 */
LIB_API void BackwardGaussianYoloLayerGpu(layer* l, NetworkState state);
/* host loss of the Gaussian head (src/gaussian_yolo_layer.cpp:518-851): fills delta, returns the cost */
LIB_API float DkGaussianYoloLossHost(const layer* l, int net_w, int net_h, float* out, const float* truth, float* delta);
LIB_API int YoloNumDetections(layer const* l, float thresh);
LIB_API int GetYoloDetections(layer const* l, int net_w, int net_h, float thresh, Detection* dets);

/* ---- additive extensions ----------------------------------------------- */
/* Inference load with batch > 1 (BN folded exactly like LoadNetwork(train=false)). */
LIB_API bool LoadNetworkBatch(Network* net, char const* model_file,
    char const* weights_file, int batch);
/* Forward from an input already resident in HBM (net->input_state_gpu when
 * input_gpu == NULL); enqueues only, yolo heads are copied to pinned host
 * memory asynchronously; NetworkSync() waits for them. */
LIB_API void NetworkPredictDevice(Network* net, float* input_gpu);
LIB_API void NetworkSync(Network* net);
/* Batch-aware detection extraction (GetNetworkBoxes reads batch item 0). */
LIB_API Detection* GetNetworkBoxesBatch(Network* net, int b, float thresh, int* num);
/* Graph-level options: conv+shortcut epilogue fusion (default on for inference
 * loads), hipGraph replay of the forward (default on), tile autotune at load. */
LIB_API void DkSetFusion(int on);
LIB_API void DkSetGraph(int on);
LIB_API void DkSetAutotune(int on);
/* fp16-operand convolutions for the layers the reference's CUDNN_HALF rule admits
 * (sets net->cudnn_half at the next load; also env DK_HALF=1) */
LIB_API void DkSetHalf(int on);
/* 1 (default; env DK_WINOGRAD): the inference plan may run 3x3/s1 layers with the fused Winograd F(2x2,3x3)
 * kernel where the per-layer autotune measures it faster (results within the fp32 tolerance, not bitwise
 * equal to the direct kernels); 0: only the k-ascending implicit-GEMM kernels. */
LIB_API void DkSetWinograd(int on);

/* Flat helpers for FFI callers (ctypes, tests, bench.py) */
LIB_API Network* DkNetworkCreate(void);            /* calloc'ed Network */
LIB_API void DkNetworkDestroy(Network* net);       /* FreeNetwork + free */
LIB_API float* DkNetworkInputGpu(Network* net);
LIB_API void DkNetworkInfo(Network* net, int* out /* [8]: n,batch,w,h,c,inputs,outputs,gpu_index */);
LIB_API void DkLayerInfo(Network* net, int i, int* out /* [24], see tests/reflib.py INFO */);
LIB_API float DkLayerBflops(Network* net, int i);
LIB_API int DkLayerOutput(Network* net, int i, float* dst, size_t n); /* D2H copy of output_gpu */
/* Mat2Image + NetworkPredict for `net->batch` interleaved u8 frames (h rows of row_step bytes each,
 * c channels, already resized to net->w x net->h): the frames cross PCIe as bytes and are
 * converted on the device (visualize.cpp:26-55 arithmetic).  Heads stay on the device; use
 * GetNetworkBoxesBatch (device extraction, DkSetPullHeads(0)) or DkSetPullHeads(1). */
LIB_API void DkNetworkPredictU8(Network* net, const unsigned char* frames_hwc, size_t row_step);
/* The same in two halves, double-buffered: DkNetworkStageU8 copies the NEXT batch of frames to pinned memory
 * and starts its H2D on the copy stream (call it right after launching the current prediction: both overlap the
 * forward pass that is running); DkNetworkPredictStaged converts the staged frames on the device and runs the
 * forward.  DkNetworkPredictU8 == Stage + PredictStaged. */
LIB_API void DkNetworkStageU8(Network* net, const unsigned char* frames_hwc, size_t row_step);
/* Frames at ANY resolution: cv::resize(INTER_LINEAR) [+ cvtColor(RGB2BGR) when swap_rb] + Mat2Image happen on the
 * device in one kernel (dk_image_resize_u8_to_chw) when DkNetworkPredictStaged consumes them: the whole input step
 * of the reference's ProcImage (src/yolo_core.cpp:104-112). */
LIB_API void DkNetworkStageFrames(Network* net, const unsigned char* frames_hwc, int src_w, int src_h, size_t row_step,
    int swap_rb);
LIB_API void DkNetworkPredictStaged(Network* net);
/* float CHW frames (what NetworkPredict takes) staged AHEAD: pinned copy by a few host threads + H2D on the staging
 * stream into a second device buffer, while the previous batch's forward runs; DkNetworkPredictStaged then copies it
 * into the input tensor on the device (71 MB in ~30 us) and runs the forward.  Loop: PredictStaged(k); StageFloat(k + 1);
 * read the results of k. */
LIB_API void DkNetworkStageFloat(Network* net, const float* input);
LIB_API float* DkLayerOutputGpu(Network* net, int i);
LIB_API float* DkLayerHostPtr(Network* net, int i, int which); /* 1 weights 2 biases 3 scales 4 mean 5 var */
/* Flattened detections of batch item b: per det [x,y,w,h,obj,prob[classes]] and
 * ids [layer, anchor, row, col]; returns the count (writes at most max_dets). */
LIB_API int DkGetBoxesBatch(Network* net, int b, float thresh, float* dets, int* ids, int max_dets);
/* GetNetworkBoxes + NmsSort with the suppression on the device (src/box.cpp:393-419; heads must stay
 * in HBM: DkSetPullHeads(0)): every candidate of image b in scan order with post-NMS probabilities */
LIB_API Detection* DkGetNetworkBoxesNms(Network* net, int b, float thresh, float nms, int* num);
LIB_API int DkGetBoxesBatchNms(Network* net, int b, float thresh, float nms, float* dets, int* ids, int max_dets);
LIB_API size_t DkWeightsFileSize(Network* net);

/* mAP of ValidateDetector (src/detector.cpp:326-562) on flattened inputs: per image n_dets[i]
 * detections AFTER NmsSort as [x, y, w, h, prob[classes]] and n_gts[i] labels as [id, x, y, w, h];
 * ap_out (may be NULL) receives the per-class APs. */
LIB_API double DkMeanAveragePrecision(int n_images, const int* n_dets, const float* dets, const int* n_gts,
    const float* gts, int classes, float iou_thresh, double* ap_out);
/* flat forms of the headless harness below (data_file = the reference's .data format) */
LIB_API float DkValidateDetectorFlat(const char* data_file, Network* net, float iou_thresh, float thresh, float nms);
LIB_API void DkTrainDetectorFlat(const char* data_file, const char* cfg, const char* weights, int num_gpus, int clear,
    int calc_map, int max_iterations, int save_every, float map_thresh);
LIB_API int GetCurrIter(Network* net);

#ifdef __cplusplus
}
LIB_API std::vector<MostProbDet> GetMostProbDets(Detection* dets, int num_dets);

/* ---- headless trainer / evaluator harness (src/detector.cpp:27-562,
 *      src/option_list.h:8-30, src/data.cpp:78-115).  Images are binary PPMs at the network's
 *      resolution (no OpenCV, no augmentation); labels and checkpoints are the reference's formats. */
#include <string>
class LIB_API Metadata
{
 public:
  Metadata() : classes_(0) {}
  explicit Metadata(std::string filename) : classes_(0) { Get(filename); }
  bool Get(std::string filename);
  int NumClasses() const { return classes_; }
  std::string TrainFile() const { return train_file_; }
  std::string ValFile() const { return val_file_; }
  std::string NameFile() const { return name_file_; }
  std::string SaveDir() const { return save_dir_; }
  std::vector<std::string> TrainImgList() const { return train_img_list_; }
  std::vector<std::string> ValImgList() const { return val_img_list_; }
  std::vector<std::string> NameList() const { return name_list_; }

 private:
  int classes_;
  std::string train_file_, val_file_, name_file_, save_dir_;
  std::vector<std::string> train_img_list_, val_img_list_, name_list_;
};
typedef struct BoxLabel
{
  int id;
  float x, y, w, h;
  float left, right, top, bottom;
} BoxLabel;
LIB_API std::vector<BoxLabel> ReadBoxAnnot(std::string filename);
LIB_API std::string ReplaceImage2Label(std::string str);
/* The reference declares these two INSIDE its extern "C" block (yolo_core.h:43-45, 640-644), so a binary built
 * against it binds the unmangled names `TrainDetector` / `ValidateDetector`: same linkage here. */
extern "C" {
LIB_API float ValidateDetector(Metadata const& md, Network* net, float const iou_thresh);
LIB_API void TrainDetector(Metadata const& md, std::string model_file, std::string weights_file, int num_gpus,
    bool clear, bool show_imgs, bool calc_map, int benchmark_layers);
}
/* the same with the constants of the reference exposed (thresh .005, nms .45 there) and bounded runs */
LIB_API float DkValidateDetector(Metadata const& md, Network* net, float iou_thresh, float thresh, float nms);
LIB_API void DkTrainDetector(Metadata const& md, std::string model_file, std::string weights_file, int num_gpus,
    bool clear, bool calc_map, int max_iterations, int save_every, float map_thresh);
#endif

#endif /* YOLO_CORE_HIP_H */
