// multigpu.cpp -- C-level data-parallel training over RCCL: TrainNetwork / TrainNetworks /
// SyncNetworks with the reference's signatures.
//
// Reference twins (Ravicmoon/darknet src/): TrainNetwork network.cpp:210-239, TrainNetworks
// network_kernels.cu:446-484 (one pthread per GPU, each training its own replica on its
// GetPartialData shard), SyncNetworks :398-427 (host-mediated weight averaging every
// `sync_interval` iterations, per layer: PullWeights / MergeWeights / ScaleWeights / PushWeights
// :295-356), get_next_batch / GetPartialData data.cpp:879-901.
//
// MI355X design: one host thread + one compute stream + one RCCL
// communicator per GPU inside this process (ncclCommInitAll).  Replicas stay IDENTICAL: every
// iteration the replicas' gradient buckets (all conv weight/bias/scale updates, one contiguous
// fp32 allocation per GPU: DkAttachGradBucket) are all-reduced (sum) over xGMI and every replica
// applies the same SGD step with B = batch x subdivisions x GPUs.  The all-reduce is cut into a
// few slices in backward order; a slice is handed to RCCL (on a communication stream ordered
// behind the compute stream by an event) as soon as the backward pass has finished the layers
// that own it, so the collective runs beside the rest of the backward pass.  This is the
// synchronous form of the reference's lossy averaging, hence `sync_interval` has nothing left to
// do (weights never diverge); SyncNetworks still exists for callers that want the per-replica
// batch-norm rolling statistics (which DO differ, as in the reference) averaged before saving.
//
// RCCL is loaded with dlopen at the first multi-GPU call, so the library itself carries no
// load-time dependency on librccl (the CPU-only test container loads it without a GPU stack).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "dk_host.h"
#include "dk_internal.h"

// ---------------------------------------------------------------------------
// data helpers (data.cpp:879-901)
// ---------------------------------------------------------------------------
void get_next_batch(data d, int n, int offset, float* X, float* y)
{
  for (int j = 0; j < n; ++j)
  {
    const int index = offset + j;
    memcpy(X + (size_t)j * d.X.cols, d.X.vals[index], d.X.cols * sizeof(float));
    if (y && d.y.vals)
      memcpy(y + (size_t)j * d.y.cols, d.y.vals[index], d.y.cols * sizeof(float));
  }
}

data GetPartialData(data d, int idx, int num_split)
{
  data p;
  memset(&p, 0, sizeof(p));
  p.shallow = 1;
  p.X.rows = d.X.rows / num_split;
  p.y.rows = d.y.rows / num_split;
  p.X.cols = d.X.cols;
  p.y.cols = d.y.cols;
  p.X.vals = d.X.vals + (size_t)d.X.rows * idx / num_split;
  p.y.vals = d.y.vals + (size_t)d.y.rows * idx / num_split;
  return p;
}

// ---------------------------------------------------------------------------
// bucket segmentation (same rule as darknet_amd/train_dist.py: bucket_segments)
// ---------------------------------------------------------------------------
struct DkSegment
{
  int hi, lo;          // backward of layers hi-1 .. lo
  size_t off, cnt;     // finalises bucket[off, off+cnt)
};

static std::vector<DkSegment> bucket_segments(Network* net, int nseg)
{
  std::vector<int> convs;
  std::vector<size_t> sizes, offs(1, 0);
  for (int i = 0; i < net->n; ++i)
  {
    const size_t a = DkGradBucketOffset(net, i), b = DkGradBucketOffset(net, i + 1);
    if (b > a)
    {
      convs.push_back(i);
      sizes.push_back(b - a);
      offs.push_back(offs.back() + (b - a));
    }
  }
  std::vector<DkSegment> segs;
  if (nseg < 1)
    nseg = 1;
  const double total = (double)offs.back(), target = total / nseg;
  int hi = net->n;
  size_t cnt_hi = convs.size();
  double acc = 0;
  for (int k = (int)convs.size() - 1; k >= 0; --k)
  {
    acc += (double)sizes[k];
    if (acc >= target && k > 0 && (int)segs.size() < nseg - 1)
    {
      const int lo = convs[k];
      segs.push_back({hi, lo, offs[k], offs[cnt_hi] - offs[k]});
      hi = lo;
      cnt_hi = (size_t)k;
      acc = 0;
    }
  }
  segs.push_back({hi, 0, 0, offs[cnt_hi]});
  std::vector<DkSegment> out;
  for (auto& s : segs)
    if (s.hi > s.lo)
      out.push_back(s);
  return out;
}

// flat view for tests: out[4*i .. 4*i+3] = hi, lo, off, cnt; returns the number of segments
extern "C" LIB_API int DkBucketSegments(Network* net, int nseg, long long* out, int max_segs)
{
  std::vector<DkSegment> s = bucket_segments(net, nseg);
  for (size_t i = 0; i < s.size() && (int)i < max_segs; ++i)
  {
    out[4 * i + 0] = s[i].hi;
    out[4 * i + 1] = s[i].lo;
    out[4 * i + 2] = (long long)s[i].off;
    out[4 * i + 3] = (long long)s[i].cnt;
  }
  return (int)s.size();
}

// ---------------------------------------------------------------------------
// RCCL, loaded on demand
// ---------------------------------------------------------------------------
namespace
{
struct Rccl
{
  void* h = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_mu;

void load_rccl()
{
  if (g_rccl.h)
    return;
  // DK_RCCL_LIB: another library with the same six entry points (tests/shim_rccl.c: host-staged sums, so that the
  // thread-per-replica / segment / stream-ordering code below runs on a box with ONE GPU)
  const char* override_lib = getenv("DK_RCCL_LIB");
  const char* names[] = {override_lib ? override_lib : "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (int k = 0; k < (override_lib ? 1 : 3); ++k)
    if ((g_rccl.h = dlopen(names[k], RTLD_NOW | RTLD_GLOBAL)))
      break;
  if (!g_rccl.h)
  {
    fprintf(stderr, "TrainNetworks: dlopen(%s): %s\n", names[0], dlerror());
    error("TrainNetworks: cannot load the collective library (multi-GPU training needs RCCL)");
  }
  auto sym = [](const char* s) {
    void* p = dlsym(g_rccl.h, s);
    if (!p)
    {
      fprintf(stderr, "librccl: missing symbol %s\n", s);
      exit(EXIT_FAILURE);
    }
    return p;
  };
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))sym("ncclCommInitAll");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))sym("ncclAllReduce");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
}

#define CHECK_RCCL(X)                                                                        \
  do                                                                                         \
  {                                                                                          \
    ncclResult_t r_ = (X);                                                                   \
    if (r_ != ncclSuccess)                                                                   \
    {                                                                                        \
      fprintf(stderr, "RCCL error %s:%d: %s\n", __FILE__, __LINE__, g_rccl.GetErrorString(r_)); \
      exit(EXIT_FAILURE);                                                                    \
    }                                                                                        \
  } while (0)

// per-replica data-parallel state (Network::dp)
struct DpState
{
  int world = 0, rank = 0;
  bool rccl = false;            // distinct devices -> RCCL; replicas sharing one device -> local sum
  ncclComm_t comm = nullptr;
  hipStream_t cs = nullptr;     // communication stream
  hipStream_t own = nullptr;    // compute stream of a replica that shares its device with another one (collective mode)
  std::vector<hipEvent_t> seg_ready;
  hipEvent_t comm_done = nullptr;
  float* bucket = nullptr;
  size_t bucket_n = 0;
  std::vector<DkSegment> segs;
  float* X = nullptr;           // host staging of one sub-batch
  float* y = nullptr;
};

int dp_segments()
{
  const char* e = getenv("DK_DP_SEGMENTS");
  const int n = e ? atoi(e) : 4;
  return n < 1 ? 1 : n;
}

// one-time set-up of the replicas nets[0..n): buckets, communicators, streams
void dp_setup(Network* nets, int n)
{
  std::lock_guard<std::mutex> lk(g_mu);
  bool ready = true;
  for (int i = 0; i < n; ++i)
    if (!nets[i].dp || ((DpState*)nets[i].dp)->world != n)
      ready = false;
  if (ready)
    return;
  bool distinct = true;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < i; ++j)
      if (nets[i].gpu_index == nets[j].gpu_index)
        distinct = false;
  // Replicas on ONE device normally take the serial local-sum path.  With DK_DP_SHARED_DEVICE_COLLECTIVE=1 (test rigs,
  // together with DK_RCCL_LIB) they run as on distinct devices -- one host thread each, the collective library between
  // them -- on compute streams and reduction scratch of their own.
  const bool shared_collective = !distinct && getenv("DK_DP_SHARED_DEVICE_COLLECTIVE") && atoi(getenv("DK_DP_SHARED_DEVICE_COLLECTIVE"));
  const bool collective = n > 1 && (distinct || shared_collective);
  std::vector<ncclComm_t> comms(n, nullptr);
  if (collective)
  {
    load_rccl();
    std::vector<int> devs(n);
    for (int i = 0; i < n; ++i) devs[i] = nets[i].gpu_index;
    CHECK_RCCL(g_rccl.CommInitAll(comms.data(), n, devs.data()));
  }
  for (int i = 0; i < n; ++i)
  {
    Network* net = &nets[i];
    if (net->gpu_index < 0 || !net->train)
      error("TrainNetworks: every replica must be a train-mode network on a HIP device");
    cuda_set_device(net->gpu_index);
    DpState* st = (DpState*)net->dp;
    if (!st)
    {
      st = new DpState();
      net->dp = st;
    }
    st->world = n;
    st->rank = i;
    st->rccl = collective;
    st->comm = comms[i];
    if (shared_collective && !st->own)
      CHECK_HIP(hipStreamCreateWithFlags(&st->own, hipStreamNonBlocking));
    if (!st->cs)
      CHECK_HIP(hipStreamCreateWithFlags(&st->cs, hipStreamNonBlocking));
    if (!st->comm_done)
      CHECK_HIP(hipEventCreateWithFlags(&st->comm_done, hipEventDisableTiming));
    if (!net->grad_bucket)
    {
      st->bucket_n = DkGradBucketSize(net);
      st->bucket = cuda_make_array(nullptr, st->bucket_n);
      CHECK_HIP(hipMemsetAsync(st->bucket, 0, st->bucket_n * sizeof(float), get_cuda_stream()));
      DkAttachGradBucket(net, st->bucket);
    }
    else
    {
      st->bucket = net->grad_bucket;
      st->bucket_n = DkGradBucketSize(net);
    }
    DkSetReplicas(net, n);
    st->segs = bucket_segments(net, dp_segments());
    for (auto e : st->seg_ready) (void)hipEventDestroy(e);
    st->seg_ready.resize(st->segs.size());
    for (auto& e : st->seg_ready) CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    free(st->X);
    free(st->y);
    st->X = (float*)xcalloc((size_t)net->batch * net->inputs, sizeof(float));
    st->y = (float*)xcalloc((size_t)net->batch * (net->truths > 0 ? net->truths : 1), sizeof(float));
    CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
  }
}

// forward/backward of one replica over its subdivisions; on the last subdivision the bucket
// slices go to RCCL as the backward pass finalises them
// the calling thread's compute stream for the lifetime of the guard (replicas sharing a device)
struct ThreadStream
{
  explicit ThreadStream(hipStream_t s) : on(s != nullptr)
  {
    if (on)
      dk_set_thread_stream(s);
  }
  ~ThreadStream()
  {
    if (on)
      dk_set_thread_stream(nullptr);
  }
  bool on;
};

float replica_step(Network* net, data d)
{
  cuda_set_device(net->gpu_index);
  DpState* st = (DpState*)net->dp;
  ThreadStream ts(st->own);
  hipStream_t s = get_cuda_stream();
  const int batch = net->batch, subdiv = net->subdiv;
  if (d.X.rows != batch * subdiv)
    error("TrainNetworks: the data shard must hold batch x subdivisions rows per replica");
  float sum = 0;
  for (int i = 0; i < subdiv; ++i)
  {
    get_next_batch(d, batch, i * batch, st->X, st->y);
    net->curr_subdiv = i;
    DkTrainForward(net, st->X, st->y);
    if (i + 1 < subdiv || !st->rccl)
      DkBackwardRange(net, net->n, 0);
    else
    {
      for (size_t k = 0; k < st->segs.size(); ++k)
      {
        const DkSegment& sg = st->segs[k];
        DkBackwardRange(net, sg.hi, sg.lo);
        if (!sg.cnt)
          continue;
        CHECK_HIP(hipEventRecord(st->seg_ready[k], s));
        CHECK_HIP(hipStreamWaitEvent(st->cs, st->seg_ready[k], 0));
        CHECK_RCCL(g_rccl.AllReduce(st->bucket + sg.off, st->bucket + sg.off, sg.cnt, ncclFloat, ncclSum, st->comm, st->cs));
      }
      CHECK_HIP(hipEventRecord(st->comm_done, st->cs));
      CHECK_HIP(hipStreamWaitEvent(s, st->comm_done, 0));   // the update waits for the collectives
    }
    sum += DkTrainFinish(net);
  }
  return sum / (batch * subdiv);
}
}  // namespace

void DkFreeDpState(Network* net)
{
  DpState* st = (DpState*)net->dp;
  if (!st)
    return;
  for (auto e : st->seg_ready) (void)hipEventDestroy(e);
  if (st->comm_done) (void)hipEventDestroy(st->comm_done);
  if (st->cs) (void)hipStreamDestroy(st->cs);
  if (st->own) (void)hipStreamDestroy(st->own);
  if (st->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(st->comm);
  if (st->bucket && st->bucket == net->grad_bucket)
  {
    // the layers' gradient pointers alias the bucket: FreeNetwork drops them before this runs
    cuda_free(st->bucket);
    net->grad_bucket = nullptr;
  }
  free(st->X);
  free(st->y);
  delete st;
  net->dp = nullptr;
}

// ---------------------------------------------------------------------------
// public entry points
// ---------------------------------------------------------------------------
float TrainNetwork(Network* net, data d)
{
  // network.cpp:210-239
  if (d.X.rows % net->batch != 0 || d.X.rows / net->batch != net->subdiv)
    error("TrainNetwork: data rows must equal batch x subdivisions");
  const int batch = net->batch, subdiv = net->subdiv;
  float* X = (float*)xcalloc((size_t)batch * d.X.cols, sizeof(float));
  float* y = (float*)xcalloc((size_t)batch * (d.y.cols > 0 ? d.y.cols : 1), sizeof(float));
  float sum = 0;
  for (int i = 0; i < subdiv; ++i)
  {
    get_next_batch(d, batch, i * batch, X, y);
    net->curr_subdiv = i;
    sum += TrainNetworkDatum(net, X, y);
  }
  net->curr_iter++;
  UpdateNetworkGpu(net);
  free(X);
  free(y);
  return sum / (batch * subdiv);
}

float TrainNetworks(Network* nets, int num_gpus, data d, int sync_interval)
{
  (void)sync_interval;   // replicas never diverge: nothing to re-synchronise (see the header comment)
  if (num_gpus < 1)
    error("TrainNetworks: num_gpus < 1");
  if (d.X.rows != nets[0].batch * nets[0].subdiv * num_gpus)
    error("TrainNetworks: data rows must equal batch x subdivisions x num_gpus");
  dp_setup(nets, num_gpus);
  std::vector<float> errors(num_gpus, 0.f);
  DpState* st0 = (DpState*)nets[0].dp;
  if (st0->rccl || num_gpus == 1)
  {
    std::vector<std::thread> threads;
    for (int i = 0; i < num_gpus; ++i)
      threads.emplace_back([&, i]() { errors[i] = replica_step(&nets[i], GetPartialData(d, i, num_gpus)); });
    for (auto& t : threads) t.join();
  }
  else
  {
    // replicas sharing one device also share its stream and per-device scratch buffers:
    // run them one after the other
    for (int i = 0; i < num_gpus; ++i) errors[i] = replica_step(&nets[i], GetPartialData(d, i, num_gpus));
  }
  if (num_gpus > 1 && !st0->rccl)
  {
    // replicas sharing one device (single-GPU boxes, tests): the "collective" is a local sum
    cuda_set_device(nets[0].gpu_index);
    hipStream_t s = get_cuda_stream();
    CHECK_HIP(hipStreamSynchronize(s));
    for (int i = 1; i < num_gpus; ++i)
      dk_axpy(st0->bucket_n, 1.0f, ((DpState*)nets[i].dp)->bucket, st0->bucket, s);
    for (int i = 1; i < num_gpus; ++i)
      dk_copy(st0->bucket_n, st0->bucket, ((DpState*)nets[i].dp)->bucket, s);
    CHECK_HIP(hipStreamSynchronize(s));
  }
  {
    auto upd = [&](int i) {
      cuda_set_device(nets[i].gpu_index);
      ThreadStream ts(((DpState*)nets[i].dp)->own);
      nets[i].curr_iter++;
      UpdateNetworkGpu(&nets[i]);
      CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
    };
    if (st0->rccl)
    {
      std::vector<std::thread> threads;
      for (int i = 0; i < num_gpus; ++i) threads.emplace_back(upd, i);
      for (auto& t : threads) t.join();
    }
    else
      for (int i = 0; i < num_gpus; ++i) upd(i);
  }
  float sum = 0;
  for (float e : errors) sum += e;
  return sum / num_gpus;
}

// Averages weights, biases, scales AND the batch-norm rolling statistics of the replicas
// (network_kernels.cu:398-427 averages the first three; the rolling statistics are the only
// tensors that differ between replicas here).
void SyncNetworks(Network* nets, int num_gpus)
{
  for (int j = 1; j < num_gpus; ++j)
  {
    nets[0].seen += nets[j].seen;
    nets[j].seen = 0;
  }
  if (num_gpus < 2)
    return;
  dp_setup(nets, num_gpus);
  DpState* st0 = (DpState*)nets[0].dp;
  const float inv = 1.0f / num_gpus;
  auto tensors = [](layer* l, std::vector<std::pair<float*, size_t>>& v) {
    v.clear();
    if (l->type == BATCHNORM)
    {
      // standalone [batchnorm]: its gradients ride in the bucket, so its parameters are replicated state too
      v.push_back({l->biases_gpu, (size_t)l->c});
      v.push_back({l->scales_gpu, (size_t)l->c});
      v.push_back({l->rolling_mean_gpu, (size_t)l->c});
      v.push_back({l->rolling_variance_gpu, (size_t)l->c});
      return;
    }
    if (l->type != CONVOLUTIONAL)
      return;
    v.push_back({l->biases_gpu, (size_t)l->n});
    v.push_back({l->weights_gpu, (size_t)l->nweights});
    if (l->scales_gpu)
    {
      v.push_back({l->scales_gpu, (size_t)l->n});
      v.push_back({l->rolling_mean_gpu, (size_t)l->n});
      v.push_back({l->rolling_variance_gpu, (size_t)l->n});
    }
  };
  if (st0->rccl)
  {
    std::vector<std::thread> threads;
    for (int i = 0; i < num_gpus; ++i)
      threads.emplace_back([&, i]() {
        Network* net = &nets[i];
        cuda_set_device(net->gpu_index);
        DpState* st = (DpState*)net->dp;
        ThreadStream ts(st->own);
        hipStream_t s = get_cuda_stream();
        std::vector<std::pair<float*, size_t>> v;
        for (int j = 0; j < net->n; ++j)
        {
          tensors(&net->layers[j], v);
          for (auto& t : v)
          {
            CHECK_RCCL(g_rccl.AllReduce(t.first, t.first, t.second, ncclFloat, ncclSum, st->comm, s));
            dk_scal(t.second, inv, t.first, s);
          }
        }
        CHECK_HIP(hipStreamSynchronize(s));
      });
    for (auto& t : threads) t.join();
    return;
  }
  cuda_set_device(nets[0].gpu_index);
  hipStream_t s = get_cuda_stream();
  std::vector<std::pair<float*, size_t>> v0, vi;
  for (int j = 0; j < nets[0].n; ++j)
  {
    tensors(&nets[0].layers[j], v0);
    for (int i = 1; i < num_gpus; ++i)
    {
      tensors(&nets[i].layers[j], vi);
      for (size_t k = 0; k < v0.size(); ++k) dk_axpy(v0[k].second, 1.0f, vi[k].first, v0[k].first, s);
    }
    for (auto& t : v0) dk_scal(t.second, inv, t.first, s);
    for (int i = 1; i < num_gpus; ++i)
    {
      tensors(&nets[i].layers[j], vi);
      for (size_t k = 0; k < v0.size(); ++k) dk_copy(v0[k].second, v0[k].first, vi[k].first, s);
    }
  }
  CHECK_HIP(hipStreamSynchronize(s));
}

// ---------------------------------------------------------------------------
// flat helpers for FFI callers (ctypes tests, bench tools): replicas live in ONE array of
// Network structs, as the reference's `Network* nets` does
// ---------------------------------------------------------------------------
extern "C" LIB_API Network* DkNetworkArrayCreate(int n) { return (Network*)xcalloc(n, sizeof(Network)); }
extern "C" LIB_API Network* DkNetworkArrayAt(Network* nets, int i) { return nets + i; }
extern "C" LIB_API void DkNetworkArrayDestroy(Network* nets, int n)
{
  for (int i = 0; i < n; ++i) FreeNetwork(&nets[i]);
  free(nets);
}

// X: rows x x_cols floats, y: rows x y_cols floats (row-major); rows = batch x subdivisions x num_gpus
extern "C" LIB_API float DkTrainNetworksFlat(Network* nets, int num_gpus, float* X, int x_cols, float* y, int y_cols,
    int rows, int sync_interval)
{
  std::vector<float*> xv(rows), yv(rows);
  for (int r = 0; r < rows; ++r)
  {
    xv[r] = X + (size_t)r * x_cols;
    yv[r] = y + (size_t)r * y_cols;
  }
  data d;
  memset(&d, 0, sizeof(d));
  d.X.rows = d.y.rows = rows;
  d.X.cols = x_cols;
  d.y.cols = y_cols;
  d.X.vals = xv.data();
  d.y.vals = yv.data();
  d.shallow = 1;
  return num_gpus == 1 ? TrainNetwork(nets, d) : TrainNetworks(nets, num_gpus, d, sync_interval);
}
