// shim_rccl.cpp -- TEST-ONLY stand-in for librccl (selected with DK_RCCL_LIB=<this .so>): the six entry points
// darknet_amd/csrc/host/multigpu.cpp binds with dlsym, implemented as host-staged sums between the communicator's
// ranks (threads of one process).  It exists so that the C-level data-parallel path (TrainNetworks: one host thread per
// replica, gradient bucket all-reduced in backward-order segments on a communication stream behind events, threaded
// update, SyncNetworks) can be EXECUTED on a box with one GPU; it says nothing about RCCL's performance or its
// behaviour over xGMI.  Blocking (each call synchronises its stream), float sums only, ranks added in rank order.
// build: hipcc -shared -fPIC -O2 tests/shim_rccl.cpp -o tests/libshim_rccl.so -lpthread
#include <hip/hip_runtime.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <vector>

extern "C" {
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclFloat = 7 } ncclDataType_t;   // values of rccl.h
typedef enum { ncclSum = 0 } ncclRedOp_t;

struct ShimCtx
{
  int n;
  pthread_barrier_t bar;
  std::vector<std::vector<float>> stage;
  std::vector<float> sum;
  std::atomic<int> live;
  std::atomic<long long> calls;     // all-reduce calls seen by rank 0 (the test reads it back)
  std::atomic<long long> floats;
};
struct ncclComm
{
  ShimCtx* ctx;
  int rank, dev;
};
typedef struct ncclComm* ncclComm_t;

static std::atomic<long long> g_total_calls{0}, g_total_floats{0}, g_max_concurrency{0}, g_inside{0};

__attribute__((visibility("default"))) ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist)
{
  if (!comms || ndev < 1)
    return ncclInvalidArgument;
  ShimCtx* c = new ShimCtx();
  c->n = ndev;
  pthread_barrier_init(&c->bar, nullptr, (unsigned)ndev);
  c->stage.resize(ndev);
  c->live = ndev;
  c->calls = 0;
  c->floats = 0;
  for (int i = 0; i < ndev; ++i)
  {
    comms[i] = new ncclComm();
    comms[i]->ctx = c;
    comms[i]->rank = i;
    comms[i]->dev = devlist ? devlist[i] : i;
  }
  return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
  if (!comm)
    return ncclSuccess;
  ShimCtx* c = comm->ctx;
  if (--c->live == 0)
  {
    pthread_barrier_destroy(&c->bar);
    delete c;
  }
  delete comm;
  return ncclSuccess;
}

__attribute__((visibility("default"))) ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count,
    ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
  if (!comm || datatype != ncclFloat || op != ncclSum)
    return ncclInvalidArgument;
  ShimCtx* c = comm->ctx;
  const int r = comm->rank;
  const long long in = ++g_inside;
  long long prev = g_max_concurrency.load();
  while (in > prev && !g_max_concurrency.compare_exchange_weak(prev, in)) {}
  if (hipSetDevice(comm->dev) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess)
    return ncclUnhandledCudaError;
  c->stage[r].resize(count);
  if (hipMemcpy(c->stage[r].data(), sendbuff, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  pthread_barrier_wait(&c->bar);
  if (r == 0)
  {
    c->sum = c->stage[0];
    for (int k = 1; k < c->n; ++k)
    {
      if (c->stage[k].size() != count)
      {
        fprintf(stderr, "shim_rccl: ranks disagree on the element count of an all-reduce (%zu vs %zu)\n", c->stage[k].size(), count);
        abort();
      }
      const float* a = c->stage[k].data();
      for (size_t i = 0; i < count; ++i) c->sum[i] += a[i];
    }
    ++c->calls;
    c->floats += (long long)count;
    ++g_total_calls;
    g_total_floats += (long long)count;
  }
  pthread_barrier_wait(&c->bar);
  const hipError_t e = hipMemcpy(recvbuff, c->sum.data(), count * sizeof(float), hipMemcpyHostToDevice);
  pthread_barrier_wait(&c->bar);   // nobody starts the next call (rank 0 rewrites `sum`) before everybody has read this one
  --g_inside;
  return e == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

__attribute__((visibility("default"))) ncclResult_t ncclGroupStart() { return ncclSuccess; }
__attribute__((visibility("default"))) ncclResult_t ncclGroupEnd() { return ncclSuccess; }
__attribute__((visibility("default"))) const char* ncclGetErrorString(ncclResult_t r)
{
  return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (the shim sums floats only)" : "HIP error inside the shim";
}

// test read-back: all-reduce calls / floats summed so far, and the largest number of ranks ever inside a call at once
__attribute__((visibility("default"))) void shim_rccl_stats(long long* out)
{
  out[0] = g_total_calls.load();
  out[1] = g_total_floats.load();
  out[2] = g_max_concurrency.load();
}
}
