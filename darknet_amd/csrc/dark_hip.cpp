// dark_hip.cpp -- device runtime shim: the reference's src/dark_cuda.c re-done on
// HIP behind the same C-ABI (include/dark_hip.h).  One compute stream and one
// memcpy stream per device (src/dark_cuda.c:128-177), errors print and exit()
// (src/dark_cuda.c:85-106).  No cuBLAS/cuDNN/cuRAND handles exist.
#include "dark_hip.h"
#include "dk_kernels.h"

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <atomic>

#include <map>
#include <mutex>
#include <utility>

#define DK_EXPORT extern "C" __attribute__((visibility("default")))

int cuda_debug_sync = 0;

namespace
{
const int kMaxDevices = 16;
hipStream_t g_streams[kMaxDevices];
int g_stream_init[kMaxDevices];
hipStream_t g_copy_streams[kMaxDevices];
int g_copy_stream_init[kMaxDevices];
int g_no_device = -1;  // -1 unknown, 0 devices usable, 1 none

// pinned pool for cuda_make_array_pinned_preallocated (src/dark_cuda.c:274-408)
char* g_pinned = nullptr;
size_t g_pinned_size = 0, g_pinned_used = 0;

bool have_device()
{
  if (g_no_device < 0)
  {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess)
      (void)hipGetLastError();
    g_no_device = (e == hipSuccess && n > 0) ? 0 : 1;
  }
  return g_no_device == 0;
}
}  // namespace

DK_EXPORT void check_error_extended(
    cudaError_t status, const char* file, int line, const char* date_time)
{
  if (status != hipSuccess)
  {
    fprintf(stderr, "HIP status Error: file: %s : line: %d : build time: %s \n", file, line,
        date_time);
    fprintf(stderr, "HIP Error: %s\n", hipGetErrorString(status));
    exit(EXIT_FAILURE);
  }
  if (cuda_debug_sync)
  {
    status = hipDeviceSynchronize();
    if (status != hipSuccess)
    {
      fprintf(stderr, "HIP status = hipDeviceSynchronize() Error: file: %s : line: %d\n", file,
          line);
      fprintf(stderr, "HIP Error: %s\n", hipGetErrorString(status));
      exit(EXIT_FAILURE);
    }
  }
}

DK_EXPORT void check_error(cudaError_t status) { check_error_extended(status, "", 0, ""); }

DK_EXPORT void cuda_set_device(int n)
{
  if (!have_device())
  {
    fprintf(stderr, "cuda_set_device(%d): no HIP device is available\n", n);
    exit(EXIT_FAILURE);
  }
  CHECK_HIP(hipSetDevice(n));
}

DK_EXPORT int cuda_get_device(void)
{
  if (!have_device())
    return -1;
  int n = 0;
  CHECK_HIP(hipGetDevice(&n));
  return n;
}

DK_EXPORT int CudaGetDeviceCount(void)
{
  if (!have_device())
    return 0;
  int n = 0;
  CHECK_HIP(hipGetDeviceCount(&n));
  return n;
}

DK_EXPORT int get_gpu_compute_capability(int i)
{
  if (!have_device())
    return 0;
  hipDeviceProp_t prop;
  CHECK_HIP(hipGetDeviceProperties(&prop, i));
  // gcnArchName is e.g. "gfx950:sramecc+:xnack-" -> 950
  int v = 0;
  const char* p = strstr(prop.gcnArchName, "gfx");
  if (p)
    v = (int)strtol(p + 3, nullptr, 10);
  return v;
}

DK_EXPORT void show_cuda_cudnn_info(void)
{
  if (!have_device())
  {
    fprintf(stderr, " HIP: no device\n");
    return;
  }
  int rt = 0, drv = 0;
  (void)hipRuntimeGetVersion(&rt);
  (void)hipDriverGetVersion(&drv);
  hipDeviceProp_t prop;
  int dev = cuda_get_device();
  CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  fprintf(stderr, " HIP runtime %d, driver %d, device %d: %s (%s), %d CUs, %.1f GiB\n", rt, drv,
      dev, prop.name, prop.gcnArchName, prop.multiProcessorCount,
      prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
}

// Replicas of a data-parallel run that share a device (a test rig: two replica threads on one GPU through the real
// collective path) each get their own compute stream: the thread's override wins over the per-device stream.
static thread_local cudaStream_t t_stream_override = nullptr;
DK_EXPORT void dk_set_thread_stream(cudaStream_t s) { t_stream_override = s; }

DK_EXPORT cudaStream_t get_cuda_stream(void)
{
  if (t_stream_override)
    return t_stream_override;
  int i = cuda_get_device();
  if (i < 0 || i >= kMaxDevices)
  {
    fprintf(stderr, "get_cuda_stream: no HIP device (the HIP path has no CPU fallback)\n");
    exit(EXIT_FAILURE);
  }
  if (!g_stream_init[i])
  {
    CHECK_HIP(hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking));
    g_stream_init[i] = 1;
  }
  return g_streams[i];
}

DK_EXPORT cudaStream_t get_cuda_memcpy_stream(void)
{
  int i = cuda_get_device();
  if (i < 0 || i >= kMaxDevices)
  {
    fprintf(stderr, "get_cuda_memcpy_stream: no HIP device\n");
    exit(EXIT_FAILURE);
  }
  if (!g_copy_stream_init[i])
  {
    CHECK_HIP(hipStreamCreateWithFlags(&g_copy_streams[i], hipStreamNonBlocking));
    g_copy_stream_init[i] = 1;
  }
  return g_copy_streams[i];
}

DK_EXPORT float* cuda_make_array(float* x, size_t n)
{
  float* x_gpu = nullptr;
  size_t size = sizeof(float) * n;
  // 64 bytes of slack behind every device array: kernels that fetch 16-byte pieces of rows whose length is not a
  // multiple of 4 floats (conv3x3_wino.hip) may read up to 12 bytes past the last element (the values are masked)
  hipError_t status = hipMalloc((void**)&x_gpu, (size ? size : 4) + 64);
  if (status != hipSuccess)
    fprintf(stderr, " Try to set subdivisions=64 in your cfg-file. \n");  // dark_cuda.c:432-435
  CHECK_HIP(status);
  if (x)
  {
    // Pageable host memory: this copy is synchronous w.r.t. the host buffer.
    CHECK_HIP(hipMemcpyAsync(x_gpu, x, size, hipMemcpyHostToDevice, get_cuda_stream()));
  }
  if (!x_gpu)
  {
    fprintf(stderr, "HIP malloc failed\n");
    exit(EXIT_FAILURE);
  }
  return x_gpu;
}

DK_EXPORT int* cuda_make_int_array(size_t n)
{
  int* x_gpu = nullptr;
  CHECK_HIP(hipMalloc((void**)&x_gpu, sizeof(int) * (n ? n : 1)));
  return x_gpu;
}

DK_EXPORT int* cuda_make_int_array_new_api(int* x, size_t n)
{
  int* x_gpu = cuda_make_int_array(n);
  if (x)
    CHECK_HIP(hipMemcpyAsync(x_gpu, x, sizeof(int) * n, hipMemcpyHostToDevice, get_cuda_stream()));
  return x_gpu;
}

DK_EXPORT void** cuda_make_array_pointers(void** x, size_t n)
{
  void** x_gpu = nullptr;
  CHECK_HIP(hipMalloc((void**)&x_gpu, sizeof(void*) * (n ? n : 1)));
  if (x)
    CHECK_HIP(
        hipMemcpyAsync(x_gpu, x, sizeof(void*) * n, hipMemcpyHostToDevice, get_cuda_stream()));
  return x_gpu;
}

DK_EXPORT float* cuda_make_array_pinned(float* x, size_t n)
{
  float* p = nullptr;
  CHECK_HIP(hipHostMalloc((void**)&p, sizeof(float) * (n ? n : 1), hipHostMallocDefault));
  if (x)
    memcpy(p, x, sizeof(float) * n);
  return p;
}

DK_EXPORT void pre_allocate_pinned_memory(size_t size)
{
  if (g_pinned)
    return;
  CHECK_HIP(hipHostMalloc((void**)&g_pinned, size, hipHostMallocDefault));
  g_pinned_size = size;
  g_pinned_used = 0;
}

DK_EXPORT float* cuda_make_array_pinned_preallocated(float* x, size_t n)
{
  size_t bytes = ((sizeof(float) * n + 255) / 256) * 256;
  float* p;
  if (g_pinned && g_pinned_used + bytes <= g_pinned_size)
  {
    p = (float*)(g_pinned + g_pinned_used);
    g_pinned_used += bytes;
    if (x)
      memcpy(p, x, sizeof(float) * n);
  }
  else
    p = cuda_make_array_pinned(x, n);
  return p;
}

DK_EXPORT void free_pinned_memory(void)
{
  if (g_pinned)
    CHECK_HIP(hipHostFree(g_pinned));
  g_pinned = nullptr;
  g_pinned_size = g_pinned_used = 0;
}

DK_EXPORT void cuda_free(float* x_gpu)
{
  if (x_gpu)
    CHECK_HIP(hipFree(x_gpu));
}

DK_EXPORT void cuda_free_host(float* x_cpu)
{
  if (x_cpu)
    CHECK_HIP(hipHostFree(x_cpu));
}

DK_EXPORT void cuda_push_array(float* x_gpu, float* x, size_t n)
{
  CHECK_HIP(hipMemcpyAsync(x_gpu, x, sizeof(float) * n, hipMemcpyHostToDevice, get_cuda_stream()));
}

DK_EXPORT void cuda_pull_array(float* x_gpu, float* x, size_t n)
{
  CHECK_HIP(hipMemcpyAsync(x, x_gpu, sizeof(float) * n, hipMemcpyDeviceToHost, get_cuda_stream()));
  CHECK_HIP(hipStreamSynchronize(get_cuda_stream()));
}

DK_EXPORT void cuda_pull_array_async(float* x_gpu, float* x, size_t n)
{
  CHECK_HIP(hipMemcpyAsync(x, x_gpu, sizeof(float) * n, hipMemcpyDeviceToHost, get_cuda_stream()));
}

DK_EXPORT float cuda_compare(float* x_gpu, float* x, size_t n, char* s)
{
  float* tmp = (float*)calloc(n, sizeof(float));
  cuda_pull_array(x_gpu, tmp, n);
  double err = 0;
  for (size_t i = 0; i < n; ++i)
  {
    double d = (double)tmp[i] - x[i];
    err += d * d;
  }
  printf("Error %s: %f\n", s ? s : "", err / (n ? n : 1));
  free(tmp);
  return (float)err;
}

DK_EXPORT void cuda_random(float* x_gpu, size_t n)
{
  // src/dark_cuda.c:464-477 (curandGenerateUniform with a time(0) seed per device): per-device call counter +
  // wall-clock seed, counter-based draws (kernels/extra_layers.hip)
  static std::atomic<unsigned long long> calls[16];
  static const unsigned long long t0 = (unsigned long long)time(nullptr);
  const int dev = cuda_get_device();
  const unsigned long long c = calls[dev & 15].fetch_add(1);
  if (dk_random_uniform(x_gpu, n, (t0 << 20) ^ ((unsigned long long)dev << 56) ^ (c * 0x9E3779B97F4A7C15ULL), get_cuda_stream()))
  {
    fprintf(stderr, "cuda_random failed\n");
    exit(EXIT_FAILURE);
  }
}

DK_EXPORT dim3 cuda_gridsize(size_t n)
{
  // src/dark_cuda.c:109-126: ceil(n/BLOCK) blocks, folded to 2-D above 65535
  size_t k = (n - 1) / BLOCK + 1;
  size_t x = k, y = 1;
  if (x > 65535)
  {
    x = (size_t)ceil(sqrt((double)k));
    y = (n - 1) / (x * BLOCK) + 1;
  }
  return dim3((unsigned)x, (unsigned)y, 1);
}

DK_EXPORT int get_number_of_blocks(int array_size, int block_size)
{
  return array_size / block_size + ((array_size % block_size > 0) ? 1 : 0);
}

// ---- per-(device, kernel) dynamic-LDS limit (see dk_internal.h) --------------------------
void dk_set_max_dynamic_lds(const void* kernel, int bytes)
{
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> done;
  int dev = 0;
  CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  int& have = done[std::make_pair(dev, kernel)];
  if (have >= bytes)
    return;
  CHECK_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  have = bytes;
}
