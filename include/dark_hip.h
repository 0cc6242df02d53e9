/*
 * dark_hip.h -- device runtime C-ABI of the MI355X-native Darknet conv path.
 *
 * This is the reference's src/dark_cuda.h (:56-83) re-done on HIP: same entry
 * point names, argument meaning and error behaviour (errors print and exit(),
 * src/dark_cuda.c:85-106), so host code written against dark_cuda.h relinks
 * unchanged.  No CUDA headers, no cuBLAS/cuDNN/cuRAND: `blas_handle` and
 * `cudnn_handle` are deliberately absent (nothing on the conv hot path needs
 * them once GEMM is a hand-written MFMA kernel); `cuda_random` is kept, backed
 * by a counter-based generator instead of cuRAND (same distribution, another
 * stream: no caller can depend on cuRAND's values, the reference seeds it
 * with time(0)).
 *
 * The only types in the signatures are plain pointers/sizes plus `hipStream_t`
 * (an opaque pointer) and `dim3`; the CUDA spellings are kept as typedefs so
 * that `cudaStream_t s = get_cuda_stream();` in caller code still compiles.
 */
#ifndef DARK_HIP_H
#define DARK_HIP_H

#include <stddef.h>
#include <hip/hip_runtime_api.h>

#ifndef DK_API
#define DK_API __attribute__((visibility("default")))
#endif

#define BLOCK 512        /* src/dark_cuda.h:17 */
#define WARP_SIZE 64     /* CDNA wavefront; the reference has 32 (src/dark_cuda.h:19) */

typedef hipError_t cudaError_t;
typedef hipStream_t cudaStream_t;

#ifdef __cplusplus
extern "C" {
#endif

DK_API extern int cuda_debug_sync; /* src/dark_cuda.c:5: sync inside every check */

/* src/dark_cuda.c:85-106 */
DK_API void check_error(cudaError_t status);
DK_API void check_error_extended(cudaError_t status, const char* file, int line, const char* date_time);
#define CHECK_CUDA(X) check_error_extended(X, __FILE__, __LINE__, __DATE__ " - " __TIME__);
#define CHECK_HIP(X) CHECK_CUDA(X)

/* device selection, src/dark_cuda.c:36-56, :596 (-1 when no device is usable) */
DK_API void cuda_set_device(int n);
DK_API int cuda_get_device(void);
DK_API int CudaGetDeviceCount(void);
DK_API int get_gpu_compute_capability(int i); /* gfx950 -> 950 */
DK_API void show_cuda_cudnn_info(void);

/* streams, src/dark_cuda.c:128-177: one compute + one memcpy stream per device */
DK_API cudaStream_t get_cuda_stream(void);
DK_API cudaStream_t get_cuda_memcpy_stream(void);
/* additive: get_cuda_stream() of the CALLING THREAD returns s until reset with NULL (replicas sharing a device) */
DK_API void dk_set_thread_stream(cudaStream_t s);

/* allocation + copies, src/dark_cuda.c:259-272, 410-477, 520-559 */
DK_API float* cuda_make_array(float* x, size_t n);               /* hipMalloc (+ async H2D when x) */
DK_API int* cuda_make_int_array(size_t n);
DK_API int* cuda_make_int_array_new_api(int* x, size_t n);
DK_API void** cuda_make_array_pointers(void** x, size_t n);
DK_API float* cuda_make_array_pinned(float* x, size_t n);        /* hipHostMalloc */
DK_API float* cuda_make_array_pinned_preallocated(float* x, size_t n);
DK_API void pre_allocate_pinned_memory(size_t size);
DK_API void free_pinned_memory(void);
DK_API void cuda_free(float* x_gpu);
DK_API void cuda_free_host(float* x_cpu);
DK_API void cuda_push_array(float* x_gpu, float* x, size_t n);   /* H2D on the compute stream */
DK_API void cuda_pull_array(float* x_gpu, float* x, size_t n);   /* D2H + stream sync */
DK_API void cuda_pull_array_async(float* x_gpu, float* x, size_t n);
DK_API float cuda_compare(float* x_gpu, float* x, size_t n, char* s);
/* src/dark_cuda.c:464-477: x_gpu[0..n) = uniform draws in [0, 1) on the compute stream.  The reference seeds a
 * per-device cuRAND generator with time(0); here a per-device call counter seeds a splitmix64 hash of the index. */
DK_API void cuda_random(float* x_gpu, size_t n);

/* launch geometry helpers, src/dark_cuda.c:108-126 */
DK_API dim3 cuda_gridsize(size_t n);
DK_API int get_number_of_blocks(int array_size, int block_size);

#ifdef __cplusplus
}
#endif
#endif /* DARK_HIP_H */
