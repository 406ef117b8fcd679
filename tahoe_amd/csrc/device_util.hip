// Thin HIP runtime wrappers so host code above the C ABI (the BaseTahoeTest look-alike, language
// bindings) needs no HIP headers.  Replaces cuda_base.h:28-50 (allocate / updateDevice /
// updateHost / copy) and compare_GPU (cuda_base.h:98-111) of the reference.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "common.h"

namespace tahoe {

// compare_GPU walks the arrays with one thread; this counts mismatches with the whole chip and
// returns the count instead of printing.
__global__ void compare_kernel(const float *__restrict__ a, const float *__restrict__ b, size_t n, float tol,
                               unsigned long long *__restrict__ bad)
{
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = a[i] - b[i];
        // cuda_base.h:103: (d > tol || d < -tol); a NaN difference passes there, so it passes here.
        if (d > tol || d < -tol) ++local;
    }
    // wave64 reduction, then one atomic per wave
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(bad, local);
}

__global__ void widen_kernel(double *__restrict__ dst, const float *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (double)src[i];
}

__global__ void narrow_kernel(float *__restrict__ dst, const double *__restrict__ src, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

}  // namespace tahoe

using namespace tahoe;

extern "C" {

tahoe_status tahoe_device_count(int *count)
{
    if (!count) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return TAHOE_OK;
}

tahoe_status tahoe_device_set(int device)
{
    TAHOE_HIP_TRY(hipSetDevice(device));
    return TAHOE_OK;
}

tahoe_status tahoe_device_alloc(void **ptr, size_t bytes, int set_zero)
{
    if (!ptr) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    *ptr = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return fail(TAHOE_ERR_NO_DEVICE, "no HIP device is visible; libtahoe_amd has no CPU path");
    TAHOE_HIP_TRY(hipMalloc(ptr, bytes ? bytes : 1));
    if (set_zero && bytes) TAHOE_HIP_TRY(hipMemset(*ptr, 0, bytes));
    return TAHOE_OK;
}

tahoe_status tahoe_host_alloc(void **ptr, size_t bytes)
{
    if (!ptr) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    *ptr = nullptr;
    TAHOE_HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return TAHOE_OK;
}

tahoe_status tahoe_host_free(void *ptr)
{
    if (ptr) TAHOE_HIP_TRY(hipHostFree(ptr));
    return TAHOE_OK;
}

tahoe_status tahoe_device_free(void *ptr)
{
    if (ptr) TAHOE_HIP_TRY(hipFree(ptr));
    return TAHOE_OK;
}

tahoe_status tahoe_device_memset(void *ptr_dev, int value, size_t bytes, void *stream)
{
    if (bytes && !ptr_dev) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (bytes) TAHOE_HIP_TRY(hipMemsetAsync(ptr_dev, value, bytes, (hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_widen_f32_to_f64(double *dst_dev, const float *src_dev, size_t n, void *stream)
{
    if (n && (!dst_dev || !src_dev)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (n == 0) return TAHOE_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(widen_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst_dev, src_dev, n);
    TAHOE_HIP_TRY(hipGetLastError());
    return TAHOE_OK;
}

tahoe_status tahoe_narrow_f64_to_f32(float *dst_dev, const double *src_dev, size_t n, void *stream)
{
    if (n && (!dst_dev || !src_dev)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (n == 0) return TAHOE_OK;
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(narrow_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dst_dev, src_dev, n);
    TAHOE_HIP_TRY(hipGetLastError());
    return TAHOE_OK;
}

tahoe_status tahoe_copy_to_device(void *dst_dev, const void *src_host, size_t bytes, void *stream)
{
    if (bytes && (!dst_dev || !src_host)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (bytes) TAHOE_HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_copy_to_host(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
    if (bytes && (!dst_host || !src_dev)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (bytes) {
        TAHOE_HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
        TAHOE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    }
    return TAHOE_OK;
}

tahoe_status tahoe_stream_create(void **stream)
{
    if (!stream) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    hipStream_t s;
    TAHOE_HIP_TRY(hipStreamCreate(&s));
    *stream = (void *)s;
    return TAHOE_OK;
}

tahoe_status tahoe_stream_destroy(void *stream)
{
    if (stream) TAHOE_HIP_TRY(hipStreamDestroy((hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_stream_synchronize(void *stream)
{
    TAHOE_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_event_create(void **event)
{
    if (!event) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    hipEvent_t e;
    TAHOE_HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = (void *)e;
    return TAHOE_OK;
}

tahoe_status tahoe_event_destroy(void *event)
{
    if (event) TAHOE_HIP_TRY(hipEventDestroy((hipEvent_t)event));
    return TAHOE_OK;
}

tahoe_status tahoe_event_record(void *event, void *stream)
{
    if (!event) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    TAHOE_HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_stream_wait_event(void *stream, void *event)
{
    if (!event) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    TAHOE_HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return TAHOE_OK;
}

tahoe_status tahoe_copy_peer(void *dst_dev, int dst_device, const void *src_dev, int src_device, size_t bytes, void *stream)
{
    if (bytes && (!dst_dev || !src_dev)) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    if (bytes == 0) return TAHOE_OK;
    if (dst_device == src_device)
        TAHOE_HIP_TRY(hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    else
        TAHOE_HIP_TRY(hipMemcpyPeerAsync(dst_dev, dst_device, src_dev, src_device, bytes, (hipStream_t)stream));
    return TAHOE_OK;
}

tahoe_status tahoe_device_synchronize(void)
{
    TAHOE_HIP_TRY(hipDeviceSynchronize());
    return TAHOE_OK;
}

tahoe_status tahoe_device_lds_bytes(int *bytes)
{
    if (!bytes) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    int dev = 0;
    TAHOE_HIP_TRY(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    TAHOE_HIP_TRY(hipGetDeviceProperties(&prop, dev));
    *bytes = (int)prop.maxSharedMemoryPerMultiProcessor;
    return TAHOE_OK;
}

tahoe_status tahoe_compare_device(const float *a_dev, const float *b_dev, size_t n, float tol, size_t *num_bad,
                                  void *stream)
{
    if (!num_bad || (n && (!a_dev || !b_dev))) return fail(TAHOE_ERR_INVALID_ARG, "null argument");
    *num_bad = 0;
    if (n == 0) return TAHOE_OK;
    unsigned long long *d_bad = nullptr;
    TAHOE_HIP_TRY(hipMalloc(&d_bad, sizeof(*d_bad)));
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_bad, 0, sizeof(*d_bad), s);
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 2048);
        hipLaunchKernelGGL(compare_kernel, dim3(grid), dim3(256), 0, s, a_dev, b_dev, n, tol, d_bad);
        e = hipGetLastError();
    }
    unsigned long long h_bad = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_bad, d_bad, sizeof(h_bad), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(TAHOE_ERR_HIP, "tahoe_compare_device: %s", hipGetErrorString(e));
    *num_bad = (size_t)h_bad;
    return TAHOE_OK;
}

}  // extern "C"
