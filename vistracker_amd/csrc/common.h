// common.h -- shared helpers of libvistracker_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <atomic>
#include "../../include/vistracker.h"

extern thread_local char vt_err_buf[512];

#define VT_FAIL(code, ...)                                         \
    do {                                                           \
        snprintf(vt_err_buf, sizeof(vt_err_buf), __VA_ARGS__);    \
        return (code);                                             \
    } while (0)

#define VT_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) VT_FAIL(VT_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #call,   \
                                      hipGetErrorString(e_));                                    \
    } while (0)

#define VT_LAUNCH_CHECK()                                                                              \
    do {                                                                                               \
        hipError_t e_ = hipGetLastError();                                                             \
        if (e_ != hipSuccess) VT_FAIL(VT_ERR_HIP, "%s:%d launch -> %s", __FILE__, __LINE__,            \
                                      hipGetErrorString(e_));                                          \
    } while (0)

#define VT_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) VT_FAIL(VT_ERR_ARG, __VA_ARGS__); \
    } while (0)

static inline hipStream_t vt_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Device-side early stop (vt_stream_set_skip_flag, misc.hip): the int a fit registered for this stream, or NULL.  The heavy kernels of a step take it
// as an argument and return at once when it is non-zero: the steps the host queued behind the one whose stop rule fired cost a launch, not a pass.
const int *vt_skip_flag_of(hipStream_t st);
#define VT_SKIP_RETURN(flag_) do { if ((flag_) != nullptr && *(volatile const int *)(flag_) != 0) return; } while (0)

// Raise a kernel's dynamic-LDS limit once per (kernel, device).  The flag is a per-call-site bit mask indexed by the device ordinal:
// thread-safe (bench.py drives two host threads), and a second GPU in the same process gets its own call (the attribute is per device).
// Two threads racing on the same device both make the (idempotent) call; neither launches before its own call returned.
struct vt_lds_once { std::atomic<unsigned long long> mask{0ull}; };
static inline int vt_raise_lds_limit(vt_lds_once &once, const void *kernel, size_t bytes)
{
    int dev = 0;
    VT_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (once.mask.load(std::memory_order_acquire) & bit) return VT_OK;
    VT_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    once.mask.fetch_or(bit, std::memory_order_release);
    return VT_OK;
}
#define VT_LDS_LIMIT(kernel_, bytes_)                                                                          \
    do {                                                                                                       \
        static vt_lds_once once_;                                                                              \
        const int rc_ = vt_raise_lds_limit(once_, reinterpret_cast<const void *>(kernel_), (bytes_));          \
        if (rc_) return rc_;                                                                                   \
    } while (0)

template <typename T>
static inline int vt_upload(T **dst, const T *host, size_t n, hipStream_t st)
{
    VT_HIP(hipMalloc(reinterpret_cast<void **>(dst), n * sizeof(T)));
    VT_HIP(hipMemcpyAsync(*dst, host, n * sizeof(T), hipMemcpyHostToDevice, st));
    return VT_OK;
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block sum for blockDim.x = 64 * NW; `red` = NW floats of LDS; result valid in every thread
template <int NW>
__device__ __forceinline__ float block_sum(float v, float *red)
{
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; i++) s += red[i];
    return s;
}
