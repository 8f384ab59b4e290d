// Common definitions for the gfx950 kernels of libsgx_hip.so.
// Device build: hipcc --offload-arch=gfx950.  SGX_EMU (tests only) swaps the HIP runtime for the host
// emulation in tests/emu/ so kernel logic can be checked without a GPU; the product never defines it.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#ifdef SGX_EMU
#include "hip_emu.h"
#define SGX_LAUNCH(kernel, grid, block, smem, stream, ...) \
    sgx_emu::launch(grid, block, smem, [=]() { kernel(__VA_ARGS__); })
#define SGX_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(sgx_emu_dyn_smem())
#else
#include <hip/hip_runtime.h>
#define SGX_LAUNCH(kernel, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define SGX_DYN_SMEM(type, name)                                            \
    extern __shared__ __attribute__((aligned(16))) unsigned char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
typedef float sgx_f32x16 __attribute__((ext_vector_type(16)));
typedef float sgx_f32x4 __attribute__((ext_vector_type(4)));
#endif

#include "../../include/sgx_hip.h"

// ---- status / error plumbing -------------------------------------------------------------------
void sgx_set_error(const char* fmt, ...);
#define SGX_FAIL(code, ...)          \
    do {                             \
        sgx_set_error(__VA_ARGS__);  \
        return (code);               \
    } while (0)
#define SGX_CHECK_ARG(cond, ...) \
    do {                         \
        if (!(cond)) SGX_FAIL(SGX_ERR_BAD_ARG, __VA_ARGS__); \
    } while (0)
#define SGX_MEMSET_ASYNC(ptr, val, bytes, stream)                                                   \
    do {                                                                                            \
        hipError_t e__ = hipMemsetAsync((ptr), (val), (bytes), (hipStream_t)(stream));              \
        if (e__ != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "hipMemsetAsync: %s", hipGetErrorString(e__)); \
    } while (0)
#define SGX_CHECK_LAUNCH(what)                                                         \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "%s: %s", what, hipGetErrorString(e__)); \
    } while (0)

static inline int sgx_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation codes shared by host and device
#define SGX_ACT_NONE 0
#define SGX_ACT_RELU 1
#define SGX_ACT_SILU 2

__device__ __forceinline__ float sgx_act(float v, int act) {
    if (act == SGX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == SGX_ACT_SILU) return v / (1.f + expf(-v));
    return v;
}
// d act(v) / dv given the pre-activation v
__device__ __forceinline__ float sgx_act_grad(float v, int act) {
    if (act == SGX_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == SGX_ACT_SILU) {
        float s = 1.f / (1.f + expf(-v));
        return s * (1.f + v * (1.f - s));
    }
    return 1.f;
}

__device__ __forceinline__ float4 sgx_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sgx_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
