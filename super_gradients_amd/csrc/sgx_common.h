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
// Workgroup count of a GRID-STRIDE kernel (every workgroup loops `for (i = global id; i < n; i += grid size)`): what the kernel computes does
// not depend on it.  The host emulation pays two 256-thread barriers per emulated workgroup, and the per-step batch kernels (weight
// transposes, QARepVGG filter preparation, filter planes) launch hundreds of workgroups per job - most of a small test network's step.
#define SGX_STRIDE_GRID(n) ((n) < 2 ? (n) : 2)
#else
#define SGX_STRIDE_GRID(n) (n)
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

// n / d for 0 <= n < 2^31 and a launch-constant d >= 1 as one multiply-high, one shift and one masked add - branch-free:
// p = 31 + ceil(log2 d), mul = ceil(2^p / d) < 2^32, n / d = (n * mul) >> p; d = 1 (p would be 31): mul = 0 and the mask passes n through.
// Host side fills the triple, device side applies it.
struct sgx_fastdiv {
    unsigned mul, shr, one;
};
static inline sgx_fastdiv sgx_make_fastdiv(int d) {
    sgx_fastdiv f = {0u, 0u, 0xffffffffu};
    if (d > 1) {
        int lg = 0;
        while ((1L << lg) < d) ++lg;  // ceil(log2 d)
        const unsigned p = 31u + (unsigned)lg;
        f.mul = (unsigned)((((unsigned long long)1 << p) + (unsigned)d - 1u) / (unsigned)d);
        f.shr = p - 32u;
        f.one = 0u;
    }
    return f;
}
__device__ __forceinline__ int sgx_fdiv(int n, const sgx_fastdiv& f) {
    return (int)(((unsigned)(((unsigned long long)(unsigned)n * f.mul) >> 32) >> f.shr) + ((unsigned)n & f.one));
}

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

// ---- buffer (SRD) loads: 32-bit per-lane byte offset against a wave-uniform base, hardware bounds check ----------
// An offset >= the buffer's byte count returns zeros: the conv kernels encode "this element is padding / outside the
// tile" as SGX_BUF_OOB instead of branching around the load.  Buffers are limited to 2 GiB - 1 (offsets stay positive).
#define SGX_BUF_OOB 0x80000000u
#define SGX_BUF_MAX 0x7fffffffL
#ifdef SGX_EMU
struct sgx_buf {
    const char* base;
    unsigned bytes;
};
static inline sgx_buf sgx_make_buf(const void* p, long bytes) {
    if (bytes < 0) bytes = 0;
    return sgx_buf{(const char*)p, (unsigned)(bytes > SGX_BUF_MAX ? SGX_BUF_MAX : bytes)};
}
static inline float4 sgx_buf_ld4(const sgx_buf& b, unsigned off) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (off < b.bytes && off + 16u <= b.bytes) memcpy(&v, b.base + off, 16);
    return v;
}
// per-lane offset + a wave-uniform offset (the hardware form below: voffset VGPR + soffset SGPR, no vector add; the bounds check of a raw
// buffer looks at the per-lane offset alone, so a lane masked with SGX_BUF_OOB stays masked whatever soff is)
static inline float4 sgx_buf_ld4_so(const sgx_buf& b, unsigned voff, unsigned soff) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (voff < b.bytes && (unsigned long long)voff + soff + 16ull <= b.bytes) memcpy(&v, b.base + voff + soff, 16);
    return v;
}
static inline uint4 sgx_buf_ld4u(const sgx_buf& b, unsigned off) {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (off < b.bytes && off + 16u <= b.bytes) memcpy(&v, b.base + off, 16);
    return v;
}
#else
typedef __amdgpu_buffer_rsrc_t sgx_buf;
__device__ __forceinline__ sgx_buf sgx_make_buf(const void* p, long bytes) {
    if (bytes < 0) bytes = 0;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > SGX_BUF_MAX ? SGX_BUF_MAX : bytes), 0x00020000);
}
__device__ __forceinline__ float4 sgx_buf_ld4(sgx_buf b, unsigned off) {
    typedef unsigned int sgx_u32x4 __attribute__((ext_vector_type(4)));
    sgx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 0);
    // bit-cast the WHOLE vector: hipcc (ROCm 7.2) narrows the load to one dword when the lanes are bit-cast one by one
    sgx_f32x4 f = __builtin_bit_cast(sgx_f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
// per-lane offset (VGPR) + wave-uniform offset (SGPR, the instruction's soffset): no vector add per load.  A raw buffer's range check is
// made on the per-lane offset (+ the instruction's immediate) alone - SGX_BUF_OOB in `voff` masks the lane whatever `soff` is; callers keep
// voff + soff inside the buffer for the lanes that are not masked.
__device__ __forceinline__ float4 sgx_buf_ld4_so(sgx_buf b, unsigned voff, unsigned soff) {
    typedef unsigned int sgx_u32x4 __attribute__((ext_vector_type(4)));
    sgx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)voff, (int)soff, 0);
    sgx_f32x4 f = __builtin_bit_cast(sgx_f32x4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
// the same 16-byte load as raw dwords (bf16 operands: eight elements per lane)
__device__ __forceinline__ uint4 sgx_buf_ld4u(sgx_buf b, unsigned off) {
    typedef unsigned int sgx_u32x4 __attribute__((ext_vector_type(4)));
    const sgx_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b, (int)off, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}
#endif

// ---- cross-workgroup hand-over through HBM (arrival tickets) ---------------------------------------------------------------------------
// MI355X has one L2 per XCD and they are not coherent with each other for ordinary accesses.  A device-scope release FENCE makes a
// workgroup's stores visible by writing back the WHOLE L2 (buffer_wbl2 sc1) and an acquire fence invalidates it (buffer_inv sc1):
// measured (r3a) at ~3x the weight-gradient kernel's run time when every workgroup publishes a partial tile that way.  Instead the
// hand-over data itself moves with device-scope accesses (sc1: written through to / read from the memory side, no cache walk):
//   producer: sgx_st_dev* ... sgx_wait_stores() (stores acknowledged) ... __syncthreads() ... one atomicAdd on the ticket
//   consumer: (sees the last ticket) ... sgx_ld4_dev
#ifdef SGX_EMU
static inline void sgx_st_dev(float* p, float v) { *p = v; }
static inline float sgx_ld_dev(const float* p) { return *p; }
static inline void sgx_st4_dev(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
static inline float4 sgx_ld4_dev(const float* p) { return *reinterpret_cast<const float4*>(p); }
#define sgx_wait_stores() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define sgx_sched_fence() ((void)0)
#define SGX_SCHED_GROUP(mask, n) ((void)0)
#define SGX_PIN2(a, b) ((void)0)
#define sgx_wave_lds_sync() __syncthreads()  // (the emulation's lanes are separate fibers: a workgroup barrier, reached by every thread alike)
#else
// LDS hand-over between the lanes of ONE wave (a patch only this wave touches): a wave's LDS instructions execute in order, so the
// stores of all its lanes precede its later loads - only the compiler must keep them in program order and wait for the store counter.
// No workgroup barrier: in the conv epilogues every 32x32 block used to cost the whole workgroup two of them.
#define sgx_wave_lds_sync()                                \
    do {                                                   \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
// an ordering point for two register values: what produces them stays above, what consumes them below (an empty volatile asm statement;
// volatile asm statements and scheduling fences keep their mutual order)
#define SGX_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
// "the next `n` instructions of class `mask` (0x8 MFMA, 0x2 vector ALU, 0x100 / 0x200 LDS read / write, 0x20 vector-memory read) come here":
// a sequence of these lays out the instruction mix of a basic block - how the pipelined GEMM loop gets its split between its MFMAs
#define SGX_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
// no instruction moves across this point in the scheduler: bounds how many independent loads an unrolled loop keeps in flight (and with
// them the registers a kernel's tail claims for the WHOLE kernel: the allocation of a kernel is its hungriest region's)
#define sgx_sched_fence() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ void sgx_st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sgx_ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sgx_st4_dev(float* p, float4 v) {
    unsigned long long a, b;
    const float lo[2] = {v.x, v.y}, hi[2] = {v.z, v.w};
    memcpy(&a, lo, 8);
    memcpy(&b, hi, 8);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p) + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 sgx_ld4_dev(const float* p) {
    const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float lo[2], hi[2];
    memcpy(lo, &a, 8);
    memcpy(hi, &b, 8);
    return make_float4(lo[0], lo[1], hi[0], hi[1]);
}
#define sgx_wait_stores() __builtin_amdgcn_s_waitcnt(0)  // vmcnt(0) expcnt(0) lgkmcnt(0): every store of this wave has been acknowledged
#endif

// value of a 64-bit register in lane `src` - `src` must be the same in every lane (a scalar read: two v_readlane, no LDS crossbar)
#ifdef SGX_EMU
static inline unsigned long long sgx_readlane_u64(unsigned long long v, int src) { return __shfl(v, src); }
static inline unsigned long long sgx_uniform_u64(unsigned long long v) { return v; }
static inline int sgx_uniform_i32(int v) { return v; }
#else
__device__ __forceinline__ int sgx_uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a value that IS the same in every lane, moved to scalar registers: what is computed from it afterwards runs on the scalar unit
__device__ __forceinline__ unsigned long long sgx_uniform_u64(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long sgx_readlane_u64(unsigned long long v, int src) {
    const int s = __builtin_amdgcn_readfirstlane(src);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, s), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), s);
    return ((unsigned long long)hi << 32) | lo;
}
#endif

// Issue priority of the calling wave among the waves of its SIMD (s_setprio, 0 - 3; the default is 0).  The short dependent kernels of the
// main chain (finalize kernels, sweeps) share their SIMDs with long-running weight-gradient waves of the side stream, which the oldest-first
// arbiter otherwise prefers (r6fin2's trace: a 7 us finalize kernel takes 180 - 310 us when a wpatch_kernel launch started just before it).
#ifdef SGX_EMU
#define SGX_WAVE_PRIO(n) ((void)0)
#else
#define SGX_WAVE_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif

__device__ __forceinline__ float4 sgx_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void sgx_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
