// TEST INFRASTRUCTURE ONLY.  Host-side emulation of the small HIP/gfx950 subset our kernels use, so
// that kernel *logic* (tile indexing, MFMA fragment layouts, wave reductions, barriers) can be
// exercised by the `-m "not gpu"` test-suite in a container that has no GPU.  The product library
// (libsgx_hip.so, built by hipcc for gfx950) never contains or loads this; bench.py, smoke() and the
// `-m gpu` tests never load libsgx_emu.so.
//
// Model: one FIBER (a user-space context on its own stack) per HIP thread of ONE workgroup, all run by the launching OS thread;
// workgroups run one after another.  A fiber runs until it reaches a barrier that is not complete yet, then the scheduler resumes the
// next one - a barrier of 256 lanes is 256 register swaps, not 256 futex round trips (the OS-thread form spent 4/5 of the CPU suite in the kernel).
//  * __syncthreads()          -> workgroup barrier (participants that already returned are dropped)
//  * __shfl*/__ballot/MFMA    -> wave(64)-collective exchange through a per-wave buffer + wave barrier
//  * MFMA fragment layouts follow /opt/skills/guides/cdna_hip_programming.md section 3:
//      32x32x2 f32 : A lane l -> A[i=l&31][k=l>>5], B lane l -> B[k=l>>5][j=l&31],
//                    D reg r  -> row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31
//      16x16x4 f32 : A[l&15][k=l>>4], B[k=l>>4][l&15], D reg r -> row=(l>>4)*4+r, col=l&15
#pragma once
#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#ifndef __restrict__
#define __restrict__
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_emu {
    unsigned x, y, z;
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    memset(p, v, n);
    return 0;
}
static inline hipError_t hipMemcpyAsyncD2D(void* d, const void* s, size_t n, hipStream_t) {
    memcpy(d, s, n);
    return 0;
}

struct float4 {
    float x, y, z, w;
};
struct float2 {
    float x, y;
};
struct int4 {
    int x, y, z, w;
};
struct int2 {
    int x, y;
};
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
struct alignas(8) uint2 {
    unsigned x, y;
};
struct alignas(16) uint4 {
    unsigned x, y, z, w;
};
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }

namespace sgx_emu {

// Generation barrier of cooperative fibers (one OS thread: no locks).  The last arrival bumps the generation; the others hand the
// processor back to the scheduler, which resumes a waiting fiber only once its barrier's generation has moved.
void fiber_wait(const unsigned long* gen, unsigned long seen);  // hip_emu.cpp
class Barrier {
  public:
    void reset(int n) {
        count_ = n;
        waiting_ = 0;
    }
    void wait() {
        const unsigned long my = gen_;
        if (++waiting_ >= count_) {
            waiting_ = 0;
            ++gen_;
            return;
        }
        fiber_wait(&gen_, my);
    }
    void drop() {  // a participant that returned from the kernel no longer counts
        --count_;
        if (count_ > 0 && waiting_ >= count_) {
            waiting_ = 0;
            ++gen_;
        }
    }

  private:
    int count_ = 0, waiting_ = 0;
    unsigned long gen_ = 0;
};

struct WaveState {
    Barrier bar;
    // double-buffered exchange area (one barrier per collective)
    uint64_t xbuf[2][64][4];
    bool active[2][64];
};

struct BlockState {
    Barrier bar;
    std::vector<WaveState> waves;
    alignas(16) unsigned char dyn_smem[160 * 1024];
};

extern BlockState* g_block;
extern thread_local uint3_emu t_threadIdx, t_blockIdx;
extern thread_local dim3 t_blockDim, t_gridDim;
extern thread_local int t_lane, t_wave;
extern thread_local unsigned t_xgen;

void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);

// wave-collective exchange: every live lane deposits up to 4 x 64-bit words, then reads anyone's.
struct Xchg {
    int buf;
    WaveState* w;
};
static inline Xchg xchg_put(const uint64_t* vals, int n) {
    WaveState& w = g_block->waves[t_wave];
    int b = (t_xgen++) & 1;
    for (int i = 0; i < n; ++i) w.xbuf[b][t_lane][i] = vals[i];
    w.active[b][t_lane] = true;
    w.bar.wait();
    return Xchg{b, &w};
}
static inline void xchg_done(const Xchg& x) {
    // mark inactive for the generation after next (safe: everyone passed this generation's barrier
    // before anyone can reach generation+2's write on the same buffer)
    (void)x;
}
}  // namespace sgx_emu

#define threadIdx (sgx_emu::t_threadIdx)
#define blockIdx (sgx_emu::t_blockIdx)
#define blockDim (sgx_emu::t_blockDim)
#define gridDim (sgx_emu::t_gridDim)
#define warpSize 64

static inline void __syncthreads() { sgx_emu::g_block->bar.wait(); }
static inline void* sgx_emu_dyn_smem() { return sgx_emu::g_block->dyn_smem; }

template <typename T>
static inline T __shfl(T v, int src) {
    uint64_t u = 0;
    memcpy(&u, &v, sizeof(T));
    auto x = sgx_emu::xchg_put(&u, 1);
    T r;
    memcpy(&r, &x.w->xbuf[x.buf][src & 63][0], sizeof(T));
    return r;
}
template <typename T>
static inline T __shfl_xor(T v, int mask) {
    return __shfl(v, sgx_emu::t_lane ^ mask);
}
template <typename T>
static inline T __shfl_down(T v, int d) {
    int s = sgx_emu::t_lane + d;
    return __shfl(v, s > 63 ? sgx_emu::t_lane : s);
}
template <typename T>
static inline T __shfl_up(T v, int d) {
    int s = sgx_emu::t_lane - d;
    return __shfl(v, s < 0 ? sgx_emu::t_lane : s);
}
static inline unsigned long long __ballot(int pred) {
    uint64_t u = pred ? 1 : 0;
    // lanes that already exited contribute 0: clear first
    auto x = sgx_emu::xchg_put(&u, 1);
    unsigned long long m = 0;
    int nl = (int)(blockDim.x * blockDim.y * blockDim.z) - sgx_emu::t_wave * 64;
    if (nl > 64) nl = 64;
    for (int l = 0; l < nl; ++l)
        if (x.w->xbuf[x.buf][l][0]) m |= (1ull << l);
    return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }

typedef float sgx_f32x16 __attribute__((ext_vector_type(16)));
typedef float sgx_f32x4 __attribute__((ext_vector_type(4)));

static inline sgx_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, sgx_f32x16 c, int, int, int) {
    uint64_t u[2] = {0, 0};
    memcpy(&u[0], &a, 4);
    memcpy(&u[1], &b, 4);
    auto x = sgx_emu::xchg_put(u, 2);
    const int l = sgx_emu::t_lane;
    const int col = l & 31;
    sgx_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = d[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, &x.w->xbuf[x.buf][row + 32 * k][0], 4);  // A[i=row][k] lives in lane row+32k
            memcpy(&bv, &x.w->xbuf[x.buf][col + 32 * k][1], 4);  // B[k][j=col] lives in lane col+32k
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
static inline sgx_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, sgx_f32x4 c, int, int, int) {
    uint64_t u[2] = {0, 0};
    memcpy(&u[0], &a, 4);
    memcpy(&u[1], &b, 4);
    auto x = sgx_emu::xchg_put(u, 2);
    const int l = sgx_emu::t_lane;
    const int col = l & 15;
    sgx_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = d[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, &x.w->xbuf[x.buf][row + 16 * k][0], 4);
            memcpy(&bv, &x.w->xbuf[x.buf][col + 16 * k][1], 4);
            acc = fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// atomics (plain atomics: other processes / Python threads never touch a launch's memory, but keep the device semantics)
static inline float atomicAdd(float* p, float v) {
    auto* a = reinterpret_cast<std::atomic<float>*>(p);
    float old = a->load();
    while (!a->compare_exchange_weak(old, old + v)) {
    }
    return old;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicMax(int* p, int v) {
    int old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {
    }
    return old;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicOr(unsigned long long* p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

static inline float __fdividef(float a, float b) { return a / b; }
static inline unsigned __float_as_uint(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float __uint_as_float(unsigned u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline int __float_as_int(float f) {
    int u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float __int_as_float(int u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
using std::max;
using std::min;
