// Pooling kernels (HBM/L2-bound, no MFMA): SPP max-pool k=5/9/13 stride 1, ResNet 3x3 s2 max-pool,
// global average pool.  One thread owns a float4 channel group of one output pixel.
// Reference call sites: include/sgx_hip.h (Pooling section).
#include "sgx_common.h"
#include <atomic>

__global__ void maxpool_fwd_kernel(int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, const float* x, long x_ld_pix,
                                   long x_ld_img, float* y, long y_ld_pix, long y_ld_img, int* argmax) {
    const int C4 = C / 4;
    long n = (long)N * Ho * Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long pix = i / C4;
        int wo = (int)(pix % Wo);
        long t = pix / Wo;
        int ho = (int)(t % Ho);
        int img = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int ix = -1, iy = -1, iz = -1, iw = -1;
        for (int r = 0; r < k; ++r) {
            int hi = ho * stride - pad + r;
            if (hi < 0 || hi >= H) continue;
            for (int q = 0; q < k; ++q) {
                int wi = wo * stride - pad + q;
                if (wi < 0 || wi >= W) continue;
                float4 v = sgx_ld4(x + (long)img * x_ld_img + ((long)hi * W + wi) * x_ld_pix + c);
                int idx = hi * W + wi;
                if (v.x > m.x || ix < 0) { m.x = v.x; ix = idx; }
                if (v.y > m.y || iy < 0) { m.y = v.y; iy = idx; }
                if (v.z > m.z || iz < 0) { m.z = v.z; iz = idx; }
                if (v.w > m.w || iw < 0) { m.w = v.w; iw = idx; }
            }
        }
        sgx_st4(y + (long)img * y_ld_img + ((long)ho * Wo + wo) * y_ld_pix + c, m);
        if (argmax) {
            int* a = argmax + pix * C + c;
            a[0] = ix; a[1] = iy; a[2] = iz; a[3] = iw;
        }
    }
}

// ---- stride-1 pooling of a map that fits LDS (the SPP's 5 / 9 / 13 windows on the last backbone map): one workgroup owns MP_CG channels of one
// image.  The direct kernel above reads k x k inputs per output (169 float4 loads at k = 13: 88 us per call on the 32 x 20 x 20 x 384 map,
// r4r); here the map is staged once and the window is walked separably - per row the first largest value of the k columns, then down the k rows
// the first row holding the largest of those: exactly the first maximum in row-major window order, which is what ATen returns.
#define MP_CG 8
#define MP_TILE_MAX_LDS 65536  // dynamic LDS a launch may ask for without raising the function's limit
static long maxpool_fwd_tile_lds(int H, int W, int Wo) { return (long)H * W * MP_CG * 4 + (long)H * Wo * MP_CG * 6; }
__global__ __launch_bounds__(256) void maxpool_fwd_tile_kernel(int H, int W, int C, int k, int pad, int Ho, int Wo, const float* x, long x_ld_pix,
                                                               long x_ld_img, float* y, long y_ld_pix, long y_ld_img, int* argmax) {
    SGX_DYN_SMEM(float, smem);
    float* s_x = smem;                                                   // [H * W][MP_CG]
    float* s_v = smem + (long)H * W * MP_CG;                              // [H * Wo][MP_CG]: row maxima
    unsigned short* s_q = (unsigned short*)(s_v + (long)H * Wo * MP_CG);  // their columns
    const int groups = C / MP_CG;
    const int img = blockIdx.x / groups, c0 = (blockIdx.x % groups) * MP_CG;
    for (int i = threadIdx.x; i < H * W * (MP_CG / 4); i += blockDim.x) {
        const int pix = i / (MP_CG / 4), h4 = (i % (MP_CG / 4)) * 4;
        const float4 v = sgx_ld4(x + (long)img * x_ld_img + (long)pix * x_ld_pix + c0 + h4);
        float* d = s_x + pix * MP_CG + h4;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // (a lane owns FOUR channels: 16-byte LDS reads and 16-byte stores - as one channel per lane the kernel took 52 / 67 / 82 us for k = 5 / 9 /
    // 13 on the 32 x 20 x 20 x 384 map, r4w, bound by its instruction count and its 4-byte stores)
    for (int i = threadIdx.x; i < H * Wo * (MP_CG / 4); i += blockDim.x) {
        const int c4 = (i % (MP_CG / 4)) * 4, t = i / (MP_CG / 4), wo = t % Wo, hi = t / Wo;
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int q[4] = {-1, -1, -1, -1};
        for (int j = 0; j < k; ++j) {
            const int wi = wo - pad + j;
            if (wi < 0 || wi >= W) continue;
            const float4 v4 = sgx_ld4(s_x + (hi * W + wi) * MP_CG + c4);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (v[z] > m[z] || q[z] < 0) { m[z] = v[z]; q[z] = wi; }
        }
        sgx_st4(s_v + t * MP_CG + c4, make_float4(m[0], m[1], m[2], m[3]));
        unsigned short* sq = s_q + t * MP_CG + c4;
        sq[0] = (unsigned short)q[0]; sq[1] = (unsigned short)q[1]; sq[2] = (unsigned short)q[2]; sq[3] = (unsigned short)q[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < Ho * Wo * (MP_CG / 4); i += blockDim.x) {
        const int c4 = (i % (MP_CG / 4)) * 4, o = i / (MP_CG / 4), wo = o % Wo, ho = o / Wo;
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int idx[4] = {-1, -1, -1, -1};
        for (int r = 0; r < k; ++r) {
            const int hi = ho - pad + r;
            if (hi < 0 || hi >= H) continue;
            const int e = (hi * Wo + wo) * MP_CG + c4;
            const float4 v4 = sgx_ld4(s_v + e);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int z = 0; z < 4; ++z)
                if (v[z] > m[z] || idx[z] < 0) { m[z] = v[z]; idx[z] = hi * W + (int)s_q[e + z]; }
        }
        sgx_st4(y + (long)img * y_ld_img + (long)o * y_ld_pix + c0 + c4, make_float4(m[0], m[1], m[2], m[3]));
        if (argmax) {
            int* am = argmax + ((long)img * Ho * Wo + o) * C + c0 + c4;
            am[0] = idx[0]; am[1] = idx[1]; am[2] = idx[2]; am[3] = idx[3];
        }
    }
}

extern "C" int32_t sgx_maxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const float* x,
                                   int64_t x_ld_pix, int64_t x_ld_img, float* y, int64_t y_ld_pix, int64_t y_ld_img, int32_t* argmax,
                                   void* stream) {
    SGX_CHECK_ARG(x && y && C % 4 == 0 && k > 0 && stride > 0, "maxpool_fwd: bad args");
    SGX_CHECK_ARG(pad >= 0 && 2 * pad <= k, "maxpool_fwd: pad %d must be at most half the window %d (F.max_pool2d's own rule)", pad, k);
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (stride == 1 && C % MP_CG == 0 && Ho > 0 && Wo > 0 && (long)H * W < 65536 && maxpool_fwd_tile_lds(H, W, Wo) <= MP_TILE_MAX_LDS && k > 2) {
        SGX_LAUNCH(maxpool_fwd_tile_kernel, dim3((unsigned)(N * (C / MP_CG))), dim3(256), (unsigned)maxpool_fwd_tile_lds(H, W, Wo), stream, H, W, C, k, pad,
                   Ho, Wo, x, (long)x_ld_pix, (long)x_ld_img, y, (long)y_ld_pix, (long)y_ld_img, argmax);
        SGX_CHECK_LAUNCH("maxpool_fwd (tile)");
        return SGX_OK;
    }
    long n = (long)N * Ho * Wo * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(maxpool_fwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, k, stride, pad, Ho, Wo,
               x, (long)x_ld_pix, (long)x_ld_img, y, (long)y_ld_pix, (long)y_ld_img, argmax);
    SGX_CHECK_LAUNCH("maxpool_fwd");
    return SGX_OK;
}

__global__ void maxpool_bwd_kernel(int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, const int* argmax, const float* dy,
                                   long dy_ld_pix, long dy_ld_img, float* dx, long dx_ld_pix, long dx_ld_img, int accumulate) {
    const int C4 = C / 4;
    long n = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long pix = i / C4;
        int wi = (int)(pix % W);
        long t = pix / W;
        int hi = (int)(t % H);
        int img = (int)(t / H);
        const int me = hi * W + wi;
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // output windows containing (hi, wi): ho*stride - pad <= hi <= ho*stride - pad + k - 1
        int ho_lo = (hi + pad - k + 1 + stride - 1);
        ho_lo = ho_lo < 0 ? 0 : ho_lo / stride;
        int ho_hi = (hi + pad) / stride;
        if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        int wo_lo = (wi + pad - k + 1 + stride - 1);
        wo_lo = wo_lo < 0 ? 0 : wo_lo / stride;
        int wo_hi = (wi + pad) / stride;
        if (wo_hi > Wo - 1) wo_hi = Wo - 1;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                long op = ((long)img * Ho + ho) * Wo + wo;
                const int* a = argmax + op * C + c;
                float4 d = sgx_ld4(dy + (long)img * dy_ld_img + ((long)ho * Wo + wo) * dy_ld_pix + c);
                if (a[0] == me) g.x += d.x;
                if (a[1] == me) g.y += d.y;
                if (a[2] == me) g.z += d.z;
                if (a[3] == me) g.w += d.w;
            }
        float* o = dx + (long)img * dx_ld_img + (long)me * dx_ld_pix + c;
        if (accumulate) {
            float4 u = sgx_ld4(o);
            g.x += u.x; g.y += u.y; g.z += u.z; g.w += u.w;
        }
        sgx_st4(o, g);
    }
}

// The same for the backward pass: the direct kernel reads the argmax (16 B) and the gradient (16 B) of every window an input pixel lies in -
// 169 x 32 B per float4 at k = 13, 150 - 350 us per call (r4r: 0.68 ms per step for the SPP's three pools).  Here the output map's argmax
// (as 16-bit pixel indices) and gradient are staged once per MP_CG channels of an image and the windows are walked in LDS, in the same order:
// the sums are bit-identical to the direct kernel's.
__global__ __launch_bounds__(256) void maxpool_bwd_tile_kernel(int H, int W, int C, int k, int pad, int Ho, int Wo, const int* argmax, const float* dy,
                                                               long dy_ld_pix, long dy_ld_img, float* dx, long dx_ld_pix, long dx_ld_img, int accumulate) {
    SGX_DYN_SMEM(float, smem);
    float* s_dy = smem;                                                        // [Ho * Wo][MP_CG]
    unsigned short* s_ix = (unsigned short*)(smem + (long)Ho * Wo * MP_CG);     // [Ho * Wo][MP_CG]
    const int groups = C / MP_CG;
    const int img = blockIdx.x / groups, c0 = (blockIdx.x % groups) * MP_CG;
    for (int i = threadIdx.x; i < Ho * Wo * (MP_CG / 4); i += blockDim.x) {
        const int o = i / (MP_CG / 4), h4 = (i % (MP_CG / 4)) * 4;
        const float4 d = sgx_ld4(dy + (long)img * dy_ld_img + (long)o * dy_ld_pix + c0 + h4);
        const int* a = argmax + ((long)img * Ho * Wo + o) * C + c0 + h4;
        float* sd = s_dy + o * MP_CG + h4;
        unsigned short* si = s_ix + o * MP_CG + h4;
        sd[0] = d.x; sd[1] = d.y; sd[2] = d.z; sd[3] = d.w;
        si[0] = (unsigned short)a[0]; si[1] = (unsigned short)a[1]; si[2] = (unsigned short)a[2]; si[3] = (unsigned short)a[3];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < H * W * MP_CG; i += blockDim.x) {
        const int c = i % MP_CG, me = i / MP_CG, wi = me % W, hi = me / W;
        int ho_lo = hi + pad - k + 1, ho_hi = hi + pad, wo_lo = wi + pad - k + 1, wo_hi = wi + pad;
        if (ho_lo < 0) ho_lo = 0;
        if (wo_lo < 0) wo_lo = 0;
        if (ho_hi > Ho - 1) ho_hi = Ho - 1;
        if (wo_hi > Wo - 1) wo_hi = Wo - 1;
        float g = 0.f;
        for (int ho = ho_lo; ho <= ho_hi; ++ho)
            for (int wo = wo_lo; wo <= wo_hi; ++wo) {
                const int e = (ho * Wo + wo) * MP_CG + c;
                if ((int)s_ix[e] == me) g += s_dy[e];
            }
        float* o = dx + (long)img * dx_ld_img + (long)me * dx_ld_pix + c0 + c;
        *o = accumulate ? *o + g : g;
    }
}

// The gather forms above cost k x k window tests per input element whatever they read from (r4z: the LDS form 198 us per call against the
// direct kernel's 226 - 1.35 G tests per step for the SPP's 5 / 9 / 13 pools, bound by their ~5 vector instructions each).  The scatter form
// costs ONE add per OUTPUT element: a wave owns 64 channels of one image, lane = channel, keeps that (image, channels) slice of dx in LDS
// (H x W x 64 floats: 100 KB for the 20 x 20 map) and walks the outputs in ascending order - every lane adds its gradient at its arg-max
// pixel.  Lanes never share an address (different channels) and a lane's additions happen in output order: the sums are deterministic and in
// the order of ATen's CPU kernel.  Eight outputs of loads in flight per lane; no atomics.
// Round 5: 32 channels per workgroup by default (half a wave adds; 50 KB for the 20 x 20 map).  Alone on the chip the 64-channel form is
// as fast (the add chain is latency-bound either way), but in the train step its 100 KB workgroups waited for a CU with that much LDS
// free while weight-gradient kernels of the side stream were resident: 200 us per call in the step's trace against 35 alone (r5m).
#define MP_SCATTER_CH 32
#define MP_SCATTER_THREADS 256
#define MP_SCATTER_MAX_LDS (152 * 1024)
template <int CH>
__global__ __launch_bounds__(MP_SCATTER_THREADS) void maxpool_bwd_scatter_kernel(int HW, int C, int HoWo, const int* argmax, const float* dy, long dy_ld_pix,
                                                                                 long dy_ld_img, float* dx, long dx_ld_pix, long dx_ld_img, int accumulate) {
    SGX_DYN_SMEM(float, tile);  // [HW][CH]
    constexpr int PP = MP_SCATTER_THREADS / CH;  // pixels per pass of the load / store phases (thread = (pixel slot, channel))
    const int groups = C / CH, ch = threadIdx.x % CH, ps = threadIdx.x / CH;
    const int img = blockIdx.x / groups, c = (blockIdx.x % groups) * CH + ch;
    float* const gx = dx + (long)img * dx_ld_img + c;
    // the slice's starting value: every thread, eight pixels of loads in flight (a lane's scatter below is a chain of ~1 us memory
    // round trips as it is: r4v measured 372 us per call with one wave doing everything, eight outputs at a time)
    for (int p0 = ps * 8; p0 < HW; p0 += 8 * PP) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (accumulate && p0 + u < HW) ? gx[(long)(p0 + u) * dx_ld_pix] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + u < HW) tile[(p0 + u) * CH + ch] = v[u];
    }
    __syncthreads();
    if (threadIdx.x < CH) {  // ONE (part of a) wave adds, in output order; lane = channel
        const int* const a = argmax + (long)img * HoWo * C + c;
        const float* const g = dy + (long)img * dy_ld_img + c;
        int o = 0;
        for (; o + 32 <= HoWo; o += 32) {
            int ix[32];
            float d[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                ix[u] = a[(long)(o + u) * C];
                d[u] = g[(long)(o + u) * dy_ld_pix];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (ix[u] >= 0) tile[ix[u] * CH + ch] += d[u];  // (-1: a window that lies in the padding only)
        }
        for (; o < HoWo; ++o) {
            const int ix = a[(long)o * C];
            if (ix >= 0) tile[ix * CH + ch] += g[(long)o * dy_ld_pix];
        }
    }
    __syncthreads();
    for (int p = ps; p < HW; p += PP) gx[(long)p * dx_ld_pix] = tile[p * CH + ch];
}

extern "C" int32_t sgx_maxpool_bwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const int32_t* argmax,
                                   const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img, float* dx, int64_t dx_ld_pix, int64_t dx_ld_img,
                                   int32_t accumulate, void* stream) {
    SGX_CHECK_ARG(argmax && dy && dx && C % 4 == 0, "maxpool_bwd: bad args");
    SGX_CHECK_ARG(pad >= 0 && 2 * pad <= k, "maxpool_bwd: pad %d must be at most half the window %d", pad, k);
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (C % MP_SCATTER_CH == 0 && Ho > 0 && Wo > 0 && (long)H * W * MP_SCATTER_CH * 4 <= MP_SCATTER_MAX_LDS) {
        const unsigned lds = (unsigned)((long)H * W * MP_SCATTER_CH * 4);
#ifndef SGX_EMU
        // dynamic LDS beyond 64 KB is opt-in per function AND per device: one flag bit per device ordinal
        static std::atomic<unsigned long long> raised{0ull};
        int dev = 0;
        (void)hipGetDevice(&dev);
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute((const void*)maxpool_bwd_scatter_kernel<MP_SCATTER_CH>, hipFuncAttributeMaxDynamicSharedMemorySize, MP_SCATTER_MAX_LDS) != hipSuccess)
                SGX_FAIL(SGX_ERR_HIP, "maxpool_bwd: cannot raise the scatter kernel's dynamic LDS limit");
            raised.fetch_or(bit, std::memory_order_release);
        }
#endif
        SGX_LAUNCH(maxpool_bwd_scatter_kernel<MP_SCATTER_CH>, dim3((unsigned)(N * (C / MP_SCATTER_CH))), dim3(MP_SCATTER_THREADS), lds, stream, H * W, C, Ho * Wo, argmax, dy,
                   (long)dy_ld_pix, (long)dy_ld_img, dx, (long)dx_ld_pix, (long)dx_ld_img, accumulate);
        SGX_CHECK_LAUNCH("maxpool_bwd (scatter)");
        return SGX_OK;
    }
    if (stride == 1 && C % MP_CG == 0 && Ho > 0 && Wo > 0 && (long)H * W < 65536 && (long)Ho * Wo * MP_CG * 6 <= MP_TILE_MAX_LDS && k > 2) {
        SGX_LAUNCH(maxpool_bwd_tile_kernel, dim3((unsigned)(N * (C / MP_CG))), dim3(256), (unsigned)((long)Ho * Wo * MP_CG * 6), stream, H, W, C, k, pad, Ho, Wo,
                   argmax, dy, (long)dy_ld_pix, (long)dy_ld_img, dx, (long)dx_ld_pix, (long)dx_ld_img, accumulate);
        SGX_CHECK_LAUNCH("maxpool_bwd (tile)");
        return SGX_OK;
    }
    long n = (long)N * H * W * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(maxpool_bwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, k, stride, pad, Ho, Wo,
               argmax, dy, (long)dy_ld_pix, (long)dy_ld_img, dx, (long)dx_ld_pix, (long)dx_ld_img, accumulate);
    SGX_CHECK_LAUNCH("maxpool_bwd");
    return SGX_OK;
}

__global__ void avgpool_fwd_kernel(int N, int HW, int C, const float* x, long ld_pix, long ld_img, float* y) {
    const int C4 = C / 4;
    long n = (long)N * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        int img = (int)(i / C4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = 0; p < HW; ++p) {
            float4 v = sgx_ld4(x + (long)img * ld_img + (long)p * ld_pix + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        float inv = 1.f / (float)HW;
        sgx_st4(y + (long)img * C + c, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
    }
}
extern "C" int32_t sgx_avgpool_fwd(int32_t N, int32_t HW, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img, float* y, void* stream) {
    SGX_CHECK_ARG(x && y && C % 4 == 0, "avgpool_fwd: bad args");
    long n = (long)N * (C / 4), blocks = (n + 63) / 64;
    SGX_LAUNCH(avgpool_fwd_kernel, dim3((unsigned)blocks), dim3(64), 0, stream, N, HW, C, x, (long)x_ld_pix, (long)x_ld_img, y);
    SGX_CHECK_LAUNCH("avgpool_fwd");
    return SGX_OK;
}
__global__ void avgpool_bwd_kernel(int N, int HW, int C, const float* dy, float* dx, long ld_pix, long ld_img) {
    const int C4 = C / 4;
    long n = (long)N * HW * C4;
    float inv = 1.f / (float)HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long t = i / C4;
        int p = (int)(t % HW);
        int img = (int)(t / HW);
        float4 d = sgx_ld4(dy + (long)img * C + c);
        sgx_st4(dx + (long)img * ld_img + (long)p * ld_pix + c, make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv));
    }
}
extern "C" int32_t sgx_avgpool_bwd(int32_t N, int32_t HW, int32_t C, const float* dy, float* dx, int64_t dx_ld_pix, int64_t dx_ld_img, void* stream) {
    SGX_CHECK_ARG(dy && dx && C % 4 == 0, "avgpool_bwd: bad args");
    long n = (long)N * HW * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(avgpool_bwd_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, N, HW, C, dy, dx, (long)dx_ld_pix,
               (long)dx_ld_img);
    SGX_CHECK_LAUNCH("avgpool_bwd");
    return SGX_OK;
}
