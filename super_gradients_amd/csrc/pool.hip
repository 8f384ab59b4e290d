// Pooling kernels (HBM/L2-bound, no MFMA): SPP max-pool k=5/9/13 stride 1, ResNet 3x3 s2 max-pool,
// global average pool.  One thread owns a float4 channel group of one output pixel.
// Reference call sites: include/sgx_hip.h (Pooling section).
#include "sgx_common.h"

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

extern "C" int32_t sgx_maxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const float* x,
                                   int64_t x_ld_pix, int64_t x_ld_img, float* y, int64_t y_ld_pix, int64_t y_ld_img, int32_t* argmax,
                                   void* stream) {
    SGX_CHECK_ARG(x && y && C % 4 == 0 && k > 0 && stride > 0, "maxpool_fwd: bad args");
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
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

extern "C" int32_t sgx_maxpool_bwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const int32_t* argmax,
                                   const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img, float* dx, int64_t dx_ld_pix, int64_t dx_ld_img,
                                   int32_t accumulate, void* stream) {
    SGX_CHECK_ARG(argmax && dy && dx && C % 4 == 0, "maxpool_bwd: bad args");
    int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
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
