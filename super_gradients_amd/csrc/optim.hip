// Optimizer and EMA steps over flat fp32 arenas: one HBM-bound launch per arena instead of the reference's
// ~500 (AdamW foreach) / 921x3 (ModelEMA.update, training/utils/ema.py:139-141) tiny ATen kernels.
// AdamW follows torch.optim.AdamW (decoupled weight decay, bias correction, eps added after sqrt):
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// SGD follows torch.optim.SGD (coupled weight decay, momentum buffer initialised with the first gradient).
// Per-segment weight decay reproduces the zero-WD groups of optimizer_utils.py:32-59.
#include "sgx_common.h"

// first segment whose end is beyond i (segments ascend): binary search, ~9 L1-resident probes for a 500-tensor model
__device__ __forceinline__ float seg_wd_of(long i, const long long* seg_end, const float* seg_wd, int nseg) {
    int lo = 0, hi = nseg;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (i < (long)seg_end[mid]) hi = mid;
        else lo = mid + 1;
    }
    return lo < nseg ? seg_wd[lo] : 0.f;
}

__device__ __forceinline__ void adamw_one(float& pi, float gi, float& mi, float& vi, float wd, float lr, float b1, float b2, float eps, float bc1, float bc2s) {
    pi = pi * (1.f - lr * wd);
    mi = b1 * mi + (1.f - b1) * gi;
    vi = b2 * vi + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2s + eps;
    pi -= (lr / bc1) * (mi / denom);
}
// Four elements per lane (16-byte accesses) and ONE segment search per four: the per-element form spent nine dependent L1 probes per
// element on the weight-decay lookup and moved 4 bytes per memory instruction - 133 us for the 12.9 M parameters of YOLO-NAS-S (r5m),
// 2.7 TB/s.  A group of four that straddles a segment end looks every element up by itself.  Same arithmetic per element.
__global__ void adamw_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps, float bc1, float bc2s,
                             const long long* seg_end, const float* seg_wd, int nseg, const float* grad_scale) {
    const float gs = grad_scale ? grad_scale[0] : 1.f;
    const long n4 = n / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const long i = 4 * q;
        float4 P = sgx_ld4(p + i), G = sgx_ld4(g + i), M = sgx_ld4(m + i), V = sgx_ld4(v + i);
        int lo = 0, hi = nseg;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (i < (long)seg_end[mid]) hi = mid;
            else lo = mid + 1;
        }
        float w0, w1, w2, w3;
        w0 = lo < nseg ? seg_wd[lo] : 0.f;
        if (lo >= nseg || i + 3 < (long)seg_end[lo]) w1 = w2 = w3 = w0;
        else {
            w1 = seg_wd_of(i + 1, seg_end, seg_wd, nseg);
            w2 = seg_wd_of(i + 2, seg_end, seg_wd, nseg);
            w3 = seg_wd_of(i + 3, seg_end, seg_wd, nseg);
        }
        adamw_one(P.x, G.x * gs, M.x, V.x, w0, lr, b1, b2, eps, bc1, bc2s);
        adamw_one(P.y, G.y * gs, M.y, V.y, w1, lr, b1, b2, eps, bc1, bc2s);
        adamw_one(P.z, G.z * gs, M.z, V.z, w2, lr, b1, b2, eps, bc1, bc2s);
        adamw_one(P.w, G.w * gs, M.w, V.w, w3, lr, b1, b2, eps, bc1, bc2s);
        sgx_st4(p + i, P); sgx_st4(m + i, M); sgx_st4(v + i, V);
    }
    for (long i = 4 * n4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {  // (n % 4 trailing elements)
        float pi = p[i], mi = m[i], vi = v[i];
        adamw_one(pi, g[i] * gs, mi, vi, seg_wd_of(i, seg_end, seg_wd, nseg), lr, b1, b2, eps, bc1, bc2s);
        p[i] = pi; m[i] = mi; v[i] = vi;
    }
}
extern "C" int32_t sgx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                  int32_t step, const int64_t* seg_end, const float* seg_wd, int32_t nseg, const float* grad_scale, void* stream) {
    if (n <= 0) return SGX_OK;
    SGX_CHECK_ARG(p && g && m && v && step >= 1 && (nseg == 0 || (seg_end && seg_wd)), "adamw: bad args");
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    SGX_CHECK_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0, "adamw: arenas must be 16-byte aligned");
    long blocks = (n / 4 + 255) / 256 + 1;
    SGX_LAUNCH(adamw_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, p, g, m, v, (long)n, lr, beta1, beta2, eps, bc1, bc2s,
               (const long long*)seg_end, seg_wd, nseg, grad_scale);
    SGX_CHECK_LAUNCH("adamw");
    return SGX_OK;
}

__global__ void sgd_kernel(float* p, const float* g, float* mom, long n, float lr, float momentum, float dampening, int nesterov, int first,
                           const long long* seg_end, const float* seg_wd, int nseg) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float wd = seg_wd_of(i, seg_end, seg_wd, nseg);
        float gi = g[i] + wd * p[i];
        if (momentum != 0.f) {
            float b = first ? gi : momentum * mom[i] + (1.f - dampening) * gi;
            mom[i] = b;
            gi = nesterov ? gi + momentum * b : b;
        }
        p[i] -= lr * gi;
    }
}
extern "C" int32_t sgx_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float dampening, int32_t nesterov,
                                int32_t first_step, const int64_t* seg_end, const float* seg_wd, int32_t nseg, void* stream) {
    if (n <= 0) return SGX_OK;
    SGX_CHECK_ARG(p && g && (momentum == 0.f || mom) && (nseg == 0 || (seg_end && seg_wd)), "sgd: bad args");
    long blocks = (n + 255) / 256;
    SGX_LAUNCH(sgd_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, p, g, mom, (long)n, lr, momentum, dampening, nesterov,
               first_step, (const long long*)seg_end, seg_wd, nseg);
    SGX_CHECK_LAUNCH("sgd");
    return SGX_OK;
}

// ema = ema*decay + (1-decay)*p      (training/utils/ema.py:139-141)
__global__ void ema_kernel(float* ema, const float* p, long n, float decay) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) ema[i] = ema[i] * decay + (1.f - decay) * p[i];
}
extern "C" int32_t sgx_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream) {
    if (n <= 0) return SGX_OK;
    SGX_CHECK_ARG(ema && p, "ema: null pointer");
    long blocks = (n + 255) / 256;
    SGX_LAUNCH(ema_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, ema, p, (long)n, decay);
    SGX_CHECK_LAUNCH("ema");
    return SGX_OK;
}
