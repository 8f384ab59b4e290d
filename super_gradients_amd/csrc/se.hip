// Squeeze-excitation gates and nearest x2 up-sampling of PP-YOLOE (HBM-bound, no MFMA).
//   EffectiveSEBlock  modules/se_blocks.py:29-42           y = x * hardsigmoid(project(mean_hw x))
//   ESEAttn           pp_yolo_e/pp_yolo_head.py:79-93      conv(feat * sigmoid(fc(avg_feat)))
//   F.interpolate(x2, nearest)  pp_yolo_e/pan.py:170
// Data layout: NHWC views with explicit pixel / image strides (channel slices of concat buffers are read and written in place);
// per-image vectors ([N][C] means, gate pre-activations, their gradients) are contiguous.
// The per-image reductions are two-stage and deterministic: a (image, pixel chunk, channel strip) grid leaves fp32 partial rows,
// a finalize kernel adds them in fp64 in chunk order.  No atomics.
#include "sgx_common.h"

#define SE_THREADS 256
#define SE_MAXCG 64      // float4 channel groups per workgroup strip
#define SE_CHUNK_ROWS 512  // pixels per chunk: 160x160 maps -> 50 chunks per image, 32 images -> 1600 workgroups per strip

__device__ __forceinline__ float se_gate(float p, int gate) {
    if (gate == SGX_GATE_HARDSIGMOID) return fminf(fmaxf(p * (1.f / 6.f) + 0.5f, 0.f), 1.f);
    if (gate == SGX_GATE_SIGMOID) return 1.f / (1.f + expf(-p));
    return p;
}
__device__ __forceinline__ float se_gate_grad(float p, int gate) {
    if (gate == SGX_GATE_HARDSIGMOID) return (p > -3.f && p < 3.f) ? (1.f / 6.f) : 0.f;
    if (gate == SGX_GATE_SIGMOID) {
        float s = 1.f / (1.f + expf(-p));
        return s * (1.f - s);
    }
    return 1.f;
}

struct SeGeom {
    int N, HW, C, C4, CG, RL, chunks, ctiles;
};
static SeGeom se_geom(int N, int HW, int C) {
    SeGeom g;
    g.N = N; g.HW = HW; g.C = C; g.C4 = C / 4;
    g.CG = g.C4 < SE_MAXCG ? g.C4 : SE_MAXCG;
    g.RL = SE_THREADS / g.CG;
    g.chunks = (HW + SE_CHUNK_ROWS - 1) / SE_CHUNK_ROWS;
    g.ctiles = (g.C4 + g.CG - 1) / g.CG;
    return g;
}

// partials [N][chunks][C]
__global__ __launch_bounds__(SE_THREADS) void image_colsum_kernel(SeGeom g, const float* u, long u_ld_pix, long u_ld_img, const float* v,
                                                                  long v_ld_pix, long v_ld_img, float* partials) {
    __shared__ float4 red[SE_THREADS];
    const int tid = threadIdx.x;
    const int cg = tid % g.CG, rl = tid / g.CG;
    const int c4 = blockIdx.y * g.CG + cg;
    const int chunk = blockIdx.x % g.chunks, img = blockIdx.x / g.chunks;
    const bool live = (rl < g.RL) && (c4 < g.C4);
    const int c = c4 * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        int p0 = chunk * SE_CHUNK_ROWS, p1 = p0 + SE_CHUNK_ROWS;
        if (p1 > g.HW) p1 = g.HW;
        const float* ub = u + (long)img * u_ld_img + c;
        const float* vb = v ? v + (long)img * v_ld_img + c : nullptr;
        for (int p = p0 + rl; p < p1; p += g.RL) {
            float4 a = sgx_ld4(ub + (long)p * u_ld_pix);
            if (vb) {
                float4 b = sgx_ld4(vb + (long)p * v_ld_pix);
                q.x += a.x * b.x; q.y += a.y * b.y; q.z += a.z * b.z; q.w += a.w * b.w;
            } else {
                q.x += a.x; q.y += a.y; q.z += a.z; q.w += a.w;
            }
        }
    }
    red[tid] = q;
    __syncthreads();
    if (live && rl == 0) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < g.RL; ++k) {
            float4 a = red[k * g.CG + cg];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        sgx_st4(partials + ((long)img * g.chunks + chunk) * g.C + c, s);
    }
}
__global__ void image_colsum_finalize_kernel(int N, int chunks, int C, const float* partials, float scale, const float* pre, int gate,
                                             float* out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * C) return;
    int c = (int)(i % C), img = (int)(i / C);
    double s = 0.0;
    for (int k = 0; k < chunks; ++k) s += (double)partials[((long)img * chunks + k) * C + c];
    float r = (float)(s * (double)scale);
    if (pre) r *= se_gate_grad(pre[i], gate);
    out[i] = r;
}
extern "C" int64_t sgx_image_colsum_workspace(int32_t N, int32_t HW, int32_t C) {
    long chunks = ((long)HW + SE_CHUNK_ROWS - 1) / SE_CHUNK_ROWS;
    return (int64_t)N * chunks * C * (int64_t)sizeof(float) + 256;
}
extern "C" int32_t sgx_image_colsum(int32_t N, int32_t HW, int32_t C, const float* u, int64_t u_ld_pix, int64_t u_ld_img, const float* v,
                                    int64_t v_ld_pix, int64_t v_ld_img, float scale, const float* pre, int32_t gate, float* out, void* ws,
                                    int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(u && out && ws && N > 0 && HW > 0 && C > 0 && C % 4 == 0, "image_colsum: bad args (C=%d)", C);
    SGX_CHECK_ARG(ws_bytes >= sgx_image_colsum_workspace(N, HW, C), "image_colsum: workspace too small");
    SeGeom g = se_geom(N, HW, C);
    SGX_LAUNCH(image_colsum_kernel, dim3((unsigned)((long)N * g.chunks), g.ctiles), dim3(SE_THREADS), 0, stream, g, u, (long)u_ld_pix,
               (long)u_ld_img, v, (long)v_ld_pix, (long)v_ld_img, (float*)ws);
    SGX_CHECK_LAUNCH("image_colsum");
    SGX_LAUNCH(image_colsum_finalize_kernel, dim3(sgx_cdiv((long)N * C, 256)), dim3(256), 0, stream, N, g.chunks, C, (const float*)ws, scale,
               pre, gate, out);
    SGX_CHECK_LAUNCH("image_colsum_finalize");
    return SGX_OK;
}

// One thread: one float4 channel group of one pixel; the gate / bias of (image, channel group) come from L2-resident [N][C] vectors.
__global__ __launch_bounds__(256) void channel_gate_kernel(int N, int HW, int C, const float* x, long x_ld_pix, long x_ld_img, const float* pre,
                                                           int gate, const float* bias, float bias_scale, float* y, long y_ld_pix,
                                                           long y_ld_img, int accumulate) {
    const int C4 = C / 4;
    const long n = (long)N * HW * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long t = i / C4;
        int p = (int)(t % HW), img = (int)(t / HW);
        float4 v = sgx_ld4(x + (long)img * x_ld_img + (long)p * x_ld_pix + c);
        float4 w = sgx_ld4(pre + (long)img * C + c);
        float4 o = make_float4(v.x * se_gate(w.x, gate), v.y * se_gate(w.y, gate), v.z * se_gate(w.z, gate), v.w * se_gate(w.w, gate));
        if (bias) {
            float4 b = sgx_ld4(bias + (long)img * C + c);
            o.x += bias_scale * b.x; o.y += bias_scale * b.y; o.z += bias_scale * b.z; o.w += bias_scale * b.w;
        }
        float* yp = y + (long)img * y_ld_img + (long)p * y_ld_pix + c;
        if (accumulate) {
            float4 a = sgx_ld4(yp);
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        sgx_st4(yp, o);
    }
}
extern "C" int32_t sgx_channel_gate(int32_t N, int32_t HW, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img, const float* pre,
                                    int32_t gate, const float* bias, float bias_scale, float* y, int64_t y_ld_pix, int64_t y_ld_img,
                                    int32_t accumulate, void* stream) {
    SGX_CHECK_ARG(x && pre && y && N > 0 && HW > 0 && C > 0 && C % 4 == 0, "channel_gate: bad args (C=%d)", C);
    SGX_CHECK_ARG(gate >= SGX_GATE_NONE && gate <= SGX_GATE_SIGMOID, "channel_gate: unknown gate %d", gate);
    long n = (long)N * HW * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(channel_gate_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, HW, C, x, (long)x_ld_pix,
               (long)x_ld_img, pre, gate, bias, bias_scale, y, (long)y_ld_pix, (long)y_ld_img, accumulate);
    SGX_CHECK_LAUNCH("channel_gate");
    return SGX_OK;
}

// One thread: one float4 channel group of one INPUT pixel (read once, written to its 2x2 output block / gathered from it).
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(int N, int H, int W, int C, const float* x, long x_ld_pix, long x_ld_img, float* y,
                                                             long y_ld_pix, long y_ld_img) {
    const int C4 = C / 4;
    const long n = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long t = i / C4;
        int w = (int)(t % W);
        t /= W;
        int h = (int)(t % H), img = (int)(t / H);
        float4 v = sgx_ld4(x + (long)img * x_ld_img + ((long)h * W + w) * x_ld_pix + c);
        float* yb = y + (long)img * y_ld_img + c;
        const long W2 = 2L * W;
        sgx_st4(yb + ((2L * h) * W2 + 2 * w) * y_ld_pix, v);
        sgx_st4(yb + ((2L * h) * W2 + 2 * w + 1) * y_ld_pix, v);
        sgx_st4(yb + ((2L * h + 1) * W2 + 2 * w) * y_ld_pix, v);
        sgx_st4(yb + ((2L * h + 1) * W2 + 2 * w + 1) * y_ld_pix, v);
    }
}
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(int N, int H, int W, int C, const float* dy, long dy_ld_pix, long dy_ld_img,
                                                             float* dx, long dx_ld_pix, long dx_ld_img, int accumulate) {
    const int C4 = C / 4;
    const long n = (long)N * H * W * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % C4) * 4;
        long t = i / C4;
        int w = (int)(t % W);
        t /= W;
        int h = (int)(t % H), img = (int)(t / H);
        const float* db = dy + (long)img * dy_ld_img + c;
        const long W2 = 2L * W;
        float4 a = sgx_ld4(db + ((2L * h) * W2 + 2 * w) * dy_ld_pix), b = sgx_ld4(db + ((2L * h) * W2 + 2 * w + 1) * dy_ld_pix);
        float4 e = sgx_ld4(db + ((2L * h + 1) * W2 + 2 * w) * dy_ld_pix), f = sgx_ld4(db + ((2L * h + 1) * W2 + 2 * w + 1) * dy_ld_pix);
        float4 o = make_float4((a.x + b.x) + (e.x + f.x), (a.y + b.y) + (e.y + f.y), (a.z + b.z) + (e.z + f.z), (a.w + b.w) + (e.w + f.w));
        float* xp = dx + (long)img * dx_ld_img + ((long)h * W + w) * dx_ld_pix + c;
        if (accumulate) {
            float4 u = sgx_ld4(xp);
            o.x += u.x; o.y += u.y; o.z += u.z; o.w += u.w;
        }
        sgx_st4(xp, o);
    }
}
extern "C" int32_t sgx_upsample2x_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img, float* y,
                                      int64_t y_ld_pix, int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_fwd: bad args (C=%d)", C);
    long n = (long)N * H * W * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(upsample2x_fwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, x, (long)x_ld_pix,
               (long)x_ld_img, y, (long)y_ld_pix, (long)y_ld_img);
    SGX_CHECK_LAUNCH("upsample2x_fwd");
    return SGX_OK;
}
extern "C" int32_t sgx_upsample2x_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img,
                                      float* dx, int64_t dx_ld_pix, int64_t dx_ld_img, int32_t accumulate, void* stream) {
    SGX_CHECK_ARG(dy && dx && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "upsample2x_bwd: bad args (C=%d)", C);
    long n = (long)N * H * W * (C / 4), blocks = (n + 255) / 256;
    SGX_LAUNCH(upsample2x_bwd_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, dy, (long)dy_ld_pix,
               (long)dy_ld_img, dx, (long)dx_ld_pix, (long)dx_ld_img, accumulate);
    SGX_CHECK_LAUNCH("upsample2x_bwd");
    return SGX_OK;
}
