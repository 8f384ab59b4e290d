// Inference-side image preparation on the device: one launch turns a batch of ragged uint8 HWC images into the standardized fp32 NHWC
// batch the first convolution reads.  Per image it fuses what the reference's predict() pipeline does on the host, one numpy / cv2 pass
// each (training/processing/processing.py): ReverseImageChannels :230-257, Detection[LongestMaxSize]Rescale :510-589 (cv2.resize,
// INTER_LINEAR, transforms/utils.py:17-25), Detection{Center,BottomRight,Auto}Padding :326-471 (np.pad of the uint8 image,
// transforms/utils.py:109-158), StandardizeImage :260-295 ((image / max_value).astype(float32): the division is float64), NormalizeImage
// :298-323 ((image - mean) / std in float32), ImagePermute :205-227 (the NHWC layout is the kernels' own, the permutation is a view).
//
// Rescale arithmetic: cv2 itself is not vendored by the reference (requirements.txt: opencv-python>=4.5.1) and is not installed here, so the
// bilinear path restates OpenCV's published 8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp, 4.x): coordinates
// (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dsize / ssize) in double, 11-bit fixed-point coefficients rounded half-to-even,
// an integer horizontal pass, and the vertical pass (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2x2
// reduction takes cv::resize's INTER_AREA shortcut ((a + b + c + d + 2) >> 2).  oracle/image.py restates the same; neither can be pinned
// against cv2 output in this container (stated in DESIGN.md: "rescale parity unpinned"); every other stage is pinned against the
// reference's own processing classes.
#include "sgx_common.h"

__device__ __forceinline__ int sgx_clip_i(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ int sgx_sat_short(float v) { return sgx_clip_i((int)rintf(v), -32768, 32767); }

// source tap and the two 11-bit weights of destination index d along one axis
__device__ __forceinline__ void sgx_linear_taps(int d, int ssize, double scale, bool clamp_weights, int& s, int& w0, int& w1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    s = (int)floorf(f);
    f -= (float)s;
    if (clamp_weights) {  // the horizontal pass zeroes the fraction at the borders; the vertical pass clips the rows instead
        if (s < 0) { f = 0.f; s = 0; }
        if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    }
    w0 = sgx_sat_short((1.f - f) * 2048.f);
    w1 = sgx_sat_short(f * 2048.f);
}

__global__ void preprocess_u8_kernel(const sgx_image_job* jobs, int C, int Cpad, int H, int W, int reverse, int standardize, double max_value,
                                     const float* mean, const float* stdv, const uint8_t* pad_value, float* y) {
    const sgx_image_job j = jobs[blockIdx.y];
    float* yi = y + (long)blockIdx.y * H * W * Cpad;
    const long npix = (long)H * W;
    const bool resize = j.h != j.h0 || j.w != j.w0;
    const double scale_x = 1.0 / ((double)j.w / (double)j.w0), scale_y = 1.0 / ((double)j.h / (double)j.h0);
    const bool area2 = resize && j.w0 == 2 * j.w && j.h0 == 2 * j.h;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        const int dy = yy - j.top, dx = xx - j.left;
        const bool inside = dy >= 0 && dy < j.h && dx >= 0 && dx < j.w;
        int px[4] = {0, 0, 0, 0};
        if (!inside) {
            for (int c = 0; c < C && c < 4; ++c) px[c] = pad_value[c];
        } else if (!resize) {
            const uint8_t* s = j.src + ((long)dy * j.w0 + dx) * C;
            for (int c = 0; c < C && c < 4; ++c) px[c] = s[reverse ? C - 1 - c : c];
        } else if (area2) {
            const uint8_t* s0 = j.src + ((long)(2 * dy) * j.w0 + 2 * dx) * C;
            const uint8_t* s1 = s0 + (long)j.w0 * C;
            for (int c = 0; c < C && c < 4; ++c) {
                const int cs = reverse ? C - 1 - c : c;
                px[c] = (s0[cs] + s0[cs + C] + s1[cs] + s1[cs + C] + 2) >> 2;
            }
        } else {
            int sx, a0, a1, sy, b0, b1;
            sgx_linear_taps(dx, j.w0, scale_x, true, sx, a0, a1);
            sgx_linear_taps(dy, j.h0, scale_y, false, sy, b0, b1);
            const int sx1 = sx + 1 < j.w0 ? sx + 1 : j.w0 - 1;
            const uint8_t* r0 = j.src + (long)sgx_clip_i(sy, 0, j.h0 - 1) * j.w0 * C;
            const uint8_t* r1 = j.src + (long)sgx_clip_i(sy + 1, 0, j.h0 - 1) * j.w0 * C;
            for (int c = 0; c < C && c < 4; ++c) {
                const int cs = reverse ? C - 1 - c : c;
                const int S0 = r0[(long)sx * C + cs] * a0 + r0[(long)sx1 * C + cs] * a1;
                const int S1 = r1[(long)sx * C + cs] * a0 + r1[(long)sx1 * C + cs] * a1;
                px[c] = sgx_clip_i((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2, 0, 255);
            }
        }
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float f = 0.f;
            if (c < C) {
                f = standardize ? (float)((double)px[c] / max_value) : (float)px[c];
                if (mean) f = (f - mean[c]) / stdv[c];
            }
            v[c] = f;
        }
        sgx_st4(yi + i * Cpad, make_float4(v[0], v[1], v[2], v[3]));
        for (int c0 = 4; c0 < Cpad; c0 += 4) sgx_st4(yi + i * Cpad + c0, make_float4(0.f, 0.f, 0.f, 0.f));
    }
}

extern "C" int32_t sgx_preprocess_u8_hwc(const sgx_image_job* jobs_dev, int32_t N, int32_t C, int32_t Cpad, int32_t H, int32_t W, int32_t reverse_channels,
                                         int32_t standardize, double max_value, const float* mean, const float* stdv, const uint8_t* pad_value,
                                         float* y, void* stream) {
    SGX_CHECK_ARG(jobs_dev && y && pad_value && N > 0 && N <= 65535 && H > 0 && W > 0, "preprocess_u8: bad args (N=%d H=%d W=%d)", N, H, W);
    SGX_CHECK_ARG(C >= 1 && C <= 4 && Cpad >= C && Cpad % 4 == 0, "preprocess_u8: 1..4 image channels, Cpad a multiple of 4 (C=%d Cpad=%d)", C, Cpad);
    SGX_CHECK_ARG((!standardize || max_value > 0.0) && ((mean == nullptr) == (stdv == nullptr)), "preprocess_u8: max_value > 0, mean and std go together");
    const long npix = (long)H * W, blocks = (npix + 255) / 256;
    SGX_LAUNCH(preprocess_u8_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks), (unsigned)N), dim3(256), 0, stream, jobs_dev, C, Cpad, H, W,
               reverse_channels, standardize, max_value, mean, stdv, pad_value, y);
    SGX_CHECK_LAUNCH("preprocess_u8");
    return SGX_OK;
}
