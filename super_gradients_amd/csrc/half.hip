// Half-precision INFERENCE path for gfx950: the deployment form of the detectors (every conv + BatchNorm pair folded into one
// convolution with bias + activation, QARepVGG blocks fully fused - reference modules/qarepvgg_block.py:255-321) executed on the bf16
// matrix pipe with single-plane bf16 operands and fp32 accumulation.  This is what `predict(fp16=True)` runs - the reference wraps its
// forward in torch.autocast there (training/pipelines/pipelines.py:76,223,375); nothing on the TRAINING path uses these kernels (training
// keeps fp32 storage and fp32-accurate bf16x3 products, conv.hip).
//
// HBM layout: activations NHWC **bf16** with explicit (ld_pix, ld_img) element strides (concat slices are written in place, as on the fp32
// path), channel counts multiples of 8 (one 16-byte lane load = 8 channels; the RGB input is padded 3 -> 8 once by sgx_cast_f32_bf16);
// filters OHWI bf16, converted once per fused model; bias fp32.  Half the bytes of the fp32 path per activation.
//
// hconv_kernel = the gather GEMM of conv.hip's igemm_kernel re-cut for this arithmetic: per-lane 31-bit offsets + a 64-bit tap-validity
// mask (zero padding = the buffer bounds check), scalar tap / chunk counters, XCD-aware tile order.  Differences that matter:
//   * staging is a pure copy: global (16 B = 8 bf16) -> register -> ONE ds_write_b128, no split, no conversion;
//   * K slabs of 32 or 64 bf16 (64 / 128-byte rows), two LDS buffers, ONE barrier per slab; rows are unpadded and their 16-byte chunks
//     XOR-swizzled by row bits so that the fragment reads (ds_read_b128, 16 rows per bank pass) are conflict-free;
//   * one v_mfma_f32_32x32x16_bf16 per (32x32 block, 16-deep step) where the training path issues six;
//   * FLAT variant for the 8-channel (padded RGB) stem: the K axis is the flattened (tap, 8 channels) axis, one tap per lane chunk;
//   * epilogue: bias + activation (+ post_scale * post_add AFTER the activation: the YOLO-NAS bottleneck's shortcut, yolo_stages.py:61-63),
//     stored as bf16 - or as fp32 for the prediction convs, whose outputs feed the fp32 decode / NMS kernels unchanged.
#include "sgx_common.h"
#include "conv_mma.h"

struct HconvParams {
    const unsigned short* A;
    const unsigned short* Wt;
    const float* bias;
    void* Y;
    const unsigned short* post;
    const float* post_scale_dev;
    float post_scale;
    int y_f32;
    int M, Ha, Wa, Hin, Win, C, Nout;
    int Th, Tw, dh0, dw0;
    int si, so, ph, pw, Hout, Wout;
    long a_ld_pix, a_ld_img, y_ld_pix, y_ld_img, p_ld_pix, p_ld_img, w_ld_n;
    long a_bytes, w_bytes;
    int act, vec;
    int mt, nt, nblk, chunk;
};

__device__ __forceinline__ float sgx_bf16_to_f32(unsigned short h) { return sgx_u2f((unsigned)h << 16); }

template <int BM, int BN, int WM, int WN, int KD, bool FLAT>
__global__ __launch_bounds__(WM * WN * 64) void hconv_kernel(HconvParams p) {
    static_assert(KD == 32 || KD == 64, "slab depth: 32 or 64 bf16");
    constexpr int NTH = WM * WN * 64;
    constexpr int CPR = KD / 8;      // lanes per slab row (16 B = 8 bf16 each)
    constexpr int RPP = NTH / CPR;   // slab rows staged per pass
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AJ = (BM + RPP - 1) / RPP, BJ = (BN + RPP - 1) / RPP;
    constexpr int ROWB = KD * 2;     // bytes per LDS row
    static_assert(TM >= 1 && TN >= 1 && TM * WM * 32 == BM && TN * WN * 32 == BN, "bad tile");
    static_assert(NTH >= BM && NTH >= BN, "one thread per tile row");
    constexpr int SLABS = 2 * (BM + BN) * ROWB, STAGE = WM * WN * 32 * 32 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SLABS > STAGE ? SLABS : STAGE];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + 2 * BM * ROWB;
    __shared__ long long rowoff[BM];
    __shared__ long long rowoffP[BM];
    // 16-byte chunk c of row r lives at chunk c ^ sw(r): 16 rows of one ds_read_b128 bank pass then cover all 64 banks once
    auto sw = [](int row) { return KD == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int bid = blockIdx.x;
    const int lin = (bid & 7) * p.chunk + (bid >> 3);  // block b runs on XCD b % 8: every XCD gets a contiguous run of tiles
    if (lin >= p.nblk) return;
    const int mtile = lin / p.nt, ntile = lin - mtile * p.nt;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int hw = p.Ha * p.Wa;
    const int T = p.Th * p.Tw;
    const int img0 = m0 / hw;
    const int lrow = tid / CPR, cq = tid % CPR;

    const sgx_buf bufA = sgx_make_buf(p.A + (long)img0 * p.a_ld_img, p.a_bytes - (long)img0 * p.a_ld_img * 2);
    const sgx_buf bufB = sgx_make_buf(p.Wt, p.w_bytes);
    int aoff[AJ], boff[BJ];
    unsigned long long amask[AJ];
    bool bok[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int row = lrow + RPP * j;
        const int m = m0 + row;
        aoff[j] = 0;
        amask[j] = 0ull;
        if (row < BM && m < p.M) {
            const int img = m / hw;
            const int rem = m - img * hw;
            const int a = rem / p.Wa;
            const int b = rem - a * p.Wa;
            const int hi0 = a * p.si + p.dh0, wi0 = b * p.si + p.dw0;
            aoff[j] = (int)(((long)(img - img0) * p.a_ld_img + ((long)hi0 * p.Win + wi0) * p.a_ld_pix + (FLAT ? 0 : cq * 8)) * 2);
            unsigned long long mk = 0ull;
            for (int i = 0; i < p.Th; ++i)
                for (int jj = 0; jj < p.Tw; ++jj)
                    if (hi0 + i >= 0 && hi0 + i < p.Hin && wi0 + jj >= 0 && wi0 + jj < p.Win) mk |= 1ull << (i * p.Tw + jj);
            amask[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int row = lrow + RPP * j;
        const int n = n0 + row;
        bok[j] = (row < BN) && (n < p.Nout);
        boff[j] = (int)(((long)n * p.w_ld_n + (FLAT ? 0 : cq * 8)) * 2);
    }
    const int cpt = (p.C + KD - 1) / KD;
    const int nkt = FLAT ? (T + CPR - 1) / CPR : T * cpt;
    const int pixstep = (int)p.a_ld_pix * 2, rowstep = pixstep * p.Win;
    int s_ti = 0, s_tj = 0, s_ck = 0, s_kt = 0;

    if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1, offp = 0;
        if (m < p.M) {
            const int img = m / hw;
            const int rem = m - img * hw;
            const int a = rem / p.Wa;
            const int b = rem - a * p.Wa;
            const long long pix = (long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw);
            off = (long long)img * p.y_ld_img + pix * p.y_ld_pix;
            offp = (long long)img * p.p_ld_img + pix * p.p_ld_pix;
        }
        rowoff[tid] = off;
        rowoffP[tid] = offp;
    }

    uint4 ra[AJ], rb[BJ];
    auto load_tile = [&]() {
        if (FLAT) {
            const int t = s_kt * CPR + cq;  // this lane's tap (8 channels = one 16-byte chunk)
            const bool kok = t < T;
            const int ti = t / p.Tw, tj = t - ti * p.Tw;
            const int tapoff = ti * rowstep + tj * pixstep;
            const int tb = kok ? t : 0;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool ok = kok && ((amask[j] >> tb) & 1ull);
                ra[j] = sgx_buf_ld4u(bufA, ok ? (unsigned)(aoff[j] + tapoff) : SGX_BUF_OOB);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) rb[j] = sgx_buf_ld4u(bufB, (kok && bok[j]) ? (unsigned)(boff[j] + t * 16) : SGX_BUF_OOB);
        } else {
            const int tbit = s_ti * p.Tw + s_tj;
            const int tapoff = s_ti * rowstep + s_tj * pixstep + s_ck * (KD * 2);
            const int woff = (tbit * p.C + s_ck * KD) * 2;
            const bool cok = s_ck * KD + cq * 8 < p.C;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool ok = cok && ((amask[j] >> tbit) & 1ull);
                ra[j] = sgx_buf_ld4u(bufA, ok ? (unsigned)(aoff[j] + tapoff) : SGX_BUF_OOB);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) rb[j] = sgx_buf_ld4u(bufB, (cok && bok[j]) ? (unsigned)(boff[j] + woff) : SGX_BUF_OOB);
            if (++s_ck == cpt) {
                s_ck = 0;
                if (++s_tj == p.Tw) {
                    s_tj = 0;
                    ++s_ti;
                }
            }
        }
        ++s_kt;
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = lrow + RPP * j;
            if (BM % RPP == 0 || row < BM) *reinterpret_cast<uint4*>(As + (buf * BM + row) * ROWB + ((cq ^ sw(row)) * 16)) = ra[j];
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = lrow + RPP * j;
            if (BN % RPP == 0 || row < BN) *reinterpret_cast<uint4*>(Bs + (buf * BN + row) * ROWB + ((cq ^ sw(row)) * 16)) = rb[j];
        }
    };

    sgx_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frow = lane & 31, khalf = lane >> 5;
    auto compute = [&](int buf) {
#pragma unroll
        for (int s = 0; s < KD / 16; ++s) {
            uint4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * TM * 32 + i * 32 + frow;
                af[i] = *reinterpret_cast<const uint4*>(As + (buf * BM + row) * ROWB + (((2 * s + khalf) ^ sw(row)) * 16));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * TN * 32 + j * 32 + frow;
                bf[j] = *reinterpret_cast<const uint4*>(Bs + (buf * BN + row) * ROWB + (((2 * s + khalf) ^ sw(row)) * 16));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = sgx_mfma_bf16(af[i], bf[j], acc[i][j]);
        }
    };

    if (nkt > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile();  // the next slab travels in registers under this slab's MFMAs
        compute(buf);
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: every wave transposes its 32x32 accumulators through a private 4 KB LDS patch (the slabs are free now): a lane then
    // owns 4 consecutive columns of one row - 16-byte fp32 / 8-byte bf16 stores, the post-activation addend read the same way
    float* const stage = reinterpret_cast<float*>(smem) + wave * (32 * 32);
    const int sr = lane >> 3, sc4 = (lane & 7) * 4;
    const float psc = p.post ? p.post_scale * (p.post_scale_dev ? p.post_scale_dev[0] : 1.f) : 0.f;
    unsigned short* const Yh = reinterpret_cast<unsigned short*>(p.Y);
    float* const Yf = reinterpret_cast<float*>(p.Y);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + sc4;
        const bool colok = col < p.Nout;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && colok) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (col + t < p.Nout) bv[t] = p.bias[col + t];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rowl = q * 8 + sr;
                const int trow = wm * TM * 32 + i * 32 + rowl;
                const long long off = rowoff[trow];
                const float4 v4 = sgx_ld4(stage + rowl * 32 + sc4);
                if (off >= 0 && colok) {
                    float v[4] = {v4.x + bv[0], v4.y + bv[1], v4.z + bv[2], v4.w + bv[3]};
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = sgx_act(v[t], p.act);
                    if (p.post) {
                        const unsigned short* const pp = p.post + rowoffP[trow] + col;
                        if (p.vec) {
                            const uint2 u = *reinterpret_cast<const uint2*>(pp);
                            v[0] += psc * sgx_u2f(u.x << 16); v[1] += psc * sgx_u2f(u.x & 0xffff0000u);
                            v[2] += psc * sgx_u2f(u.y << 16); v[3] += psc * sgx_u2f(u.y & 0xffff0000u);
                        } else {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (col + t < p.Nout) v[t] += psc * sgx_bf16_to_f32(pp[t]);
                        }
                    }
                    if (p.vec) {
                        if (p.y_f32) sgx_st4(Yf + off + col, make_float4(v[0], v[1], v[2], v[3]));
                        else *reinterpret_cast<uint2*>(Yh + off + col) = make_uint2(sgx_pack_bf16(v[0], v[1]), sgx_pack_bf16(v[2], v[3]));
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (col + t < p.Nout) {
                                if (p.y_f32) Yf[off + col + t] = v[t];
                                else Yh[off + col + t] = (unsigned short)(sgx_pack_bf16(v[t], 0.f) & 0xffffu);
                            }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// ---- tile choice -------------------------------------------------------------------------------------------------------------------------
#include <atomic>
static std::atomic<int> g_h_bm{0}, g_h_bn{0}, g_h_kd{0};
extern "C" int32_t sgx_hconv_debug_set_tile(int32_t bm, int32_t bn, int32_t kd) {  // measurement aid (tools/predict_bench.py): 0 = heuristic
    SGX_CHECK_ARG((bm == 0 || bm == 64 || bm == 128) && (bn == 0 || bn == 32 || bn == 64 || bn == 96 || bn == 128) && (kd == 0 || kd == 32 || kd == 64),
                  "hconv_debug_set_tile: tile %dx%d, slab depth %d", bm, bn, kd);
    g_h_bm = bm; g_h_bn = bn; g_h_kd = kd;
    return SGX_OK;
}
template <int BM, int BN, int WM, int WN, int KD, bool FLAT>
static void launch_hconv(HconvParams& p, void* stream) {
    p.mt = sgx_cdiv(p.M, BM);
    p.nt = sgx_cdiv(p.Nout, BN);
    p.nblk = p.mt * p.nt;
    p.chunk = sgx_cdiv(p.nblk, 8);
    SGX_LAUNCH((hconv_kernel<BM, BN, WM, WN, KD, FLAT>), dim3(p.chunk * 8), dim3(WM * WN * 64), 0, stream, p);
}
template <int KD>
static int32_t launch_hconv_tile(HconvParams& p, int bm, int bn, void* stream) {
    if (bm == 128 && bn == 128) launch_hconv<128, 128, 2, 2, KD, false>(p, stream);
    else if (bm == 128 && bn == 64) launch_hconv<128, 64, 2, 2, KD, false>(p, stream);
    else if (bm == 64 && bn == 128) launch_hconv<64, 128, 2, 2, KD, false>(p, stream);
    else if (bm == 64 && bn == 64) launch_hconv<64, 64, 2, 2, KD, false>(p, stream);
    else if (bm == 128 && bn == 96) launch_hconv<128, 96, 4, 1, KD, false>(p, stream);
    else if (bm == 64 && bn == 96) launch_hconv<64, 96, 2, 1, KD, false>(p, stream);
    else if (bm == 128 && bn == 32) launch_hconv<128, 32, 4, 1, KD, false>(p, stream);
    else if (bm == 64 && bn == 32) launch_hconv<64, 32, 2, 1, KD, false>(p, stream);
    else SGX_FAIL(SGX_ERR_UNSUPPORTED, "hconv: no tile %dx%d", bm, bn);
    return SGX_OK;
}
static int32_t launch_hconv_flat(HconvParams& p, int bm, int bn, void* stream) {  // the 8-channel stem: few output channels, narrow tiles only
    if (bm == 128 && bn == 64) launch_hconv<128, 64, 2, 2, 32, true>(p, stream);
    else if (bm == 64 && bn == 64) launch_hconv<64, 64, 2, 2, 32, true>(p, stream);
    else if (bm == 128 && bn == 32) launch_hconv<128, 32, 4, 1, 32, true>(p, stream);
    else if (bm == 64 && bn == 32) launch_hconv<64, 32, 2, 1, 32, true>(p, stream);
    else SGX_FAIL(SGX_ERR_UNSUPPORTED, "hconv (flat): no tile %dx%d", bm, bn);
    return SGX_OK;
}
static int32_t run_hconv(HconvParams& p, void* stream) {
    SGX_CHECK_ARG(p.C % 8 == 0 && p.C >= 8, "hconv: C=%d must be a multiple of 8 bf16 (16-byte lane loads; pad the input)", p.C);
    SGX_CHECK_ARG(p.a_ld_pix % 8 == 0 && p.a_ld_img % 8 == 0 && ((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.Wt % 16) == 0,
                  "hconv: the input's strides and the operand addresses must be multiples of 16 bytes");
    const int T = p.Th * p.Tw;
    if (T > 64) SGX_FAIL(SGX_ERR_UNSUPPORTED, "hconv: more than 64 taps");
    if (p.w_bytes > SGX_BUF_MAX) SGX_FAIL(SGX_ERR_UNSUPPORTED, "hconv: weight tensor larger than 2 GiB");
    const int yal = p.y_f32 ? 16 : 8;
    p.vec = (p.Nout % 4 == 0 && p.y_ld_pix % 4 == 0 && p.y_ld_img % 4 == 0 && ((uintptr_t)p.Y % yal) == 0 &&
             (!p.post || (p.p_ld_pix % 4 == 0 && p.p_ld_img % 4 == 0 && ((uintptr_t)p.post % 8) == 0)))
                ? 1
                : 0;
    // Tile and slab depth, fitted to the replay of every convolution of the fused YOLO-NAS-S / M forward under every instantiation
    // (tools/hconv_lab.py, profiles/r5b_hconv_lab.txt): the loop moves bytes, not FLOPs (one MFMA per 16-deep step and block), so what a
    // tile buys is fewer re-reads of the other operand - worth it only while enough workgroups remain to fill 256 CUs.
    const long depth = (long)T * p.C;
    const long mt128 = sgx_cdiv(p.M, 128);
    int bm = 64, bn = 64;
    if (p.Nout <= 32) bm = 128, bn = 32;
    else if (p.Nout % 64 != 0 && p.Nout % 96 == 0) {  // 96, 288: three 32-wide blocks per wave, nothing padded
        if (p.M >= 100000) bm = 128, bn = 96;
        else if (T > 1) bm = 64, bn = 96;
    } else if (p.Nout % 128 == 0 && depth >= 1024 && mt128 * (p.Nout / 128) >= 256) bm = 128, bn = 128;  // deep reductions: operand reuse pays
    else if (mt128 * sgx_cdiv(p.Nout, 64) >= 2048 || (depth >= 512 && mt128 * sgx_cdiv(p.Nout, 64) >= 512)) bm = 128, bn = 64;
    if (const int o = g_h_bm.load(std::memory_order_relaxed)) bm = o;
    if (const int o = g_h_bn.load(std::memory_order_relaxed)) bn = o;
    {
        const long hw = (long)p.Ha * p.Wa;
        const long imgs = (bm + hw - 1) / hw + 1;  // images a pixel tile can touch
        if (imgs * p.a_ld_img * 2 > SGX_BUF_MAX) SGX_FAIL(SGX_ERR_UNSUPPORTED, "hconv: one pixel tile spans more than 2 GiB of input");
    }
    const bool flat = p.C == 8 && T > 1;
    // 64-deep slabs (whole 128-byte lines per row) pay on the small maps with many channels only; elsewhere their LDS costs occupancy
    int kd = (!flat && p.C % 64 == 0 && p.C >= 192 && p.M <= 51200 && !(bm == 128 && bn == 128)) ? 64 : 32;
    if (const int o = g_h_kd.load(std::memory_order_relaxed)) kd = flat ? 32 : o;
    int32_t rc;
    if (flat) rc = launch_hconv_flat(p, bm, bn > 64 ? 64 : bn, stream);
    else if (kd == 64) rc = launch_hconv_tile<64>(p, bm, bn, stream);
    else rc = launch_hconv_tile<32>(p, bm, bn, stream);
    if (rc) return rc;
    SGX_CHECK_LAUNCH("hconv");
    return SGX_OK;
}

extern "C" int32_t sgx_hconv2d_fwd(const sgx_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int32_t y_is_f32, int32_t act,
                                   const void* post_add, int64_t post_ld_pix, int64_t post_ld_img, float post_scale, const float* post_scale_dev,
                                   void* stream) {
    SGX_CHECK_ARG(d && x && w && y, "hconv2d_fwd: null pointer");
    SGX_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0 && d->R > 0 && d->S > 0 && d->stride >= 1 && d->pad >= 0, "hconv2d_fwd: bad dims");
    SGX_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->S) / d->stride + 1, "hconv2d_fwd: Ho/Wo do not match (H+2p-R)/s+1");
    SGX_CHECK_ARG(d->x_ld_pix >= d->C && d->y_ld_pix >= d->K, "hconv2d_fwd: bad pixel strides");
    HconvParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const unsigned short*)x; p.Wt = (const unsigned short*)w; p.bias = bias; p.Y = y; p.y_f32 = y_is_f32 ? 1 : 0;
    p.post = (const unsigned short*)post_add; p.post_scale = post_scale; p.post_scale_dev = post_scale_dev;
    p.p_ld_pix = post_ld_pix; p.p_ld_img = post_ld_img;
    p.M = d->N * d->Ho * d->Wo; p.Ha = d->Ho; p.Wa = d->Wo; p.Hin = d->H; p.Win = d->W;
    p.C = d->C; p.Nout = d->K; p.Th = d->R; p.Tw = d->S; p.dh0 = -d->pad; p.dw0 = -d->pad;
    p.si = d->stride; p.so = 1; p.ph = 0; p.pw = 0; p.Hout = d->Ho; p.Wout = d->Wo;
    p.a_ld_pix = d->x_ld_pix; p.a_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
    p.w_ld_n = (long)d->R * d->S * d->C;
    p.a_bytes = ((long)(d->N - 1) * d->x_ld_img + ((long)d->H * d->W - 1) * d->x_ld_pix + d->C) * 2;
    p.w_bytes = (long)d->K * p.w_ld_n * 2;
    p.act = act;
    return run_hconv(p, stream);
}

// ConvTranspose2d(kernel 2, stride 2) + bias (reference modules/sampling.py:72-73, the YOLO-NAS up stages): every output pixel has exactly
// one tap, so the four output-parity classes are four 1x1 convolutions writing with stride 2.  w4: [2][2][K][C] bf16 (parity-major).
extern "C" int32_t sgx_hconvT2x2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const void* x, int64_t x_ld_pix, int64_t x_ld_img,
                                     const void* w4, const float* bias, void* y, int64_t y_ld_pix, int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(x && w4 && y && N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "hconvT2x2_fwd: bad args");
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
            HconvParams p;
            memset(&p, 0, sizeof(p));
            p.A = (const unsigned short*)x; p.Wt = (const unsigned short*)w4 + (long)(i * 2 + j) * K * C; p.bias = bias; p.Y = y; p.y_f32 = 0;
            p.M = N * H * W; p.Ha = H; p.Wa = W; p.Hin = H; p.Win = W; p.C = C; p.Nout = K; p.Th = 1; p.Tw = 1;
            p.si = 1; p.so = 2; p.ph = i; p.pw = j; p.Hout = 2 * H; p.Wout = 2 * W;
            p.a_ld_pix = x_ld_pix; p.a_ld_img = x_ld_img; p.y_ld_pix = y_ld_pix; p.y_ld_img = y_ld_img;
            p.w_ld_n = C;
            p.a_bytes = ((long)(N - 1) * x_ld_img + ((long)H * W - 1) * x_ld_pix + C) * 2;
            p.w_bytes = (long)K * C * 2;
            p.act = SGX_ACT_NONE;
            const int32_t rc = run_hconv(p, stream);
            if (rc) return rc;
        }
    return SGX_OK;
}

// ---- the few non-convolution ops of the deployment form, on bf16 ---------------------------------------------------------------------------
// max pooling (SPP, csp_darknet53.py:136-157): one thread per (output pixel, 8 channels); windows reach outside the image as -inf.  Max is
// exact in any precision: no rounding question.
__global__ void hmaxpool_kernel(int N, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, const unsigned short* x, long x_ld_pix,
                                long x_ld_img, unsigned short* y, long y_ld_pix, long y_ld_img) {
    const int C8 = C / 8;
    const long n = (long)N * Ho * Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        long t = i / C8;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const int img = (int)(t / Ho);
        float m[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
        for (int r = 0; r < k; ++r) {
            const int hi = ho * stride - pad + r;
            if (hi < 0 || hi >= H) continue;
            for (int s = 0; s < k; ++s) {
                const int wi = wo * stride - pad + s;
                if (wi < 0 || wi >= W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (long)img * x_ld_img + ((long)hi * W + wi) * x_ld_pix + c);
                const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    m[2 * e] = fmaxf(m[2 * e], sgx_u2f(u[e] << 16));
                    m[2 * e + 1] = fmaxf(m[2 * e + 1], sgx_u2f(u[e] & 0xffff0000u));
                }
            }
        }
        uint4 o;
        o.x = sgx_pack_bf16(m[0], m[1]); o.y = sgx_pack_bf16(m[2], m[3]); o.z = sgx_pack_bf16(m[4], m[5]); o.w = sgx_pack_bf16(m[6], m[7]);
        *reinterpret_cast<uint4*>(y + (long)img * y_ld_img + ((long)ho * Wo + wo) * y_ld_pix + c) = o;
    }
}
extern "C" int32_t sgx_hmaxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const void* x, int64_t x_ld_pix,
                                    int64_t x_ld_img, void* y, int64_t y_ld_pix, int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(x && y && C % 8 == 0 && k > 0 && stride > 0 && pad >= 0 && 2 * pad <= k, "hmaxpool_fwd: bad args (C % 8 == 0, pad <= k / 2)");
    SGX_CHECK_ARG(x_ld_pix % 8 == 0 && x_ld_img % 8 == 0 && y_ld_pix % 8 == 0 && y_ld_img % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                  "hmaxpool_fwd: strides and addresses must be multiples of 16 bytes");
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    SGX_CHECK_ARG(Ho > 0 && Wo > 0, "hmaxpool_fwd: empty output");
    const long n = (long)N * Ho * Wo * (C / 8), blocks = (n + 255) / 256;
    SGX_LAUNCH(hmaxpool_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, k, stride, pad, Ho, Wo,
               (const unsigned short*)x, (long)x_ld_pix, (long)x_ld_img, (unsigned short*)y, (long)y_ld_pix, (long)y_ld_img);
    SGX_CHECK_LAUNCH("hmaxpool_fwd");
    return SGX_OK;
}

// rows of C bf16 from one uniformly strided view into another (the skip tensor of a YOLO-NAS down stage into its concat slice)
__global__ void hcopy_kernel(long M, int C, const unsigned short* x, long x_ld, unsigned short* y, long y_ld, int vec) {
    if (vec) {
        const int C8 = C / 8;
        const long n = M * C8;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            const long m = i / C8;
            const int c = (int)(i - m * C8) * 8;
            *reinterpret_cast<uint4*>(y + m * y_ld + c) = *reinterpret_cast<const uint4*>(x + m * x_ld + c);
        }
    } else {
        const long n = M * C;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            const long m = i / C;
            y[m * y_ld + (i - m * C)] = x[m * x_ld + (i - m * C)];
        }
    }
}
extern "C" int32_t sgx_hcopy(const void* x, int64_t x_ld, int64_t M, int32_t C, void* y, int64_t y_ld, void* stream) {
    SGX_CHECK_ARG(x && y && M > 0 && C > 0 && x_ld >= C && y_ld >= C, "hcopy: bad args");
    const int vec = (C % 8 == 0 && x_ld % 8 == 0 && y_ld % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0) ? 1 : 0;
    const long n = vec ? M * (C / 8) : M * C, blocks = (n + 255) / 256;
    SGX_LAUNCH(hcopy_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, (long)M, C, (const unsigned short*)x, (long)x_ld,
               (unsigned short*)y, (long)y_ld, vec);
    SGX_CHECK_LAUNCH("hcopy");
    return SGX_OK;
}

// fp32 rows [M][Cs] (row pitch x_ld) -> bf16 rows [M][Cd], Cd >= Cs, the extra channels zero (the 4-channel fp32 image batch of the
// pre-processing launch -> the 8-channel bf16 batch the first convolution reads); round-to-nearest-even like torch's .to(bfloat16)
__global__ void cast_f32_bf16_kernel(long M, int Cs, int Cd, const float* x, long x_ld, unsigned short* y, long y_ld) {
    const int C2 = Cd / 2;
    const long n = M * C2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long m = i / C2;
        const int c = (int)(i - m * C2) * 2;
        const float a = c < Cs ? x[m * x_ld + c] : 0.f, b = c + 1 < Cs ? x[m * x_ld + c + 1] : 0.f;
        *reinterpret_cast<unsigned*>(y + m * y_ld + c) = sgx_pack_bf16(a, b);
    }
}
extern "C" int32_t sgx_cast_f32_bf16(const float* x, int64_t x_ld, int64_t M, int32_t Cs, void* y, int64_t y_ld, int32_t Cd, void* stream) {
    SGX_CHECK_ARG(x && y && M > 0 && Cs > 0 && Cd >= Cs && Cd % 2 == 0 && x_ld >= Cs && y_ld >= Cd && y_ld % 2 == 0 && ((uintptr_t)y % 4) == 0,
                  "cast_f32_bf16: bad args (even destination channel count and pitch)");
    const long n = M * (Cd / 2), blocks = (n + 255) / 256;
    SGX_LAUNCH(cast_f32_bf16_kernel, dim3((unsigned)(blocks > 32768 ? 32768 : blocks)), dim3(256), 0, stream, (long)M, Cs, Cd, x, (long)x_ld,
               (unsigned short*)y, (long)y_ld);
    SGX_CHECK_LAUNCH("cast_f32_bf16");
    return SGX_OK;
}

// ---- PP-YOLOE's deployment form on bf16 (round 6): the three ops around its convolutions that YOLO-NAS does not have ------------------------
// Reference: EffectiveSEBlock (modules/se_blocks.py:39-42: x * hardsigmoid(conv1x1(mean_hw(x)))), ESEAttn (pp_yolo_head.py:90-92: sigmoid),
// adaptive_avg_pool2d (pp_yolo_head.py:203), F.interpolate(scale_factor=2, mode="nearest") (pp_yolo_e/pan.py:170).  Under torch.autocast the
// reference keeps the mean in fp32 (autocast's fp32 list) and the gate product in the activation type: here the per-image channel means
// are fp32 sums of the bf16 activations in a fixed order, the 1x1 convolution on the [N,1,1,C] means stays on the fp32 path (N x C values),
// and the gate multiplies in fp32 and rounds once to bf16.
__device__ __forceinline__ float hgate_fn(float p, int gate) {
    if (gate == SGX_GATE_HARDSIGMOID) return fminf(fmaxf(p * (1.f / 6.f) + 0.5f, 0.f), 1.f);
    if (gate == SGX_GATE_SIGMOID) return 1.f / (1.f + expf(-p));
    return p;
}
// out[n][c] = scale * sum over the image's pixels: one workgroup per (image, 64 channels) - 8 channel octets x 32 pixel lanes, a lane walks
// every 32nd pixel, the 32 partial sums of a channel fold through LDS in lane order (deterministic)
__global__ __launch_bounds__(256) void hcolsum_kernel(int HW, int C, const unsigned short* x, long x_ld_pix, long x_ld_img, float scale, float* out) {
    __shared__ float part[32][64 + 1];
    const int img = blockIdx.y, c0 = blockIdx.x * 64;
    const int oct = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = c0 + oct * 8;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const unsigned short* xp = x + (long)img * x_ld_img + c;
        for (int p = pl; p < HW; p += 32) {
            const uint4 v = *reinterpret_cast<const uint4*>(xp + (long)p * x_ld_pix);
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += sgx_u2f(u[e] << 16);
                acc[2 * e + 1] += sgx_u2f(u[e] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[pl][oct * 8 + e] = acc[e];
    __syncthreads();
    if (threadIdx.x < 64 && c0 + (int)threadIdx.x < C) {
        float s = 0.f;
        for (int q = 0; q < 32; ++q) s += part[q][threadIdx.x];
        out[(long)img * C + c0 + threadIdx.x] = s * scale;
    }
}
extern "C" int32_t sgx_himage_colsum(int32_t N, int32_t HW, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, float scale, float* out, void* stream) {
    SGX_CHECK_ARG(x && out && N > 0 && HW > 0 && C > 0, "himage_colsum: bad args");
    SGX_CHECK_ARG(C % 8 == 0 && x_ld_pix % 8 == 0 && x_ld_img % 8 == 0 && ((uintptr_t)x % 16) == 0, "himage_colsum: channel count / strides / address must be multiples of 8 elements (16 bytes)");
    SGX_LAUNCH(hcolsum_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0, stream, HW, C, (const unsigned short*)x, (long)x_ld_pix, (long)x_ld_img, scale, out);
    SGX_CHECK_LAUNCH("himage_colsum");
    return SGX_OK;
}
// y = x * f(pre[n][c]): one thread per (pixel, 8 channels); the product in fp32, one rounding to bf16
__global__ void hgate_kernel(int N, int HW, int C, const unsigned short* x, long x_ld_pix, long x_ld_img, const float* pre, int gate, unsigned short* y,
                             long y_ld_pix, long y_ld_img) {
    const int C8 = C / 8;
    const long n = (long)N * HW * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        const long t = i / C8;
        const int p = (int)(t % HW), img = (int)(t / HW);
        const uint4 v = *reinterpret_cast<const uint4*>(x + (long)img * x_ld_img + (long)p * x_ld_pix + c);
        const float* pp = pre + (long)img * C + c;
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
        float r[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            r[2 * e] = sgx_u2f(u[e] << 16) * hgate_fn(pp[2 * e], gate);
            r[2 * e + 1] = sgx_u2f(u[e] & 0xffff0000u) * hgate_fn(pp[2 * e + 1], gate);
        }
        uint4 o;
        o.x = sgx_pack_bf16(r[0], r[1]); o.y = sgx_pack_bf16(r[2], r[3]); o.z = sgx_pack_bf16(r[4], r[5]); o.w = sgx_pack_bf16(r[6], r[7]);
        *reinterpret_cast<uint4*>(y + (long)img * y_ld_img + (long)p * y_ld_pix + c) = o;
    }
}
extern "C" int32_t sgx_hchannel_gate(int32_t N, int32_t HW, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, const float* pre, int32_t gate, void* y,
                                     int64_t y_ld_pix, int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(x && y && pre && N > 0 && HW > 0 && C > 0, "hchannel_gate: bad args");
    SGX_CHECK_ARG(gate >= SGX_GATE_NONE && gate <= SGX_GATE_SIGMOID, "hchannel_gate: unknown gate %d", gate);
    SGX_CHECK_ARG(C % 8 == 0 && x_ld_pix % 8 == 0 && x_ld_img % 8 == 0 && y_ld_pix % 8 == 0 && y_ld_img % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                  "hchannel_gate: channel count / strides / addresses must be multiples of 8 elements (16 bytes)");
    const long n = (long)N * HW * (C / 8), blocks = (n + 255) / 256;
    SGX_LAUNCH(hgate_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, HW, C, (const unsigned short*)x, (long)x_ld_pix, (long)x_ld_img, pre, gate,
               (unsigned short*)y, (long)y_ld_pix, (long)y_ld_img);
    SGX_CHECK_LAUNCH("hchannel_gate");
    return SGX_OK;
}
// nearest x2 up-sampling: one thread per (OUTPUT pixel, 8 channels) - a copy
__global__ void hupsample2x_kernel(int N, int H, int W, int C, const unsigned short* x, long x_ld_pix, long x_ld_img, unsigned short* y, long y_ld_pix, long y_ld_img) {
    const int C8 = C / 8, W2 = 2 * W, H2 = 2 * H;
    const long n = (long)N * H2 * W2 * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C8) * 8;
        long t = i / C8;
        const int wo = (int)(t % W2);
        t /= W2;
        const int ho = (int)(t % H2), img = (int)(t / H2);
        *reinterpret_cast<uint4*>(y + (long)img * y_ld_img + ((long)ho * W2 + wo) * y_ld_pix + c) =
            *reinterpret_cast<const uint4*>(x + (long)img * x_ld_img + ((long)(ho >> 1) * W + (wo >> 1)) * x_ld_pix + c);
    }
}
extern "C" int32_t sgx_hupsample2x_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, void* y, int64_t y_ld_pix,
                                       int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0, "hupsample2x: bad args");
    SGX_CHECK_ARG(C % 8 == 0 && x_ld_pix % 8 == 0 && x_ld_img % 8 == 0 && y_ld_pix % 8 == 0 && y_ld_img % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0,
                  "hupsample2x: channel count / strides / addresses must be multiples of 8 elements (16 bytes)");
    const long n = (long)N * 4 * H * W * (C / 8), blocks = (n + 255) / 256;
    SGX_LAUNCH(hupsample2x_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, N, H, W, C, (const unsigned short*)x, (long)x_ld_pix, (long)x_ld_img,
               (unsigned short*)y, (long)y_ld_pix, (long)y_ld_img);
    SGX_CHECK_LAUNCH("hupsample2x");
    return SGX_OK;
}
