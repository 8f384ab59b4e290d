// Implicit-GEMM convolution for gfx950 on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 fma
// chains, 157 TFLOP/s peak) - forward, data gradient and weight gradient.
//
// HBM layout: activations NHWC (+ explicit pixel/image strides), weights OHWI, so the GEMM-K axis
// (input channels of one filter tap) is contiguous for BOTH operands: a workgroup stages
// [BM pixels x 16 ch] and [BN filters x 16 ch] slabs with 16-byte coalesced loads, keeps them in LDS
// with a 20-float row pitch (conflict-free ds_read_b128: 16 consecutive rows hit 16 disjoint 4-bank
// groups), and every wave reads its 32x32x2 MFMA fragments as two 128-bit LDS loads per 8 k-steps.
// Double-buffered LDS, register-staged prefetch of the next slab issued before the MFMA block, one
// barrier per slab.  Workgroup -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a
// contiguous run of tiles, N-tiles of the same pixel slab adjacent, so halo rows and the slab itself
// are re-read from that XCD's L2.
//
// One "gather GEMM" kernel serves forward and data-gradient: output row m <-> a pixel of an output
// grid, tap t contributes input pixel (a*si + dh[t], b*si + dw[t]).  Stride-s data gradients are
// decomposed into s*s output-parity classes, each a dense stride-1 problem over the dY grid with its
// own tap subset (no multiply-by-zero work).  Reference call sites: see include/sgx_hip.h.
#include "sgx_common.h"

#define SGX_MAX_TAPS 64

// ------------------------------------------------------------------------------------------------
// Optional per-launch timing of the two MFMA kernel classes (bench.py's roofline leg): HIP events recorded on the
// launch stream around every igemm / wgrad launch, algorithmic FLOPs tallied next to them.  Off by default.
// ------------------------------------------------------------------------------------------------
#ifndef SGX_EMU
#include <mutex>
#include <vector>
namespace {
struct ProfRec {
    hipEvent_t a, b;
    double flops;
    int cls;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
std::vector<ProfRec> g_prof_recs;
std::vector<hipEvent_t> g_prof_pool;
hipEvent_t prof_event() {
    if (!g_prof_pool.empty()) {
        hipEvent_t e = g_prof_pool.back();
        g_prof_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct ProfScope {
    bool on;
    ProfRec r;
    hipStream_t st;
    ProfScope(int cls, double flops, void* stream) : on(false), st((hipStream_t)stream) {
        std::lock_guard<std::mutex> g(g_prof_mu);
        if (!g_prof_on) return;
        on = true;
        r.cls = cls;
        r.flops = flops;
        r.a = prof_event();
        r.b = prof_event();
        (void)hipEventRecord(r.a, st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(r.b, st);
        std::lock_guard<std::mutex> g(g_prof_mu);
        g_prof_recs.push_back(r);
    }
};
}  // namespace
extern "C" int32_t sgx_prof_enable(int32_t on) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    for (auto& r : g_prof_recs) {
        g_prof_pool.push_back(r.a);
        g_prof_pool.push_back(r.b);
    }
    g_prof_recs.clear();
    g_prof_on = on != 0;
    return SGX_OK;
}
extern "C" int32_t sgx_prof_summary(int32_t cls, double* ms, double* flops, int64_t* launches) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    double t = 0.0, f = 0.0;
    long n = 0;
    for (auto& r : g_prof_recs) {
        if (r.cls != cls) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "prof: event sync failed");
        float e = 0.f;
        if (hipEventElapsedTime(&e, r.a, r.b) != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "prof: elapsed failed");
        t += e;
        f += r.flops;
        ++n;
    }
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (launches) *launches = n;
    return SGX_OK;
}
#define SGX_PROF(cls, flops, stream) ProfScope prof_scope__((cls), (flops), (stream))
#else
extern "C" int32_t sgx_prof_enable(int32_t) { return SGX_OK; }
extern "C" int32_t sgx_prof_summary(int32_t, double* ms, double* flops, int64_t* launches) {
    if (ms) *ms = 0;
    if (flops) *flops = 0;
    if (launches) *launches = 0;
    return SGX_OK;
}
#define SGX_PROF(cls, flops, stream)
#endif

struct IgemmParams {
    const float* A;
    const float* Wt;
    const float* bias;
    const float* addend;
    float* Y;
    float* stat_partials;
    int M, Ha, Wa, Hin, Win, C, Nout, T;
    int si, so, ph, pw, Hout, Wout;
    long a_ld_pix, a_ld_img, y_ld_pix, y_ld_img;
    long w_ld_n;
    int act, accumulate;
    int mt, nt, nblk, chunk;  // tile counts and XCD chunk
    int stat_nblk;
    signed char dh[SGX_MAX_TAPS], dw[SGX_MAX_TAPS];
};

#define IG_BK 16
#define IG_LD 20

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(IgemmParams p) {
    constexpr int NTH = WM * WN * 64;   // threads per workgroup
    constexpr int RPP = NTH / 4;        // slab rows staged per pass (4 threads x 16 B per 16-float row)
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AJ = (BM + RPP - 1) / RPP, BJ = (BN + RPP - 1) / RPP;
    static_assert(TM >= 1 && TN >= 1 && TM * WM * 32 == BM && TN * WN * 32 == BN, "bad tile");
    static_assert(NTH >= BM && NTH >= BN, "epilogue helpers need one thread per tile row/col");
    __shared__ float As[2 * BM * IG_LD];
    __shared__ float Bs[2 * BN * IG_LD];
    __shared__ long long rowoff[BM];
    __shared__ float red[2 * WM * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile assignment (block b runs on XCD b%8; give each XCD a contiguous run of tiles)
    const int bid = blockIdx.x;
    const int lin = (bid & 7) * p.chunk + (bid >> 3);
    if (lin >= p.nblk) return;  // whole workgroup leaves together (before any barrier)
    const int mtile = lin / p.nt, ntile = lin - mtile * p.nt;
    const int m0 = mtile * BM, n0 = ntile * BN;

    const int lrow = tid >> 2, chunk4 = (tid & 3) * 4;
    long abase[AJ];
    int ah[AJ], aw[AJ];
    const int hw = p.Ha * p.Wa;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        int row = lrow + RPP * j;
        int m = m0 + row;
        if (row < BM && m < p.M) {
            int img = m / hw;
            int rem = m - img * hw;
            int a = rem / p.Wa;
            int b = rem - a * p.Wa;
            abase[j] = (long)img * p.a_ld_img;
            ah[j] = a * p.si;
            aw[j] = b * p.si;
        } else {
            abase[j] = 0;
            ah[j] = -(1 << 28);
            aw[j] = 0;
        }
    }
    if (tid < BM) {
        int m = m0 + tid;
        long long off = -1;
        if (m < p.M) {
            int img = m / hw;
            int rem = m - img * hw;
            int a = rem / p.Wa;
            int b = rem - a * p.Wa;
            off = (long long)img * p.y_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.y_ld_pix;
        }
        rowoff[tid] = off;
    }
    long bbase[BJ];
    bool bok[BJ];
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        int row = lrow + RPP * j;
        int n = n0 + row;
        bok[j] = (row < BN) && (n < p.Nout);
        bbase[j] = (long)n * p.w_ld_n;
    }

    const int cpt = (p.C + IG_BK - 1) / IG_BK;
    const int nkt = p.T * cpt;

    float4 ra[AJ], rb[BJ];
    auto load_tile = [&](int kt) {
        int tap = kt / cpt;
        int c0 = (kt - tap * cpt) * IG_BK + chunk4;
        int dh = p.dh[tap], dw = p.dw[tap];
        bool cok = c0 < p.C;
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            int hi = ah[j] + dh, wi = aw[j] + dw;
            bool ok = cok && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
            ra[j] = ok ? sgx_ld4(p.A + abase[j] + ((long)hi * p.Win + wi) * p.a_ld_pix + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            rb[j] = (cok && bok[j]) ? sgx_ld4(p.Wt + bbase[j] + (long)tap * p.C + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            int row = lrow + RPP * j;
            if (row < BM) sgx_st4(&As[buf * BM * IG_LD + row * IG_LD + chunk4], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            int row = lrow + RPP * j;
            if (row < BN) sgx_st4(&Bs[buf * BN * IG_LD + row * IG_LD + chunk4], rb[j]);
        }
    };

    sgx_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nkt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);  // global loads in flight under the MFMA block

        float af[TM][8], bf[TN][8];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float* s = &As[buf * BM * IG_LD + (wm * TM * 32 + i * 32 + frow) * IG_LD + fk];
            float4 v0 = sgx_ld4(s), v1 = sgx_ld4(s + 4);
            af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
            af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* s = &Bs[buf * BN * IG_LD + (wn * TN * 32 + j * 32 + frow) * IG_LD + fk];
            float4 v0 = sgx_ld4(s), v1 = sgx_ld4(s + 4);
            bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
            bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);

        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + addend + accumulate + activation, optional BN partial statistics ----
    float csum[TN], csq[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) csum[j] = csq[j] = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
        const bool colok = col < p.Nout;
        const float bv = (p.bias && colok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const long long off = rowoff[row];
                if (off >= 0 && colok) {
                    float v = acc[i][j][r] + bv;
                    if (p.addend) v += p.addend[off + col];
                    if (p.accumulate) v += p.Y[off + col];
                    csum[j] += v;
                    csq[j] += v * v;
                    p.Y[off + col] = sgx_act(v, p.act);
                }
            }
        }
    }
    if (p.stat_partials) {
        // lane l and l^32 hold the same column: fold, then fold the WM waves that share the column
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            csum[j] += __shfl_xor(csum[j], 32);
            csq[j] += __shfl_xor(csq[j], 32);
        }
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                int c = wn * TN * 32 + j * 32 + lane;
                red[(0 * WM + wm) * BN + c] = csum[j];
                red[(1 * WM + wm) * BN + c] = csq[j];
            }
        }
        __syncthreads();
        if (tid < BN) {
            int col = n0 + tid;
            if (col < p.Nout) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    s += red[(0 * WM + w) * BN + tid];
                    q += red[(1 * WM + w) * BN + tid];
                }
                p.stat_partials[(long)mtile * p.Nout + col] = s;
                p.stat_partials[((long)p.stat_nblk + mtile) * p.Nout + col] = q;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// tile-shape selection
// ------------------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
};
static TileCfg pick_tile(long M, int N) {
    // N tile: least channel padding among {32,64,96,128}, ties to the wider tile (more operand reuse).
    // M tile: 128 pixels unless that leaves the 256 CUs with fewer than two workgroups each.
    const int cand[4] = {32, 64, 96, 128};
    int bn = 128, best = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        int padded = ((N + cand[i] - 1) / cand[i]) * cand[i];
        if (padded <= best) {
            best = padded;
            bn = cand[i];
        }
    }
    int bm = 128;
    long blocks = ((M + 127) / 128) * ((N + bn - 1) / bn);
    if (blocks < 512) bm = 64;
    return TileCfg{bm, bn};
}

template <int BM, int BN, int WM, int WN>
static void launch_igemm(IgemmParams& p, void* stream) {
    p.mt = sgx_cdiv(p.M, BM);
    p.nt = sgx_cdiv(p.Nout, BN);
    p.nblk = p.mt * p.nt;
    p.chunk = sgx_cdiv(p.nblk, 8);
    int grid = p.chunk * 8;
    SGX_LAUNCH((igemm_kernel<BM, BN, WM, WN>), dim3(grid), dim3(WM * WN * 64), 0, stream, p);
}

static int32_t run_igemm(IgemmParams& p, int bm, int bn, void* stream) {
    if (p.T > SGX_MAX_TAPS) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: more than %d taps", SGX_MAX_TAPS);
    SGX_PROF(0, 2.0 * (double)p.M * (double)p.Nout * (double)p.C * (double)p.T, stream);
    if (bm == 128 && bn == 128) launch_igemm<128, 128, 2, 2>(p, stream);
    else if (bm == 128 && bn == 96) launch_igemm<128, 96, 4, 1>(p, stream);
    else if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2>(p, stream);
    else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1>(p, stream);
    else if (bm == 64 && bn == 128) launch_igemm<64, 128, 2, 2>(p, stream);
    else if (bm == 64 && bn == 96) launch_igemm<64, 96, 2, 1>(p, stream);
    else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2>(p, stream);
    else if (bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1>(p, stream);
    else SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: no tile %dx%d", bm, bn);
    SGX_CHECK_LAUNCH("igemm");
    return SGX_OK;
}

static int32_t check_desc(const sgx_conv_desc* d) {
    SGX_CHECK_ARG(d != nullptr, "conv: null desc");
    SGX_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0, "conv: non-positive dims");
    SGX_CHECK_ARG(d->C % 4 == 0, "conv: C=%d must be a multiple of 4 (pad the input)", d->C);
    SGX_CHECK_ARG(d->R > 0 && d->S > 0 && d->R * d->S <= SGX_MAX_TAPS, "conv: filter %dx%d unsupported", d->R, d->S);
    SGX_CHECK_ARG(d->stride >= 1 && d->pad >= 0, "conv: bad stride/pad");
    SGX_CHECK_ARG(d->Ho == (d->H + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->S) / d->stride + 1,
                  "conv: Ho/Wo do not match (H+2p-R)/s+1");
    SGX_CHECK_ARG(d->x_ld_pix >= d->C && d->x_ld_pix % 4 == 0 && d->y_ld_pix >= d->K, "conv: bad pixel strides");
    return SGX_OK;
}

static int env_tile(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}

extern "C" int32_t sgx_conv2d_fwd_stat_blocks(const sgx_conv_desc* d) {
    long M = (long)d->N * d->Ho * d->Wo;
    TileCfg t = pick_tile(M, d->K);
    if (env_tile("SGX_CONV_BM")) t.bm = env_tile("SGX_CONV_BM");
    return sgx_cdiv(M, t.bm);
}

extern "C" int32_t sgx_conv2d_fwd(const sgx_conv_desc* d, const float* x, const float* w, const float* bias,
                                  const float* addend, float* y, int32_t act, float* stat_partials, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(x && w && y, "conv fwd: null pointer");
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.Wt = w; p.bias = bias; p.addend = addend; p.Y = y; p.stat_partials = stat_partials;
    p.M = d->N * d->Ho * d->Wo; p.Ha = d->Ho; p.Wa = d->Wo; p.Hin = d->H; p.Win = d->W;
    p.C = d->C; p.Nout = d->K; p.T = d->R * d->S;
    p.si = d->stride; p.so = 1; p.ph = 0; p.pw = 0; p.Hout = d->Ho; p.Wout = d->Wo;
    p.a_ld_pix = d->x_ld_pix; p.a_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
    p.w_ld_n = (long)p.T * d->C;
    p.act = act; p.accumulate = 0;
    for (int r = 0; r < d->R; ++r)
        for (int s = 0; s < d->S; ++s) {
            p.dh[r * d->S + s] = (signed char)(r - d->pad);
            p.dw[r * d->S + s] = (signed char)(s - d->pad);
        }
    TileCfg t = pick_tile(p.M, p.Nout);
    if (env_tile("SGX_CONV_BM")) t.bm = env_tile("SGX_CONV_BM");
    if (env_tile("SGX_CONV_BN")) t.bn = env_tile("SGX_CONV_BN");
    p.stat_nblk = sgx_cdiv(p.M, t.bm);
    return run_igemm(p, t.bm, t.bn, stream);
}

// ------------------------------------------------------------------------------------------------
// data gradient
// ------------------------------------------------------------------------------------------------
struct TapList {
    unsigned char idx[SGX_MAX_TAPS];
};
// wt[c][t][k] = w[k][tap_t][c] for the taps of one output-parity class
__global__ void wtrans_kernel(const float* w, float* wt, int K, int C, int RS, int T, TapList taps) {
    long n = (long)C * T * K;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int k = (int)(i % K);
        long r = i / K;
        int t = (int)(r % T);
        int c = (int)(r / T);
        wt[i] = w[((long)k * RS + taps.idx[t]) * C + c];
    }
}

extern "C" int64_t sgx_conv2d_bwd_data_workspace(const sgx_conv_desc* d) {
    return (int64_t)d->R * d->S * d->C * d->K * sizeof(float) + 256;
}

static int32_t conv_bwd_data_impl(const sgx_conv_desc* d, const float* dy, const float* w, const float* bias, const float* addend,
                                  float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream);
extern "C" int32_t sgx_conv2d_bwd_data(const sgx_conv_desc* d, const float* dy, const float* w, const float* addend,
                                       float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream) {
    return conv_bwd_data_impl(d, dy, w, nullptr, addend, dx, accumulate, ws, ws_bytes, stream);
}
static int32_t conv_bwd_data_impl(const sgx_conv_desc* d, const float* dy, const float* w, const float* bias, const float* addend,
                                  float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(dy && w && dx, "conv bwd_data: null pointer");
    SGX_CHECK_ARG(d->K % 4 == 0 && d->y_ld_pix % 4 == 0, "conv bwd_data: K and dy pixel stride must be multiples of 4");
    if (ws_bytes < sgx_conv2d_bwd_data_workspace(d) || !ws) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_data: workspace too small");
    const int s = d->stride;
    float* wt = (float*)ws;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            IgemmParams p;
            memset(&p, 0, sizeof(p));
            int T = 0;
            TapList taps;
            memset(&taps, 0, sizeof(taps));
            for (int r = 0; r < d->R; ++r) {
                int vh = ph + d->pad - r;
                if (((vh % s) + s) % s) continue;
                for (int q = 0; q < d->S; ++q) {
                    int vw = pw + d->pad - q;
                    if (((vw % s) + s) % s) continue;
                    p.dh[T] = (signed char)(vh / s);
                    p.dw[T] = (signed char)(vw / s);
                    taps.idx[T] = (unsigned char)(r * d->S + q);
                    ++T;
                }
            }
            const int Ha = (d->H - ph + s - 1) / s, Wa = (d->W - pw + s - 1) / s;
            if (Ha <= 0 || Wa <= 0) continue;
            p.A = dy; p.Wt = wt; p.bias = bias; p.addend = addend; p.Y = dx; p.stat_partials = nullptr;
            p.M = d->N * Ha * Wa; p.Ha = Ha; p.Wa = Wa; p.Hin = d->Ho; p.Win = d->Wo;
            p.C = d->K; p.Nout = d->C; p.T = T;
            p.si = 1; p.so = s; p.ph = ph; p.pw = pw; p.Hout = d->H; p.Wout = d->W;
            p.a_ld_pix = d->y_ld_pix; p.a_ld_img = d->y_ld_img; p.y_ld_pix = d->x_ld_pix; p.y_ld_img = d->x_ld_img;
            p.w_ld_n = (long)T * d->K;
            p.act = SGX_ACT_NONE; p.accumulate = accumulate;
            if (T > 0) {
                long n = (long)d->C * T * d->K;
                int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
                SGX_LAUNCH(wtrans_kernel, dim3(grid), dim3(256), 0, stream, w, wt, d->K, d->C, d->R * d->S, T, taps);
                SGX_CHECK_LAUNCH("wtrans");
            }
            // T == 0: the class receives no contribution; the kernel still writes addend/accumulate/zero.
            TileCfg t = pick_tile(p.M, p.Nout);
            if (env_tile("SGX_CONV_BM")) t.bm = env_tile("SGX_CONV_BM");
            if (env_tile("SGX_CONV_BN")) t.bn = env_tile("SGX_CONV_BN");
            rc = run_igemm(p, t.bm, t.bn, stream);
            if (rc) return rc;
            wt += (long)d->C * T * d->K;
        }
    return SGX_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: for one tap, dW[k][c] = sum_m dY[m][k] * X[pix(m,tap)][c]; split over pixel chunks,
// partial slabs reduced by a second kernel in a fixed order (deterministic, no float atomics).
// ------------------------------------------------------------------------------------------------
struct WgradParams {
    const float* X;
    const float* DY;
    float* part;  // [ksplit][K][T][C]
    int N, H, W, C, K, R, S, stride, pad, Ho, Wo;
    long x_ld_pix, x_ld_img, y_ld_pix, y_ld_img;
    int M, ksplit, mchunk;  // pixels per split (multiple of 16)
    int kt_tiles, ct_tiles;
};

#define WG_BKP 16

template <int BNK, int BC, int WK, int WC>
__global__ __launch_bounds__(WK * WC * 64) void wgrad_kernel(WgradParams p) {
    constexpr int NTH = WK * WC * 64;
    constexpr int TK = BNK / (WK * 32), TC = BC / (WC * 32);
    static_assert(TK >= 1 && TC >= 1 && TK * WK * 32 == BNK && TC * WC * 32 == BC, "bad tile");
    constexpr int DJ = (WG_BKP * BNK / 4 + NTH - 1) / NTH, XJ = (WG_BKP * BC / 4 + NTH - 1) / NTH;
    __shared__ float Ds[2 * WG_BKP * BNK];
    __shared__ float Xs[2 * WG_BKP * BC];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / WC, wc = wave % WC;
    // blockIdx.x = ((split * kt_tiles + ktile) * ct_tiles + ctile) * T + tap   (taps of one slab adjacent)
    const int T = p.R * p.S;
    int b = blockIdx.x;
    const int tap = b % T; b /= T;
    const int ctile = b % p.ct_tiles; b /= p.ct_tiles;
    const int ktile = b % p.kt_tiles; b /= p.kt_tiles;
    const int split = b;
    const int k0 = ktile * BNK, c0 = ctile * BC;
    const int tr = tap / p.S, ts = tap - tr * p.S;
    const int mbeg = split * p.mchunk;
    const int mend = min(p.M, mbeg + p.mchunk);
    const int nkt = (mend > mbeg) ? (mend - mbeg + WG_BKP - 1) / WG_BKP : 0;
    const int hw = p.Ho * p.Wo;

    float4 rd[DJ], rx[XJ];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            int idx = tid + NTH * j;
            int row = idx / (BNK / 4), c4 = (idx % (BNK / 4)) * 4;
            int m = mbeg + kt * WG_BKP + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < WG_BKP * BNK / 4 && m < mend && k0 + c4 < p.K) {
                int img = m / hw;
                int rem = m - img * hw;
                v = sgx_ld4(p.DY + (long)img * p.y_ld_img + (long)rem * p.y_ld_pix + k0 + c4);
            }
            rd[j] = v;
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            int idx = tid + NTH * j;
            int row = idx / (BC / 4), c4 = (idx % (BC / 4)) * 4;
            int m = mbeg + kt * WG_BKP + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < WG_BKP * BC / 4 && m < mend && c0 + c4 < p.C) {
                int img = m / hw;
                int rem = m - img * hw;
                int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                int hi = ho * p.stride + tr - p.pad, wi = wo * p.stride + ts - p.pad;
                if (hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    v = sgx_ld4(p.X + (long)img * p.x_ld_img + ((long)hi * p.W + wi) * p.x_ld_pix + c0 + c4);
            }
            rx[j] = v;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            int idx = tid + NTH * j;
            if (idx < WG_BKP * BNK / 4) sgx_st4(&Ds[buf * WG_BKP * BNK + idx * 4], rd[j]);
        }
#pragma unroll
        for (int j = 0; j < XJ; ++j) {
            int idx = tid + NTH * j;
            if (idx < WG_BKP * BC / 4) sgx_st4(&Xs[buf * WG_BKP * BC + idx * 4], rx[j]);
        }
    };

    sgx_f32x16 acc[TK][TC];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nkt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    const int fcol = lane & 31, fkh = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < WG_BKP / 2; ++kk) {
            float af[TK], bf[TC];
#pragma unroll
            for (int i = 0; i < TK; ++i) af[i] = Ds[buf * WG_BKP * BNK + (2 * kk + fkh) * BNK + wk * TK * 32 + i * 32 + fcol];
#pragma unroll
            for (int j = 0; j < TC; ++j) bf[j] = Xs[buf * WG_BKP * BC + (2 * kk + fkh) * BC + wc * TC * 32 + j * 32 + fcol];
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }
    // partial slab: part[split][k][tap][c]
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const int c = c0 + wc * TC * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * TK * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < p.K && c < p.C) p.part[(((long)split * p.K + k) * T + tap) * p.C + c] = acc[i][j][r];
            }
        }
}

// dw[i] += sum_k part[k][i]: a workgroup owns EL consecutive elements and walks the split slabs with KL lanes per element
// (KL * EL = 256), then folds the KL lane sums through LDS in a fixed order (deterministic, no atomics).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* part, float* dw, long n, int ksplit, int KL) {
    __shared__ float red[256];
    const int EL = 256 / KL;
    const int el = threadIdx.x % EL, kl = threadIdx.x / EL;
    const long i = (long)blockIdx.x * EL + el;
    float s = 0.f;
    if (i < n) {
#pragma unroll 8
        for (int k = kl; k < ksplit; k += KL) s += part[(long)k * n + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (kl == 0 && i < n) {
        float t = 0.f;
        for (int k = 0; k < KL; ++k) t += red[k * EL + el];
        dw[i] += t;
    }
}

struct WgradPlan {
    int bnk, bc, kt_tiles, ct_tiles, ksplit, mchunk;
};
static int wg_tile(int n) {  // least padding among {32,64,128}, ties to the wider tile
    const int cand[3] = {32, 64, 128};
    int b = 128, best = 1 << 30;
    for (int i = 0; i < 3; ++i) {
        int padded = ((n + cand[i] - 1) / cand[i]) * cand[i];
        if (padded <= best) {
            best = padded;
            b = cand[i];
        }
    }
    return b;
}
extern "C" int32_t sgx_stats_blocks(int64_t M);
extern "C" int64_t sgx_colsum_workspace(int64_t M, int32_t C);
static WgradPlan wgrad_plan(const sgx_conv_desc* d) {
    WgradPlan pl;
    pl.bnk = wg_tile(d->K);
    pl.bc = wg_tile(d->C);
    pl.kt_tiles = sgx_cdiv(d->K, pl.bnk);
    pl.ct_tiles = sgx_cdiv(d->C, pl.bc);
    long M = (long)d->N * d->Ho * d->Wo;
    long tiles = (long)pl.kt_tiles * pl.ct_tiles * d->R * d->S;
    long ks = (1024 + tiles - 1) / tiles;   // ~1024 workgroups (4 per CU)
    long maxsplit = (M + 255) / 256;         // at least 256 pixels (16 slabs) per split
    if (ks > maxsplit) ks = maxsplit;
    if (ks < 1) ks = 1;
    long mchunk = (M + ks - 1) / ks;
    mchunk = ((mchunk + WG_BKP - 1) / WG_BKP) * WG_BKP;
    ks = (M + mchunk - 1) / mchunk;
    pl.ksplit = (int)ks;
    pl.mchunk = (int)mchunk;
    return pl;
}

extern "C" int64_t sgx_conv2d_bwd_weight_workspace(const sgx_conv_desc* d) {
    WgradPlan pl = wgrad_plan(d);
    int64_t slabs = (int64_t)pl.ksplit * d->K * d->R * d->S * d->C * sizeof(float);
    int64_t bias = sgx_colsum_workspace((int64_t)d->N * d->Ho * d->Wo, d->K);
    return (slabs > bias ? slabs : bias) + 256;
}


extern "C" int32_t sgx_conv2d_bwd_weight(const sgx_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                         void* ws, int64_t ws_bytes, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(x && dy && dw, "conv bwd_weight: null pointer");
    SGX_CHECK_ARG(d->K % 4 == 0 && d->y_ld_pix % 4 == 0, "conv bwd_weight: K and dy pixel stride must be multiples of 4");
    if (!ws || ws_bytes < sgx_conv2d_bwd_weight_workspace(d)) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_weight: workspace too small");
    WgradPlan pl = wgrad_plan(d);
    WgradParams p;
    p.X = x; p.DY = dy; p.part = (float*)ws;
    p.N = d->N; p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.R = d->R; p.S = d->S; p.stride = d->stride; p.pad = d->pad;
    p.Ho = d->Ho; p.Wo = d->Wo;
    p.x_ld_pix = d->x_ld_pix; p.x_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
    p.M = d->N * d->Ho * d->Wo; p.ksplit = pl.ksplit; p.mchunk = pl.mchunk; p.kt_tiles = pl.kt_tiles; p.ct_tiles = pl.ct_tiles;
    long nblk = (long)pl.ksplit * pl.kt_tiles * pl.ct_tiles * d->R * d->S;
    dim3 grid((unsigned)nblk);
    {
    SGX_PROF(1, 2.0 * (double)p.M * (double)d->K * (double)d->C * (double)(d->R * d->S), stream);
#define WG_CASE(BK_, BC_, WK_, WC_) \
    if (pl.bnk == BK_ && pl.bc == BC_) SGX_LAUNCH((wgrad_kernel<BK_, BC_, WK_, WC_>), grid, dim3(WK_ * WC_ * 64), 0, stream, p)
    WG_CASE(128, 128, 2, 2);
    WG_CASE(128, 64, 2, 2);
    WG_CASE(128, 32, 4, 1);
    WG_CASE(64, 128, 2, 2);
    WG_CASE(64, 64, 2, 2);
    WG_CASE(64, 32, 2, 1);
    WG_CASE(32, 128, 1, 4);
    WG_CASE(32, 64, 1, 2);
    WG_CASE(32, 32, 1, 1);
#undef WG_CASE
    }
    SGX_CHECK_LAUNCH("wgrad");
    long n = (long)d->K * d->R * d->S * d->C;
    int KL = 1;
    while (KL < 16 && KL < pl.ksplit) KL *= 2;
    const int EL = 256 / KL;
    SGX_LAUNCH(wgrad_reduce_kernel, dim3((unsigned)((n + EL - 1) / EL)), dim3(256), 0, stream, (const float*)ws, dw, n, pl.ksplit, KL);
    SGX_CHECK_LAUNCH("wgrad_reduce");
    if (dbias) {
        // column sum of dy: reuse the partial buffer tail is not safe while reduce may still read -> stream order makes it safe
        return sgx_colsum(dy, d->y_ld_pix, (int64_t)d->N * d->Ho * d->Wo, d->K, (int64_t)d->Ho * d->Wo, d->y_ld_img, dbias, 1, (float*)ws, stream);
    }
    return SGX_OK;
}

// ------------------------------------------------------------------------------------------------
// ConvTranspose2d k=2 s=2: the adjoint of conv(in = [N,2H,2W,K], weight wt[C][2][2][K], stride 2, pad 0)
// ------------------------------------------------------------------------------------------------
static sgx_conv_desc convT_adjoint_desc(int N, int H, int W, int C, int K, long small_ld_pix, long small_ld_img, long big_ld_pix,
                                        long big_ld_img) {
    sgx_conv_desc d;
    d.N = N; d.H = 2 * H; d.W = 2 * W; d.C = K; d.K = C; d.R = 2; d.S = 2; d.stride = 2; d.pad = 0; d.Ho = H; d.Wo = W;
    d.x_ld_pix = big_ld_pix; d.x_ld_img = big_ld_img; d.y_ld_pix = small_ld_pix; d.y_ld_img = small_ld_img;
    return d;
}
extern "C" int64_t sgx_convT2x2_workspace(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K) {
    sgx_conv_desc d = convT_adjoint_desc(N, H, W, C, K, C, (long)H * W * C, K, 4L * H * W * K);
    int64_t a = sgx_conv2d_bwd_data_workspace(&d), b = sgx_conv2d_bwd_weight_workspace(&d);
    int64_t c = sgx_colsum_workspace(4L * N * H * W, K) + 256;
    a = a > b ? a : b;
    return a > c ? a : c;
}
extern "C" int32_t sgx_convT2x2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix, int64_t x_ld_img,
                                    const float* wt, const float* bias, float* y, int64_t y_ld_pix, int64_t y_ld_img, void* ws,
                                    int64_t ws_bytes, void* stream) {
    sgx_conv_desc d = convT_adjoint_desc(N, H, W, C, K, x_ld_pix, x_ld_img, y_ld_pix, y_ld_img);
    return conv_bwd_data_impl(&d, x, wt, bias, nullptr, y, 0, ws, ws_bytes, stream);
}
extern "C" int32_t sgx_convT2x2_bwd_data(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* dy, int64_t dy_ld_pix,
                                         int64_t dy_ld_img, const float* wt, float* dx, int64_t dx_ld_pix, int64_t dx_ld_img, void* stream) {
    sgx_conv_desc d = convT_adjoint_desc(N, H, W, C, K, dx_ld_pix, dx_ld_img, dy_ld_pix, dy_ld_img);
    return sgx_conv2d_fwd(&d, dy, wt, nullptr, nullptr, dx, SGX_ACT_NONE, nullptr, stream);
}
extern "C" int32_t sgx_convT2x2_bwd_weight(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix,
                                           int64_t x_ld_img, const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img, float* dwt, float* dbias,
                                           void* ws, int64_t ws_bytes, void* stream) {
    sgx_conv_desc d = convT_adjoint_desc(N, H, W, C, K, x_ld_pix, x_ld_img, dy_ld_pix, dy_ld_img);
    int32_t rc = sgx_conv2d_bwd_weight(&d, dy, x, dwt, nullptr, ws, ws_bytes, stream);
    if (rc) return rc;
    if (dbias) {
        // bias gradient = column sum of dy over all 4*N*H*W output pixels (contiguous-image layout required)
        int64_t M = 4L * N * H * W;
        if (ws_bytes < sgx_colsum_workspace(M, K)) SGX_FAIL(SGX_ERR_WORKSPACE, "convT bwd_weight: workspace too small");
        return sgx_colsum(dy, dy_ld_pix, M, K, 4L * H * W, dy_ld_img, dbias, 1, (float*)ws, stream);
    }
    return SGX_OK;
}
