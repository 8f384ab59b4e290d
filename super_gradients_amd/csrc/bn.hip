// BatchNorm (training) statistics, the fused affine/residual/activation passes around it, and their
// backward - HBM-bound row x channel sweeps over NHWC tensors (rows = pixels, columns = channels).
//
// One sweep skeleton serves all of them: a 256-thread workgroup owns a contiguous run of rows and a
// strip of up to 256 channels; a thread owns one float4 channel group and every RL-th row, so a wave's
// loads are 16-byte, channel-contiguous (coalesced), and per-channel reductions are lane-local until a
// single LDS fold at the end.  Reductions are two-stage and deterministic: per-workgroup partials in
// fp32, a tiny finalize kernel accumulating them in fp64 in a fixed order (no float atomics).
// Reference call sites: include/sgx_hip.h (BatchNorm section).
#include "sgx_common.h"
#include <atomic>
#include <cstdlib>

#define SW_THREADS 256
#define SGX_WAVE_PRIO_DEFAULT 0
#define SW_MAXCG 64  // float4 channel groups per workgroup strip (256 channels)

struct SweepGeom {
    long M;
    int C, C4, CG, RL, nblk, rows_per_blk, ctiles;
    int prio;  // raise the waves' issue priority (SGX_WAVE_PRIO bit 1; sgx_common.h)
};
// SGX_WAVE_PRIO (environment, read once): bit 0 the finalize kernels, bit 1 the sweeps
static int wave_prio_mode() {
    static const int mode = [] {
        const char* e = getenv("SGX_WAVE_PRIO");
        return e ? atoi(e) : SGX_WAVE_PRIO_DEFAULT;
    }();
    return mode;
}

// Row blocks of a sweep: 64 rows per workgroup until the grid reaches 1024 workgroups.  (Round 1 used 256 rows: the 40x40 and 20x20
// levels of the network then ran their sweeps on 200 / 50 workgroups - fewer than the chip has CUs; 64 rows: +4 % on the whole train
// step, profiles/r2q_sweep_rows.txt.)
extern "C" int32_t sgx_stats_blocks(int64_t M) {
    long n = (M + 63) / 64;
    if (n < 1) n = 1;
    if (n > 1024) n = 1024;
    return (int32_t)n;
}

static SweepGeom sweep_geom(long M, int C) {
    SweepGeom g;
    g.M = M;
    g.C = C;
    g.C4 = C / 4;
    g.CG = g.C4 < SW_MAXCG ? g.C4 : SW_MAXCG;
    g.RL = SW_THREADS / g.CG;
    g.nblk = sgx_stats_blocks(M);
    g.rows_per_blk = (int)((M + g.nblk - 1) / g.nblk);
    g.ctiles = (g.C4 + g.CG - 1) / g.CG;
    g.prio = (wave_prio_mode() >> 1) & 1;
    return g;
}

// F: struct with  In load(long r, int c)  (all global loads of row r, channels c..c+3),  Cst consts(int c)  (the per-channel
// constants of the lane's four channels - scale / shift / coefficient rows - loaded ONCE, ahead of the row loop) and
// void apply(long r, int c, const In&, const Cst&, float4& q0, float4& q1)  (the arithmetic, the stores and up to two per-channel
// accumulations).  (Round 5: the constants used to be re-read inside apply for every row - the compiler cannot hoist them past the
// row's stores, which may alias them for all it knows: 7 of the 9 load instructions per row of the BatchNorm-backward apply, 14 of 17
// of the QARepVGG one, all L1 hits but each a trip through the texture path, which at 64 B/clk/CU was busier with them than with the
// data.)  The split lets the sweep issue the loads of FOUR rows before the first store: with one row in flight
// per lane these streaming kernels sat at ~40 % of the HBM rate (r1b profile) - latency-bound, not bandwidth-bound.
// In-place use (output aliasing an input) stays correct: a row is completely read before it is written, rows are disjoint.
template <typename F, int NQ>
__global__ __launch_bounds__(SW_THREADS) void sweep_kernel(F f, SweepGeom g, float* partials) {
    __shared__ float4 red[NQ > 0 ? NQ : 1][SW_THREADS];
    if (g.prio) SGX_WAVE_PRIO(2);
    const int tid = threadIdx.x;
    const int cg = tid % g.CG, rl = tid / g.CG;
    const int c4 = blockIdx.y * g.CG + cg;
    const bool live = (rl < g.RL) && (c4 < g.C4);
    const int c = c4 * 4;
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (live) {
        long r0 = (long)blockIdx.x * g.rows_per_blk;
        long r1 = r0 + g.rows_per_blk;
        if (r1 > g.M) r1 = g.M;
        long r = r0 + rl;
        const long st = g.RL;
        const typename F::Cst k = f.consts(c);
        for (; r + 3 * st < r1; r += 4 * st) {
            typename F::In i0 = f.load(r, c), i1 = f.load(r + st, c), i2 = f.load(r + 2 * st, c), i3 = f.load(r + 3 * st, c);
            f.apply(r, c, i0, k, q0, q1);
            f.apply(r + st, c, i1, k, q0, q1);
            f.apply(r + 2 * st, c, i2, k, q0, q1);
            f.apply(r + 3 * st, c, i3, k, q0, q1);
        }
        for (; r < r1; r += st) {
            typename F::In i0 = f.load(r, c);
            f.apply(r, c, i0, k, q0, q1);
        }
    }
    if (NQ > 0 && partials) {
        red[0][tid] = q0;
        if (NQ > 1) red[NQ > 1 ? 1 : 0][tid] = q1;
        __syncthreads();
        if (live && rl == 0) {
            // the lane sums meet in double: the partial row carries ONE fp32 rounding (random sign), not a chain of them - the per-channel
            // means the finalize kernels form from these rows (mean of g in the BatchNorm backward above all) are then good to ~1e-9 relative
            double s0[4] = {0.0, 0.0, 0.0, 0.0}, s1[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < g.RL; ++k) {
                float4 a = red[0][k * g.CG + cg];
                s0[0] += a.x; s0[1] += a.y; s0[2] += a.z; s0[3] += a.w;
                if (NQ > 1) {
                    float4 b = red[NQ > 1 ? 1 : 0][k * g.CG + cg];
                    s1[0] += b.x; s1[1] += b.y; s1[2] += b.z; s1[3] += b.w;
                }
            }
            sgx_st4(partials + (long)blockIdx.x * g.C + c, make_float4((float)s0[0], (float)s0[1], (float)s0[2], (float)s0[3]));
            if (NQ > 1) sgx_st4(partials + ((long)g.nblk + blockIdx.x) * g.C + c, make_float4((float)s1[0], (float)s1[1], (float)s1[2], (float)s1[3]));
        }
    }
}

template <typename F, int NQ>
static int32_t run_sweep(const F& f, long M, int C, float* partials, void* stream, const char* what) {
    SGX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "%s: need M>0 and C%%4==0 (C=%d)", what, C);
    SweepGeom g = sweep_geom(M, C);
    SGX_LAUNCH((sweep_kernel<F, NQ>), dim3(g.nblk, g.ctiles), dim3(SW_THREADS), 0, stream, f, g, partials);
    SGX_CHECK_LAUNCH(what);
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
struct StatsF {
    const float* x;
    long ld;
    struct In { float4 v; };
    __device__ In load(long r, int c) const { return In{sgx_ld4(x + r * ld + c)}; }
    struct Cst {};
    __device__ Cst consts(int) const { return Cst{}; }
    __device__ void apply(long, int, const In& in, const Cst&, float4& q0, float4& q1) const {
        const float4 v = in.v;
        q0.x += v.x; q0.y += v.y; q0.z += v.z; q0.w += v.w;
        q1.x += v.x * v.x; q1.y += v.y * v.y; q1.z += v.z * v.z; q1.w += v.w * v.w;
    }
};
extern "C" int32_t sgx_channel_stats_partial(const float* x, int64_t M, int32_t C, int64_t ld, float* partials, void* stream) {
    SGX_CHECK_ARG(x && partials, "channel_stats: null pointer");
    StatsF f{x, ld};
    return run_sweep<StatsF, 2>(f, M, C, partials, stream, "channel_stats");
}

// ---------------------------------------------------------------------------------------------
// Second stage of the per-channel reductions.  The producers (conv epilogue, sweeps) leave up to tens of thousands of
// fp32 partial rows per plane ([planes][nblk][C]); a single thread per channel walking them serially was the slowest
// kernel of the whole train step (r1a profile: bn_finalize 30 % of GPU time).  So: when nblk is large a wide
// pre-reduction (grid = channel strips x slices, one wave per row lane, 256-byte coalesced row reads, fp64 accumulate,
// fixed order) folds the rows into <= CR_MAX_SLICES fp64 rows per plane in the caller's workspace, and the finalize
// kernels then read at most that many.  Deterministic: every sum has a fixed association order.
// ---------------------------------------------------------------------------------------------
#define CR_MAX_SLICES 64
#define CR_DIRECT 32  // up to this many partial rows the finalize kernels read the fp32 partials directly

struct ColSrc {  // what a finalize kernel sums over: either the fp32 partials or the fp64 slices
    const float* f;
    const double* d;
    int n;     // rows per plane
    int coop;  // 1: the finalize kernel is launched with CO_CH x CO_RL threads per workgroup and folds the fp32 partial rows itself
    int prio;  // raise the waves' issue priority (SGX_WAVE_PRIO bit 0)
};
// ---- one-launch finalize (default; sgx_bn_set_fused_finalize(0) restores the two-launch form): instead of a pre-reduction
// launch + a finalize launch, the finalize kernel runs with CO_CH channels x CO_RL row lanes per workgroup; every lane folds its rows
// (b = lane, lane + CO_RL, ...) in fp64, the lane sums meet in LDS and are added in lane order (deterministic, no atomics).  Worth it
// while one workgroup can stream the partial rows of its channels faster than a second launch costs: nblk <= CR_COOP_MAX.
// These kernels are a handful of workgroups on an otherwise idle chip (they sit between a convolution and the sweep that needs its
// statistics), so their time is load LATENCY x dependent rounds, not bytes: all planes of FOUR rows are loaded before the first add
// (r3: the per-plane form measured 27 us for the five-moment finalize of an 80 x 80 map - 500 loads per lane, eight in flight).
// Workgroup shape (round 4): 256 threads = 4 channels x 64 row lanes, LDS 32 doubles per plane.  The first form was 1024 threads (16
// channels x 64 row lanes, 8 KB of LDS per plane): during the backward pass the weight-gradient kernels of the side stream fill every CU's
// LDS and wave slots, and a 1024-thread workgroup then waits until ONE CU has sixteen wave slots and its LDS free at the same moment -
// r4t: 250-450 us for a 7 us kernel, 1.35 ms per step on the critical path (profiles/r4t_*).  Four waves with half a KB fit anywhere.
#define CO_CH 4
#define CO_RL 64
#define CO_WAVES (CO_CH * CO_RL / 64)
#define CR_COOP_MAX 4096
static std::atomic<int> g_fused_finalize{1};
extern "C" int32_t sgx_bn_set_fused_finalize(int32_t on) {
    g_fused_finalize = on != 0;
    return SGX_OK;
}
extern "C" int32_t sgx_bn_get_fused_finalize(void) { return g_fused_finalize; }
static dim3 fin_grid(const ColSrc& s, int C) { return dim3(sgx_cdiv(C, s.coop ? CO_CH : 64)); }
static dim3 fin_block(const ColSrc& s) { return dim3(s.coop ? CO_CH * CO_RL : 64); }
// channel of this thread, whether it exists, whether this thread writes the channel's results
#define SGX_FIN_THREAD(src, C)                                                                                            \
    if ((src).prio) SGX_WAVE_PRIO(3);                                                                                      \
    const int c = (src).coop ? blockIdx.x * CO_CH + (threadIdx.x % CO_CH) : blockIdx.x * blockDim.x + threadIdx.x;        \
    const bool cok = c < (C);                                                                                            \
    const bool writer = cok && (!(src).coop || threadIdx.x < CO_CH)
__device__ __forceinline__ double colsrc_sum(const ColSrc& s, int plane, int C, int c) {
    double acc = 0.0;
    if (s.d) {
        const double* p = s.d + (long)plane * s.n * C + c;
#pragma unroll 8
        for (int b = 0; b < s.n; ++b) acc += p[(long)b * C];
    } else {
        const float* p = s.f + (long)plane * s.n * C + c;
#pragma unroll 8
        for (int b = 0; b < s.n; ++b) acc += (double)p[(long)b * C];
    }
    return acc;
}
// this lane's rows of P planes, ascending (a fixed order).  Round 5: SIXTEEN rows x P planes of loads in flight (eight for the five-plane
// form) before the first add, then four, then one - the adds stay in ascending row order, so every sum keeps its bits.  During the
// backward pass these kernels run beside the weight-gradient patch kernels of the side stream, which keep the memory system busy: a load
// then takes several microseconds, and with four rows in flight a lane's 64-100 rows were 16-25 dependent rounds - the launches that the
// trace shows at 100-300 us instead of 8 (profiles/r5aa: every one of them overlaps a wpatch_kernel; ~0.5 ms per step on the critical path).
template <int P, typename T>
__device__ __forceinline__ void col_lane_sums(const T* p0, long plane_ld, int n, int C, int rl, double (&acc)[P]) {
    constexpr int U = P <= 2 ? 16 : 8;
    int b = rl;
    for (; b + (U - 1) * CO_RL < n; b += U * CO_RL) {
        T v[U][P];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q) v[u][q] = p0[q * plane_ld + (long)(b + u * CO_RL) * C];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q) acc[q] += (double)v[u][q];
    }
    for (; b + 3 * CO_RL < n; b += 4 * CO_RL) {
        T v[4][P];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q) v[u][q] = p0[q * plane_ld + (long)(b + u * CO_RL) * C];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q) acc[q] += (double)v[u][q];
    }
    for (; b < n; b += CO_RL)
#pragma unroll
        for (int q = 0; q < P; ++q) acc[q] += (double)p0[q * plane_ld + (long)b * C];
}
// column totals of the first P planes; in coop mode EVERY thread of the workgroup must call it (barriers); the totals are valid in the
// writer lanes (SGX_FIN_THREAD) - the only ones that use them
template <int P>
__device__ __forceinline__ void col_totals(const ColSrc& s, int C, int c, bool cok, double (&out)[P]) {
    if (!s.coop) {
#pragma unroll
        for (int q = 0; q < P; ++q) out[q] = cok ? colsrc_sum(s, q, C, c) : 0.0;
        return;
    }
    __shared__ double red[P][CO_WAVES][CO_CH];
    const int cl = threadIdx.x % CO_CH, rl = threadIdx.x / CO_CH;
    double acc[P];
#pragma unroll
    for (int q = 0; q < P; ++q) acc[q] = 0.0;
    if (cok) {
        if (s.d) col_lane_sums<P>(s.d + c, (long)s.n * C, s.n, C, rl, acc);
        else col_lane_sums<P>(s.f + c, (long)s.n * C, s.n, C, rl, acc);
    }
    // lane sums -> totals in a fixed order: the sixteen row lanes of a wave meet by exchange (lane = row lane x CO_CH + channel), the
    // waves' sums through LDS, added in wave order by the writer lanes
#pragma unroll
    for (int q = 0; q < P; ++q) {
#pragma unroll
        for (int m = CO_CH; m < 64; m <<= 1) acc[q] += __shfl_xor(acc[q], m);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < CO_CH)
#pragma unroll
        for (int q = 0; q < P; ++q) red[q][wave][cl] = acc[q];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < P; ++q) {
        double t = 0.0;
        if (rl == 0)
#pragma unroll
            for (int k = 0; k < CO_WAVES; ++k) t += red[q][k][cl];
        out[q] = t;  // valid in the writer lanes (row lane 0) only
    }
}
static int cr_slices(int nblk) {
    if (nblk <= CR_DIRECT) return 0;
    int s = (nblk + 31) / 32;
    return s > CR_MAX_SLICES ? CR_MAX_SLICES : s;
}
extern "C" int64_t sgx_reduce_workspace(int32_t nblk, int32_t C) { return (int64_t)2 * cr_slices(nblk) * C * (int64_t)sizeof(double) + 256; }

template <int PLANES>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* partials, int nblk, int C, int S, int chunk, double* out) {
    __shared__ double red[PLANES][256];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl, s = blockIdx.y;
    const int b0 = s * chunk, b1 = min(nblk, b0 + chunk);
    double acc[PLANES];
#pragma unroll
    for (int p = 0; p < PLANES; ++p) acc[p] = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int b = b0 + rl; b < b1; b += 4) {
#pragma unroll
            for (int p = 0; p < PLANES; ++p) acc[p] += (double)partials[((long)p * nblk + b) * C + c];
        }
    }
#pragma unroll
    for (int p = 0; p < PLANES; ++p) red[p][threadIdx.x] = acc[p];
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int p = 0; p < PLANES; ++p)
            out[((long)p * S + s) * C + c] = ((red[p][cl] + red[p][64 + cl]) + red[p][128 + cl]) + red[p][192 + cl];
    }
}
// -> the source the finalize kernel should read; launches the pre-reduction when it pays
template <int PLANES>
static int32_t col_prereduce(const float* partials, int nblk, int C, void* ws, int64_t ws_bytes, void* stream, ColSrc* src) {
    const int S = cr_slices(nblk);
    if (S == 0) {
        *src = ColSrc{partials, nullptr, nblk, 0, wave_prio_mode() & 1};
        return SGX_OK;
    }
    if (g_fused_finalize && nblk <= CR_COOP_MAX) {
        *src = ColSrc{partials, nullptr, nblk, 1, wave_prio_mode() & 1};
        return SGX_OK;
    }
    if (!ws || ws_bytes < (int64_t)PLANES * S * C * (int64_t)sizeof(double)) SGX_FAIL(SGX_ERR_WORKSPACE, "column reduce: workspace too small (sgx_reduce_workspace)");
    const int chunk = (nblk + S - 1) / S;
    SGX_LAUNCH((colreduce_kernel<PLANES>), dim3(sgx_cdiv(C, 64), S), dim3(256), 0, stream, partials, nblk, C, S, chunk, (double*)ws);
    SGX_CHECK_LAUNCH("colreduce");
    *src = ColSrc{nullptr, (const double*)ws, S, g_fused_finalize ? 1 : 0, wave_prio_mode() & 1};  // the fp64 slices are folded by row lanes as well
    return SGX_OK;
}

__global__ void bn_finalize_kernel(ColSrc src, long M, int C, const float* gamma, const float* beta, float eps,
                                   float momentum, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                   float* scale, float* shift) {
    SGX_FIN_THREAD(src, C);
    double tot[2];
    col_totals<2>(src, C, c, cok, tot);
    if (!writer) return;
    const double s = tot[0], q = tot[1];
    double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    double invstd = 1.0 / sqrt(var + (double)eps);
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    if (running_var) {
        double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unb);
    }
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)invstd;
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float sc = g * (float)invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
}
extern "C" int32_t sgx_bn_finalize(const float* partials, int32_t nblk, int64_t M, int32_t C, const float* gamma, const float* beta,
                                   float eps, float momentum, float* running_mean, float* running_var, float* save_mean,
                                   float* save_invstd, float* scale, float* shift, void* ws, int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(partials && scale && shift && nblk > 0, "bn_finalize: bad args");
    ColSrc src;
    int32_t rc = col_prereduce<2>(partials, nblk, C, ws, ws_bytes, stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(bn_finalize_kernel, fin_grid(src, C), fin_block(src), 0, stream, src, (long)M, C, gamma, beta, eps, momentum,
               running_mean, running_var, save_mean, save_invstd, scale, shift);
    SGX_CHECK_LAUNCH("bn_finalize");
    return SGX_OK;
}

// ---- cross-rank (synchronised) BatchNorm: the per-channel sums leave the library as fp64 [planes][C] so that the host can
// all-reduce them over RCCL (ONE small collective per BN layer and direction), then come back for the finalisation.
template <int P>
__global__ void colsum_f64_kernel(ColSrc src, int C, double* out) {
    SGX_FIN_THREAD(src, C);
    double tot[P];
    col_totals<P>(src, C, c, cok, tot);
    if (writer)
#pragma unroll
        for (int p = 0; p < P; ++p) out[(long)p * C + c] = tot[p];
}
extern "C" int32_t sgx_bn_reduce_sums(const float* partials, int32_t nblk, int32_t C, double* sums, void* ws, int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(partials && sums && nblk > 0, "bn_reduce_sums: bad args");
    ColSrc src;
    int32_t rc = col_prereduce<2>(partials, nblk, C, ws, ws_bytes, stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(colsum_f64_kernel<2>, fin_grid(src, C), fin_block(src), 0, stream, src, C, sums);
    SGX_CHECK_LAUNCH("bn_reduce_sums");
    return SGX_OK;
}
extern "C" int32_t sgx_bn_finalize_sums(const double* sums, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                                        float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift,
                                        void* stream) {
    SGX_CHECK_ARG(sums && scale && shift && M > 0, "bn_finalize_sums: bad args");
    ColSrc src{nullptr, sums, 1, 0};
    SGX_LAUNCH(bn_finalize_kernel, dim3(sgx_cdiv(C, 64)), dim3(64), 0, stream, src, (long)M, C, gamma, beta, eps, momentum, running_mean, running_var,
               save_mean, save_invstd, scale, shift);
    SGX_CHECK_LAUNCH("bn_finalize_sums");
    return SGX_OK;
}

__global__ void bn_eval_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = (float)(1.0 / sqrt((double)rv[c] + (double)eps));
    float sc = (gamma ? gamma[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}
extern "C" int32_t sgx_bn_eval_scale_shift(int32_t C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                           float eps, float* scale, float* shift, void* stream) {
    SGX_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_eval: bad args");
    SGX_LAUNCH(bn_eval_kernel, dim3(sgx_cdiv(C, 64)), dim3(64), 0, stream, C, gamma, beta, running_mean, running_var, eps, scale, shift);
    SGX_CHECK_LAUNCH("bn_eval");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
struct AffineActF {
    const float* x; long x_ld;
    const float* scale; const float* shift;
    const float* r1; long r1_ld; float a1; const float* a1_dev;
    const float* r2; long r2_ld; float a2;
    float* y; long y_ld;
    int act;
    struct In { float4 v, u1, u2; };
    __device__ In load(long r, int c) const {
        In in;
        in.v = sgx_ld4(x + r * x_ld + c);
        in.u1 = r1 ? sgx_ld4(r1 + r * r1_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        in.u2 = r2 ? sgx_ld4(r2 + r * r2_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        return in;
    }
    struct Cst { float4 s, t; float a; };
    __device__ Cst consts(int c) const {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        return Cst{scale ? sgx_ld4(scale + c) : z, scale ? sgx_ld4(shift + c) : z, (r1 && a1_dev) ? a1_dev[0] : a1};
    }
    __device__ void apply(long r, int c, const In& in, const Cst& k, float4& q0, float4& q1) const {
        float4 v = in.v;
        if (scale) {
            const float4 s = k.s, t = k.t;
            v.x = s.x * v.x + t.x; v.y = s.y * v.y + t.y; v.z = s.z * v.z + t.z; v.w = s.w * v.w + t.w;
        }
        if (r1) {
            const float a = k.a;
            const float4 u = in.u1;
            v.x += a * u.x; v.y += a * u.y; v.z += a * u.z; v.w += a * u.w;
        }
        if (r2) {
            const float4 u = in.u2;
            v.x += a2 * u.x; v.y += a2 * u.y; v.z += a2 * u.z; v.w += a2 * u.w;
        }
        q0.x += v.x; q0.y += v.y; q0.z += v.z; q0.w += v.w;
        q1.x += v.x * v.x; q1.y += v.y * v.y; q1.z += v.z * v.z; q1.w += v.w * v.w;
        float4 o = make_float4(sgx_act(v.x, act), sgx_act(v.y, act), sgx_act(v.z, act), sgx_act(v.w, act));
        sgx_st4(y + r * y_ld + c, o);
    }
};
extern "C" int32_t sgx_affine_act_fwd(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* r1,
                                      int64_t r1_ld, float a1, const float* a1_dev, const float* r2, int64_t r2_ld, float a2, float* y,
                                      int64_t y_ld, int64_t M, int32_t C, int32_t act, float* partials, void* stream) {
    SGX_CHECK_ARG(x && y, "affine_act: null pointer");
    SGX_CHECK_ARG((scale == nullptr) == (shift == nullptr), "affine_act: scale and shift go together");
    AffineActF f{x, x_ld, scale, shift, r1, r1_ld, a1, a1_dev, r2, r2_ld, a2, y, y_ld, act};
    return run_sweep<AffineActF, 2>(f, M, C, partials, stream, "affine_act");
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float bn_masked(float dy, float x, float s, float t, int act) {
    return act == SGX_ACT_NONE ? dy : dy * sgx_act_grad(s * x + t, act);
}
struct BnBwdReduceF {
    const float* dy; long dy_ld; const float* x; long x_ld; const float* scale; const float* shift; const float* mean; int act;
    struct In { float4 d, v; };
    __device__ In load(long r, int c) const { return In{sgx_ld4(dy + r * dy_ld + c), sgx_ld4(x + r * x_ld + c)}; }
    struct Cst { float4 s, t, mu; };
    __device__ Cst consts(int c) const { return Cst{sgx_ld4(scale + c), sgx_ld4(shift + c), sgx_ld4(mean + c)}; }
    __device__ void apply(long, int, const In& in, const Cst& k, float4& q0, float4& q1) const {
        const float4 d = in.d, v = in.v;
        const float4 s = k.s, t = k.t, mu = k.mu;
        float gx = bn_masked(d.x, v.x, s.x, t.x, act), gy = bn_masked(d.y, v.y, s.y, t.y, act);
        float gz = bn_masked(d.z, v.z, s.z, t.z, act), gw = bn_masked(d.w, v.w, s.w, t.w, act);
        q0.x += gx; q0.y += gy; q0.z += gz; q0.w += gw;
        // centred second moment: sum g*(x - mean) (no large-term cancellation later, cf. ATen's batch_norm backward)
        q1.x += gx * (v.x - mu.x); q1.y += gy * (v.y - mu.y); q1.z += gz * (v.z - mu.z); q1.w += gw * (v.w - mu.w);
    }
};
extern "C" int32_t sgx_bn_bwd_reduce(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* scale, const float* shift,
                                     const float* save_mean, int64_t M, int32_t C, int32_t act, float* partials, void* stream) {
    SGX_CHECK_ARG(dy && x && scale && shift && save_mean && partials, "bn_bwd_reduce: null pointer");
    BnBwdReduceF f{dy, dy_ld, x, x_ld, scale, shift, save_mean, act};
    return run_sweep<BnBwdReduceF, 2>(f, M, C, partials, stream, "bn_bwd_reduce");
}

// src: the sums that define dx (all ranks' when BatchNorm is synchronised); loc: this rank's own sums, which is what dgamma / dbeta
// accumulate (the data-parallel gradient exchange adds the other ranks' later) - loc.n == 0 means "same as src"
__global__ void bn_bwd_finalize_kernel(ColSrc src, ColSrc loc, long M, int C, const float* gamma, const float* save_mean,
                                       const float* save_invstd, float* dgamma, float* dbeta, float* coef) {
    SGX_FIN_THREAD(src, C);
    double tot[2];
    col_totals<2>(src, C, c, cok, tot);
    const double sg = tot[0], sgx = tot[1];
    // (loc: the synchronised-BatchNorm path, one fp64 row per plane, never cooperative - read by the writer lanes only)
    if (!writer) return;
    double mean = save_mean[c], invstd = save_invstd[c], g = gamma ? (double)gamma[c] : 1.0;
    const double lsg = loc.n ? colsrc_sum(loc, 0, C, c) : sg, lsgx = loc.n ? colsrc_sum(loc, 1, C, c) : sgx;
    double sgxhat = invstd * lsgx;  // sgx is already centred: sum g*(x - mean)
    if (dgamma) dgamma[c] += (float)sgxhat;
    if (dbeta) dbeta[c] += (float)lsg;
    // dx = c1 * ((g - mg) - (x - mean) * k): differences first, then the scale - the order ATen's CPU kernel uses, so a
    // nearly constant upstream gradient does not lose its small remainder to cancellation between large products
    coef[c] = (float)(g * invstd);
    // The mean of g as TWO floats (hi + lo): (g - mean g) must sum to zero over the pixels to ~2^-48, not 2^-24.  With one float the
    // rounding of the mean is a constant per-channel offset in every element of dx; the weight gradient of the producing convolution sums
    // dx against activations whose per-channel mean need not be small, so M * offset * mean(x) competes with a sum that grows like
    // sqrt(M): measured (r3b, 32 x 640^2, activations at mean 4) as a 60x larger distance from fp64 than ATen's CPU kernel, which forms
    // (g - mean g) in double.  The x-hat coefficient and the scale are relative factors: their rounding is harmless.
    const double mgd = sg / (double)M;
    const float mg_hi = (float)mgd;
    coef[C + c] = mg_hi;
    coef[2 * C + c] = (float)(invstd * invstd * sgx / (double)M);
    coef[3 * C + c] = (float)mean;
    coef[4 * C + c] = (float)(mgd - (double)mg_hi);
}
extern "C" int32_t sgx_bn_bwd_finalize(const float* partials, int32_t nblk, int64_t M, int32_t C, const float* gamma, const float* save_mean,
                                       const float* save_invstd, float* dgamma, float* dbeta, float* coef, void* ws, int64_t ws_bytes,
                                       void* stream) {
    SGX_CHECK_ARG(partials && save_mean && save_invstd && coef, "bn_bwd_finalize: null pointer");
    ColSrc src;
    int32_t rc = col_prereduce<2>(partials, nblk, C, ws, ws_bytes, stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(bn_bwd_finalize_kernel, fin_grid(src, C), fin_block(src), 0, stream, src, ColSrc{nullptr, nullptr, 0, 0, 0}, (long)M, C, gamma, save_mean,
               save_invstd, dgamma, dbeta, coef);
    SGX_CHECK_LAUNCH("bn_bwd_finalize");
    return SGX_OK;
}
extern "C" int32_t sgx_bn_bwd_finalize_sums(const double* local_sums, const double* global_sums, int64_t M_total, int32_t C, const float* gamma,
                                            const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef, void* stream) {
    SGX_CHECK_ARG(local_sums && global_sums && save_mean && save_invstd && coef && M_total > 0, "bn_bwd_finalize_sums: bad args");
    SGX_LAUNCH(bn_bwd_finalize_kernel, dim3(sgx_cdiv(C, 64)), dim3(64), 0, stream, ColSrc{nullptr, global_sums, 1, 0, 0}, ColSrc{nullptr, local_sums, 1, 0, 0},
               (long)M_total, C, gamma, save_mean, save_invstd, dgamma, dbeta, coef);
    SGX_CHECK_LAUNCH("bn_bwd_finalize_sums");
    return SGX_OK;
}

struct BnBwdApplyF {
    const float* dy; long dy_ld; const float* x; long x_ld; const float* scale; const float* shift; const float* coef; int C;
    float* dx; long dx_ld; float* g_out; long g_ld; int act;
    struct In { float4 d, v; };
    __device__ In load(long r, int c) const { return In{sgx_ld4(dy + r * dy_ld + c), sgx_ld4(x + r * x_ld + c)}; }
    struct Cst { float4 s, t, c1, mg, k, mu, ml; };
    __device__ Cst consts(int c) const {
        return Cst{sgx_ld4(scale + c), sgx_ld4(shift + c), sgx_ld4(coef + c), sgx_ld4(coef + C + c), sgx_ld4(coef + 2 * C + c), sgx_ld4(coef + 3 * C + c),
                   sgx_ld4(coef + 4 * C + c)};
    }
    __device__ void apply(long r, int c, const In& in, const Cst& kc, float4& q0, float4& q1) const {
        (void)q0; (void)q1;
        const float4 d = in.d, v = in.v;
        const float4 s = kc.s, t = kc.t;
        const float4 c1 = kc.c1, mg = kc.mg, k = kc.k, mu = kc.mu, ml = kc.ml;
        float4 g = make_float4(bn_masked(d.x, v.x, s.x, t.x, act), bn_masked(d.y, v.y, s.y, t.y, act),
                               bn_masked(d.z, v.z, s.z, t.z, act), bn_masked(d.w, v.w, s.w, t.w, act));
        float4 o = make_float4(c1.x * (((g.x - mg.x) - ml.x) - (v.x - mu.x) * k.x), c1.y * (((g.y - mg.y) - ml.y) - (v.y - mu.y) * k.y),
                               c1.z * (((g.z - mg.z) - ml.z) - (v.z - mu.z) * k.z), c1.w * (((g.w - mg.w) - ml.w) - (v.w - mu.w) * k.w));
        sgx_st4(dx + r * dx_ld + c, o);
        if (g_out) sgx_st4(g_out + r * g_ld + c, g);
    }
};
extern "C" int32_t sgx_bn_bwd_apply(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* scale, const float* shift,
                                    const float* coef, float* dx, int64_t dx_ld, float* g_out, int64_t g_ld, int64_t M, int32_t C,
                                    int32_t act, void* stream) {
    SGX_CHECK_ARG(dy && x && scale && shift && coef && dx, "bn_bwd_apply: null pointer");
    BnBwdApplyF f{dy, dy_ld, x, x_ld, scale, shift, coef, C, dx, dx_ld, g_out, g_ld, act};
    return run_sweep<BnBwdApplyF, 0>(f, M, C, nullptr, stream, "bn_bwd_apply");
}

// ---------------------------------------------------------------------------------------------
// out (+)= scale * <a, b> over NHWC views: the gradient of the learnable scalars (the bottlenecks' alpha, yolo_stages.py:61-63: d alpha =
// <x, dz>).  Those sums cancel ~1e3x, so every lane accumulates with an error-free transformation (TwoSum: running sum + the rounding
// errors it dropped), the workgroup folds its 256 (sum, error) pairs in fp64 into ONE value, and a single-workgroup second stage adds the
// <= 6 144 workgroup values in fp64 in a fixed order (deterministic; the earlier form summed 2 x nblk x C fp32 partials with one workgroup:
// 68 us per call at 160x160x96, r2k).
template <int DUMMY>
__global__ __launch_bounds__(SW_THREADS) void dot_kernel(const float* a, long a_ld, const float* b, long b_ld, SweepGeom g, double* out) {
    __shared__ double red[SW_THREADS];
    const int tid = threadIdx.x;
    const int cg = tid % g.CG, rl = tid / g.CG;
    const int c4 = blockIdx.y * g.CG + cg;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f), e = s;
    auto add = [](float& s, float& e, float p) {
        const float t = s + p, bp = t - s;
        e += (s - (t - bp)) + (p - bp);
        s = t;
    };
    if (rl < g.RL && c4 < g.C4) {
        const int c = c4 * 4;
        long r0 = (long)blockIdx.x * g.rows_per_blk, r1 = r0 + g.rows_per_blk;
        if (r1 > g.M) r1 = g.M;
        const long st = g.RL;
        long r = r0 + rl;
        for (; r + st < r1; r += 2 * st) {
            const float4 u0 = sgx_ld4(a + r * a_ld + c), v0 = sgx_ld4(b + r * b_ld + c), u1 = sgx_ld4(a + (r + st) * a_ld + c), v1 = sgx_ld4(b + (r + st) * b_ld + c);
            add(s.x, e.x, u0.x * v0.x); add(s.y, e.y, u0.y * v0.y); add(s.z, e.z, u0.z * v0.z); add(s.w, e.w, u0.w * v0.w);
            add(s.x, e.x, u1.x * v1.x); add(s.y, e.y, u1.y * v1.y); add(s.z, e.z, u1.z * v1.z); add(s.w, e.w, u1.w * v1.w);
        }
        for (; r < r1; r += st) {
            const float4 u0 = sgx_ld4(a + r * a_ld + c), v0 = sgx_ld4(b + r * b_ld + c);
            add(s.x, e.x, u0.x * v0.x); add(s.y, e.y, u0.y * v0.y); add(s.z, e.z, u0.z * v0.z); add(s.w, e.w, u0.w * v0.w);
        }
    }
    red[tid] = (((double)s.x + (double)s.y) + ((double)s.z + (double)s.w)) + (((double)e.x + (double)e.y) + ((double)e.z + (double)e.w));
    __syncthreads();
    for (int w = SW_THREADS / 2; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
    }
    if (tid == 0) out[(long)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void dot_final_kernel(const double* parts, int n, float scale, float* out, int accumulate) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += parts[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float v = (float)(red[0] * (double)scale);
        out[0] = accumulate ? out[0] + v : v;
    }
}
extern "C" int64_t sgx_dot_workspace(int64_t M, int32_t C) {
    SweepGeom g = sweep_geom(M, C > 0 ? C : 4);
    return (int64_t)g.nblk * g.ctiles * (int64_t)sizeof(double) + 256;
}
extern "C" int32_t sgx_dot(const float* a, int64_t a_ld, const float* b, int64_t b_ld, int64_t M, int32_t C, float scale, float* out,
                           int32_t accumulate, void* ws, int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(a && b && out && ws, "dot: null pointer");
    SGX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "dot: need M>0 and C%%4==0 (C=%d)", C);
    if (ws_bytes < sgx_dot_workspace(M, C)) SGX_FAIL(SGX_ERR_WORKSPACE, "dot: workspace too small (sgx_dot_workspace)");
    SweepGeom g = sweep_geom(M, C);
    SGX_LAUNCH((dot_kernel<0>), dim3(g.nblk, g.ctiles), dim3(SW_THREADS), 0, stream, a, (long)a_ld, b, (long)b_ld, g, (double*)ws);
    SGX_CHECK_LAUNCH("dot");
    SGX_LAUNCH(dot_final_kernel, dim3(1), dim3(256), 0, stream, (const double*)ws, g.nblk * g.ctiles, scale, out, accumulate);
    SGX_CHECK_LAUNCH("dot_final");
    return SGX_OK;
}

struct AxpyF {
    const float* x; long x_ld; float a; const float* a_dev; float* y; long y_ld; int accumulate;
    struct In { float4 v, u; };
    __device__ In load(long r, int c) const {
        return In{sgx_ld4(x + r * x_ld + c), accumulate ? sgx_ld4(y + r * y_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f)};
    }
    struct Cst { float s; };
    __device__ Cst consts(int) const { return Cst{a_dev ? a_dev[0] : a}; }
    __device__ void apply(long r, int c, const In& in, const Cst& k, float4& q0, float4& q1) const {
        (void)q0; (void)q1;
        const float s = k.s;
        const float4 v = in.v;
        float4 o = make_float4(s * v.x, s * v.y, s * v.z, s * v.w);
        if (accumulate) {
            const float4 u = in.u;
            o.x += u.x; o.y += u.y; o.z += u.z; o.w += u.w;
        }
        sgx_st4(y + r * y_ld + c, o);
    }
};
extern "C" int32_t sgx_axpy(const float* x, int64_t x_ld, float a, const float* a_dev, float* y, int64_t y_ld, int64_t M, int32_t C,
                            int32_t accumulate, void* stream) {
    SGX_CHECK_ARG(x && y, "axpy: null pointer");
    AxpyF f{x, x_ld, a, a_dev, y, y_ld, accumulate};
    return run_sweep<AxpyF, 0>(f, M, C, nullptr, stream, "axpy");
}

// g = dy where the (post-activation) output y is positive: backward of a ReLU that sits AFTER a residual add
// (classification_models/resnet.py:43-50, 72-84: out = relu(bn(conv) + shortcut)), where the mask is not a function of one BN output.
struct ReluBwdF {
    const float* dy; long dy_ld; const float* y; long y_ld; float* g; long g_ld;
    struct In { float4 d, v; };
    __device__ In load(long r, int c) const { return In{sgx_ld4(dy + r * dy_ld + c), sgx_ld4(y + r * y_ld + c)}; }
    struct Cst {};
    __device__ Cst consts(int) const { return Cst{}; }
    __device__ void apply(long r, int c, const In& in, const Cst&, float4& q0, float4& q1) const {
        (void)q0; (void)q1;
        const float4 d = in.d, v = in.v;
        sgx_st4(g + r * g_ld + c, make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f));
    }
};
extern "C" int32_t sgx_relu_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, float* g, int64_t g_ld, int64_t M, int32_t C,
                                void* stream) {
    SGX_CHECK_ARG(dy && y && g, "relu_bwd: null pointer");
    ReluBwdF f{dy, dy_ld, y, y_ld, g, g_ld};
    return run_sweep<ReluBwdF, 0>(f, M, C, nullptr, stream, "relu_bwd");
}

// The same mask AND the BatchNorm-backward reduce of the layer underneath in one sweep (round 6): out = relu(bn(conv) + shortcut) hands
// g = dy * (y > 0) to bn's backward, whose reduce sweep would read g and the saved conv output x again - here the two per-channel sums
// (sum g, sum g * (x - mean): the partials sgx_bn_bwd_reduce leaves for act = none) come out of the pass that writes g.
struct ReluBwdBnReduceF {
    const float* dy; long dy_ld; const float* y; long y_ld; const float* x; long x_ld; const float* mean; float* g; long g_ld;
    struct In { float4 d, v, u; };
    __device__ In load(long r, int c) const { return In{sgx_ld4(dy + r * dy_ld + c), sgx_ld4(y + r * y_ld + c), sgx_ld4(x + r * x_ld + c)}; }
    struct Cst { float4 mu; };
    __device__ Cst consts(int c) const { return Cst{sgx_ld4(mean + c)}; }
    __device__ void apply(long r, int c, const In& in, const Cst& k, float4& q0, float4& q1) const {
        const float4 d = in.d, v = in.v, u = in.u, mu = k.mu;
        const float4 o = make_float4(v.x > 0.f ? d.x : 0.f, v.y > 0.f ? d.y : 0.f, v.z > 0.f ? d.z : 0.f, v.w > 0.f ? d.w : 0.f);
        sgx_st4(g + r * g_ld + c, o);
        q0.x += o.x; q0.y += o.y; q0.z += o.z; q0.w += o.w;
        q1.x += o.x * (u.x - mu.x); q1.y += o.y * (u.y - mu.y); q1.z += o.z * (u.z - mu.z); q1.w += o.w * (u.w - mu.w);
    }
};
extern "C" int32_t sgx_relu_bwd_bn_reduce(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, const float* x, int64_t x_ld,
                                          const float* save_mean, float* g, int64_t g_ld, int64_t M, int32_t C, float* partials, void* stream) {
    SGX_CHECK_ARG(dy && y && x && save_mean && g && partials, "relu_bwd_bn_reduce: null pointer");
    ReluBwdBnReduceF f{dy, dy_ld, y, y_ld, x, x_ld, save_mean, g, g_ld};
    return run_sweep<ReluBwdBnReduceF, 2>(f, M, C, partials, stream, "relu_bwd_bn_reduce");
}

// RepVGG-style two-branch BatchNorm sum + activation (+ post-activation residual) in one sweep, and the gradient through the
// activation (the pre-activation is recomputed from the two saved conv outputs; nothing else is stored).
struct DualAffineF {
    const float* x1; long x1_ld; const float* s1; const float* t1;
    const float* x2; long x2_ld; const float* s2; const float* t2;
    const float* r; long r_ld; float* y; long y_ld; int act;
    float r_scale; const float* r_scale_dev;  // post-activation residual: y = act(..) + r_scale * r_scale_dev[0] * r
    struct In { float4 a, b, u; };
    __device__ In load(long row, int c) const {
        In in;
        in.a = sgx_ld4(x1 + row * x1_ld + c);
        in.b = x2 ? sgx_ld4(x2 + row * x2_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        in.u = r ? sgx_ld4(r + row * r_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        return in;
    }
    struct Cst { float4 s, t, p, q; float sc; };
    __device__ Cst consts(int c) const {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        return Cst{sgx_ld4(s1 + c), sgx_ld4(t1 + c), x2 ? sgx_ld4(s2 + c) : z, x2 ? sgx_ld4(t2 + c) : z, r_scale * ((r && r_scale_dev) ? r_scale_dev[0] : 1.f)};
    }
    __device__ float4 pre(const Cst& k, const float4& a, const float4& b) const {
        const float4 s = k.s, t = k.t;
        float4 v = make_float4(s.x * a.x + t.x, s.y * a.y + t.y, s.z * a.z + t.z, s.w * a.w + t.w);
        if (x2) {
            const float4 p = k.p, q = k.q;
            v.x += p.x * b.x + q.x; v.y += p.y * b.y + q.y; v.z += p.z * b.z + q.z; v.w += p.w * b.w + q.w;
        }
        return v;
    }
    __device__ void apply(long row, int c, const In& in, const Cst& k, float4& q0, float4& q1) const {
        (void)q0; (void)q1;
        float4 v = pre(k, in.a, in.b);
        float4 o = make_float4(sgx_act(v.x, act), sgx_act(v.y, act), sgx_act(v.z, act), sgx_act(v.w, act));
        if (r) {
            const float sc = k.sc;
            o.x += sc * in.u.x; o.y += sc * in.u.y; o.z += sc * in.u.z; o.w += sc * in.u.w;
        }
        sgx_st4(y + row * y_ld + c, o);
    }
};
extern "C" int32_t sgx_dual_affine_act_fwd(const float* x1, int64_t x1_ld, const float* s1, const float* t1, const float* x2, int64_t x2_ld,
                                           const float* s2, const float* t2, const float* r, int64_t r_ld, float r_scale, const float* r_scale_dev,
                                           float* y, int64_t y_ld, int64_t M, int32_t C, int32_t act, void* stream) {
    SGX_CHECK_ARG(x1 && s1 && t1 && y, "dual_affine_act_fwd: null pointer");
    SGX_CHECK_ARG(!x2 || (s2 && t2), "dual_affine_act_fwd: second branch needs scale and shift");
    DualAffineF f{x1, x1_ld, s1, t1, x2, x2_ld, s2, t2, r, r_ld, y, y_ld, act, r_scale, r_scale_dev};
    return run_sweep<DualAffineF, 0>(f, M, C, nullptr, stream, "dual_affine_act_fwd");
}
struct DualAffineBwdF {
    DualAffineF p; const float* dy; long dy_ld; float* g; long g_ld;
    struct In { float4 a, b, d; };
    __device__ In load(long row, int c) const {
        In in;
        in.a = sgx_ld4(p.x1 + row * p.x1_ld + c);
        in.b = p.x2 ? sgx_ld4(p.x2 + row * p.x2_ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        in.d = sgx_ld4(dy + row * dy_ld + c);
        return in;
    }
    typedef DualAffineF::Cst Cst;
    __device__ Cst consts(int c) const { return p.consts(c); }
    __device__ void apply(long row, int c, const In& in, const Cst& k, float4& q0, float4& q1) const {
        (void)q0; (void)q1;
        float4 v = p.pre(k, in.a, in.b);
        sgx_st4(g + row * g_ld + c, make_float4(in.d.x * sgx_act_grad(v.x, p.act), in.d.y * sgx_act_grad(v.y, p.act),
                                                 in.d.z * sgx_act_grad(v.z, p.act), in.d.w * sgx_act_grad(v.w, p.act)));
    }
};
extern "C" int32_t sgx_dual_affine_act_bwd(const float* dy, int64_t dy_ld, const float* x1, int64_t x1_ld, const float* s1, const float* t1,
                                           const float* x2, int64_t x2_ld, const float* s2, const float* t2, float* g, int64_t g_ld, int64_t M,
                                           int32_t C, int32_t act, void* stream) {
    SGX_CHECK_ARG(dy && x1 && s1 && t1 && g, "dual_affine_act_bwd: null pointer");
    SGX_CHECK_ARG(!x2 || (s2 && t2), "dual_affine_act_bwd: second branch needs scale and shift");
    DualAffineBwdF f{DualAffineF{x1, x1_ld, s1, t1, x2, x2_ld, s2, t2, nullptr, 0, nullptr, 0, act, 1.f, nullptr}, dy, dy_ld, g, g_ld};
    return run_sweep<DualAffineBwdF, 0>(f, M, C, nullptr, stream, "dual_affine_act_bwd");
}

// ---------------------------------------------------------------------------------------------
// QARepVGG block, training form, on the two-output convolution (conv.hip: sgx_conv2d_fwd_dual):
//     y = conv3x3(x)          u = conv1x1(x; alpha * W1 + I) + b1          (identity branch folded into the 1x1 filter)
//     s = bn3(y) + u          out = act(post_bn(s))
// Both BatchNorms are finalised from the FIVE per-channel moments the convolution epilogue leaves (sum y, y^2, u0, u0^2, y*u0, u0 = u - b1):
// s is an affine function of (y, u) per channel, so mean / variance of s follow from the moments of (y, u) - no pass over s for its
// statistics, and s itself is never written: the forward is ONE sweep, out = act(a*y + scp*u + c) (sgx_dual_affine_act_fwd with the
// coefficient rows cf), instead of two sweeps and a residual read.  Backward: ONE reduce sweep (4 sums) + ONE apply sweep that writes the
// upstream gradients of BOTH convolutions, instead of two reduce and two apply sweeps:
//     g = dout * act'(z),  ds = cp * ((g - mean g) - shat * mean(g * shat)),  dy = c3 * (ds - mean ds - yhat * mean(ds * yhat))
// with mean ds = 0 (a BatchNorm's input gradient sums to zero per channel) and mean(ds * yhat) = cp * (mean(g * yhat) - mean(g * shat) *
// mean(shat * yhat)): every mean comes out of the one reduce sweep.  Reference arithmetic: modules/qarepvgg_block.py:184-204.
// ---------------------------------------------------------------------------------------------
template <typename F, int NQ>
__global__ __launch_bounds__(SW_THREADS) void sweepq_kernel(F f, SweepGeom g, float* partials) {
    __shared__ float4 red[NQ][SW_THREADS];
    if (g.prio) SGX_WAVE_PRIO(2);
    const int tid = threadIdx.x;
    const int cg = tid % g.CG, rl = tid / g.CG;
    const int c4 = blockIdx.y * g.CG + cg;
    const bool live = (rl < g.RL) && (c4 < g.C4);
    const int c = c4 * 4;
    float4 q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
        long r0 = (long)blockIdx.x * g.rows_per_blk;
        long r1 = r0 + g.rows_per_blk;
        if (r1 > g.M) r1 = g.M;
        long r = r0 + rl;
        const long st = g.RL;
        const typename F::Cst k = f.consts(c);
        for (; r + st < r1; r += 2 * st) {  // three tensors per row: two rows of loads in flight per lane
            typename F::In i0 = f.load(r, c), i1 = f.load(r + st, c);
            f.apply(r, c, i0, k, q);
            f.apply(r + st, c, i1, k, q);
        }
        for (; r < r1; r += st) {
            typename F::In i0 = f.load(r, c);
            f.apply(r, c, i0, k, q);
        }
    }
    if (!partials) return;  // no reduction (apply sweeps): uniform across the workgroup
#pragma unroll
    for (int k = 0; k < NQ; ++k) red[k][tid] = q[k];
    __syncthreads();
    if (live && rl == 0) {
#pragma unroll
        for (int k = 0; k < NQ; ++k) {
            double t[4] = {0.0, 0.0, 0.0, 0.0};  // (lane sums meet in double: see sweep_kernel)
            for (int j = 0; j < g.RL; ++j) {
                const float4 a = red[k][j * g.CG + cg];
                t[0] += a.x; t[1] += a.y; t[2] += a.z; t[3] += a.w;
            }
            sgx_st4(partials + ((long)k * g.nblk + blockIdx.x) * g.C + c, make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]));
        }
    }
}

extern "C" int64_t sgx_qarep_workspace(int32_t nblk, int32_t C) { return (int64_t)5 * cr_slices(nblk) * C * (int64_t)sizeof(double) + 256; }

// sv rows: 0 mean3, 1 invstd3, 2 scale3, 3 shift3, 4 mean_s, 5 invstd_p, 6 scale_p, 7 shift_p;  cf rows: 0 a = scale_p * scale3, 1 c = scale_p * shift3 +
// shift_p, 2 scale_p, 3 zeros  (out = act(cf0 * y + cf1 + cf2 * u + cf3): the operand rows of sgx_dual_affine_act_fwd)
__global__ void qarep_fwd_finalize_kernel(ColSrc src, long M, int C, const float* bias1, const float* g3, const float* b3, float eps3, float mom3,
                                          float* rm3, float* rv3, const float* gp, const float* bp, float epsp, float momp, float* rmp, float* rvp,
                                          float* cf, float* sv) {
    SGX_FIN_THREAD(src, C);
    double tot[5];
    col_totals<5>(src, C, c, cok, tot);
    if (!writer) return;
    const double Sy = tot[0], Syy = tot[1], Su = tot[2], Suu = tot[3], Syu = tot[4];
    const double m = (double)M, unb = M > 1 ? m / (m - 1.0) : 1.0;
    const double mean3 = Sy / m;
    double var3 = Syy / m - mean3 * mean3;
    if (var3 < 0.0) var3 = 0.0;
    const double invstd3 = 1.0 / sqrt(var3 + (double)eps3);
    const float sc3 = (g3 ? g3[c] : 1.f) * (float)invstd3;  // the same fp32 roundings as bn_finalize_kernel
    const float sh3 = (b3 ? b3[c] : 0.f) - (float)mean3 * sc3;
    const double mu0 = Su / m, bias = bias1 ? (double)bias1[c] : 0.0;
    double varu = Suu / m - mu0 * mu0;
    if (varu < 0.0) varu = 0.0;
    const double cov = Syu / m - mean3 * mu0;
    // s = sc3 * y + sh3 + u0 + bias, per channel
    const double mean_s = (double)sc3 * mean3 + (double)sh3 + mu0 + bias;
    double var_s = (double)sc3 * (double)sc3 * var3 + 2.0 * (double)sc3 * cov + varu;
    if (var_s < 0.0) var_s = 0.0;
    const double invstdp = 1.0 / sqrt(var_s + (double)epsp);
    const float scp = (gp ? gp[c] : 1.f) * (float)invstdp;
    const float shp = (bp ? bp[c] : 0.f) - (float)mean_s * scp;
    if (rm3) rm3[c] = (float)((1.0 - mom3) * (double)rm3[c] + mom3 * mean3);
    if (rv3) rv3[c] = (float)((1.0 - mom3) * (double)rv3[c] + mom3 * var3 * unb);
    if (rmp) rmp[c] = (float)((1.0 - momp) * (double)rmp[c] + momp * mean_s);
    if (rvp) rvp[c] = (float)((1.0 - momp) * (double)rvp[c] + momp * var_s * unb);
    cf[c] = scp * sc3;
    cf[C + c] = scp * sh3 + shp;
    cf[2 * C + c] = scp;
    cf[3 * C + c] = 0.f;
    sv[c] = (float)mean3; sv[C + c] = (float)invstd3; sv[2 * C + c] = sc3; sv[3 * C + c] = sh3;
    sv[4 * C + c] = (float)mean_s; sv[5 * C + c] = (float)invstdp; sv[6 * C + c] = scp; sv[7 * C + c] = shp;
}
extern "C" int32_t sgx_qarep_fwd_finalize(const float* stat5, int32_t nblk, int64_t M, int32_t C, const float* bias1, const float* gamma3,
                                          const float* beta3, float eps3, float mom3, float* rmean3, float* rvar3, const float* gammap,
                                          const float* betap, float epsp, float momp, float* rmeanp, float* rvarp, float* cf, float* sv, void* ws,
                                          int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(stat5 && cf && sv && nblk > 0 && M > 0, "qarep_fwd_finalize: bad args");
    ColSrc src;
    int32_t rc = col_prereduce<5>(stat5, nblk, C, ws, ws_bytes, stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(qarep_fwd_finalize_kernel, fin_grid(src, C), fin_block(src), 0, stream, src, (long)M, C, bias1, gamma3, beta3, eps3, mom3, rmean3, rvar3,
               gammap, betap, epsp, momp, rmeanp, rvarp, cf, sv);
    SGX_CHECK_LAUNCH("qarep_fwd_finalize");
    return SGX_OK;
}

struct QarepBwdIn { float4 d, y, u; };
struct QarepBwdBase {
    const float* dout; long d_ld; const float* y; long y_ld; const float* u; long u_ld; const float* cf; const float* sv; int C; int act;
    __device__ QarepBwdIn load(long r, int c) const { return QarepBwdIn{sgx_ld4(dout + r * d_ld + c), sgx_ld4(y + r * y_ld + c), sgx_ld4(u + r * u_ld + c)}; }
    // masked upstream gradient g and the centred s, y of one float4 (the pre-activation with the forward sweep's own roundings)
    struct Cst { float4 a, cc, b, t0, m3, s3, h3, ms; };
    __device__ Cst consts(int c) const {
        return Cst{sgx_ld4(cf + c), sgx_ld4(cf + C + c), sgx_ld4(cf + 2 * C + c), sgx_ld4(cf + 3 * C + c),
                   sgx_ld4(sv + c), sgx_ld4(sv + 2 * C + c), sgx_ld4(sv + 3 * C + c), sgx_ld4(sv + 4 * C + c)};
    }
    __device__ void terms(const Cst& k, const QarepBwdIn& in, float4& g, float4& sc, float4& yc) const {
        const float4 a = k.a, cc = k.cc, b = k.b, t0 = k.t0;
        float4 z = make_float4(a.x * in.y.x + cc.x, a.y * in.y.y + cc.y, a.z * in.y.z + cc.z, a.w * in.y.w + cc.w);
        z.x += b.x * in.u.x + t0.x; z.y += b.y * in.u.y + t0.y; z.z += b.z * in.u.z + t0.z; z.w += b.w * in.u.w + t0.w;
        g = make_float4(in.d.x * sgx_act_grad(z.x, act), in.d.y * sgx_act_grad(z.y, act), in.d.z * sgx_act_grad(z.z, act), in.d.w * sgx_act_grad(z.w, act));
        const float4 m3 = k.m3, s3 = k.s3, h3 = k.h3, ms = k.ms;
        // s = bn3(y) + u as the reference forms it (qarepvgg_block.py:197-202), then centred
        sc = make_float4((s3.x * in.y.x + h3.x + in.u.x) - ms.x, (s3.y * in.y.y + h3.y + in.u.y) - ms.y, (s3.z * in.y.z + h3.z + in.u.z) - ms.z,
                         (s3.w * in.y.w + h3.w + in.u.w) - ms.w);
        yc = make_float4(in.y.x - m3.x, in.y.y - m3.y, in.y.z - m3.z, in.y.w - m3.w);
    }
};
struct QarepBwdReduceF {
    QarepBwdBase b;
    typedef QarepBwdIn In;
    __device__ In load(long r, int c) const { return b.load(r, c); }
    typedef QarepBwdBase::Cst Cst;
    __device__ Cst consts(int c) const { return b.consts(c); }
    __device__ void apply(long, int, const In& in, const Cst& k, float4 (&q)[4]) const {
        float4 g, sc, yc;
        b.terms(k, in, g, sc, yc);
        q[0].x += g.x; q[0].y += g.y; q[0].z += g.z; q[0].w += g.w;
        q[1].x += g.x * sc.x; q[1].y += g.y * sc.y; q[1].z += g.z * sc.z; q[1].w += g.w * sc.w;
        q[2].x += g.x * yc.x; q[2].y += g.y * yc.y; q[2].z += g.z * yc.z; q[2].w += g.w * yc.w;
        q[3].x += sc.x * yc.x; q[3].y += sc.y * yc.y; q[3].z += sc.z * yc.z; q[3].w += sc.w * yc.w;
    }
};
// partials4: [4][sgx_stats_blocks(M)][C] = sum g, g*(s - mean_s), g*(y - mean3), (s - mean_s)*(y - mean3)
extern "C" int32_t sgx_qarep_bwd_reduce(const float* dout, int64_t d_ld, const float* y, int64_t y_ld, const float* u, int64_t u_ld, const float* cf,
                                        const float* sv, int64_t M, int32_t C, int32_t act, float* partials4, void* stream) {
    SGX_CHECK_ARG(dout && y && u && cf && sv && partials4, "qarep_bwd_reduce: null pointer");
    SGX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "qarep_bwd_reduce: need M>0 and C%%4==0 (C=%d)", C);
    QarepBwdReduceF f{QarepBwdBase{dout, d_ld, y, y_ld, u, u_ld, cf, sv, C, act}};
    SweepGeom g = sweep_geom(M, C);
    SGX_LAUNCH((sweepq_kernel<QarepBwdReduceF, 4>), dim3(g.nblk, g.ctiles), dim3(SW_THREADS), 0, stream, f, g, partials4);
    SGX_CHECK_LAUNCH("qarep_bwd_reduce");
    return SGX_OK;
}
// cb rows: 0 cp = gamma_p * invstd_p, 1 mean g, 2 kp = invstd_p^2 * mean(g * (s - mean_s)), 3 c3 = gamma3 * invstd3, 4 k3 = invstd3^2 * mean(ds * (y - mean3))
__global__ void qarep_bwd_finalize_kernel(ColSrc src, long M, int C, const float* gamma3, const float* gammap, const float* sv, float* dgamma3,
                                          float* dgammap, float* dbetap, float* cb) {
    SGX_FIN_THREAD(src, C);
    double tot[4];
    col_totals<4>(src, C, c, cok, tot);
    if (!writer) return;
    const double Sg = tot[0], Sgs = tot[1], Sgy = tot[2], Ssy = tot[3];
    const double m = (double)M, invstd3 = sv[C + c], invstdp = sv[5 * C + c];
    const double gp = gammap ? (double)gammap[c] : 1.0, g3 = gamma3 ? (double)gamma3[c] : 1.0;
    if (dgammap) dgammap[c] += (float)(invstdp * Sgs);
    if (dbetap) dbetap[c] += (float)Sg;
    const double cp = gp * invstdp, kp = invstdp * invstdp * Sgs / m;
    // sum ds * (y - mean3) with ds = cp * ((g - mean g) - (s - mean_s) * kp); sum (y - mean3) = 0
    const double Sdy = cp * (Sgy - kp * Ssy);
    if (dgamma3) dgamma3[c] += (float)(invstd3 * Sdy);
    // (d beta3 = sum ds = 0: the input gradient of post_bn sums to zero per channel)
    cb[c] = (float)cp;
    const double mgd = Sg / m;  // mean g as hi + lo floats: see bn_bwd_finalize_kernel
    const float mg_hi = (float)mgd;
    cb[C + c] = mg_hi;
    cb[2 * C + c] = (float)kp;
    cb[3 * C + c] = (float)(g3 * invstd3);
    cb[4 * C + c] = (float)(invstd3 * invstd3 * Sdy / m);
    cb[5 * C + c] = (float)(mgd - (double)mg_hi);
}
extern "C" int32_t sgx_qarep_bwd_finalize(const float* partials4, int32_t nblk, int64_t M, int32_t C, const float* gamma3, const float* gammap,
                                          const float* sv, float* dgamma3, float* dgammap, float* dbetap, float* cb, void* ws, int64_t ws_bytes,
                                          void* stream) {
    SGX_CHECK_ARG(partials4 && sv && cb && nblk > 0 && M > 0, "qarep_bwd_finalize: bad args");
    ColSrc src;
    int32_t rc = col_prereduce<4>(partials4, nblk, C, ws, ws_bytes, stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(qarep_bwd_finalize_kernel, fin_grid(src, C), fin_block(src), 0, stream, src, (long)M, C, gamma3, gammap, sv, dgamma3, dgammap, dbetap, cb);
    SGX_CHECK_LAUNCH("qarep_bwd_finalize");
    return SGX_OK;
}
struct QarepBwdApplyF {
    QarepBwdBase b;
    const float* cb; float* ds; long ds_ld; float* dy; long dy_ld;
    typedef QarepBwdIn In;
    __device__ In load(long r, int c) const { return b.load(r, c); }
    struct Cst { QarepBwdBase::Cst t; float4 cp, mg, kp, c3, k3, ml; };
    __device__ Cst consts(int c) const {
        const int C = b.C;
        return Cst{b.consts(c), sgx_ld4(cb + c), sgx_ld4(cb + C + c), sgx_ld4(cb + 2 * C + c), sgx_ld4(cb + 3 * C + c), sgx_ld4(cb + 4 * C + c), sgx_ld4(cb + 5 * C + c)};
    }
    __device__ void apply(long r, int c, const In& in, const Cst& k, float4 (&)[1]) const {
        float4 g, sc, yc;
        b.terms(k.t, in, g, sc, yc);
        const float4 cp = k.cp, mg = k.mg, kp = k.kp, c3 = k.c3, k3 = k.k3, ml = k.ml;
        // differences first, then the scale (the order ATen's CPU batch-norm backward uses)
        const float4 s = make_float4(cp.x * (((g.x - mg.x) - ml.x) - sc.x * kp.x), cp.y * (((g.y - mg.y) - ml.y) - sc.y * kp.y),
                                     cp.z * (((g.z - mg.z) - ml.z) - sc.z * kp.z), cp.w * (((g.w - mg.w) - ml.w) - sc.w * kp.w));
        sgx_st4(ds + r * ds_ld + c, s);
        sgx_st4(dy + r * dy_ld + c, make_float4(c3.x * (s.x - yc.x * k3.x), c3.y * (s.y - yc.y * k3.y), c3.z * (s.z - yc.z * k3.z), c3.w * (s.w - yc.w * k3.w)));
    }
};
// ds (gradient of the 1x1 branch output u = upstream gradient of bn3) and dy (gradient of the 3x3 convolution output); either may alias its input
// (ds over u, dy over y): a row is read completely before it is written
extern "C" int32_t sgx_qarep_bwd_apply(const float* dout, int64_t d_ld, const float* y, int64_t y_ld, const float* u, int64_t u_ld, const float* cf,
                                       const float* sv, const float* cb, float* ds, int64_t ds_ld, float* dy, int64_t dy_ld, int64_t M, int32_t C,
                                       int32_t act, void* stream) {
    SGX_CHECK_ARG(dout && y && u && cf && sv && cb && ds && dy, "qarep_bwd_apply: null pointer");
    SGX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "qarep_bwd_apply: need M>0 and C%%4==0 (C=%d)", C);
    QarepBwdApplyF f{QarepBwdBase{dout, d_ld, y, y_ld, u, u_ld, cf, sv, C, act}, cb, ds, ds_ld, dy, dy_ld};
    SweepGeom g = sweep_geom(M, C);
    SGX_LAUNCH((sweepq_kernel<QarepBwdApplyF, 1>), dim3(g.nblk, g.ctiles), dim3(SW_THREADS), 0, stream, f, g, (float*)nullptr);
    SGX_CHECK_LAUNCH("qarep_bwd_apply");
    return SGX_OK;
}

// RepVGG two-branch block, backward through the activation AND the reduce of both BatchNorm backward passes in one sweep (round 5):
//   g = dy * act'(s1*x1 + t1 + s2*x2 + t2)   (written: both BatchNorm backward applies read it),
//   partials [4][blocks][C] = sum g, sum g (x1 - mean1), sum g, sum g (x2 - mean2)   - i.e. the two [2][blocks][C] row sets sgx_bn_bwd_reduce
// would have produced for (g, x1) and (g, x2) with two more passes over g and the saved conv outputs (modules/repvgg_block.py:94-104:
// PP-YOLOE runs ~30 such blocks per step).  The sums are taken from the SAME g values that are stored.
struct DualAffineBwdReduceF {
    DualAffineF p; const float* dy; long dy_ld; float* g; long g_ld; const float* mean1; const float* mean2;
    struct In { float4 a, b, d; };
    struct Cst { DualAffineF::Cst k; float4 m1, m2; };
    __device__ In load(long row, int c) const {
        return In{sgx_ld4(p.x1 + row * p.x1_ld + c), sgx_ld4(p.x2 + row * p.x2_ld + c), sgx_ld4(dy + row * dy_ld + c)};
    }
    __device__ Cst consts(int c) const { return Cst{p.consts(c), sgx_ld4(mean1 + c), sgx_ld4(mean2 + c)}; }
    __device__ void apply(long row, int c, const In& in, const Cst& k, float4 (&q)[4]) const {
        const float4 v = p.pre(k.k, in.a, in.b);
        const float4 o = make_float4(in.d.x * sgx_act_grad(v.x, p.act), in.d.y * sgx_act_grad(v.y, p.act), in.d.z * sgx_act_grad(v.z, p.act),
                                     in.d.w * sgx_act_grad(v.w, p.act));
        sgx_st4(g + row * g_ld + c, o);
        q[0].x += o.x; q[0].y += o.y; q[0].z += o.z; q[0].w += o.w;
        q[1].x += o.x * (in.a.x - k.m1.x); q[1].y += o.y * (in.a.y - k.m1.y); q[1].z += o.z * (in.a.z - k.m1.z); q[1].w += o.w * (in.a.w - k.m1.w);
        q[2].x += o.x; q[2].y += o.y; q[2].z += o.z; q[2].w += o.w;
        q[3].x += o.x * (in.b.x - k.m2.x); q[3].y += o.y * (in.b.y - k.m2.y); q[3].z += o.z * (in.b.z - k.m2.z); q[3].w += o.w * (in.b.w - k.m2.w);
    }
};
extern "C" int32_t sgx_dual_affine_act_bwd_reduce(const float* dy, int64_t dy_ld, const float* x1, int64_t x1_ld, const float* s1, const float* t1,
                                                  const float* mean1, const float* x2, int64_t x2_ld, const float* s2, const float* t2,
                                                  const float* mean2, float* g, int64_t g_ld, int64_t M, int32_t C, int32_t act, float* partials4,
                                                  void* stream) {
    SGX_CHECK_ARG(dy && x1 && s1 && t1 && mean1 && x2 && s2 && t2 && mean2 && g && partials4, "dual_affine_act_bwd_reduce: null pointer");
    SGX_CHECK_ARG(M > 0 && C > 0 && C % 4 == 0, "dual_affine_act_bwd_reduce: need M>0 and C%%4==0 (C=%d)", C);
    DualAffineBwdReduceF f{DualAffineF{x1, x1_ld, s1, t1, x2, x2_ld, s2, t2, nullptr, 0, nullptr, 0, act, 1.f, nullptr}, dy, dy_ld, g, g_ld, mean1, mean2};
    SweepGeom gm = sweep_geom(M, C);
    SGX_LAUNCH((sweepq_kernel<DualAffineBwdReduceF, 4>), dim3(gm.nblk, gm.ctiles), dim3(SW_THREADS), 0, stream, f, gm, partials4);
    SGX_CHECK_LAUNCH("dual_affine_act_bwd_reduce");
    return SGX_OK;
}

struct ColsumF {
    const float* x; long ld; long rows_per_img; long ld_img;
    struct In { float4 v; };
    __device__ In load(long r, int c) const {
        long img = r / rows_per_img;
        return In{sgx_ld4(x + img * ld_img + (r - img * rows_per_img) * ld + c)};
    }
    struct Cst {};
    __device__ Cst consts(int) const { return Cst{}; }
    __device__ void apply(long, int, const In& in, const Cst&, float4& q0, float4& q1) const {
        (void)q1;
        const float4 v = in.v;
        q0.x += v.x; q0.y += v.y; q0.z += v.z; q0.w += v.w;
    }
};
__global__ void colsum_finalize_kernel(ColSrc src, int C, float* out, int accumulate) {
    SGX_FIN_THREAD(src, C);
    double tot[1];
    col_totals<1>(src, C, c, cok, tot);
    if (!writer) return;
    const double s = tot[0];
    out[c] = accumulate ? out[c] + (float)s : (float)s;
}
extern "C" int64_t sgx_colsum_workspace(int64_t M, int32_t C) {
    int nblk = sgx_stats_blocks(M);
    return (((int64_t)nblk * C * (int64_t)sizeof(float) + 255) & ~255L) + sgx_reduce_workspace(nblk, C);
}
// ws: sgx_colsum_workspace(M, C) bytes = the fp32 partial rows followed by the fp64 slices of the pre-reduction
extern "C" int32_t sgx_colsum(const float* x, int64_t ld, int64_t M, int32_t C, int64_t rows_per_img, int64_t ld_img, float* out,
                              int32_t accumulate, float* ws, void* stream) {
    SGX_CHECK_ARG(x && out && ws && rows_per_img > 0, "colsum: bad args");
    ColsumF f{x, ld, rows_per_img, ld_img};
    int32_t rc = run_sweep<ColsumF, 1>(f, M, C, ws, stream, "colsum");
    if (rc) return rc;
    const int nblk = sgx_stats_blocks(M);
    const int64_t part_bytes = ((int64_t)nblk * C * (int64_t)sizeof(float) + 255) & ~255L;
    ColSrc src;
    rc = col_prereduce<1>(ws, nblk, C, (char*)ws + part_bytes, sgx_reduce_workspace(nblk, C), stream, &src);
    if (rc) return rc;
    SGX_LAUNCH(colsum_finalize_kernel, fin_grid(src, C), fin_block(src), 0, stream, src, C, out, accumulate);
    SGX_CHECK_LAUNCH("colsum_finalize");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* p, long n, float v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int32_t sgx_fill(float* p, int64_t n, float v, void* stream) {
    if (n <= 0) return SGX_OK;
    SGX_CHECK_ARG(p, "fill: null pointer");
    long blocks = (n + 255) / 256;
    SGX_LAUNCH(fill_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, p, (long)n, v);
    SGX_CHECK_LAUNCH("fill");
    return SGX_OK;
}

__global__ void scale_dev_kernel(const float* x, const float* s, const float* t, float* y, long n) {
    float f = s[0] * (t ? t[0] : 1.f);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * f;
}
extern "C" int32_t sgx_scale_by_device_scalar(const float* x, const float* s, const float* t, float* y, int64_t n, void* stream) {
    if (n <= 0) return SGX_OK;
    SGX_CHECK_ARG(x && s && y, "scale: null pointer");
    long blocks = (n + 255) / 256;
    SGX_LAUNCH(scale_dev_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, stream, x, s, t, y, (long)n);
    SGX_CHECK_LAUNCH("scale_by_device_scalar");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
// layout changes at the model boundary
__global__ void nchw_to_nhwc_kernel(int N, int C, int H, int W, int Cpad, const float* x, float* y) {
    long n = (long)N * H * W * Cpad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % Cpad);
        long pix = i / Cpad;
        long hw = (long)H * W;
        long img = pix / hw, rem = pix - img * hw;
        y[i] = c < C ? x[(img * C + c) * hw + rem] : 0.f;
    }
}
extern "C" int32_t sgx_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad, const float* x, float* y, void* stream) {
    SGX_CHECK_ARG(x && y && Cpad >= C && Cpad % 4 == 0, "nchw_to_nhwc: bad args");
    long n = (long)N * H * W * Cpad, blocks = (n + 255) / 256;
    SGX_LAUNCH(nchw_to_nhwc_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, N, C, H, W, Cpad, x, y);
    SGX_CHECK_LAUNCH("nchw_to_nhwc");
    return SGX_OK;
}
__global__ void nhwc_to_nchw_kernel(int N, int C, int H, int W, const float* x, long ld_pix, long ld_img, float* y) {
    long n = (long)N * C * H * W;
    long hw = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long rem = i % hw;
        long t = i / hw;
        int c = (int)(t % C);
        long img = t / C;
        y[i] = x[img * ld_img + rem * ld_pix + c];
    }
}
extern "C" int32_t sgx_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float* x, int64_t x_ld_pix, int64_t x_ld_img, float* y,
                                    void* stream) {
    SGX_CHECK_ARG(x && y, "nhwc_to_nchw: null pointer");
    long n = (long)N * C * H * W, blocks = (n + 255) / 256;
    SGX_LAUNCH(nhwc_to_nchw_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, N, C, H, W, x, (long)x_ld_pix,
               (long)x_ld_img, y);
    SGX_CHECK_LAUNCH("nhwc_to_nchw");
    return SGX_OK;
}

// uint8 HWC batch -> standardized fp32 NHWC (channels padded to Cpad): one thread per pixel, 16-byte stores.
__global__ void standardize_u8_kernel(long npix, int C, int Cpad, const uint8_t* x, float max_value, const float* mean, const float* stdv, float* y) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const uint8_t* px = x + i * C;
        float* py = y + i * Cpad;
        for (int c0 = 0; c0 < Cpad; c0 += 4) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c0 + t;
                float f = 0.f;
                if (c < C) {
                    f = (float)px[c] / max_value;
                    if (mean) f = (f - mean[c]) / stdv[c];
                }
                v[t] = f;
            }
            sgx_st4(py + c0, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}
extern "C" int32_t sgx_standardize_u8_hwc(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cpad, const uint8_t* x, float max_value,
                                          const float* mean, const float* stdv, float* y, void* stream) {
    SGX_CHECK_ARG(x && y && N > 0 && H > 0 && W > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0, "standardize_u8: bad args (C=%d Cpad=%d)", C, Cpad);
    SGX_CHECK_ARG(max_value > 0.f && ((mean == nullptr) == (stdv == nullptr)), "standardize_u8: max_value > 0, mean and std go together");
    const long npix = (long)N * H * W;
    const long blocks = (npix + 255) / 256;
    SGX_LAUNCH(standardize_u8_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, npix, C, Cpad, x, max_value, mean,
               stdv, y);
    SGX_CHECK_LAUNCH("standardize_u8");
    return SGX_OK;
}

// One ragged image into its slot of the padded batch: dst pixel (yy, xx) = standardized src pixel (yy - top, xx - left) inside the image,
// the standardized pad value outside (DetectionPadIfNeeded / DetectionPadToSize on the device: transforms.py:846-941 with the padding
// coordinates of transforms/utils.py:79-106, followed by DetectionStandardize :490-510).  Same arithmetic as standardize_u8_kernel.
__global__ void pad_standardize_u8_kernel(int h, int w, int C, const uint8_t* x, int H, int W, int Cpad, int top, int left, float max_value,
                                          const float* mean, const float* stdv, const float* pad_value, float* y) {
    const long npix = (long)H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const int yy = (int)(i / W), xx = (int)(i - (long)yy * W);
        const int sy = yy - top, sx = xx - left;
        const bool inside = sy >= 0 && sy < h && sx >= 0 && sx < w;
        const uint8_t* px = x + ((long)(inside ? sy : 0) * w + (inside ? sx : 0)) * C;
        float* py = y + i * Cpad;
        for (int c0 = 0; c0 < Cpad; c0 += 4) {
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int c = c0 + t;
                float f = 0.f;
                if (c < C) {
                    f = (inside ? (float)px[c] : pad_value[c]) / max_value;
                    if (mean) f = (f - mean[c]) / stdv[c];
                }
                v[t] = f;
            }
            sgx_st4(py + c0, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
}
extern "C" int32_t sgx_pad_standardize_u8_hwc(int32_t h, int32_t w, int32_t C, const uint8_t* x, int32_t H, int32_t W, int32_t Cpad, int32_t top,
                                              int32_t left, float max_value, const float* mean, const float* stdv, const float* pad_value, float* y,
                                              void* stream) {
    SGX_CHECK_ARG(x && y && pad_value && h > 0 && w > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0, "pad_standardize_u8: bad args (C=%d Cpad=%d)", C, Cpad);
    SGX_CHECK_ARG(top >= 0 && left >= 0 && top + h <= H && left + w <= W, "pad_standardize_u8: the %dx%d image at (%d, %d) does not fit %dx%d", h, w, top, left, H, W);
    SGX_CHECK_ARG(max_value > 0.f && ((mean == nullptr) == (stdv == nullptr)), "pad_standardize_u8: max_value > 0, mean and std go together");
    const long npix = (long)H * W, blocks = (npix + 255) / 256;
    SGX_LAUNCH(pad_standardize_u8_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, h, w, C, x, H, W, Cpad, top, left,
               max_value, mean, stdv, pad_value, y);
    SGX_CHECK_LAUNCH("pad_standardize_u8");
    return SGX_OK;
}
