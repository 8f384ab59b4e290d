// Implicit-GEMM convolution for gfx950 on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32 fma
// chains, 157 TFLOP/s peak) - forward, data gradient and weight gradient.
//
// HBM layout: activations NHWC (+ explicit pixel/image strides), weights OHWI, so the GEMM-K axis
// (input channels of one filter tap) is contiguous for BOTH operands: a workgroup stages
// [BM pixels x 16 ch] and [BN filters x 16 ch] slabs with 16-byte coalesced loads, keeps them in LDS
// with a 20-float row pitch (conflict-free ds_read_b128: 16 consecutive rows hit 16 disjoint 4-bank
// groups), and every wave reads its 32x32x2 MFMA fragments as two 128-bit LDS loads per 8 k-steps.
// Double-buffered LDS, register-staged prefetch of the next slab issued before the MFMA block, one
// barrier per slab.  Global operands are fetched with buffer loads (32-bit lane offset, hardware
// bounds check returning zeros for padding) so that no load in the main loop waits on another.  Workgroup -> tile mapping is XCD-aware: the 8 XCDs (private L2s) each get a
// contiguous run of tiles, N-tiles of the same pixel slab adjacent, so halo rows and the slab itself
// are re-read from that XCD's L2.
//
// One "gather GEMM" kernel serves forward and data-gradient: output row m <-> a pixel of an output
// grid, tap (i,j) contributes input pixel (a*si + dh0 + dstep*i, b*si + dw0 + dstep*j).  Stride-s data gradients are
// decomposed into s*s output-parity classes, each a dense stride-1 problem over the dY grid with its
// own tap subset (no multiply-by-zero work).  Reference call sites: see include/sgx_hip.h.
#include "sgx_common.h"

#define SGX_MAX_TAPS 64
#include <algorithm>
#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cstdlib>
#define SGX_CONV_WAVE_PRIO_DEFAULT 0

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
    double flops, bytes;
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
    ProfScope(int cls, double flops, double bytes, void* stream) : on(false), st((hipStream_t)stream) {
        std::lock_guard<std::mutex> g(g_prof_mu);
        if (!g_prof_on) return;
        on = true;
        r.cls = cls;
        r.flops = flops;
        r.bytes = bytes;
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
extern "C" int32_t sgx_prof_bytes(int32_t cls, double* bytes) {
    std::lock_guard<std::mutex> g(g_prof_mu);
    double b = 0.0;
    for (auto& r : g_prof_recs)
        if (r.cls == cls) b += r.bytes;
    if (bytes) *bytes = b;
    return SGX_OK;
}
extern "C" int32_t sgx_prof_bound_ms(int32_t cls, double peak_flops, double hbm_bytes_per_s, double* ms) {
    SGX_CHECK_ARG(peak_flops > 0 && hbm_bytes_per_s > 0 && ms, "prof_bound_ms: bad args");
    std::lock_guard<std::mutex> g(g_prof_mu);
    double t = 0.0;
    for (auto& r : g_prof_recs)
        if (r.cls == cls) t += fmax(r.flops / peak_flops, r.bytes / hbm_bytes_per_s);
    *ms = t * 1e3;
    return SGX_OK;
}
#define SGX_PROF(cls, flops, bytes, stream) ProfScope prof_scope__((cls), (flops), (bytes), (stream))
#else
extern "C" int32_t sgx_prof_bound_ms(int32_t, double, double, double* ms) {
    if (ms) *ms = 0;
    return SGX_OK;
}
extern "C" int32_t sgx_prof_bytes(int32_t, double* bytes) {
    if (bytes) *bytes = 0;
    return SGX_OK;
}
extern "C" int32_t sgx_prof_enable(int32_t) { return SGX_OK; }
extern "C" int32_t sgx_prof_summary(int32_t, double* ms, double* flops, int64_t* launches) {
    if (ms) *ms = 0;
    if (flops) *flops = 0;
    if (launches) *launches = 0;
    return SGX_OK;
}
#define SGX_PROF(cls, flops, bytes, stream)
#endif

// ------------------------------------------------------------------------------------------------
// Gather GEMM (forward and data gradient).
//
// Addressing is split so that NO memory instruction of the main loop depends on another one (the r1a kernel read
// its tap table from kernel-argument memory each slab and hipcc serialised every load behind s_waitcnt vmcnt(0)):
//   per lane, once   : a 32-bit byte offset of the lane's 16-byte chunk at tap (0,0) relative to the workgroup's first
//                      image, and a 64-bit mask "tap t reads inside the image" (zero padding / tile edge);
//   per slab, scalar : the tap's byte offset and bit index from three SALU counters (tap row, tap column, channel
//                      chunk) - taps are described arithmetically: tap (i,j) reads pixel (a*si+dh0+dstep*i, b*si+dw0+dstep*j);
//   per load         : one v_add + one v_cndmask (masked lanes get SGX_BUF_OOB and the buffer bounds check returns 0).
// FLAT (C < 16, the RGB stem): the GEMM-K axis is the flattened (tap, channel) axis, a 16-wide slab spans 16/C taps.
// ------------------------------------------------------------------------------------------------
// SGX_WAVE_PRIO (environment, read once) bit 2: the forward / data-gradient kernels run their waves at issue priority 1
static int conv_wave_prio() {
    static const int mode = [] {
        const char* e = getenv("SGX_WAVE_PRIO");
        return ((e ? atoi(e) : SGX_CONV_WAVE_PRIO_DEFAULT) >> 2) & 1;
    }();
    return mode;
}
struct IgemmParams {
    const float* A;
    const float* Wt;
    const float* bias;
    const float* addend;
    float* Y;
    float* stat_partials;
    int M, Ha, Wa, Hin, Win, C, Nout;
    int Th, Tw;            // taps along h / w
    int dh0, dw0, dstep;   // tap (i,j) -> input pixel (a*si + dh0 + dstep*i, b*si + dw0 + dstep*j)
    int si, so, ph, pw, Hout, Wout;
    long a_ld_pix, a_ld_img, y_ld_pix, y_ld_img;
    long w_ld_n;           // weight row pitch (elements); row n holds [tap][c]
    long a_bytes, w_bytes; // extents for the buffer descriptors
    int act, accumulate;
    int vec;                  // 1: Nout, output strides and pointers allow 16-byte epilogue accesses
    int mt, nt, nblk, chunk;  // tile counts and XCD chunk
    // divisions by launch constants as multiply-high + shift (sgx_fastdiv, filled by the launch helpers): nt, Ha * Wa, Wa, and the patch
    // kernel's tiles per image / per tile row - a 32-bit division is a ~40-instruction sequence, and the prologue ahead of a workgroup's
    // first global load held ten of them
    sgx_fastdiv fd_nt, fd_hw, fd_wa, fd_txy, fd_tx;
    int stat_nblk;
    // second addend with its own strides and a scale (host value x optional device scalar): the residual branch of a YOLO-NAS bottleneck,
    // dx += alpha * dz, folded into the data gradient of its first block instead of an axpy pass + an accumulate pass over dx
    const float* addend2;
    const float* addend2_scale_dev;
    float addend2_scale;
    long a2d_ld_pix, a2d_ld_img;
    // Optional SECOND K-axis source (template PH2): after the taps of A / Wt the GEMM continues over another input tensor with its own
    // taps and filter, on the same output pixel grid and with the same channel count C.
    //   PH2 = 1: same accumulator - dx = dgrad3x3(dy3) + dgrad1x1(ds) of a QARepVGG block as ONE launch (no accumulate pass over dx);
    //   PH2 = 2: second accumulator and second output (Y2, bias2) - conv3x3(x) and conv1x1(x) + b of a QARepVGG block as ONE launch
    //            (x is staged once per tap by the same workgroup); stat_partials then holds FIVE planes: sum y, y^2, u0, u0^2, y*u0 (u0 = u - bias2).
    const float* A2;
    const float* Wt2;
    const float* bias2;
    float* Y2;
    int Hin2, Win2, Th2, Tw2, dh02, dw02, dstep2;
    long a2_ld_pix, a2_ld_img, w2_ld_n, a2_bytes, w2_bytes;
    // Pre-split filter planes (template WPL; sgx_filter_planes_batch, round 5): the bf16x3 pieces of Wt / Wt2 as the per-step producer left
    // them - [ch / 16][hi | mid | lo][tap][row][16 bf16] - so that the filter half of a bf16x3 launch's staging is a 16-byte copy
    // (no vector split: the filter is split by every pixel tile of a launch, the activations once per filter tile).  NULL: split while staging.
    const unsigned char* Wp;
    const unsigned char* Wp2;
    long wp_bytes, wp2_bytes;
    // BatchNorm-backward REDUCE of the layer(s) whose output gradient this launch finalises (data gradients; sgx_bn_reduce_req): per request
    // a channel range of Y, that layer's saved conv output t (its own strides) and its BatchNorm scale / shift / mean.  The epilogue forms
    // g = v * act'(scale t + shift) from the value v it is about to store and leaves sum g, sum g (t - mean) per column in row block
    // (req_row0 + tile row) of the request's partials - the rows sgx_bn_bwd_reduce would have produced with a pass over dy and t.
    int nreq, req_row0;
    int lab;  // measurement builds (-DSGX_IGEMM_LAB, tools/conv_lab.py --ablate) only: see IGL below; the product build never reads it
    int prio;  // raise the waves' issue priority over the side stream's weight-gradient waves (SGX_WAVE_PRIO bit 2; sgx_common.h)
    struct {
        const float* t;
        const float* scale;
        const float* shift;
        const float* mean;
        float* parts;
        long t_ld_pix, t_ld_img;
        int c_lo, c_hi, act, rows;
    } req[SGX_MAX_BN_REQ];
};
// this lane's request for the four output columns col .. col + 3 (ranges are multiples of 4): index or -1, the per-channel constants
struct BnReqLane {
    int rq;
    int act;
    const float* tp;  // t + (col - c_lo)
    float4 sc, sh, mu;
};
__device__ __forceinline__ BnReqLane sgx_bnreq_lane(const IgemmParams& p, int col) {
    BnReqLane L;
    L.rq = -1;
    L.act = 0;
    L.tp = nullptr;
    L.sc = L.sh = L.mu = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < SGX_MAX_BN_REQ; ++r)
        if (r < p.nreq && col >= p.req[r].c_lo && col < p.req[r].c_hi) {
            const int c = col - p.req[r].c_lo;
            L.rq = r;
            L.act = p.req[r].act;
            L.tp = p.req[r].t + c;
            L.sc = sgx_ld4(p.req[r].scale + c);
            L.sh = sgx_ld4(p.req[r].shift + c);
            L.mu = sgx_ld4(p.req[r].mean + c);
        }
    return L;
}
__device__ __forceinline__ void sgx_bnreq_acc(const BnReqLane& L, const float4& v, const float4& t, float4& cs, float4& cq) {
    const float gx = L.act == SGX_ACT_NONE ? v.x : v.x * sgx_act_grad(L.sc.x * t.x + L.sh.x, L.act);
    const float gy = L.act == SGX_ACT_NONE ? v.y : v.y * sgx_act_grad(L.sc.y * t.y + L.sh.y, L.act);
    const float gz = L.act == SGX_ACT_NONE ? v.z : v.z * sgx_act_grad(L.sc.z * t.z + L.sh.z, L.act);
    const float gw = L.act == SGX_ACT_NONE ? v.w : v.w * sgx_act_grad(L.sc.w * t.w + L.sh.w, L.act);
    cs.x += gx; cs.y += gy; cs.z += gz; cs.w += gw;
    cq.x += gx * (t.x - L.mu.x); cq.y += gy * (t.y - L.mu.y); cq.z += gz * (t.z - L.mu.z); cq.w += gw * (t.w - L.mu.w);
}
// the finished column sums of a tile row -> the request's partial rows (one thread per tile column)
__device__ __forceinline__ void sgx_bnreq_publish(const IgemmParams& p, int col, int mtile, float s, float q) {
#pragma unroll
    for (int r = 0; r < SGX_MAX_BN_REQ; ++r)
        if (r < p.nreq && col >= p.req[r].c_lo && col < p.req[r].c_hi) {
            const int C = p.req[r].c_hi - p.req[r].c_lo;
            float* const o = p.req[r].parts + (long)(p.req_row0 + mtile) * C + (col - p.req[r].c_lo);
            o[0] = s;
            o[(long)p.req[r].rows * C] = q;
        }
}

// Ablation lab of the implicit-GEMM loop (measurement builds only: -DSGX_IGEMM_LAB; sgx_debug_set_igemm_lab; tools/conv_lab.py --ablate).
// Bits of IgemmParams::lab - each removes ONE component of the K loop (results are garbage, timings are what the lab is for):
//   1 no global loads behind the first slab (the first slab's registers are staged again and again)   2 no LDS stores behind the first slab
//   4 no fragment reads / MFMAs   8 no bf16 split (raw bits are stored: same stores, no vector work)   16 no epilogue (nothing is written)
// The product build compiles IGL(b) to `false`.
#ifdef SGX_IGEMM_LAB
#define IGL(b) ((p.lab & (b)) != 0)
static std::atomic<int> g_ig_lab{0};
extern "C" int32_t sgx_debug_set_igemm_lab(int32_t bits) {
    g_ig_lab = bits;
    return SGX_OK;
}
#else
#define IGL(b) false
#endif
#define IG_BK 16
#define IG_LD 20
// ---- "bf16x3" arithmetic (MATH = 1): fp32 operands split into three bf16 pieces (round-to-nearest, each residual exact in fp32), products hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, fp32 accumulate): the dropped terms are <= 2^-24 of the product, i.e. the result carries fp32
// accuracy, while six 32-cycle bf16 MFMAs replace eight 64-cycle fp32 ones per 16-deep slab (2.7x fewer matrix-pipe cycles).
// The split happens ONCE per element, when a slab is staged from registers into LDS: three bf16 planes of [rows][16 bf16 = 8 dwords],
// no padding; the two 16-byte halves of a row are swapped on rows with bit 3 set (IG_SWZ), which makes the 16-byte fragment reads of
// 16 consecutive rows hit 16 distinct 4-bank groups (conflict-free) at 2/3 of the LDS a padded pitch would need.
#define IG_LDP 8
#define IG_SWZ(row, dw) ((dw) ^ ((((row) >> 3) & 1) << 2))
#include "conv_mma.h"
#include "wgrad_patch.h"

// KD = slab depth, NBUF = LDS buffers.  (16, 2): one barrier per slab (the round-1 loop).  KD = 32 (fp32 arithmetic, C % 32 == 0;
// experiment switches sgx_debug_set_variant(5 | 6), see run_igemm): every global load instruction covers whole 128-byte lines (8 lanes
// x 16 B per slab row instead of 4 x 16 B = half a line - MI355X's load path handles half-line "fragment-shaped" requests at about half
// the rate), half the load instructions and address arithmetic per FLOP, and a register prefetch that is 16 MFMAs (1024 matrix-pipe
// cycles) ahead instead of 8.  (32, 1): ONE LDS buffer (18 KB for 64x64: occupancy stays VGPR-bound, 7 workgroups per CU) with a
// write-after-barrier hand-over - two barriers per slab, i.e. as many per FLOP as (16, 2).  (32, 2): two buffers (37 KB for 64x64:
// 4 workgroups per CU), one barrier per slab - half the barriers per FLOP at lower occupancy.
// Waves per SIMD the register allocation aims for.  The bf16x3 32-deep loop with one 32x32 block per wave (64x64, 128x32, 64x32 tiles) needs
// 70 + 32 registers as the compiler allocates it unprompted - 4 waves per SIMD where its 27 KB of LDS would admit 5 workgroups per CU; the
// loop lives on occupancy (r4y: a variant that cost one wave of it lost 8 %), so those forms ask for 5 (<= 96 registers).
#ifndef IG_BF3_MIN_WAVES
#define IG_BF3_MIN_WAVES 5
#endif
#ifndef IG_BF3_MIN_WAVES_PH2
#define IG_BF3_MIN_WAVES_PH2 1  // (the two-source form as well: 93 registers; the two-output form would spill - three accumulators)
#endif
#ifndef IG_WPR_MIN_WAVES_6464
#define IG_WPR_MIN_WAVES_6464 4  // (5 - round 5 - spilled 4 dwords)
#endif
template <int BM, int BN, int WM, int WN, int MATH, int KD, int PH2, int NBUF, int WPL = 0, int PP = 0>
constexpr int igemm_min_waves() {
    // (ping-pong form: 512-thread workgroups, two per CU by their LDS - four waves per SIMD, 128 registers)
    if (PP) return 4;
    // Round 6: NO instantiation that a launch can select may spill (tools/kernel_regs.py --check, tests/test_tools.py): round 5 shipped the
    // two-source 64x64 form at 96 registers with 7 spilled dwords (6 launches per step), the register-fragment forms with 1-4.  A bound is
    // lowered by one wave wherever the allocation under it spilled.
    // (register fragments of the filter, WPL = 2: 24 registers of fragments one slab ahead)
    if (WPL == 2) return (BM == 64 && BN == 64 && PH2 == 0) ? IG_WPR_MIN_WAVES_6464 : (BM / (WM * 32) == 1 && BN / (WN * 32) == 1 && PH2 == 0) ? 4 : (BM / (WM * 32) == 1 && BN / (WN * 32) == 1 && PH2 == 1) ? 3 : 1;
    // (the pipelined two-buffer loop: three workgroups' LDS per CU; a bound >= 2 also keeps the accumulators in ordinary registers -
    // with 512 registers on offer hipcc parks them in AGPRs and copies 32 registers in and out per slab)
    if (NBUF == 2 && KD == 32) return (BM + BN) * 192 * 2 > 52 * 1024 ? 2 : 3;
    // (the two-source form carries a second source's offsets and masks: four waves - 128 registers - is what it fits without spilling)
    return (MATH == 1 && KD == 32 && PH2 <= IG_BF3_MIN_WAVES_PH2 && BM / (WM * 32) == 1 && BN / (WN * 32) == 1) ? (PH2 == 1 ? IG_BF3_MIN_WAVES - 1 : IG_BF3_MIN_WAVES) : 1;
}
// PP = 1 (round 6, "ping-pong"): a 512-thread workgroup is TWO wave groups of WM x WN waves, each with its own output tile, LDS slabs and
// epilogue scratch, running the one-buffer loop's two phases of a slab - STAGE (wait for the slab's loads, split, LDS stores, issue the
// next slab's loads) and COMPUTE (fragment reads + MFMAs) - half a period apart: while group 0 computes slab k, group 1 stages its slab k,
// and vice versa, one workgroup-wide barrier per phase.  Why: the ablation lab (profiles/r6c_igemm_ablation_*.txt) has the staging-only
// and the compute-only versions of a launch at ~0.63 of the whole launch EACH - five independent workgroups per CU do not overlap their
// phases (they convoy: loads return together, everybody splits together, everybody queues for the matrix pipe together); here the
// complementary pairing (matrix beside memory / vector work on every SIMD, MI355X_MICROARCH.md "Two waves per SIMD") is by construction.
// Same products in the same order per tile as PP = 0: bit-identical results.
template <int BM, int BN, int WM, int WN, bool FLAT, int MATH = 0, int KD = IG_BK, int NBUF = 2, int PH2 = 0, int WPL = 0, int PP = 0>
__global__ __launch_bounds__(WM * WN * 64 * (PP ? 2 : 1), (igemm_min_waves<BM, BN, WM, WN, MATH, KD, PH2, NBUF, WPL, PP>())) void igemm_kernel(IgemmParams p) {
    static_assert(!WPL || (MATH == 1 && KD == 32 && NBUF == 1 && !FLAT), "pre-split filter planes: the one-buffer 32-deep bf16x3 loop");
    static_assert(!PP || (MATH == 1 && KD == 32 && NBUF == 1 && !FLAT && PH2 == 0 && WPL <= 1), "ping-pong: the one-buffer 32-deep bf16x3 loop, one source");
    if (p.prio) SGX_WAVE_PRIO(1);
    constexpr int G = PP ? 2 : 1;  // wave groups (tiles) per workgroup
    static_assert((KD == 16 && NBUF == 2) || (KD == 32 && !FLAT && NBUF == 1) || (KD == 32 && !FLAT && NBUF == 2 && MATH == 1) ||
                      (KD == 32 && !FLAT && NBUF == 3 && MATH == 0 && PH2 == 0),
                  "32-deep slabs: channel-chunked K axis; one LDS buffer, (bf16x3) the two-buffer pipelined loop, or (fp32, NBUF = 3) all slabs up front");
    constexpr int LBUF = NBUF == 3 ? 1 : NBUF;  // LDS buffers
    // (round 6: the two-output form also runs on the flattened K axis - the RGB stem's QARepVGG block: its second source is the 1x1 branch,
    // one 16-wide slab of which 4 columns are live)
    static_assert(PH2 == 0 || PH2 == 2 || !FLAT, "second K-axis source into the same accumulator: channel-chunked K axis");
    static_assert(MATH == 0 || MATH == 1, "arithmetic: 0 = fp32 matrix pipe, 1 = bf16x3");
    // (Round 4 tried MATH = 2: the five correction products of the bf16x3 scheme added into the SAME accumulator as the leading one - it
    // frees 16 registers per block and ran the step 4 % faster, but the bf16 MFMA's accumulate floors what it shifts out: small terms added
    // to a large accumulator leave a NEGATIVE bias of 1e-8 .. 8e-8 of the output's rms per convolution (tools/conv_error_probe.py,
    // profiles/r4m_*: two accumulators 1e-10, fp32 pipe 1e-10), which adds up coherently through a hundred layers - YOLO-NAS-L's
    // element-wise gradient check came out at 5e-4.  The corrections therefore keep their own accumulator, also in the two-source /
    // two-output forms: three accumulators per block there.)
    constexpr bool BF3 = MATH == 1;
    constexpr int NTH = WM * WN * 64;   // threads per workgroup
    constexpr int CPR = KD / 4;         // threads per slab row (16 B each)
    constexpr int RPP = NTH / CPR;      // slab rows staged per pass
    constexpr int LD = KD + 4;          // fp32 LDS row pitch: 20 / 36 floats -> conflict-free ds_read_b128 over 16 consecutive rows
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    // WPL (IgemmParams::Wp): the filter slab arrives as 16-byte pieces of the pre-split planes - item = (plane, row, 8-channel octet) - and
    // is copied into the LDS planes as it is: BN x 4 x 3 items per slab instead of BN x 8 float4 items with a 22-instruction split each
    // Which (plane, 16-row block) a thread's j-th piece belongs to is the same for its whole wave: a wave copies 16 rows x 4 octets, the
    // BN / 16 blocks of a plane follow one another, and only the place inside the block (row lane / 4, octet lane % 4) is per lane.
    // WPL = 2: the filter does not pass through LDS at all.  A 16-byte piece of the planes - row n, one half of a 16-channel block - IS
    // the B fragment of v_mfma_f32_32x32x16_bf16 for lane (n mod 32, half): every wave loads the fragments of its own 32-filter columns
    // straight from the planes (L2 / L1 hits: a filter is a few hundred KB, shared by every workgroup of the launch) into the registers
    // the MFMAs read, one slab ahead - no filter stores, no filter fragment reads, half the LDS bytes per MFMA, an LDS slab of A only.
    constexpr bool WPR = WPL == 2;
    constexpr int BPI = 3 * BN * 4, WBK = BN / 16;   // pieces per slab; 16-row blocks per plane
    constexpr int AJ = (BM + RPP - 1) / RPP, BJ = WPR ? 1 : WPL ? (BPI + NTH - 1) / NTH : (BN + RPP - 1) / RPP;
    static_assert(TM >= 1 && TN >= 1 && TM * WM * 32 == BM && TN * WN * 32 == BN, "bad tile");
    static_assert(NTH >= BM && NTH >= BN, "epilogue helpers need one thread per tile row/col");
    constexpr int LDPW = KD / 2;        // bf16x3: dwords per slab row and plane (KD bf16), unpadded
    constexpr int ROWW = BF3 ? 3 * LDPW : LD;                                  // dwords of LDS per slab row (all planes)
    // bf16x3 plane swizzle: 16-byte chunks of a row are permuted by row bits so that the fragment reads of 16 consecutive rows hit 16
    // distinct 4-bank groups - 32-byte rows: halves swapped on rows with bit 3 set; 64-byte rows: chunk ^= row bits 2-3
    auto swz = [](int row, int dw) { return KD == 16 ? IG_SWZ(row, dw) : (dw ^ (((row >> 2) & 3) << 2)); };
    constexpr int SLABS = LBUF * (BM + (WPR ? 0 : BN)) * ROWW, STAGE = WM * WN * 32 * 32;   // operand slabs; epilogue staging patches (reuse the slabs)
    constexpr int SMEM1 = SLABS > STAGE ? SLABS : STAGE;
    __shared__ __attribute__((aligned(16))) float smem_s[G * SMEM1];
    __shared__ long long rowoff_s[G][BM];
    __shared__ long long rowoff2[PH2 == 1 ? BM : 1];  // offsets into addend2 (two-source data gradient only)
    __shared__ long long rowoffT_s[G][PH2 == 2 ? 1 : SGX_MAX_BN_REQ][PH2 == 2 ? 1 : BM];  // offsets into the requests' saved conv outputs
    __shared__ float red_s[G][(PH2 == 2 ? 5 : 2) * WM * BN];

    // (PP: everything below is written for ONE wave group - `tid`, `wave` count inside the group, the LDS objects are the group's own)
    const int grp = PP ? (int)threadIdx.x / NTH : 0;
    const int tid = PP ? (int)threadIdx.x - grp * NTH : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    float* const smem = smem_s + grp * SMEM1;
    float* const As = smem;
    float* const Bs = smem + LBUF * BM * ROWW;
    long long* const rowoff = rowoff_s[grp];
    auto rowoffT = rowoffT_s[grp];
    float* const red = red_s[grp];

    // XCD-aware tile assignment (block b runs on XCD b%8; give each XCD a contiguous run of tiles)
    const int bid = blockIdx.x;
    const int lin0 = ((bid & 7) * p.chunk + (bid >> 3)) * G;
    if (lin0 >= p.nblk) return;  // whole workgroup leaves together (before any barrier)
    // (PP: an odd tile count leaves the last workgroup's second group without a tile - it repeats the first group's and writes nothing: the
    // barriers of both groups must match)
    const bool active = lin0 + grp < p.nblk;
    const int lin = active ? lin0 + grp : lin0;
    const int mtile = sgx_fdiv(lin, p.fd_nt), ntile = lin - mtile * p.nt;
    const int m0 = mtile * BM, n0 = ntile * BN;
    const int hw = p.Ha * p.Wa;
    const int T = p.Th * p.Tw;

    // ---- K-axis source state: (re)initialised by setup_src() for the primary source and, with PH2, for the second one -----------
    const int img0 = sgx_fdiv(m0, p.fd_hw);
    const int lrow = tid / CPR, chunk4 = (tid % CPR) * 4;
    sgx_buf bufA, bufB;
    int aoff[AJ], boff[WPL ? 1 : BJ];
    unsigned long long amask[AJ];
    bool bok[WPL ? 1 : BJ];
    int wp_lane = 0, wp_lds = 0, wp_rows = 0;  // WPL: planes byte offset / LDS dword offset of this lane's place in a 16-row block; filter rows from n0 on
    // WPL = 1 (round 6): everything about a lane's j-th piece that does not change from slab to slab - which (plane, 16-row block) its wave
    // copies, whether its row exists, the planes offset of that place at (tap 0, chunk 0) - is formed ONCE per source; per slab the load adds
    // the slab's wave-uniform (tap, chunk) offset through the instruction's scalar offset.  (The loop used to recompute it for every
    // slab: three signed divisions by the block count, the row predicates and their exec-mask juggling - 45 of the loop's 75 scalar and 6 of
    // its 69 vector instructions per slab, in a loop that issues 12 MFMAs; ISA of igemm_kernel<64,64,2,2,false,1,32,1,0,1>.)
    unsigned wp_voff[(WPL == 1) ? BJ : 1];
    int Tw_, T_, nkt, pixstep, rowstep;
    const int cpt = (p.C + KD - 1) / KD;
    // scalar slab state of the NEXT slab to load (non-FLAT): tap row, tap column, channel chunk
    int s_ti = 0, s_tj = 0, s_ck = 0, s_kt = 0;
    int wp_tapstep = 0, wp_ckstep = 0;  // WPL: bytes from one tap's rows to the next, from one 32-channel chunk's planes to the next
    int wp_woff = 0;                    // WPR: planes offset (tap, chunk) of the slab whose A tile load_tile issued last
    uint4 bfr[WPR ? 2 : 1][WPR ? 3 : 1][WPR ? TN : 1];  // WPR: B fragments [16-deep half][plane][column block] of the slab to compute next
    auto setup_src = [&](const float* A, const float* Wt, int Hin, int Win, int Th, int Tw, int dh0, int dw0, int dstep, long a_ld_pix, long a_ld_img,
                         long w_ld_n, long a_bytes, long w_bytes, const unsigned char* Wp, long wp_bytes) {
        // buffer descriptors: A is re-based at the workgroup's first image so that lane offsets fit 31 bits
        bufA = sgx_make_buf(A + (long)img0 * a_ld_img, a_bytes - (long)img0 * a_ld_img * 4);
        bufB = WPL ? sgx_make_buf(Wp, wp_bytes) : sgx_make_buf(Wt, w_bytes);
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = lrow + RPP * j;
            const int m = m0 + row;
            aoff[j] = 0;
            amask[j] = 0ull;
            if (row < BM && m < p.M) {
                const int img = sgx_fdiv(m, p.fd_hw);
                const int rem = m - img * hw;
                const int a = sgx_fdiv(rem, p.fd_wa);
                const int b = rem - a * p.Wa;
                const int hi0 = a * p.si + dh0, wi0 = b * p.si + dw0;
                aoff[j] = (int)(((long)(img - img0) * a_ld_img + ((long)hi0 * Win + wi0) * a_ld_pix + (FLAT ? 0 : chunk4)) * 4);
                unsigned long long mk = 0ull;
                for (int i = 0; i < Th; ++i) {
                    const int hi = hi0 + dstep * i;
                    for (int jj = 0; jj < Tw; ++jj) {
                        const int wi = wi0 + dstep * jj;
                        if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win) mk |= 1ull << (i * Tw + jj);
                    }
                }
                amask[j] = mk;
            }
        }
        if constexpr (WPL) {
            // planes [ch / 16][plane][tap][row][32 B]: a 32-deep slab (tap, chunk ck) is the 16-channel blocks 2 ck and 2 ck + 1
            wp_tapstep = p.Nout * 32;
            wp_ckstep = 6 * Th * Tw * wp_tapstep;
            if constexpr (WPR) {
                const int nrow = n0 + wn * TN * 32 + (lane & 31);  // fragment row of column block 0; block j: 32 j rows on
                wp_lane = nrow * 32 + (lane >> 5) * 16;
                wp_rows = p.Nout - nrow;                           // block j exists for this lane if 32 j < wp_rows
            } else {
                const int lrow16 = lane >> 2, oct = lane & 3;
                wp_lane = (oct >> 1) * 3 * Th * Tw * wp_tapstep + (n0 + lrow16) * 32 + (oct & 1) * 16;
                wp_lds = lrow16 * LDPW + ((oct ^ ((lrow16 >> 2) & 3)) << 2);  // swz(row, oct * 4): a block starts at a multiple of 16 rows
                wp_rows = p.Nout - n0 - lrow16;                               // this lane's row of block b exists if 16 b < wp_rows
                const int wv = sgx_uniform_i32(tid >> 6);
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const int blk = wv + (NTH / 64) * j;             // (plane, 16-row block) of this wave's j-th piece
                    const int plane = blk / WBK, rb16 = (blk - plane * WBK) * 16;
                    const bool ok = (BPI % NTH == 0 || blk < 3 * WBK) && rb16 < wp_rows;
                    wp_voff[(WPL == 1) ? j : 0] = ok ? (unsigned)(wp_lane + plane * Th * Tw * wp_tapstep + rb16 * 32) : SGX_BUF_OOB;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int row = lrow + RPP * j;
                const int n = n0 + row;
                bok[j] = (row < BN) && (n < p.Nout);
                boff[j] = (int)(((long)n * w_ld_n + (FLAT ? 0 : chunk4)) * 4);
            }
        }
        Tw_ = Tw;
        T_ = Th * Tw;
        nkt = FLAT ? (T_ * p.C + IG_BK - 1) / IG_BK : T_ * cpt;
        pixstep = dstep * (int)a_ld_pix * 4;  // bytes per tap step along w
        rowstep = pixstep * Win;               // bytes per tap step along h
        s_ti = s_tj = s_ck = s_kt = 0;
    };
    setup_src(p.A, p.Wt, p.Hin, p.Win, p.Th, p.Tw, p.dh0, p.dw0, p.dstep, p.a_ld_pix, p.a_ld_img, p.w_ld_n, p.a_bytes, p.w_bytes, p.Wp, p.wp_bytes);
    // output-row offsets: needed by the epilogue only - computed AFTER the first slab's loads are in flight (64-bit divisions and
    // multiplies that used to sit between the kernel's start and its first global load); the K loop's barriers publish them
    auto compute_rowoff = [&]() {
      if (tid < BM) {
        const int m = m0 + tid;
        long long off = -1;
        if ((!PP || active) && m < p.M) {
            const int img = sgx_fdiv(m, p.fd_hw);
            const int rem = m - img * hw;
            const int a = sgx_fdiv(rem, p.fd_wa);
            const int b = rem - a * p.Wa;
            off = (long long)img * p.y_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.y_ld_pix;
            if (PH2 == 1 && p.addend2) rowoff2[tid] = (long long)img * p.a2d_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.a2d_ld_pix;
            if (PH2 != 2) {
#pragma unroll
                for (int r = 0; r < SGX_MAX_BN_REQ; ++r)
                    if (r < p.nreq) rowoffT[PH2 == 2 ? 0 : r][PH2 == 2 ? 0 : tid] = (long long)img * p.req[r].t_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.req[r].t_ld_pix;
            }
        }
        rowoff[tid] = off;
      }
    };

    float4 ra[AJ], rb[BJ];
    auto load_tile_to = [&](float4* ra, float4* rb, bool live = true) {  // live = false: every lane out of bounds (a branch-free "no slab left")
        if (FLAT) {
            const int kk = s_kt * IG_BK + chunk4;  // flattened (tap, c) index of this lane's chunk
            const int t = kk / p.C;
            const int c = kk - t * p.C;
            const int ti = t / Tw_, tj = t - ti * Tw_;
            const bool kok = t < T_;
            const int tapoff = ti * rowstep + tj * pixstep + c * 4;
            const int tb = kok ? t : 0;
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool ok = kok && ((amask[j] >> tb) & 1ull);
                ra[j] = sgx_buf_ld4(bufA, ok ? (unsigned)(aoff[j] + tapoff) : SGX_BUF_OOB);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) rb[j] = sgx_buf_ld4(bufB, (kok && bok[j]) ? (unsigned)(boff[j] + kk * 4) : SGX_BUF_OOB);
        } else {
            const int tbit = s_ti * Tw_ + s_tj;
            const int tapoff = s_ti * rowstep + s_tj * pixstep + s_ck * (KD * 4);
            const int woff = WPL ? s_ck * wp_ckstep + tbit * wp_tapstep : (tbit * p.C + s_ck * KD) * 4;
            const bool cok = live && s_ck * KD + chunk4 < p.C;
            const bool lab_noload = IGL(1) && s_kt > 0;
            if (!lab_noload) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const bool ok = cok && ((amask[j] >> (tbit & 63)) & 1ull);
                ra[j] = sgx_buf_ld4(bufA, ok ? (unsigned)(aoff[j] + tapoff) : SGX_BUF_OOB);
            }
            }
            // (WPL: C is a multiple of 32 - no ragged channel chunk to mask on the filter side)
            if constexpr (WPR) {
                wp_woff = woff;  // the fragments of this slab are fetched by load_bfrags, behind the MFMAs that read the current ones
            } else if constexpr (WPL) {
                if (!lab_noload) {
#pragma unroll
                for (int j = 0; j < BJ; ++j) rb[j] = sgx_buf_ld4_so(bufB, live ? wp_voff[(WPL == 1) ? j : 0] : SGX_BUF_OOB, (unsigned)woff);
                }
            } else {
#pragma unroll
                for (int j = 0; j < BJ; ++j) rb[j] = sgx_buf_ld4(bufB, (cok && bok[j]) ? (unsigned)(boff[j] + woff) : SGX_BUF_OOB);
            }
            const bool wck = ++s_ck == cpt;  // (selects, not branches: the pipelined loop wants its body in one basic block)
            s_ck = wck ? 0 : s_ck;
            s_tj += wck ? 1 : 0;
            const bool wtj = s_tj == Tw_;
            s_tj = wtj ? 0 : s_tj;
            s_ti += wtj ? 1 : 0;
        }
        ++s_kt;
    };
    auto load_tile = [&]() { load_tile_to(ra, rb); };
    auto store_tile_from = [&](int buf, const float4* ra, const float4* rb) {
        if (BF3) {  // planes [buf][hi|mid|lo][row][IG_LDP dwords]; this lane's 4 k-values are 2 dwords of a row
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int row = lrow + RPP * j;
                if (BM % RPP == 0 || row < BM) {
                    uint2 h, m, l;
                    if (IGL(8)) {
                        h = make_uint2(sgx_f2u(ra[j].x), sgx_f2u(ra[j].y));
                        m = make_uint2(sgx_f2u(ra[j].z), sgx_f2u(ra[j].w));
                        l = h;
                    } else
                    sgx_split3(ra[j], h, m, l);
                    unsigned* d = reinterpret_cast<unsigned*>(As) + (buf * 3 * BM + row) * LDPW + swz(row, chunk4 >> 1);
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + BM * LDPW) = m;
                    *reinterpret_cast<uint2*>(d + 2 * BM * LDPW) = l;
                }
            }
            if constexpr (WPR) return;
            if constexpr (WPL) {
                const int wv = sgx_uniform_i32(tid >> 6);
#pragma unroll
                for (int j = 0; j < BJ; ++j) {
                    const int blk = wv + (NTH / 64) * j;  // plane * WBK + block: the planes of a buffer follow one another (BN = 16 WBK rows each)
                    if (BPI % NTH == 0 || blk < 3 * WBK) sgx_st4(Bs + (buf * 3 * BN + blk * 16) * LDPW + wp_lds, rb[j]);
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int row = lrow + RPP * j;
                if (BN % RPP == 0 || row < BN) {
                    uint2 h, m, l;
                    sgx_split3(rb[j], h, m, l);
                    unsigned* d = reinterpret_cast<unsigned*>(Bs) + (buf * 3 * BN + row) * LDPW + swz(row, chunk4 >> 1);
                    *reinterpret_cast<uint2*>(d) = h;
                    *reinterpret_cast<uint2*>(d + BN * LDPW) = m;
                    *reinterpret_cast<uint2*>(d + 2 * BN * LDPW) = l;
                }
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int row = lrow + RPP * j;
            if (BM % RPP == 0 || row < BM) sgx_st4(&As[buf * BM * LD + row * LD + chunk4], ra[j]);
        }
#pragma unroll
        for (int j = 0; j < BJ; ++j) {
            const int row = lrow + RPP * j;
            if (BN % RPP == 0 || row < BN) sgx_st4(&Bs[buf * BN * LD + row * LD + chunk4], rb[j]);
        }
    };
    auto store_tile = [&](int buf) { store_tile_from(buf, ra, rb); };
    // WPR: the B fragments of 16-deep half h of the slab load_tile addressed last: 16-channel block 2 ck + h, three planes, TN column blocks
    auto load_bfrags = [&](int h) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const bool ok = j * 32 < wp_rows;
                bfr[WPR ? h : 0][WPR ? pl : 0][WPR ? j : 0] =
                    sgx_buf_ld4u(bufB, ok ? (unsigned)(wp_lane + j * (32 * 32) + wp_woff + (h * 3 + pl) * T_ * wp_tapstep) : SGX_BUF_OOB);
            }
    };

    sgx_f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // bf16x3: the five correction products (<= 2^-8 of the leading one) get their own accumulator, so that their fp32 additions round
    // at 2^-8 of the result's magnitude; the leading hi*hi products are exact and add into `acc` once per 16-deep slab - fewer
    // roundings at full magnitude than the fp32 matrix pipe's eight per slab.  The two are summed once, before the epilogue.
    // PH2 = 2 (fp32 arithmetic only) reuses the name for the accumulator of the second output.
    // acc2: the second output of the PH2 = 2 form; accc: the bf16x3 correction products of the source being accumulated
    constexpr bool ACC2 = PH2 == 2;
    sgx_f32x16 acc2[ACC2 ? TM : 1][ACC2 ? TN : 1];
    sgx_f32x16 accc[BF3 ? TM : 1][BF3 ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (ACC2) acc2[ACC2 ? i : 0][ACC2 ? j : 0][r] = 0.f;
                if (BF3) accc[BF3 ? i : 0][BF3 ? j : 0][r] = 0.f;
            }

    const int frow = lane & 31, fk = (lane >> 5) * 8;
    // bf16x3 fragment reads + the six cross-product MFMAs of one slab (smallest terms first)
    auto compute_bf3 = [&](int buf, int half) {  // half: which 16-deep step of the slab (0 for 16-deep slabs)
        uint4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned* s = reinterpret_cast<const unsigned*>(As) + (buf * 3 * BM + wm * TM * 32 + i * 32 + frow) * LDPW + swz(frow, half * 8 + (lane >> 5) * 4);
            ah[i] = *reinterpret_cast<const uint4*>(s);
            am[i] = *reinterpret_cast<const uint4*>(s + BM * LDPW);
            al[i] = *reinterpret_cast<const uint4*>(s + 2 * BM * LDPW);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            if constexpr (WPR) {
                bh[j] = bfr[WPR ? half : 0][0][WPR ? j : 0];
                bm[j] = bfr[WPR ? half : 0][WPR ? 1 : 0][WPR ? j : 0];
                bl[j] = bfr[WPR ? half : 0][WPR ? 2 : 0][WPR ? j : 0];
            } else {
                const unsigned* s = reinterpret_cast<const unsigned*>(Bs) + (buf * 3 * BN + wn * TN * 32 + j * 32 + frow) * LDPW + swz(frow, half * 8 + (lane >> 5) * 4);
                bh[j] = *reinterpret_cast<const uint4*>(s);
                bm[j] = *reinterpret_cast<const uint4*>(s + BN * LDPW);
                bl[j] = *reinterpret_cast<const uint4*>(s + 2 * BN * LDPW);
            }
        }
        // smallest terms first; the six products of one (i, j) are interleaved across the tile's accumulators
        auto corr = [&](int i, int j) -> sgx_f32x16& { return accc[BF3 ? i : 0][BF3 ? j : 0]; };
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr(i, j) = sgx_mfma_bf16(al[i], bh[j], corr(i, j));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr(i, j) = sgx_mfma_bf16(ah[i], bl[j], corr(i, j));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr(i, j) = sgx_mfma_bf16(am[i], bm[j], corr(i, j));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr(i, j) = sgx_mfma_bf16(am[i], bh[j], corr(i, j));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) corr(i, j) = sgx_mfma_bf16(ah[i], bm[j], corr(i, j));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = sgx_mfma_bf16(ah[i], bh[j], acc[i][j]);
    };
    // fp32 arithmetic: fragments of one 16-deep step (columns kofs .. kofs+15 of the slab) and its 8 x TM x TN MFMAs
    auto compute_f32 = [&](int buf, int kofs) {
        float af[TM][8], bf[TN][8];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float* s = &As[buf * BM * LD + (wm * TM * 32 + i * 32 + frow) * LD + kofs + fk];
            float4 v0 = sgx_ld4(s), v1 = sgx_ld4(s + 4);
            af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
            af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float* s = &Bs[buf * BN * LD + (wn * TN * 32 + j * 32 + frow) * LD + kofs + fk];
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
    };
    // The K loop, once per source.  (Written as an unrolled loop over sources rather than a callable taking the accumulator: hipcc spends
    // 14 more VGPRs on the lambda form - one wave per SIMD less; left rolled, the PH2 = 2 form needs 100 VGPRs instead of 63.)  PH2 = 2: the first source's result moves to acc2 before the second source
    // starts from zero, so after the loop acc2 = first output (y), acc = second output (u).  Every pass ends behind a barrier: the LDS
    // slabs are free for the next source's first slab and for the epilogue's staging patches.
#pragma unroll
    for (int src = 0; src < (PH2 ? 2 : 1); ++src) {
        if (PH2 && src == 1) {
            if (!p.A2) break;
            setup_src(p.A2, p.Wt2, p.Hin2, p.Win2, p.Th2, p.Tw2, p.dh02, p.dw02, p.dstep2, p.a2_ld_pix, p.a2_ld_img, p.w2_ld_n, p.a2_bytes, p.w2_bytes, p.Wp2, p.wp2_bytes);
            if (PH2 == 2) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            acc2[ACC2 ? i : 0][ACC2 ? j : 0][r] = acc[i][j][r] + (BF3 ? accc[BF3 ? i : 0][BF3 ? j : 0][r] : 0.f);
                            acc[i][j][r] = 0.f;
                            if (BF3) accc[BF3 ? i : 0][BF3 ? j : 0][r] = 0.f;
                        }
            }
        }
        if constexpr (KD == 32 && NBUF == 2) {
            // ---- round 5: the pipelined bf16x3 loop (variant 6).  Two LDS buffers and two register stages: while the matrix pipe works on
            // slab k out of buffer k & 1, THE SAME WAVE splits slab k + 1 (in registers since the previous iteration) and writes its planes
            // into the other buffer, and the global loads of slab k + 2 are in flight - one barrier per slab, and the split's vector
            // instructions / LDS stores are placed between the MFMAs (an MFMA occupies the matrix pipe for 32 cycles; the wave issues
            // ~8 two-cycle vector instructions in its shadow) instead of in a phase of their own behind a barrier.  The store is
            // unconditional (branch-free block): behind the last slab it writes stale registers into the buffer nobody reads again.
            // Same products in the same order as the one-buffer loop: bit-identical results.
            float4 ra2[AJ], rb2[BJ];
            load_tile_to(ra, rb, nkt > 0);
            load_tile_to(ra2, rb2, nkt > 1);
            if (src == 0) compute_rowoff();
            store_tile_from(0, ra, rb);
            __syncthreads();
            // One slab = 12 x TM x TN MFMA steps (two 16-deep halves x six products, the order of compute_bf3); the split of the next slab is
            // cut into pieces of one item half (two elements: 11 vector instructions) that follow the MFMA steps one by one, an item's
            // three 8-byte LDS stores after its second half; scheduling fences keep the pieces where they are put.
            constexpr int NMF = 12 * TM * TN;
            constexpr int NPC = 2 * (AJ + BJ);  // split pieces
            auto slab = [&](int cbuf, const float4* sa, const float4* sb) {
                uint4 fa[2][3][TM], fb[2][3][TN];  // fragments [half][hi | mid | lo]
                auto read_frags = [&](int half) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        const unsigned* q = reinterpret_cast<const unsigned*>(As) + (cbuf * 3 * BM + wm * TM * 32 + i * 32 + frow) * LDPW + swz(frow, half * 8 + (lane >> 5) * 4);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) fa[half][pl][i] = *reinterpret_cast<const uint4*>(q + pl * BM * LDPW);
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned* q = reinterpret_cast<const unsigned*>(Bs) + (cbuf * 3 * BN + wn * TN * 32 + j * 32 + frow) * LDPW + swz(frow, half * 8 + (lane >> 5) * 4);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) fb[half][pl][j] = *reinterpret_cast<const uint4*>(q + pl * BN * LDPW);
                    }
                };
                unsigned hp[AJ + BJ][2], mp[AJ + BJ][2], lp[AJ + BJ][2];
                auto split_piece = [&](int pc) {
                    const int it = pc >> 1, hf = pc & 1;
                    const float4& v = it < AJ ? sa[it < AJ ? it : 0] : sb[it < AJ ? 0 : it - AJ];
                    const float a = hf ? v.z : v.x, b = hf ? v.w : v.y;
                    hp[it][hf] = sgx_pack_bf16(a, b);
                    const float r0 = a - sgx_u2f(hp[it][hf] << 16), r1 = b - sgx_u2f(hp[it][hf] & 0xffff0000u);
                    mp[it][hf] = sgx_pack_bf16(r0, r1);
                    const float s0 = r0 - sgx_u2f(mp[it][hf] << 16), s1 = r1 - sgx_u2f(mp[it][hf] & 0xffff0000u);
                    lp[it][hf] = sgx_pack_bf16(s0, s1);
                    if (hf) {
                        const bool isa = it < AJ;
                        const int row = lrow + RPP * (isa ? it : it - AJ);
                        const int rows = isa ? BM : BN;
                        if ((isa ? BM % RPP == 0 : BN % RPP == 0) || row < rows) {
                            unsigned* d = reinterpret_cast<unsigned*>(isa ? As : Bs) + ((cbuf ^ 1) * 3 * rows + row) * LDPW + swz(row, chunk4 >> 1);
                            *reinterpret_cast<uint2*>(d) = make_uint2(hp[it][0], hp[it][1]);
                            *reinterpret_cast<uint2*>(d + rows * LDPW) = make_uint2(mp[it][0], mp[it][1]);
                            *reinterpret_cast<uint2*>(d + 2 * rows * LDPW) = make_uint2(lp[it][0], lp[it][1]);
                        }
                    }
                };
                read_frags(0);
                read_frags(1);
                sgx_sched_fence();
                // MFMA order = compute_bf3's (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid -> corrections; hi*hi -> leading).  The five correction
                // products of a half chain into one accumulator and must stay back to back: two MFMAs on the same accumulator run on a
                // forwarding path, and ONE vector instruction between them costs ~43 cycles (MI355X_MICROARCH.md; r5c: the first form of
                // this loop - a split piece after every MFMA - was 7 % slower than the one-buffer loop on the deep problems).  So the split
                // pieces come in two groups, each AHEAD of a half's six MFMAs: they issue while the previous half's chain is still in the
                // matrix pipe, their LDS stores have long landed when the barrier comes, and the wave reaches the barrier with its last
                // MFMAs still executing.  SGX_PIN2 ties a register of each side into an empty volatile asm - the only ordering the
                // optimiser honours for pure instructions (scheduling fences alone do not hold IR-level code motion).
                int pc = 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    static_assert(TM == 1 && TN == 1, "the pipelined loop is laid out for one 32x32 block per wave");
                    sgx_f32x16& cc = accc[0][0];
                    SGX_PIN2(acc[0][0], cc);
                    sgx_sched_fence();
#pragma unroll
                    for (; pc < (half + 1) * NPC / 2; ++pc) split_piece(pc);
                    sgx_sched_fence();
                    SGX_PIN2(acc[0][0], cc);
                    cc = sgx_mfma_bf16(fa[half][2][0], fb[half][0][0], cc);
                    cc = sgx_mfma_bf16(fa[half][0][0], fb[half][2][0], cc);
                    cc = sgx_mfma_bf16(fa[half][1][0], fb[half][1][0], cc);
                    cc = sgx_mfma_bf16(fa[half][1][0], fb[half][0][0], cc);
                    cc = sgx_mfma_bf16(fa[half][0][0], fb[half][1][0], cc);
                    acc[0][0] = sgx_mfma_bf16(fa[half][0][0], fb[half][0][0], acc[0][0]);
                    SGX_PIN2(acc[0][0], cc);
                    sgx_sched_fence();
                }
                SGX_PIN2(acc[0][0], accc[0][0]);  // (the second chain stays ahead of the barrier and of the next trip's address arithmetic)
            };
            // (Straight-line body, two slabs per trip: the loads are issued unconditionally - out of bounds, i.e. zeros, behind the last slab -
            // and an odd slab count is rounded up with one all-zero slab (x + 0: the results stay bit-identical), so that the loop is ONE
            // basic block and the compiler's vmcnt waits are exact: each split waits for the OLDER register set only.)
            for (int kt = 0; kt < nkt; kt += 2) {
                load_tile_to(ra, rb, kt + 2 < nkt);
                slab(0, ra2, rb2);
                __syncthreads();
                load_tile_to(ra2, rb2, kt + 3 < nkt);
                slab(1, ra, rb);
                __syncthreads();
            }
            continue;
        }
        if constexpr (NBUF == 3) {
            // ---- round 5, variant 11: reductions of at most FOUR slabs (1x1 layers with <= 128 channels - the fp32-pipe launches of the
            // step) issue the loads of ALL their slabs before anything else.  The one-ahead loop waits for memory once per slab, and a
            // workgroup that lives for three slabs between a prologue and an epilogue spends most of its life in those waits (r5final:
            // these launches run at ~40 % of the HBM rate that bounds them).  64 more registers; same products in the same order.
            float4 sa[4][AJ], sb[4][BJ];
#pragma unroll
            for (int q = 0; q < 4; ++q) load_tile_to(sa[q], sb[q], q < nkt);
            compute_rowoff();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nkt) {  // (uniform)
                    store_tile_from(0, sa[q], sb[q]);
                    __syncthreads();
                    compute_f32(0, 0);
                    compute_f32(0, 16);
                    __syncthreads();
                }
            }
            continue;
        }
        if constexpr (PP != 0) {
            // ---- round 6: the ping-pong loop.  Phase ph: group g is at q = ph - g of its own sequence STAGE(0) COMPUTE(0) STAGE(1) COMPUTE(1) ...
            // (even q: stage slab q / 2 out of the registers and issue the loads of slab q / 2 + 1; odd q: the MFMAs of slab q / 2) - the
            // groups are one phase apart, so every phase pairs one group's matrix work with the other's memory / vector work; one
            // workgroup-wide barrier per phase orders a group's own LDS hand-overs (stores -> reads -> next stores).  Group 0 idles in the
            // last phase, group 1 in the first: both then run the epilogue with matching barriers.
            if (nkt > 0) load_tile();
            compute_rowoff();
            const int nph = 2 * nkt + 1;
            for (int ph = 0; ph < nph; ++ph) {
                const int q = ph - grp;
                if (q >= 0 && q < 2 * nkt) {
                    if ((q & 1) == 0) {
                        if (!IGL(2) || q == 0) store_tile(0);
                        if ((q >> 1) + 1 < nkt) load_tile();
                    } else if (!IGL(4)) {
                        compute_bf3(0, 0);
                        compute_bf3(0, 1);
                    }
                }
                __syncthreads();
            }
            continue;
        }
        if (nkt > 0) load_tile();
        if constexpr (WPR) {
            if (nkt > 0) {
                load_bfrags(0);
                load_bfrags(1);
            }
        }
        if (src == 0) compute_rowoff();
        if (nkt > 0) store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            if (KD == 32) {
                // one LDS buffer: the next slab travels in registers under 16 x TM x TN MFMAs and is written after every wave has read this one
                if (kt + 1 < nkt) load_tile();
                if constexpr (WPR) {
                    // each half's fragment registers are refilled for the next slab as soon as this slab's MFMAs have read them
                    compute_bf3(0, 0);
                    if (kt + 1 < nkt) load_bfrags(0);
                    compute_bf3(0, 1);
                    if (kt + 1 < nkt) load_bfrags(1);
                } else if (BF3) {
                    if (!IGL(4)) {
                        compute_bf3(0, 0);
                        compute_bf3(0, 1);
                    }
                } else {
                    compute_f32(0, 0);
                    compute_f32(0, 16);
                }
                __syncthreads();
                if (kt + 1 < nkt) {
                    if (!IGL(2)) store_tile(0);
                    __syncthreads();
                }
                continue;
            }
            const int buf = kt & 1;
            if (kt + 1 < nkt) load_tile();  // global loads in flight under the MFMA block
            // (bf16x3: a second register stage - slabs fetched two iterations ahead, counted vmcnt - was measured: no gain, r1z/r1z2)
            if (BF3) compute_bf3(buf, 0);
            else compute_f32(buf, 0);
            if (kt + 1 < nkt) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    if (BF3) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += accc[BF3 ? i : 0][BF3 ? j : 0][r];
    }
    if constexpr (PH2 == 2) {
        // ---- two-output epilogue (QARepVGG forward): y = acc2 -> Y, u = acc + bias2 -> Y2 (same strides), and the five per-channel sums both
        // BatchNorms of the block are finalised from: sum y, y^2, u0, u0^2, y*u0 with u0 = u - bias2 (rows outside the image are exact zeros
        // in both accumulators, so the sums need no row mask; the finalize kernel adds the bias terms in fp64).  The sums are taken in the
        // accumulator layout (a lane owns one column and 16 rows of a 32x32 sub-tile), the stores after the usual transpose through LDS.
        float* const stage = smem + wave * (32 * 32);
        const int sr = lane >> 3, sc4 = (lane & 7) * 4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float y = acc2[ACC2 ? i : 0][ACC2 ? j : 0][r], u = acc[i][j][r];
                    st[0] += y; st[1] += y * y; st[2] += u; st[3] += u * u; st[4] += y * u;
                }
#pragma unroll
            for (int t = 0; t < 5; ++t) st[t] += __shfl_xor(st[t], 32);
            if (lane < 32) {
#pragma unroll
                for (int t = 0; t < 5; ++t) red[(t * WM + wm) * BN + wn * TN * 32 + j * 32 + lane] = st[t];
            }
            const int col = n0 + wn * TN * 32 + j * 32 + sc4;
            const bool colok = col < p.Nout;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias2 && colok) bv = sgx_ld4(p.bias2 + col);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = o == 0 ? acc2[ACC2 ? i : 0][ACC2 ? j : 0][r] : acc[i][j][r];
                    sgx_wave_lds_sync();  // the staging patch is private to the wave
                    long long offq[4];
                    float4 vq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // (all eight LDS reads of the block ahead of its stores: see the one-output epilogue)
                        offq[q] = rowoff[wm * TM * 32 + i * 32 + q * 8 + sr];
                        vq[q] = sgx_ld4(stage + (q * 8 + sr) * 32 + sc4);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long off = offq[q];
                        float4 v = vq[q];
                        if (off >= 0 && colok) {
                            if (o == 1) {
                                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            }
                            sgx_st4((o == 0 ? p.Y : p.Y2) + off + col, v);
                        }
                    }
                    sgx_wave_lds_sync();  // the staging patch is private to the wave
                }
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n0 + tid;
            if (col < p.Nout) {
#pragma unroll
                for (int pl = 0; pl < 5; ++pl) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) t += red[(pl * WM + w) * BN + tid];
                    p.stat_partials[((long)pl * p.stat_nblk + mtile) * p.Nout + col] = t;
                }
            }
        }
        return;
    }
    // ---- epilogue: bias + addend + accumulate + activation, optional BN partial statistics --------------------------
    // The MFMA accumulator layout gives a lane ONE column and 16 scattered rows (4-byte stores, 128-byte runs).  Each wave
    // therefore transposes its 32x32 sub-tiles through a private 4 KB LDS patch (the operand slabs are free now) so that
    // a lane owns 4 consecutive columns of one row: 16-byte loads/stores, 8 lanes covering a 128-byte row segment, and
    // the addend / accumulate reads vectorised the same way.  Layers with few input channels are bound by exactly this
    // traffic, not by the matrix pipe.
    float* const stage = smem + wave * (32 * 32);
    const int sr = lane >> 3, sc4 = (lane & 7) * 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int coll = wn * TN * 32 + j * 32 + sc4;  // column inside the tile
        const int col = n0 + coll;
        const bool colok = col < p.Nout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && colok) {
            if (p.vec) bv = sgx_ld4(p.bias + col);
            else {
                bv.x = p.bias[col];
                if (col + 1 < p.Nout) bv.y = p.bias[col + 1];
                if (col + 2 < p.Nout) bv.z = p.bias[col + 2];
                if (col + 3 < p.Nout) bv.w = p.bias[col + 3];
            }
        }
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
        const bool bnr = p.nreq > 0;
        BnReqLane rql;
        rql.rq = -1;
        if (bnr && colok) rql = sgx_bnreq_lane(p, col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][r];
            sgx_wave_lds_sync();  // the staging patch is private to the wave
            // (Round 5: the four row offsets and the four staged rows of a block come out of LDS TOGETHER, ahead of the per-row work.  The
            // per-row form read the offset -> waited -> branched -> read the staged row -> waited -> stored, four times over: two exposed
            // LDS round trips per row, 1000-1500 cycles per block in the stamps of tools/pconv_timing.py, r5t.)
            long long offq[4];
            float4 vq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                offq[q] = rowoff[wm * TM * 32 + i * 32 + q * 8 + sr];
                vq[q] = sgx_ld4(stage + (q * 8 + sr) * 32 + sc4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rowl = q * 8 + sr;
                const long long off = offq[q];
                float4 v = vq[q];
                if (off >= 0 && colok) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    float* yp = p.Y + off + col;
                    if (p.vec) {
                        if (p.addend) {
                            float4 u = sgx_ld4(p.addend + off + col);
                            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                        }
                        if (PH2 == 1 && p.addend2) {
                            const float4 u = sgx_ld4(p.addend2 + rowoff2[wm * TM * 32 + i * 32 + rowl] + col);
                            const float sc = p.addend2_scale * (p.addend2_scale_dev ? p.addend2_scale_dev[0] : 1.f);
                            v.x += sc * u.x; v.y += sc * u.y; v.z += sc * u.z; v.w += sc * u.w;
                        }
                        if (p.accumulate) {
                            float4 u = sgx_ld4(yp);
                            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                        }
                        if (bnr) {
                            if (rql.rq >= 0) sgx_bnreq_acc(rql, v, sgx_ld4(rql.tp + rowoffT[rql.rq][wm * TM * 32 + i * 32 + rowl]), cs, cq);
                        } else {
                            cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                            cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                        }
                        if (!IGL(16)) sgx_st4(yp, make_float4(sgx_act(v.x, p.act), sgx_act(v.y, p.act), sgx_act(v.z, p.act), sgx_act(v.w, p.act)));
                    } else {
                        float e[4] = {v.x, v.y, v.z, v.w};
                        float* se[4] = {&cs.x, &cs.y, &cs.z, &cs.w};
                        float* qe[4] = {&cq.x, &cq.y, &cq.z, &cq.w};
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            if (col + t < p.Nout) {
                                float w = e[t];
                                if (p.addend) w += p.addend[off + col + t];
                                if (p.accumulate) w += yp[t];
                                *se[t] += w;
                                *qe[t] += w * w;
                                yp[t] = sgx_act(w, p.act);
                            }
                        }
                    }
                }
            }
            sgx_wave_lds_sync();  // the staging patch is private to the wave
        }
        if (p.stat_partials || bnr) {
            // the 8 lanes with equal (lane & 7) hold the same 4 columns: fold them, lanes 0-7 publish
            float vals[8] = {cs.x, cs.y, cs.z, cs.w, cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                vals[t] += __shfl_xor(vals[t], 8);
                vals[t] += __shfl_xor(vals[t], 16);
                vals[t] += __shfl_xor(vals[t], 32);
            }
            if (lane < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    red[(0 * WM + wm) * BN + coll + t] = vals[t];
                    red[(1 * WM + wm) * BN + coll + t] = vals[4 + t];
                }
            }
        }
    }
    if (p.stat_partials || p.nreq > 0) {
        __syncthreads();
        if (tid < BN && (!PP || active)) {
            int col = n0 + tid;
            if (col < p.Nout) {
                float s = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    s += red[(0 * WM + w) * BN + tid];
                    q += red[(1 * WM + w) * BN + tid];
                }
                if (p.nreq > 0) sgx_bnreq_publish(p, col, mtile, s, q);
                else {
                    p.stat_partials[(long)mtile * p.Nout + col] = s;
                    p.stat_partials[((long)p.stat_nblk + mtile) * p.Nout + col] = q;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3x3 convolution (stride-1 forward and data gradient, and the parity classes of a stride-2 data gradient) from an LDS-resident input
// PATCH, bf16x3 arithmetic ("pconv").
//
// Why a second kernel: on the bf16 matrix pipe the six cross products of a 16-deep step cost 192 cycles where the fp32 pipe needs 512,
// and the slab-per-tap loop above cannot cash that in - every input element is staged (global -> VGPR -> split -> 3 LDS stores) once
// PER TAP, and the VGPR->LDS store path (~80 B/clk/CU, MI355X_MICROARCH.md LDS table) plus the split's VALU work bound the loop at about
// half the bf16x3 rate (measured r1y-r2: 1.15x over fp32 where the matrix pipe alone would give 2.7x).  Here a workgroup owns a 2-D tile
// of 8 x 16 output pixels (one image) and, per 16-channel chunk, stages the 10 x 18 input patch ONCE (split into three bf16 planes)
// together with the filter slabs of ALL taps of the chunk; the taps then read their MFMA A-fragments from the patch (a tap is an LDS
// address offset) and run back to back - one barrier pair per CHUNK (108 MFMAs per wave for 3x3), none per tap, and no global-memory
// wait inside a chunk: the next chunk's patch and filters travel in registers from the start of this chunk's MFMAs.
// (r3b-r3d measured the first form - a double-buffered filter slab per tap, one barrier per tap, 32-channel chunks - at 35 % matrix-pipe
// utilisation: 30-50 % of the wave cycles parked in s_waitcnt / s_barrier.  A wave's loads return in issue order, so every per-tap wait
// for a short filter load also waited for the long patch prefetch issued behind it.)
// Zero padding = the hardware bounds check of the patch loads (rows / columns outside the image arrive as zeros) - no per-tap masks.
//
// LDS image: plane[3][pixel][16 bf16]; the two 16-byte halves of a pixel are swapped on pixels with bit 3 set, so that the ds_read_b128
// of 16 lanes with 16 distinct (pixel mod 16) hit 16 distinct 4-bank groups; MFMA row r of a 32-row sub-tile is output pixel (ty, tx) =
// (2s + r/16, (r - 2 ty) mod 16): the rotation makes (patch pixel index mod 16) = (r mod 16) + const for every tap, i.e. conflict-free
// fragment reads with the 18-pixel patch pitch (measured: SQ_LDS_BANK_CONFLICT = 0).  Filter slabs: plane[3][tap][n][16 bf16], same swap by n.
// Arithmetic: the bf16x3 scheme of igemm_kernel<MATH = 1> (round-to-nearest split, leading products and corrections in separate
// accumulators).  PH2 as in igemm_kernel: 1 = second source (1x1, its own tensor) into the same accumulator - the QARepVGG data
// gradient; 2 = second filter on the centre tap into a second output - the QARepVGG forward pair, with the five BatchNorm moments.
// ------------------------------------------------------------------------------------------------
// Phase timing of the patch kernel (measurement builds only: -DSGX_PCONV_TIMING[=2], tools/pconv_timing.py, tools/visits/r5_visit12.sh): lane 0
// of wave 0 of every workgroup stamps s_memtime at ten points (=2: a second set inside the two-output epilogue) and leaves the stamps in
// a caller-provided buffer [workgroups][16].  The product build compiles none of it.
#ifdef SGX_PCONV_TIMING
__device__ unsigned long long* g_pc_timing = nullptr;
extern "C" int32_t sgx_debug_set_pconv_timing(void* buf) {
    unsigned long long* b = (unsigned long long*)buf;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_pc_timing), &b, sizeof(b)) == hipSuccess ? SGX_OK : SGX_ERR_HIP;
}
#define PC_T(i) (pc_ts[i] = __builtin_readcyclecounter())
#define PC_TOUT()                                                                                      \
    do {                                                                                               \
        PC_T(9);                                                                                       \
        if (threadIdx.x == 0 && g_pc_timing)                                                           \
            for (int q_ = 0; q_ < 12; ++q_) g_pc_timing[(long)blockIdx.x * 16 + q_] = pc_ts[q_];       \
    } while (0)
#else
#define PC_T(i) ((void)0)
#define PC_TOUT() ((void)0)
#endif
#if defined(SGX_PCONV_TIMING) && SGX_PCONV_TIMING == 2
#define PC_E(i) (pc_ts[i] = __builtin_readcyclecounter())
#else
#define PC_E(i) ((void)0)
#endif
#define PC_TH 8
#define PC_TW 16
#define PC_PW (PC_TW + 2)              // patch pitch (pixels)
#define PC_NPIX ((PC_TH + 2) * PC_PW)  // 180 patch pixels
#define PC_KC 16                       // channels per chunk
template <int BN, int WM, int WN, int PH2, bool FPIPE = true, bool WPL = false>
__global__ __launch_bounds__(WM * WN * 64, BN == 32 ? 3 : 2) void pconv_kernel(IgemmParams p) {
    static_assert(4 % WM == 0, "WM divides the four 32-row sub-tiles");
    if (p.prio) SGX_WAVE_PRIO(1);
    constexpr int NTH = WM * WN * 64;
    constexpr int BM = PC_TH * PC_TW;           // 128 output pixels = 4 sub-tiles of 32 MFMA rows
    constexpr int TM = 4 / WM, TN = BN / (32 * WN);
    static_assert(TM >= 1 && TN >= 1 && TN * WN * 32 == BN, "bad tile");
    constexpr int KC = PC_KC, ROWB = KC * 2, C4 = KC / 4;   // bytes per pixel / filter row and plane; float4 items per row
    constexpr int NF = PH2 == 2 ? 10 : 9;                    // filter slabs per chunk (nine taps [+ the second filter])
    // staging items: every thread takes AR patch items and BR filter items per chunk; slots past the real data are padding (branch-free)
    // WPL (pre-split filter planes, IgemmParams::Wp): the filter items are 16-byte pieces of the three planes, copied global -> register ->
    // LDS: 3 x NF slab images of BN x 2 pieces per chunk instead of NF x BN x 4 float4 items that each cost a 22-instruction split.  A slab
    // image (one plane of one tap: BN rows x 32 bytes) is a contiguous run in the planes and in LDS, and NTH / (2 BN) of them are copied per
    // pass - which image a thread works on is the same for its whole wave (scalar arithmetic, no per-item offset registers).
    constexpr int SPP = NTH / (BN * 2);                      // WPL: slab images per pass
    constexpr int NSL = 3 * NF;                              // WPL: slab images per chunk
    static_assert(!WPL || (NTH % (BN * 2) == 0 && (BN * 2) % 64 == 0), "a wave copies pieces of one slab image");
    constexpr int AR = (PC_NPIX * C4 + NTH - 1) / NTH, BR = WPL ? (NSL + SPP - 1) / SPP : (NF * BN * C4 + NTH - 1) / NTH;
    constexpr int NPIXP = AR * NTH / C4;                     // patch pixel slots incl. padding (192)
    constexpr int NROWP = WPL ? NF * BN : BR * NTH / C4;     // filter row slots incl. padding
    constexpr int A_PLANE = NPIXP * ROWB, B_PLANE = NROWP * ROWB;
    constexpr int A_BYTES = 3 * A_PLANE, B_BYTES = 3 * B_PLANE;
    constexpr int STAGE_BYTES = WM * WN * 32 * 32 * 4;
    constexpr int SMEM_BYTES = (A_BYTES + B_BYTES) > STAGE_BYTES ? (A_BYTES + B_BYTES) : STAGE_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_BYTES];
    unsigned char* const As = smem_raw;
    unsigned char* const Bs = smem_raw + A_BYTES;
    float* const smem = reinterpret_cast<float*>(smem_raw);
    __shared__ long long rowoff[BM];
    __shared__ long long rowoff2[PH2 == 1 ? BM : 1];
    __shared__ long long rowoffT[PH2 == 2 ? 1 : SGX_MAX_BN_REQ][PH2 == 2 ? 1 : BM];
    __shared__ float red[(PH2 == 2 ? 5 : 2) * WM * BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SGX_PCONV_TIMING
    unsigned long long pc_ts[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    PC_T(0);
    const int wm = wave / WN, wn = wave % WN;
    const int bid = blockIdx.x;
    const int lin = (bid & 7) * p.chunk + (bid >> 3);
    if (lin >= p.nblk) return;
    const int mtile = sgx_fdiv(lin, p.fd_nt), ntile = lin - mtile * p.nt;
    const int n0 = ntile * BN;
    const int tiles_x = (p.Wa + PC_TW - 1) / PC_TW, tiles_y = (p.Ha + PC_TH - 1) / PC_TH;
    const int img = sgx_fdiv(mtile, p.fd_txy);
    const int trem = mtile - img * (tiles_x * tiles_y);
    const int trow = sgx_fdiv(trem, p.fd_tx);
    const int oy0 = trow * PC_TH, ox0 = (trem - trow * tiles_x) * PC_TW;
    const int cpt = p.C / KC;
    auto swz = [](int q, int row) { return q ^ ((row >> 3) & 1); };  // which 16-byte half of a 32-byte row holds k-half q

    // output rows: MFMA row r of sub-tile s <-> pixel (oy0 + ty, ox0 + tx), ty = 2 s + (r >> 4), tx = (r - 2 ty) & 15
    // (epilogue data: computed once the first chunk's loads are in flight, published by the chunk loop's barriers)
    auto compute_rowoff = [&]() {
      if (tid < BM) {
        const int s = tid >> 5, r = tid & 31;
        const int ty = 2 * s + (r >> 4), tx = (r - 2 * ty) & 15;
        const int a = oy0 + ty, b = ox0 + tx;
        long long off = -1;
        if (a < p.Ha && b < p.Wa) {
            off = (long long)img * p.y_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.y_ld_pix;
            if (PH2 == 1 && p.addend2) rowoff2[tid] = (long long)img * p.a2d_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.a2d_ld_pix;
            if (PH2 != 2) {
#pragma unroll
                for (int r = 0; r < SGX_MAX_BN_REQ; ++r)
                    if (r < p.nreq) rowoffT[PH2 == 2 ? 0 : r][PH2 == 2 ? 0 : tid] = (long long)img * p.req[r].t_ld_img + ((long long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.req[r].t_ld_pix;
            }
        }
        rowoff[tid] = off;
      }
    };

    // ---- source state ------------------------------------------------------------------------------------------------------------
    sgx_buf bufA, bufB, bufB2;
    int aoff[AR];   // byte offset of this thread's patch items at chunk 0 (-1: outside the image / padding slot)
    int boff[WPL ? 1 : BR];   // byte offset of this thread's filter items at chunk 0 (-1: no such filter row / tap); bit 30 set: second filter (bufB2)
    int wp_lane = -1, wp_lds = 0;  // WPL: this thread's piece inside a slab image - planes byte offset (-1: no such filter row) and LDS byte offset
    int taps_w, dh0_, dw0_, dstep_, ntaps;
    int bstep = 0;  // WPL: bytes from one channel chunk's planes to the next (primary filter)
    auto setup_src = [&](const float* A, const float* Wt, int Hin, int Win, int Th, int Tw, int dh0, int dw0, int dstep, long a_ld_pix, long a_ld_img,
                         long w_ld_n, long a_bytes, long w_bytes, const unsigned char* Wp, long wp_bytes) {
        bufA = sgx_make_buf(A + (long)img * a_ld_img, a_bytes - (long)img * a_ld_img * 4);
        bufB = WPL ? sgx_make_buf(Wp, wp_bytes) : sgx_make_buf(Wt, w_bytes);
        if (PH2 == 2) bufB2 = WPL ? sgx_make_buf(p.Wp2, p.wp2_bytes) : sgx_make_buf(p.Wt2, p.w2_bytes);
        ntaps = Th * Tw;
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int idx = tid + NTH * r;
            const int pp = idx / C4, c4 = (idx - pp * C4) * 4;
            const int py = pp / PC_PW, px = pp - py * PC_PW;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool ok = pp < PC_NPIX && iy >= 0 && iy < Hin && ix >= 0 && ix < Win;
            aoff[r] = ok ? (int)((((long)iy * Win + ix) * a_ld_pix + c4) * 4) : -1;
        }
        if constexpr (WPL) {
            // planes [chunk][plane][tap][row][32 B]: a chunk further is 3 * taps * Nout rows on.  LDS slot (n, s) of a slab image holds half
            // s ^ (n bit 3) of row n (the fragment reads' bank swizzle): the thread that fills the slot fetches that half.
            bstep = 3 * ntaps * p.Nout * ROWB;
            const int w = tid & (BN * 2 - 1), n = w >> 1, s_ = w & 1;
            wp_lane = n0 + n < p.Nout ? (n0 + n) * ROWB + (s_ ^ ((n >> 3) & 1)) * 16 : -1;
            wp_lds = w * 16;
        } else {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int idx = tid + NTH * r;
                const int row = idx / C4, c4 = (idx - row * C4) * 4;
                const int slab = row / BN, n = row - slab * BN;
                const bool nok = n0 + n < p.Nout;
                int o = -1;
                if (nok && slab < ntaps) o = (int)(((long)(n0 + n) * w_ld_n + (long)slab * p.C + c4) * 4);
                if (PH2 == 2 && nok && slab == NF - 1) o = (int)(((long)(n0 + n) * p.w2_ld_n + c4) * 4) | (1 << 30);
                boff[r] = o;
            }
        }
        taps_w = Tw; dh0_ = dh0; dw0_ = dw0; dstep_ = dstep;
    };

    float4 ra[AR], rb[BR];
    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int r = 0; r < AR; ++r) ra[r] = sgx_buf_ld4(bufA, aoff[r] >= 0 ? (unsigned)(aoff[r] + chunk * (KC * 4)) : SGX_BUF_OOB);
        if constexpr (WPL) {
            const int sl0 = sgx_uniform_i32(tid / (BN * 2));
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int sl = sl0 + SPP * r;              // slab image = plane * NF + slab (same for the whole wave)
                const int plane = sl / NF, slab = sl - plane * NF;
                const int rowstep = p.Nout * ROWB;
                if (PH2 == 2 && slab == NF - 1) {          // the second filter (one tap): its own planes
                    const bool ok = (NSL % SPP == 0 || sl < NSL) && wp_lane >= 0;
                    rb[r] = sgx_buf_ld4(bufB2, ok ? (unsigned)(wp_lane + (chunk * 3 + plane) * rowstep) : SGX_BUF_OOB);
                } else {
                    const bool ok = (NSL % SPP == 0 || sl < NSL) && slab < ntaps && wp_lane >= 0;
                    rb[r] = sgx_buf_ld4(bufB, ok ? (unsigned)(wp_lane + chunk * bstep + (plane * ntaps + slab) * rowstep) : SGX_BUF_OOB);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const unsigned o = boff[r] >= 0 ? (unsigned)((boff[r] & ~(1 << 30)) + chunk * (KC * 4)) : SGX_BUF_OOB;
                if (PH2 == 2 && boff[r] >= 0 && (boff[r] & (1 << 30))) rb[r] = sgx_buf_ld4(bufB2, o);
                else rb[r] = sgx_buf_ld4(bufB, o);
            }
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int r = 0; r < AR; ++r) {
            const int idx = tid + NTH * r;
            const int pp = idx / C4, c4 = (idx - pp * C4) * 4;
            unsigned char* const d = As + pp * ROWB + swz(c4 >> 3, pp) * 16 + ((c4 >> 2) & 1) * 8;
            uint2 h, m, l;
            sgx_split3(ra[r], h, m, l);
            *reinterpret_cast<uint2*>(d) = h;
            *reinterpret_cast<uint2*>(d + A_PLANE) = m;
            *reinterpret_cast<uint2*>(d + 2 * A_PLANE) = l;
        }
        if constexpr (WPL) {
            const int sl0 = sgx_uniform_i32(tid / (BN * 2));
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int sl = sl0 + SPP * r;  // (plane * NF + slab) * BN rows: B_PLANE = NF * BN rows, so slab images follow one another in LDS
                if (NSL % SPP == 0 || sl < NSL) sgx_st4(reinterpret_cast<float*>(Bs + sl * (BN * ROWB) + wp_lds), rb[r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < BR; ++r) {
                const int idx = tid + NTH * r;
                const int row = idx / C4, c4 = (idx - row * C4) * 4;
                unsigned char* const d = Bs + row * ROWB + swz(c4 >> 3, row) * 16 + ((c4 >> 2) & 1) * 8;
                uint2 h, m, l;
                sgx_split3(rb[r], h, m, l);
                *reinterpret_cast<uint2*>(d) = h;
                *reinterpret_cast<uint2*>(d + B_PLANE) = m;
                *reinterpret_cast<uint2*>(d + 2 * B_PLANE) = l;
            }
        }
    };

    sgx_f32x16 acc[TM][TN], acc2[TM][TN];
    constexpr bool DUAL = PH2 == 2;
    sgx_f32x16 accu[DUAL ? TM : 1][DUAL ? TN : 1], accu2[DUAL ? TM : 1][DUAL ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                acc2[i][j][r] = 0.f;
                if (DUAL) accu[DUAL ? i : 0][DUAL ? j : 0][r] = accu2[DUAL ? i : 0][DUAL ? j : 0][r] = 0.f;
            }

    // this lane's fragment rows: patch pixel of tap offset (0, 0) per sub-tile, filter row per N sub-tile
    const int frow = lane & 31, khalf = lane >> 5;
    int pbase[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int s = wm * TM + i;
        const int ty = 2 * s + (frow >> 4), tx = (frow - 2 * ty) & 15;
        pbase[i] = ty * PC_PW + tx;
    }
    // One tap = (TM + TN) x 3 fragment reads (16 bytes each) and 6 x TM x TN MFMAs.  The fragments of tap t + 1 are read from LDS BEFORE the
    // MFMAs of tap t are issued (two fragment register sets): with two waves per SIMD nothing else hides the LDS latency, and the first
    // form - read, wait, multiply, per tap - left 30 % of the wave cycles in s_waitcnt lgkmcnt (r3e counters).
    struct Frags {
        uint4 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
    };
    auto load_frags = [&](Frags& f, int slab, int tapoff) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pix = pbase[i] + tapoff;
            const unsigned char* s = As + pix * ROWB + swz(khalf, pix) * 16;
            f.ah[i] = *reinterpret_cast<const uint4*>(s);
            f.am[i] = *reinterpret_cast<const uint4*>(s + A_PLANE);
            f.al[i] = *reinterpret_cast<const uint4*>(s + 2 * A_PLANE);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wn * TN * 32 + j * 32 + frow;
            const unsigned char* s = Bs + (slab * BN + n) * ROWB + swz(khalf, slab * BN + n) * 16;
            f.bh[j] = *reinterpret_cast<const uint4*>(s);
            f.bm[j] = *reinterpret_cast<const uint4*>(s + B_PLANE);
            f.bl[j] = *reinterpret_cast<const uint4*>(s + 2 * B_PLANE);
        }
    };
    // products in "smallest first" order into (c1 = leading products, c2 = corrections), each product across all sub-tiles before the next
    // (independent accumulators back to back)
    auto mfma_tap = [&](const Frags& f, sgx_f32x16 (&c1)[TM][TN], sgx_f32x16 (&c2)[TM][TN]) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c2[i][j] = sgx_mfma_bf16(f.al[i], f.bh[j], c2[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c2[i][j] = sgx_mfma_bf16(f.ah[i], f.bl[j], c2[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c2[i][j] = sgx_mfma_bf16(f.am[i], f.bm[j], c2[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c2[i][j] = sgx_mfma_bf16(f.am[i], f.bh[j], c2[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c2[i][j] = sgx_mfma_bf16(f.ah[i], f.bm[j], c2[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) c1[i][j] = sgx_mfma_bf16(f.ah[i], f.bh[j], c1[i][j]);
    };
    // patch offset of tap t: input pixel (a + dh0 + dstep * ti, b + dw0 + dstep * tj), patch origin (oy0 - 1, ox0 - 1)
    auto tap_off = [&](int t) {
        // (t / taps_w without the scalar unit's 25-instruction division sequence per tap: taps_w is 1, 2 or 3 and t < 9)
        const int ti = taps_w == 3 ? (t * 11) >> 5 : taps_w == 2 ? t >> 1 : t, tj = t - ti * taps_w;
        return (dh0_ + dstep_ * ti + 1) * PC_PW + dw0_ + dstep_ * tj + 1;
    };

#pragma unroll
    for (int src = 0; src < (PH2 == 1 ? 2 : 1); ++src) {
        if (src == 0) setup_src(p.A, p.Wt, p.Hin, p.Win, p.Th, p.Tw, p.dh0, p.dw0, p.dstep, p.a_ld_pix, p.a_ld_img, p.w_ld_n, p.a_bytes, p.w_bytes, p.Wp, p.wp_bytes);
        else {
            if (!p.A2) break;
            setup_src(p.A2, p.Wt2, p.Hin2, p.Win2, p.Th2, p.Tw2, p.dh02, p.dw02, p.dstep2, p.a2_ld_pix, p.a2_ld_img, p.w2_ld_n, p.a2_bytes, p.w2_bytes, p.Wp2, p.wp2_bytes);
        }
        if (src == 0) PC_T(1);
        load_chunk(0);
        if (src == 0) compute_rowoff();
        if (src == 0) PC_T(2);
        for (int chunk = 0; chunk < cpt; ++chunk) {
            // every wave is past its last fragment read of the previous chunk (barrier below); this chunk has been travelling in registers
            store_chunk();
            if (src == 0 && chunk == 0) PC_T(3);
            if (src == 0 && chunk == 1) PC_T(7);
            __syncthreads();
            if (src == 0 && chunk == 0) PC_T(4);
            if (chunk + 1 < cpt) load_chunk(chunk + 1);
            if constexpr (!FPIPE) {  // measurement variant 8: read - wait - multiply per tap (one fragment set: fewer registers)
                Frags f;
                for (int t = 0; t < ntaps; ++t) {
                    load_frags(f, t, tap_off(t));
                    mfma_tap(f, acc, acc2);
                }
                if constexpr (DUAL) {
                    load_frags(f, NF - 1, PC_PW + 1);
                    mfma_tap(f, accu, accu2);
                }
                if (src == 0 && chunk == 0) PC_T(5);
                __syncthreads();
                if (src == 0 && chunk == 0) PC_T(6);
                continue;
            }
            Frags fa, fb;
            load_frags(fa, 0, tap_off(0));
            int t = 0;
            for (; t + 1 < ntaps; t += 2) {
                load_frags(fb, t + 1, tap_off(t + 1));
                mfma_tap(fa, acc, acc2);
                if (t + 2 < ntaps) load_frags(fa, t + 2, tap_off(t + 2));
                else if (DUAL) load_frags(fa, NF - 1, PC_PW + 1);
                mfma_tap(fb, acc, acc2);
            }
            if (t < ntaps) {  // odd tap count: the last tap is in fa
                if (DUAL) load_frags(fb, NF - 1, PC_PW + 1);
                mfma_tap(fa, acc, acc2);
                if constexpr (DUAL) mfma_tap(fb, accu, accu2);  // the centre tap again, with the second filter
            } else if constexpr (DUAL) {
                mfma_tap(fa, accu, accu2);
            }
            if (src == 0 && chunk == 0) PC_T(5);
            __syncthreads();
            if (src == 0 && chunk == 0) PC_T(6);
        }
    }
    PC_T(8);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] += acc2[i][j][r];
                if (DUAL) accu[DUAL ? i : 0][DUAL ? j : 0][r] += accu2[DUAL ? i : 0][DUAL ? j : 0][r];
            }
    // ---- epilogues: as igemm_kernel's (32x32 accumulators transposed through a per-wave LDS patch -> 16-byte stores) ------------------------
    PC_T(11);
    float* const stage = smem + wave * (32 * 32);
    const int sr = lane >> 3, sc4 = (lane & 7) * 4;
    const bool full = oy0 + PC_TH <= p.Ha && ox0 + PC_TW <= p.Wa;
    if constexpr (PH2 == 2) {
        // y = acc -> Y, u = accu + bias2 -> Y2, and the five per-channel sums both BatchNorms of the block are finalised from (rows outside
        // the image / columns outside the filter are exact zeros in both accumulators: no masks in the sums)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float st[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (a 2-D tile that sticks out of the image computes real values for its outside pixels - their neighbours are inside - so
                    // the sums take the rows of the image only)
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int ty = 2 * (wm * TM + i) + (row >> 4), tx = (row - 2 * ty) & 15;
                    if (!full && (oy0 + ty >= p.Ha || ox0 + tx >= p.Wa)) continue;
                    const float y = acc[i][j][r], u = accu[DUAL ? i : 0][DUAL ? j : 0][r];
                    st[0] += y; st[1] += y * y; st[2] += u; st[3] += u * u; st[4] += y * u;
                }
#pragma unroll
            for (int t = 0; t < 5; ++t) st[t] += __shfl_xor(st[t], 32);
            if (lane < 32) {
#pragma unroll
                for (int t = 0; t < 5; ++t) red[(t * WM + wm) * BN + wn * TN * 32 + j * 32 + lane] = st[t];
            }
            PC_E(1);
            const int col = n0 + wn * TN * 32 + j * 32 + sc4;
            const bool colok = col < p.Nout;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias2 && colok) bv = sgx_ld4(p.bias2 + col);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int o = 0; o < 2; ++o) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = o == 0 ? acc[i][j][r] : accu[DUAL ? i : 0][DUAL ? j : 0][r];
                    sgx_wave_lds_sync();  // the staging patch is private to the wave
                    if (o == 0) PC_E(2); else PC_E(4);
                    long long offq[4];
                    float4 vq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {  // (all eight LDS reads of the block ahead of its stores: see the one-output epilogue)
                        offq[q] = rowoff[(wm * TM + i) * 32 + q * 8 + sr];
                        vq[q] = sgx_ld4(stage + (q * 8 + sr) * 32 + sc4);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const long long off = offq[q];
                        float4 v = vq[q];
                        if (off >= 0 && colok) {
                            if (o == 1) {
                                v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                            }
                            sgx_st4((o == 0 ? p.Y : p.Y2) + off + col, v);
                        }
                    }
                    if (o == 0) PC_E(3); else PC_E(5);
                    sgx_wave_lds_sync();  // the staging patch is private to the wave
                }
            }
        }
        PC_T(10);
        __syncthreads();
        if (tid < BN) {
            const int col = n0 + tid;
            if (col < p.Nout) {
#pragma unroll
                for (int pl = 0; pl < 5; ++pl) {
                    float t = 0.f;
#pragma unroll
                    for (int w = 0; w < WM; ++w) t += red[(pl * WM + w) * BN + tid];
                    p.stat_partials[((long)pl * p.stat_nblk + mtile) * p.Nout + col] = t;
                }
            }
        }
        PC_TOUT();
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int coll = wn * TN * 32 + j * 32 + sc4;  // column inside the tile
        const int col = n0 + coll;
        const bool colok = col < p.Nout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && colok) bv = sgx_ld4(p.bias + col);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f), cq = cs;
        const bool bnr = p.nreq > 0;
        BnReqLane rql;
        rql.rq = -1;
        if (bnr && colok) rql = sgx_bnreq_lane(p, col);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) stage[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[i][j][r];
            sgx_wave_lds_sync();  // the staging patch is private to the wave
            // (Round 5: the four row offsets and the four staged rows of a block come out of LDS TOGETHER, ahead of the per-row work.  The
            // per-row form read the offset -> waited -> branched -> read the staged row -> waited -> stored, four times over: two exposed
            // LDS round trips per row, 1000-1500 cycles per block in the stamps of tools/pconv_timing.py, r5t.)
            long long offq[4];
            float4 vq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                offq[q] = rowoff[(wm * TM + i) * 32 + q * 8 + sr];
                vq[q] = sgx_ld4(stage + (q * 8 + sr) * 32 + sc4);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rowl = q * 8 + sr;
                const long long off = offq[q];
                float4 v = vq[q];
                if (off >= 0 && colok) {
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    float* yp = p.Y + off + col;
                    if (p.addend) {
                        float4 u = sgx_ld4(p.addend + off + col);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    if (PH2 == 1 && p.addend2) {
                        const float4 u = sgx_ld4(p.addend2 + rowoff2[(wm * TM + i) * 32 + rowl] + col);
                        const float sc = p.addend2_scale * (p.addend2_scale_dev ? p.addend2_scale_dev[0] : 1.f);
                        v.x += sc * u.x; v.y += sc * u.y; v.z += sc * u.z; v.w += sc * u.w;
                    }
                    if (p.accumulate) {
                        float4 u = sgx_ld4(yp);
                        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                    }
                    if (bnr) {
                        if (rql.rq >= 0) sgx_bnreq_acc(rql, v, sgx_ld4(rql.tp + rowoffT[rql.rq][(wm * TM + i) * 32 + rowl]), cs, cq);
                    } else {
                        cs.x += v.x; cs.y += v.y; cs.z += v.z; cs.w += v.w;
                        cq.x += v.x * v.x; cq.y += v.y * v.y; cq.z += v.z * v.z; cq.w += v.w * v.w;
                    }
                    sgx_st4(yp, make_float4(sgx_act(v.x, p.act), sgx_act(v.y, p.act), sgx_act(v.z, p.act), sgx_act(v.w, p.act)));
                }
            }
            sgx_wave_lds_sync();  // the staging patch is private to the wave
        }
        if (p.stat_partials || bnr) {
            float vals[8] = {cs.x, cs.y, cs.z, cs.w, cq.x, cq.y, cq.z, cq.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                vals[t] += __shfl_xor(vals[t], 8);
                vals[t] += __shfl_xor(vals[t], 16);
                vals[t] += __shfl_xor(vals[t], 32);
            }
            if (lane < 8) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    red[(0 * WM + wm) * BN + coll + t] = vals[t];
                    red[(1 * WM + wm) * BN + coll + t] = vals[4 + t];
                }
            }
        }
    }
    if (p.stat_partials || p.nreq > 0) {
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
                if (p.nreq > 0) sgx_bnreq_publish(p, col, mtile, s, q);
                else {
                    p.stat_partials[(long)mtile * p.Nout + col] = s;
                    p.stat_partials[((long)p.stat_nblk + mtile) * p.Nout + col] = q;
                }
            }
        }
    }
    PC_TOUT();
}

// ------------------------------------------------------------------------------------------------
// tile-shape selection
// ------------------------------------------------------------------------------------------------
struct TileCfg {
    int bm, bn;
};
// measurement aid (tools/conv_tune.py): force tile shapes / split target; 0 = heuristic
// (process-wide settings are atomics: a change made while another thread is inside a convolution call applies to the calls that follow -
// the library stays re-entrant, as include/sgx_hip.h promises)
static std::atomic<int> g_ovr_bm{0}, g_ovr_bn{0}, g_ovr_wk{0}, g_ovr_wj{0}, g_ovr_split{0}, g_ovr_var{0};
// arithmetic of the forward / data-gradient GEMMs: 0 = fp32 MFMA (exact fp32 FMA chains), 1 = bf16x3 split (see IG_LDP above)
// mode 3 (the default of round 3): fp32 matrix pipe everywhere except the 3x3 stride-1 problems on maps of 40 x 40 and larger, which run
// pconv_kernel (bf16x3 from an LDS-resident patch): 1.25-1.6x on those launches, every GPU parity test green in that mode (r3g)
// default 5 (round 4): mode 3, and every other problem whose reduction is deep enough (taps x channels >= 192: the stride-2 3x3 layers, the
// deep 1x1 layers, the QARepVGG two-branch forward / two-source data gradient) on the bf16 pipe as well - +3.9 % of the step (r4n: 737
// against 709 img/s), single convolutions 3-4x CLOSER to fp64 than the fp32 pipe (profiles/r4n_conv_arithmetic_error_probe.txt)
static std::atomic<int> g_conv_math{5};
extern "C" int32_t sgx_conv_set_math(int32_t mode) {
    SGX_CHECK_ARG(mode >= 0 && mode <= 5, "conv math mode %d (0 = fp32 MFMA, 1 = bf16x3 split, 2 = per problem, 3 = patch kernel for 3x3 stride-1, 4 = 3 + 2, 5 = 4 + the two-source / two-output launches)", mode);
    g_conv_math = mode;
    return SGX_OK;
}
extern "C" int32_t sgx_conv_get_math(void) { return g_conv_math; }
// Mode 2 picks per GEMM: the split arithmetic pays for its extra staging work (VALU split, 1.5x LDS bytes) only when the reduction
// is deep enough to be matrix-pipe bound.  Measured on all YOLO-NAS-S problems (profiles/r1y_conv_bench_bf16x3.txt vs r1n): bf16x3
// wins from a depth (taps x channels) of ~192 (1.2-1.3x on the 3x3 layers), loses 5-30 % on shallow 1x1 layers.
#define SGX_BF3_MIN_DEPTH 192
static std::atomic<int> g_bf3_min_depth{SGX_BF3_MIN_DEPTH};
extern "C" int32_t sgx_debug_set_bf3_min_depth(int32_t depth) {  // measurement: 0 = the default
    SGX_CHECK_ARG(depth >= 0, "bf3_min_depth: negative");
    g_bf3_min_depth = depth > 0 ? depth : SGX_BF3_MIN_DEPTH;
    return SGX_OK;
}
static int conv_math_for(int taps, int C) {
    const int m = g_conv_math.load(std::memory_order_relaxed);
    if (m == 3) return 0;  // the patch kernel takes the 3x3 stride-1 problems (pconv_ok), everything else stays on the fp32 pipe
    // mode 4 (measurement; r4a: +0.5 %): the patch kernel on its problems AND the per-problem rule of mode 2 for the rest
    // mode 5 (round 4): mode 4, and the two-source / two-output (QARepVGG) launches follow the same rule
    return (m == 2 || m == 4 || m == 5) ? ((long)taps * C >= g_bf3_min_depth.load(std::memory_order_relaxed) ? 1 : 0) : m;
}
// Mode 3: 3x3, stride 1, pad 1, channel counts in 16s -> pconv_kernel (bf16x3 from an LDS-resident patch).  Decidable from the descriptor,
// so that the forward statistics rows (one per 8 x 16 pixel tile and image) are known before the launch.
static int conv_variant();
static bool pconv_shape_ok(int R, int S, int stride, int pad, int C, int K, long HoWo) {
    return g_conv_math.load(std::memory_order_relaxed) >= 3 && R == 3 && S == 3 && stride == 1 && pad == 1 && C % 16 == 0 && C >= 16 && K % 4 == 0 &&
           (HoWo >= 1600 || conv_variant() == 9);
}
static int pconv_tiles(int N, int H, int W) { return N * sgx_cdiv(H, PC_TH) * sgx_cdiv(W, PC_TW); }
extern "C" int32_t sgx_debug_set_variant(int32_t v) {
    g_ovr_var = v;
    return SGX_OK;
}
extern "C" int32_t sgx_debug_set_tiles(int32_t bm, int32_t bn, int32_t wgrad_bnk, int32_t wgrad_bj, int32_t wgrad_split_target) {
    g_ovr_bm = bm; g_ovr_bn = bn; g_ovr_wk = wgrad_bnk; g_ovr_wj = wgrad_bj; g_ovr_split = wgrad_split_target;
    return SGX_OK;
}
// ---- per-problem tuning table (sgx_conv_tuning_load): what tools/conv_tune.py measured as the best (tile, variant) of a convolution
// problem replaces the heuristic for exactly that problem.  Keyed at the API level - (kind, N, H, W, C, K, R, stride, pad) - so that every
// launch of one call (the parity classes of a strided data gradient) uses the same entry, and sgx_conv2d_fwd_stat_blocks agrees with the
// forward launch.  Load before the first launch; lookups are lock-free reads of an immutable map.
struct TuneVal {
    int bm, bn, var;
};
static std::map<std::array<int, 9>, TuneVal> g_tune;
static std::mutex g_tune_mu;  // lookups copy the entry out under the lock: a concurrent sgx_conv_tuning_load cannot invalidate it
static std::atomic<int> g_tune_size{0};
struct TuneSlot {
    TuneVal v;
    bool set;
};
static thread_local TuneSlot t_slot = {{0, 0, 0}, false};
static thread_local const TuneVal* t_tune = nullptr;  // entry of the API call running on this thread
struct TuneScope {
    TuneSlot prev_slot;
    const TuneVal* prev;
    TuneScope(int kind, const sgx_conv_desc* d) : prev_slot(t_slot), prev(t_tune) {
        t_tune = nullptr;
        t_slot.set = false;
        if (g_tune_size.load(std::memory_order_acquire) && d) {
            std::lock_guard<std::mutex> g(g_tune_mu);
            auto it = g_tune.find(std::array<int, 9>{kind, d->N, d->H, d->W, d->C, d->K, d->R, d->stride, d->pad});
            if (it != g_tune.end()) {
                t_slot.v = it->second;
                t_slot.set = true;
            }
        }
        if (t_slot.set) t_tune = &t_slot.v;
    }
    ~TuneScope() {
        t_slot = prev_slot;
        t_tune = prev ? &t_slot.v : nullptr;
    }
};
extern "C" int32_t sgx_conv_tuning_load(const int32_t* entries, int32_t n) {
    SGX_CHECK_ARG(n >= 0 && (n == 0 || entries), "conv_tuning_load: bad args");
    std::map<std::array<int, 9>, TuneVal> m;
    for (int i = 0; i < n; ++i) {
        const int32_t* e = entries + 12 * i;
        SGX_CHECK_ARG(e[0] >= 0 && e[0] <= 2, "conv_tuning_load: entry %d: kind %d (0 = forward, 1 = data gradient, 2 = weight gradient)", i, e[0]);
        const bool wide = e[10] == 0 || e[10] == 32 || e[10] == 64 || e[10] == 96 || e[10] == 128;
        if (e[0] == 2)  // weight gradient: filter tile, flattened (tap, channel) tile, split target (waves over the chip; 0 = heuristic)
            SGX_CHECK_ARG((e[9] == 0 || e[9] == 32 || e[9] == 64 || e[9] == 96 || e[9] == 128) && wide && e[11] >= 0 &&
                              (e[9] == 0) == (e[10] == 0),  // all 16 tiles of {32, 64, 96, 128}^2 are instantiated
                          "conv_tuning_load: entry %d: no weight-gradient kernel (tile %dx%d, split target %d)", i, e[9], e[10], e[11]);
        else
            SGX_CHECK_ARG((e[9] == 0 || e[9] == 64 || e[9] == 128) && wide && (e[11] == 0 || e[11] == 6 || e[11] == 7 || e[11] == 11 || e[11] == 12 || e[11] == 14),
                          "conv_tuning_load: entry %d: no kernel (tile %dx%d, variant %d)", i, e[9], e[10], e[11]);
        m[std::array<int, 9>{e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[8]}] = TuneVal{e[9], e[10], e[11]};
    }
    {
        std::lock_guard<std::mutex> g(g_tune_mu);
        g_tune.swap(m);
        g_tune_size.store((int)g_tune.size(), std::memory_order_release);
    }
    return SGX_OK;
}
extern "C" int32_t sgx_conv_tuning_size(void) { return g_tune_size.load(std::memory_order_acquire); }
// the experiment switch wins over the table (measurements must see what they ask for)
static int conv_variant() {
    const int v = g_ovr_var.load(std::memory_order_relaxed);
    return v ? v : (t_tune ? t_tune->var : 0);
}

static TileCfg pick_tile_heuristic(long M, int N);
static TileCfg pick_tile(long M, int N, int math) {
    TileCfg t = pick_tile_heuristic(M, N);
    if (t_tune && t_tune->bm) t.bm = t_tune->bm;
    if (t_tune && t_tune->bn) t.bn = t_tune->bn;
    if (const int o = g_ovr_bm.load(std::memory_order_relaxed)) t.bm = o;
    if (const int o = g_ovr_bn.load(std::memory_order_relaxed)) t.bn = o;
    return t;
}
// tile of the two-source / two-output launches: the heuristic's, unless the measurement override (sgx_debug_set_tiles) names one of the tiles
// those kernels are instantiated for
static TileCfg ph2_tile(long M, int N) {
    TileCfg t = pick_tile_heuristic(M, N);
    const int bm = g_ovr_bm.load(std::memory_order_relaxed), bn = g_ovr_bn.load(std::memory_order_relaxed);
    if ((bm == 128 && bn == 96) || (bm == 128 && bn == 32) || (bm == 64 && bn == 64) || (bm == 64 && bn == 32)) t = TileCfg{bm, bn};
    return t;
}
static TileCfg pick_tile_heuristic(long M, int N) {
    // Measured (tools/conv_tune.py, profiles/r1m_conv_tune.txt: exhaustive search over the 201 conv problems of a YOLO-NAS-S
    // step): the 64x64 tile (4 waves x one 32x32 accumulator, 36 VGPRs, 22 KB LDS -> 7 workgroups per CU) wins on 83 of 140
    // forward / data-gradient problems, large ones included - occupancy and fine tail granularity beat operand reuse on this
    // matrix pipe (64 cycles per MFMA leave the LDS and L2 idle anyway).  Exceptions: output-channel counts that 64 pads badly
    // (96 = 3 x 32: a 128x32 tile; large-M 96-wide layers keep 128x96).
    const int p64 = ((N + 63) / 64) * 64, p32 = ((N + 31) / 32) * 32;
    if (p64 == p32) return TileCfg{64, 64};
    if (N % 96 == 0 && M >= 200000) return TileCfg{128, 96};
    return TileCfg{M >= 16384 ? 128 : 64, 32};
}

// 32-deep slabs (one LDS buffer) are the default wherever they apply - channel-chunked K axis, C a multiple of 32 (a ragged last chunk
// would multiply zeros for up to half a slab): measured on MI355X (profiles/r2d_conv_tune_variants.txt) they win on 3x3 and deep 1x1
// layers alike (+5-10 %, whole-line loads, half the load instructions per FLOP).  Variant 7 (tuning table / sgx_debug_set_variant) = the
// 16-deep loop; two LDS buffers with 32-deep slabs measured slower (lower occupancy) and were removed.
static bool igemm_deep_slabs(const IgemmParams& p) { return conv_variant() != 7 && p.C % 32 == 0; }
// dispatch over the eight non-flat tile shapes for one (MATH, KD, NBUF, PH2)
#define SGX_IGEMM_TILES(MATH_, KD_, NBUF_, PH2_) SGX_IGEMM_TILES_W(MATH_, KD_, NBUF_, PH2_, 0)
#define SGX_IGEMM_TILES_W(MATH_, KD_, NBUF_, PH2_, WPL_)                                                                \
    do {                                                                                                                \
        if (bm == 128 && bn == 128) launch_igemm<128, 128, 2, 2, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);      \
        else if (bm == 128 && bn == 96) launch_igemm<128, 96, 4, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);   \
        else if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);   \
        else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);   \
        else if (bm == 64 && bn == 128) launch_igemm<64, 128, 2, 2, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);   \
        else if (bm == 64 && bn == 96) launch_igemm<64, 96, 2, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);     \
        else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);     \
        else if (bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);     \
        else SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: no tile %dx%d (math %d, %d-deep slabs)", bm, bn, MATH_, KD_);         \
    } while (0)
// the two-source kernels exist for the tiles the heuristic picks (pick_tile_heuristic): overrides / table entries do not apply to them
#define SGX_IGEMM_TILES_PH2(MATH_, KD_, NBUF_, PH2_) SGX_IGEMM_TILES_PH2_W(MATH_, KD_, NBUF_, PH2_, 0)
#define SGX_IGEMM_TILES_PH2_W(MATH_, KD_, NBUF_, PH2_, WPL_)                                                            \
    do {                                                                                                                \
        if (bm == 128 && bn == 96) launch_igemm<128, 96, 4, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);        \
        else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);   \
        else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);     \
        else if (bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1, false, MATH_, KD_, NBUF_, PH2_, WPL_>(p, stream);     \
        else SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv (two sources): no tile %dx%d", bm, bn);                                \
    } while (0)
// measurement: dynamic LDS added to every implicit-GEMM launch (bytes, <= 32 KB) - fewer workgroups per CU without touching the kernel
static std::atomic<int> g_ig_lds_pad{0};
extern "C" int32_t sgx_debug_set_igemm_lds_pad(int32_t bytes) {
    SGX_CHECK_ARG(bytes >= 0 && bytes <= 32768, "igemm LDS pad %d (0 .. 32768 bytes)", bytes);
    g_ig_lds_pad = bytes;
    return SGX_OK;
}
template <int BM, int BN, int WM, int WN, bool FLAT, int MATH = 0, int KD = IG_BK, int NBUF = 2, int PH2 = 0, int WPL = 0, int PP = 0>
static void launch_igemm(IgemmParams& p, void* stream) {
    p.prio = conv_wave_prio();
    p.mt = sgx_cdiv(p.M, BM);
    p.nt = sgx_cdiv(p.Nout, BN);
    p.nblk = p.mt * p.nt;
    p.chunk = sgx_cdiv(PP ? sgx_cdiv(p.nblk, 2) : p.nblk, 8);  // (PP: a workgroup owns two consecutive tiles)
    p.fd_nt = sgx_make_fastdiv(p.nt);
    p.fd_hw = sgx_make_fastdiv(p.Ha * p.Wa);
    p.fd_wa = sgx_make_fastdiv(p.Wa);
    int grid = p.chunk * 8;
#ifdef SGX_IGEMM_LAB
    p.lab = g_ig_lab.load(std::memory_order_relaxed);
#endif
    SGX_LAUNCH((igemm_kernel<BM, BN, WM, WN, FLAT, MATH, KD, NBUF, PH2, WPL, PP>), dim3(grid), dim3(WM * WN * 64 * (PP ? 2 : 1)), (size_t)g_ig_lds_pad.load(std::memory_order_relaxed), stream, p);
}

// ---- pconv dispatch ------------------------------------------------------------------------------------------------------------------
static bool pconv_ok(const IgemmParams& p, int ph2) {
    if (g_conv_math.load(std::memory_order_relaxed) < 3) return false;
    // dense taps whose input offsets all lie in [-1, +1]: the 3x3 stride-1 forward / data gradient, and the output-parity classes of a 3x3
    // stride-2 data gradient (2x2 / 2x1 / 1x2 taps at offsets {0, +1}; the output grid is then written with stride `so`)
    if (p.si != 1 || p.Th > 3 || p.Tw > 3 || p.Th * p.Tw < 2 || (p.dstep != 1 && p.dstep != -1)) return false;
    for (int i = 0; i < p.Th; ++i)
        if (p.dh0 + p.dstep * i < -1 || p.dh0 + p.dstep * i > 1) return false;
    for (int j = 0; j < p.Tw; ++j)
        if (p.dw0 + p.dstep * j < -1 || p.dw0 + p.dstep * j > 1) return false;
    if (p.C % 16 || p.C < 16 || !p.vec) return false;
    // A forward launch that emits BatchNorm statistic rows must be a problem sgx_conv2d_fwd_stat_blocks sized those rows for (one row per
    // patch tile): the 3x3 pad-1 shape of pconv_shape_ok and nothing wider - a 2x2 / 1x3 stride-1 filter would otherwise get patch-tile row
    // counts written into a buffer allocated for cdiv(M, bm) rows
    if (p.stat_partials && !(p.Th == 3 && p.Tw == 3 && p.dh0 == -1 && p.dw0 == -1 && p.dstep == 1 && p.so == 1 && p.Nout % 4 == 0)) return false;
    // Where it pays (r3g, replay of every conv problem of a YOLO-NAS-S step): the stride-1 problems on maps of 40 x 40 and larger (1.25-1.6x
    // over the fp32 pipe).  20 x 20 maps fill their 8 x 16 tiles to 52 %, and the parity classes of a stride-2 data gradient carry
    // 2x2 / 2x1 taps only - a quarter of the reuse the patch is staged for, and 0.65x on the 768-channel layer.  Variant 9 lifts both.
    if (conv_variant() != 9 && (p.so != 1 || (long)p.Ha * p.Wa < 1600)) return false;
    if (ph2 && p.A2 && !(p.Th2 == 1 && p.Tw2 == 1 && p.dh02 == 0 && p.dw02 == 0)) return false;
    if (ph2 == 2 && (p.A2 != p.A || p.a2_ld_pix != p.a_ld_pix || p.a2_ld_img != p.a_ld_img)) return false;  // the second filter reads the same patch
    if ((long)p.Hin * p.Win * p.a_ld_pix * 4 > SGX_BUF_MAX) return false;
    return true;
}
static std::atomic<long> g_fp_hits{0};  // launches that read pre-split filter planes (see fplanes_attach)
static std::atomic<int> g_pconv_pipe32{1};  // r5f: 716.0 -> 718.8 images/s (twice each, same box)
extern "C" int32_t sgx_debug_set_pconv_pipe(int32_t on) {  // measurement switch, see launch_pconv
    g_pconv_pipe32 = on ? 1 : 0;
    return SGX_OK;
}
template <int BN, int WM, int WN, int PH2>
static void launch_pconv(IgemmParams& p, void* stream) {
    // two fragment sets where they do not cost a wave of occupancy (r3f lab: the 64-filter tile gains 4-6 %, the 32-filter tiles - three
    // workgroups per CU with one set, two with two - lose 5-15 %); measurement: variant 8 = never, 9 = always
    // (round 5: the 32-filter tiles of the one- / two-source forms take the second fragment set as well - under their launch bound of three
    // waves per SIMD they fit it without spilling, 137 / 165 registers; the two-output form does not.  sgx_debug_set_pconv_pipe(0): off)
    const int var = conv_variant();
    const bool fpipe = var == 9 || (var != 8 && (BN == 64 || (PH2 != 2 && g_pconv_pipe32.load(std::memory_order_relaxed))));
    p.mt = pconv_tiles(p.M / (p.Ha * p.Wa), p.Ha, p.Wa);
    p.nt = sgx_cdiv(p.Nout, BN);
    p.nblk = p.mt * p.nt;
    p.chunk = sgx_cdiv(p.nblk, 8);
    p.fd_nt = sgx_make_fastdiv(p.nt);
    p.fd_txy = sgx_make_fastdiv(sgx_cdiv(p.Wa, PC_TW) * sgx_cdiv(p.Ha, PC_TH));
    p.fd_tx = sgx_make_fastdiv(sgx_cdiv(p.Wa, PC_TW));
    p.stat_nblk = p.mt;
    p.prio = conv_wave_prio();
    // pre-split filter planes (fplanes_attach): every filter of the launch has them, or none is used
    const bool wpl = p.Wp && (!(PH2 && p.A2) || p.Wp2);
    if (wpl) g_fp_hits.fetch_add(1, std::memory_order_relaxed);
    // Forms whose second fragment set spills are not instantiated at all (round 6; tools/kernel_regs.py --check): the 32-filter two-output
    // form (50 / 29 spilled dwords under its three-wave bound - only the measurement switch "always" reached it) and the 64-filter
    // two-source form that splits its filters in the kernel (10 spilled dwords at 256 registers - reached with the planes switched off).
    constexpr bool PIPE_PL = !(BN == 32 && PH2 == 2), PIPE_SPLIT = PIPE_PL && !(BN == 64 && PH2 == 1);
    if (wpl) {
        if constexpr (PIPE_PL) {
            if (fpipe) {
                SGX_LAUNCH((pconv_kernel<BN, WM, WN, PH2, true, true>), dim3(p.chunk * 8), dim3(WM * WN * 64), 0, stream, p);
                return;
            }
        }
        SGX_LAUNCH((pconv_kernel<BN, WM, WN, PH2, false, true>), dim3(p.chunk * 8), dim3(WM * WN * 64), 0, stream, p);
        return;
    }
    if constexpr (PIPE_SPLIT) {
        if (fpipe) {
            SGX_LAUNCH((pconv_kernel<BN, WM, WN, PH2, true>), dim3(p.chunk * 8), dim3(WM * WN * 64), 0, stream, p);
            return;
        }
    }
    SGX_LAUNCH((pconv_kernel<BN, WM, WN, PH2, false>), dim3(p.chunk * 8), dim3(WM * WN * 64), 0, stream, p);
}
template <int PH2>
static void launch_pconv_n(IgemmParams& p, void* stream) {
    // N tile: 64 filters (two 32-wide accumulator columns x two 64-row halves) where that wastes nothing, else 32 (four 32-row waves)
    // (two outputs: ten filter slabs of 64 rows would leave room for one workgroup per CU only - 32-filter tiles, three per CU)
    if (PH2 != 2 && (p.Nout % 64 == 0 || (p.Nout > 32 && p.Nout % 32 != 0))) launch_pconv<64, 2, 2, PH2>(p, stream);
    else launch_pconv<32, 4, 1, PH2>(p, stream);
}
static int32_t run_pconv(IgemmParams& p, void* stream, int ph2) {
    if (ph2 == 0 || !p.A2) launch_pconv_n<0>(p, stream);
    else if (ph2 == 1) launch_pconv_n<1>(p, stream);
    else launch_pconv_n<2>(p, stream);
    return SGX_OK;
}

// ------------------------------------------------------------------------------------------------
// Pre-split filter planes (round 5, review item 1a for the weights).
//
// A bf16x3 launch splits its FILTER operand once per pixel tile - thousands of times per launch for a filter that changes once per
// optimizer step - and the split is a third of the staging loop's vector work (r5v: a launch whose filter is not split runs 3-11 %
// faster).  sgx_filter_planes_batch splits every registered filter ONCE per step into three bf16 planes laid out for 16-byte loads:
//     planes[ch / 16][hi | mid | lo][tap][row][16 bf16]          (rows x taps x ch x 6 bytes)
// - per (16-channel block, plane, tap) the rows of a filter tile are one contiguous run of 32-byte pieces, which is exactly what the
// patch kernel stages per chunk and what the 32-deep GEMM loop stages per slab (two blocks); the pieces are the values sgx_split3 would
// have produced in the kernel, so a launch that reads planes is bit-identical to one that splits.  (Round 4's r4h laid the planes out as
// 8-byte pieces of 24 bytes per item and lost to the load instructions it added.)
// The registry maps a filter's fp32 address to its planes.  An entry serves a launch only while (a) a training step's scope is open
// (sgx_filter_planes_scope: the network's forward / backward - nothing else may trust a device address to still mean the same tensor),
// (b) it is valid (produced since the weights last changed: sgx_filter_planes_batch validates, _invalidate drops), and (c) the launch's
// filter has the entry's shape.  Anything else falls back to splitting in the kernel.
// ------------------------------------------------------------------------------------------------
struct FplanesEntry {
    const unsigned char* planes;
    int rows, taps, ch;
    bool valid;
};
static std::mutex g_fp_mu;
static std::unordered_map<const void*, FplanesEntry> g_fp_map;
static std::atomic<int> g_fp_on{1};
// Scope depth of the CALLING thread (a process-global boolean let an inner close - nested autograd, a second network on another
// thread - switch the outer step's planes off, or a launch outside the owning step see them on): a launch looks planes up only while the
// thread that issues it is inside a training step's forward / backward.
static thread_local int t_fp_scope = 0;
__global__ void fplanes_batch_kernel(const sgx_fplanes_job* jobs) {
    __shared__ sgx_fplanes_job job;
    if (threadIdx.x == 0) job = jobs[blockIdx.y];
    __syncthreads();
    const unsigned c4n = (unsigned)job.ch / 4, T = (unsigned)job.taps, R = (unsigned)job.rows;
    const unsigned n = R * T * c4n;
    unsigned char* const dst = reinterpret_cast<unsigned char*>(job.planes);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned q = i % c4n, rt = i / c4n, t = rt % T, row = rt / T;
        uint2 h, m, l;
        sgx_split3(sgx_ld4(job.src + (size_t)i * 4), h, m, l);
        const size_t plane = (size_t)T * R * 32;
        unsigned char* const d = dst + (size_t)(q >> 2) * 3 * plane + ((size_t)t * R + row) * 32 + (q & 3) * 8;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + plane) = m;
        *reinterpret_cast<uint2*>(d + 2 * plane) = l;
    }
}
extern "C" int64_t sgx_filter_planes_bytes(int32_t rows, int32_t taps, int32_t ch) {
    if (rows <= 0 || taps <= 0 || ch <= 0 || ch % 16) return 0;
    return (int64_t)rows * taps * ch * 6;
}
extern "C" int32_t sgx_filter_planes_batch(const sgx_fplanes_job* jobs_host, const sgx_fplanes_job* jobs_dev, int32_t njobs, void* stream) {
    SGX_CHECK_ARG(jobs_host && jobs_dev && njobs > 0 && njobs <= 65535, "filter_planes_batch: bad args (njobs=%d)", njobs);
    for (int i = 0; i < njobs; ++i) {
        const sgx_fplanes_job& j = jobs_host[i];
        SGX_CHECK_ARG(j.src && j.planes && j.rows > 0 && j.taps > 0 && j.ch > 0 && j.ch % 16 == 0, "filter_planes_batch: job %d: rows / taps / ch (a multiple of 16) / pointers", i);
        SGX_CHECK_ARG(((uintptr_t)j.src % 16) == 0 && ((uintptr_t)j.planes % 16) == 0, "filter_planes_batch: job %d: 16-byte aligned pointers", i);
        SGX_CHECK_ARG(sgx_filter_planes_bytes(j.rows, j.taps, j.ch) < (1 << 30), "filter_planes_batch: job %d: planes of 1 GiB or more", i);
    }
    // grid: 64 workgroups per job for the step's largest filters (768 x 384 x 3 x 3: ~160 four-channel items per thread), fewer when every job is small
    long items = 0;
    for (int i = 0; i < njobs; ++i) items = std::max(items, (long)jobs_host[i].rows * jobs_host[i].taps * (jobs_host[i].ch / 4));
    const unsigned gx = (unsigned)std::min(64L, std::max(1L, (items + 1023) / 1024));
    SGX_LAUNCH(fplanes_batch_kernel, dim3(SGX_STRIDE_GRID(gx), (unsigned)njobs), dim3(256), 0, stream, jobs_dev);
    SGX_CHECK_LAUNCH("filter_planes_batch");
    std::lock_guard<std::mutex> lk(g_fp_mu);
    for (int i = 0; i < njobs; ++i) {
        const sgx_fplanes_job& j = jobs_host[i];
        g_fp_map[j.src] = FplanesEntry{reinterpret_cast<const unsigned char*>(j.planes), j.rows, j.taps, j.ch, true};
    }
    return SGX_OK;
}
// jobs_host == NULL: every entry (a new step of any network starts by dropping what earlier steps left valid)
extern "C" int32_t sgx_filter_planes_invalidate(const sgx_fplanes_job* jobs_host, int32_t njobs) {
    std::lock_guard<std::mutex> lk(g_fp_mu);
    if (!jobs_host) {
        g_fp_map.clear();
        return SGX_OK;
    }
    for (int i = 0; i < njobs; ++i) g_fp_map.erase(jobs_host[i].src);
    return SGX_OK;
}
extern "C" int32_t sgx_filter_planes_scope(int32_t open) {
    if (open) ++t_fp_scope;
    else if (t_fp_scope > 0) --t_fp_scope;
    return SGX_OK;
}
// 0 = every launch splits its filter while staging; 1 = planes copied into the LDS slabs; 2 = 1, and the GEMM loop's one-block-per-wave
// tiles read their filter fragments straight from the planes into registers (igemm_kernel WPL = 2)
extern "C" int32_t sgx_debug_set_filter_planes(int32_t mode) {
    SGX_CHECK_ARG(mode >= 0 && mode <= 2, "filter planes mode %d (0 off, 1 LDS copy, 2 register fragments)", mode);
    g_fp_on = mode;
    return SGX_OK;
}
static const unsigned char* fplanes_lookup(const float* w, int rows, int taps, int ch, long ld_n, long* bytes) {
    if (!w || ld_n != (long)taps * ch) return nullptr;
    std::lock_guard<std::mutex> lk(g_fp_mu);
    auto it = g_fp_map.find(w);
    if (it == g_fp_map.end()) return nullptr;
    const FplanesEntry& e = it->second;
    if (!e.valid || e.rows != rows || e.taps != taps || e.ch != ch) return nullptr;
    *bytes = (long)rows * taps * ch * 6;
    return e.planes;
}
static void fplanes_attach(IgemmParams& p, int ph2) {
    p.Wp = p.Wp2 = nullptr;
    p.wp_bytes = p.wp2_bytes = 0;
    if (t_fp_scope <= 0 || !g_fp_on.load(std::memory_order_relaxed) || p.C % 16) return;
    p.Wp = fplanes_lookup(p.Wt, p.Nout, p.Th * p.Tw, p.C, p.w_ld_n, &p.wp_bytes);
    if (ph2 && p.A2) p.Wp2 = fplanes_lookup(p.Wt2, p.Nout, p.Th2 * p.Tw2, p.C, p.w2_ld_n, &p.wp2_bytes);
}
// launches that read planes since the process started (tests: a planes-path parity check must not pass by never taking the path)
extern "C" int64_t sgx_debug_filter_planes_hits(void) { return g_fp_hits.load(); }

// ph2: 0 = one source; 1 = second source into the same accumulator; 2 = second source into a second output (see IgemmParams)
static int32_t run_igemm(IgemmParams& p, int bm, int bn, void* stream, int ph2 = 0) {
    const int T = p.Th * p.Tw;
    if (T > SGX_MAX_TAPS) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: more than %d taps", SGX_MAX_TAPS);
    if (p.w_bytes > SGX_BUF_MAX) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: weight tensor larger than 2 GiB");
    {
        const long hw = (long)p.Ha * p.Wa;
        const long imgs = (bm + hw - 1) / hw + 1;  // images a pixel tile can touch
        if (imgs * p.a_ld_img * 4 > SGX_BUF_MAX) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: one pixel tile spans more than 2 GiB of input");
        if (ph2 && p.A2 && imgs * p.a2_ld_img * 4 > SGX_BUF_MAX) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: one pixel tile spans more than 2 GiB of input");
    }
    p.vec = (p.Nout % 4 == 0 && p.y_ld_pix % 4 == 0 && p.y_ld_img % 4 == 0 && ((uintptr_t)p.Y % 16) == 0 && ((uintptr_t)p.addend % 16) == 0 &&
             ((uintptr_t)p.bias % 16) == 0)
                ? 1
                : 0;
    if (p.nreq && !p.vec) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: BatchNorm-reduce requests need 16-byte aligned outputs");
    fplanes_attach(p, ph2);
    // algorithmic work: every input element, weight and output element once (fp32); the second source adds its taps
    const double T2 = (ph2 && p.A2) ? (double)p.Th2 * p.Tw2 : 0.0;
    // (profiling class 0 = fp32-MFMA implicit GEMM, 2 = the bf16x3 patch kernel, 3 = the implicit GEMM in bf16x3 arithmetic: same algorithmic
    // FLOPs, priced separately by bench.py)
    const bool gemm_bf3 = conv_math_for(T, p.C) == 1 && (!ph2 || g_conv_math.load(std::memory_order_relaxed) == 5);
    SGX_PROF(pconv_ok(p, ph2) ? 2 : gemm_bf3 ? 3 : 0, 2.0 * (double)p.M * (double)p.Nout * (double)p.C * ((double)T + T2),
             4.0 * ((double)p.M / ((double)p.Ha * p.Wa) * p.Hin * p.Win * p.C * (ph2 == 1 && p.A2 ? 2.0 : 1.0) + (double)p.Nout * p.C * (T + T2) +
                    (double)p.M * p.Nout * (ph2 == 2 ? 2.0 : 1.0)), stream);
    const bool flat = p.C < IG_BK && T > 1;
    if (!pconv_ok(p, ph2) && p.stat_partials && p.dstep == 1 && p.so == 1 && pconv_shape_ok(p.Th, p.Tw, p.si, -p.dh0, p.C, p.Nout, (long)p.Ha * p.Wa))
        SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv (math mode 3): the statistics rows of this forward follow the patch kernel's tiles, which needs 16-byte aligned operands");
    if (pconv_ok(p, ph2)) {
        int32_t rc = run_pconv(p, stream, ph2);
        if (rc) return rc;
        SGX_CHECK_LAUNCH("pconv");
        return SGX_OK;
    }
    if (ph2 == 2 && flat && p.vec && !gemm_bf3) {
        // the two-output forward of a block with few input channels (the RGB stem, C = 4): flattened (tap, channel) K axis, fp32 pipe
        if (bm == 128) launch_igemm<128, 64, 2, 2, true, 0, IG_BK, 2, 2>(p, stream);
        else launch_igemm<64, 64, 2, 2, true, 0, IG_BK, 2, 2>(p, stream);
    } else if (ph2) {
        if (flat || !p.vec) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv (two sources): needs C >= 16 and 16-byte aligned outputs");
        const bool bf3 = gemm_bf3;  // (mode 5: the primary source's depth decides for the launch)
        if (p.C % 32 == 0) {
            if (bf3 && p.Wp && (!p.A2 || p.Wp2)) {
                g_fp_hits.fetch_add(1, std::memory_order_relaxed);
                // register fragments (mode 2, or variant 12; the tiles of one 32-filter block per wave) - by default the two-output forward
                // pair takes them (r5ac lab: -6 %) and the two-source data gradient does not (+8 %)
                const int fpm = g_fp_on.load(std::memory_order_relaxed);
                const bool wpr = (fpm == 2 || conv_variant() == 12 || (fpm == 1 && ph2 == 2 && conv_variant() != 13)) && bn <= 64;
                if (wpr && ph2 == 1) SGX_IGEMM_TILES_PH2_W(1, 32, 1, 1, 2);
                else if (wpr) SGX_IGEMM_TILES_PH2_W(1, 32, 1, 2, 2);
                else if (ph2 == 1) SGX_IGEMM_TILES_PH2_W(1, 32, 1, 1, 1);
                else SGX_IGEMM_TILES_PH2_W(1, 32, 1, 2, 1);
            } else if (bf3) {
                if (ph2 == 1) SGX_IGEMM_TILES_PH2(1, 32, 1, 1);
                else SGX_IGEMM_TILES_PH2(1, 32, 1, 2);
            } else if (ph2 == 1) SGX_IGEMM_TILES_PH2(0, 32, 1, 1);
            else SGX_IGEMM_TILES_PH2(0, 32, 1, 2);
        } else {
            if (bf3) {
                if (ph2 == 1) SGX_IGEMM_TILES_PH2(1, 16, 2, 1);
                else SGX_IGEMM_TILES_PH2(1, 16, 2, 2);
            } else if (ph2 == 1) SGX_IGEMM_TILES_PH2(0, 16, 2, 1);
            else SGX_IGEMM_TILES_PH2(0, 16, 2, 2);
        }
    } else if (gemm_bf3) {
        if (flat && bn > 64) bn = 64;
        if (flat && bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2, true, 1>(p, stream);
        else if (flat && bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, true, 1>(p, stream);
        else if (flat && bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, true, 1>(p, stream);
        else if (flat && bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1, true, 1>(p, stream);
        else if (flat) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv (bf16x3): no flat tile %dx%d", bm, bn);
        else if (igemm_deep_slabs(p) && conv_variant() == 6 && bm * bn <= 128 * 32) {
            // the pipelined loop (two LDS buffers: tiles whose two slabs fit 64 KB)
            if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, 1, 32, 2, 0>(p, stream);
            else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, 1, 32, 2, 0>(p, stream);
            else launch_igemm<64, 32, 2, 1, false, 1, 32, 2, 0>(p, stream);
        } else if (igemm_deep_slabs(p) && p.Wp && conv_variant() == 14 && ((bm == 64 && bn == 64) || (bm == 128 && bn == 32) || (bm == 64 && bn == 32))) {
            // the ping-pong loop (round 6; variant 14): two tiles per 512-thread workgroup, staging and matrix phases half a period apart
            g_fp_hits.fetch_add(1, std::memory_order_relaxed);
            if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, 1, 32, 1, 0, 1, 1>(p, stream);
            else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, 1, 32, 1, 0, 1, 1>(p, stream);
            else launch_igemm<64, 32, 2, 1, false, 1, 32, 1, 0, 1, 1>(p, stream);
        } else if (igemm_deep_slabs(p) && p.Wp) {
            g_fp_hits.fetch_add(1, std::memory_order_relaxed);
            // register fragments: mode 2, or the problem's tuning-table variant 12 (tools/conv_tune.py measures both forms per problem)
            if ((g_fp_on.load(std::memory_order_relaxed) == 2 || conv_variant() == 12) && bn <= 64) {
                if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2, false, 1, 32, 1, 0, 2>(p, stream);
                else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, 1, 32, 1, 0, 2>(p, stream);
                else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, 1, 32, 1, 0, 2>(p, stream);
                else if (bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1, false, 1, 32, 1, 0, 2>(p, stream);
                else SGX_IGEMM_TILES_W(1, 32, 1, 0, 1);
            } else SGX_IGEMM_TILES_W(1, 32, 1, 0, 1);
        }
        else if (igemm_deep_slabs(p)) SGX_IGEMM_TILES(1, 32, 1, 0);
        else SGX_IGEMM_TILES(1, 16, 2, 0);
    } else if (flat) {
        if (bn > 64) bn = 64;  // the flat variants exist for the narrow tiles only (stem layers have few output channels)
        if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2, true>(p, stream);
        else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, true>(p, stream);
        else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, true>(p, stream);
        else if (bm == 64 && bn == 32) launch_igemm<64, 32, 2, 1, true>(p, stream);
        else SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv: no flat tile %dx%d", bm, bn);
    } else if (igemm_deep_slabs(p) && conv_variant() == 11 && T * ((p.C + 31) / 32) <= 4 && bm * bn <= 128 * 64 && bn != 96 && bn != 128) {
        // all slabs up front (reductions of <= 4 slabs: shallow 1x1 layers)
        if (bm == 128 && bn == 64) launch_igemm<128, 64, 2, 2, false, 0, 32, 3, 0>(p, stream);
        else if (bm == 64 && bn == 64) launch_igemm<64, 64, 2, 2, false, 0, 32, 3, 0>(p, stream);
        else if (bm == 128 && bn == 32) launch_igemm<128, 32, 4, 1, false, 0, 32, 3, 0>(p, stream);
        else launch_igemm<64, 32, 2, 1, false, 0, 32, 3, 0>(p, stream);
    } else if (igemm_deep_slabs(p)) SGX_IGEMM_TILES(0, 32, 1, 0);  // 32-deep slabs (see igemm_kernel): whole-line loads
    else SGX_IGEMM_TILES(0, 16, 2, 0);
    SGX_CHECK_LAUNCH("igemm");
    return SGX_OK;
}
static TileCfg igemm_tile(const IgemmParams& p) {
    TileCfg t = pick_tile(p.M, p.Nout, conv_math_for(p.Th * p.Tw, p.C));
    if (p.C < IG_BK && p.Th * p.Tw > 1 && t.bn > 64) t.bn = 64;
    return t;
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
// bytes from the first element of an NHWC view to one past its last
static long view_bytes(int N, int H, int W, int C, long ld_pix, long ld_img) {
    return ((long)(N - 1) * ld_img + ((long)H * W - 1) * ld_pix + C) * 4;
}

extern "C" int32_t sgx_conv2d_fwd_stat_blocks(const sgx_conv_desc* d) {
    if (pconv_shape_ok(d->R, d->S, d->stride, d->pad, d->C, d->K, (long)d->Ho * d->Wo)) return pconv_tiles(d->N, d->Ho, d->Wo);
    TuneScope tune(0, d);
    long M = (long)d->N * d->Ho * d->Wo;
    TileCfg t = pick_tile(M, d->K, conv_math_for(d->R * d->S, d->C));
    return sgx_cdiv(M, t.bm);
}

extern "C" int32_t sgx_conv2d_fwd(const sgx_conv_desc* d, const float* x, const float* w, const float* bias,
                                  const float* addend, float* y, int32_t act, float* stat_partials, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(x && w && y, "conv fwd: null pointer");
    TuneScope tune(0, d);
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.Wt = w; p.bias = bias; p.addend = addend; p.Y = y; p.stat_partials = stat_partials;
    p.M = d->N * d->Ho * d->Wo; p.Ha = d->Ho; p.Wa = d->Wo; p.Hin = d->H; p.Win = d->W;
    p.C = d->C; p.Nout = d->K; p.Th = d->R; p.Tw = d->S;
    p.dh0 = -d->pad; p.dw0 = -d->pad; p.dstep = 1;
    p.si = d->stride; p.so = 1; p.ph = 0; p.pw = 0; p.Hout = d->Ho; p.Wout = d->Wo;
    p.a_ld_pix = d->x_ld_pix; p.a_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
    p.w_ld_n = (long)d->R * d->S * d->C;
    p.a_bytes = view_bytes(d->N, d->H, d->W, d->C, d->x_ld_pix, d->x_ld_img);
    p.w_bytes = (long)d->K * p.w_ld_n * 4;
    p.act = act; p.accumulate = 0;
    TileCfg t = pick_tile(p.M, p.Nout, conv_math_for(p.Th * p.Tw, p.C));  // the statistics rows follow the M tile: keep it in step with sgx_conv2d_fwd_stat_blocks
    p.stat_nblk = sgx_cdiv(p.M, t.bm);
    TileCfg u = igemm_tile(p);
    return run_igemm(p, t.bm, u.bn, stream);
}

// y = conv RxS(x, w) (no bias), u = conv1x1(x, w1) + bias1 with the same stride, as ONE launch: the 1x1 filter reads exactly the centre
// tap of the RxS one (pad = R / 2), so the workgroup that owns an output tile walks the taps of w into one accumulator and then the
// centre tap again with w1 into a second one.  stat5: [5][sgx_conv2d_fwd_dual_stat_blocks(d)][K] = sum y, y^2, u0, u0^2, y*u0 per row block (u0 = u - bias1).
extern "C" int32_t sgx_conv2d_fwd_dual_stat_blocks(const sgx_conv_desc* d) {
    if (pconv_shape_ok(d->R, d->S, d->stride, d->pad, d->C, d->K, (long)d->Ho * d->Wo)) return pconv_tiles(d->N, d->Ho, d->Wo);
    long M = (long)d->N * d->Ho * d->Wo;
    return sgx_cdiv(M, ph2_tile(M, d->K).bm);
}
extern "C" int32_t sgx_conv2d_fwd_dual(const sgx_conv_desc* d, const float* x, const float* w, const float* w1, const float* bias1, float* y,
                                       float* u, float* stat5, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(x && w && w1 && y && u && stat5, "conv fwd_dual: null pointer");
    SGX_CHECK_ARG(d->R == d->S && (d->R & 1) && d->pad == d->R / 2, "conv fwd_dual: odd square filter with pad = R / 2 (the 1x1 branch reads its centre tap)");
    SGX_CHECK_ARG(((uintptr_t)u % 16) == 0, "conv fwd_dual: needs a 16-byte aligned second output");
    SGX_CHECK_ARG(d->C >= IG_BK || d->K <= 64, "conv fwd_dual: C < 16 (the flattened K axis) has the 64-filter tiles only (K = %d)", d->K);
    IgemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = x; p.Wt = w; p.Y = y; p.stat_partials = stat5;
    p.M = d->N * d->Ho * d->Wo; p.Ha = d->Ho; p.Wa = d->Wo; p.Hin = d->H; p.Win = d->W;
    p.C = d->C; p.Nout = d->K; p.Th = d->R; p.Tw = d->S;
    p.dh0 = -d->pad; p.dw0 = -d->pad; p.dstep = 1;
    p.si = d->stride; p.so = 1; p.ph = 0; p.pw = 0; p.Hout = d->Ho; p.Wout = d->Wo;
    p.a_ld_pix = d->x_ld_pix; p.a_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
    p.w_ld_n = (long)d->R * d->S * d->C;
    p.a_bytes = view_bytes(d->N, d->H, d->W, d->C, d->x_ld_pix, d->x_ld_img);
    p.w_bytes = (long)d->K * p.w_ld_n * 4;
    p.act = SGX_ACT_NONE; p.accumulate = 0;
    p.A2 = x; p.Wt2 = w1; p.bias2 = bias1; p.Y2 = u;
    p.Hin2 = d->H; p.Win2 = d->W; p.Th2 = 1; p.Tw2 = 1; p.dh02 = 0; p.dw02 = 0; p.dstep2 = 1;
    p.a2_ld_pix = d->x_ld_pix; p.a2_ld_img = d->x_ld_img; p.w2_ld_n = d->C; p.a2_bytes = p.a_bytes; p.w2_bytes = (long)d->K * d->C * 4;
    TileCfg t = ph2_tile(p.M, p.Nout);
    p.stat_nblk = sgx_cdiv(p.M, t.bm);
    return run_igemm(p, t.bm, t.bn, stream, 2);
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
__global__ void dgrad_fill_kernel(IgemmParams p) {
    // a parity class no tap reaches: dx = addend (+ dx if accumulate), else 0
    const int hw = p.Ha * p.Wa;
    const long n = (long)p.M * (p.Nout / 4);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (p.Nout / 4)) * 4;
        const int m = (int)(i / (p.Nout / 4));
        const int img = m / hw, rem = m - img * hw, a = rem / p.Wa, b = rem - a * p.Wa;
        const long off = (long)img * p.y_ld_img + ((long)(a * p.so + p.ph) * p.Wout + (b * p.so + p.pw)) * p.y_ld_pix + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) v = sgx_ld4(p.bias + c4);
        if (p.addend) {
            float4 u = sgx_ld4(p.addend + off);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        if (p.accumulate) {
            float4 u = sgx_ld4(p.Y + off);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        sgx_st4(p.Y + off, v);
    }
}

extern "C" int64_t sgx_conv2d_bwd_data_workspace(const sgx_conv_desc* d) {
    return (int64_t)d->R * d->S * d->C * d->K * sizeof(float) + 256;
}

// mode 0: transpose the weights into ws, then run; mode 1: only transpose (dy/dx unused); mode 2: ws already holds the transposed
// weights of sgx_conv2d_transpose_weights (the host mirror prepares them on a side stream during the forward pass)
struct DgradSecond {  // the 1x1 branch of a QARepVGG block: dx += dgrad1x1(ds) inside the same launch
    const float* ds;
    long ld_pix, ld_img;
    const float* w1t;  // [C][K]
    const float* addend2;  // optional: dx += scale * addend2 (its own strides)
    long a2_ld_pix, a2_ld_img;
    float a2_scale;
    const float* a2_scale_dev;
};
static int32_t conv_bwd_data_impl(const sgx_conv_desc* d, const float* dy, const float* w, const float* bias, const float* addend,
                                  float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream, int mode, const DgradSecond* sec = nullptr,
                                  const sgx_bn_reduce_req* reqs = nullptr, int nreq = 0, int* rows_out = nullptr);
extern "C" int32_t sgx_conv2d_bwd_data(const sgx_conv_desc* d, const float* dy, const float* w, const float* addend,
                                       float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream) {
    return conv_bwd_data_impl(d, dy, w, nullptr, addend, dx, accumulate, ws, ws_bytes, stream, 0);
}
extern "C" int32_t sgx_conv2d_transpose_weights(const sgx_conv_desc* d, const float* w, float* wt, int64_t wt_bytes, void* stream) {
    return conv_bwd_data_impl(d, nullptr, w, nullptr, nullptr, nullptr, 0, wt, wt_bytes, stream, 1);
}
extern "C" int32_t sgx_conv2d_bwd_data_wt(const sgx_conv_desc* d, const float* dy, const float* wt, const float* addend, float* dx,
                                          int32_t accumulate, void* stream) {
    return conv_bwd_data_impl(d, dy, nullptr, nullptr, addend, dx, accumulate, const_cast<float*>(wt), sgx_conv2d_bwd_data_workspace(d), stream, 2);
}
// batched form of wtrans_kernel: blockIdx.y = job, blockIdx.x strides over the job's elements
__global__ void wtrans_batch_kernel(const sgx_wtrans_job* jobs) {
    __shared__ sgx_wtrans_job job;
    if (threadIdx.x == 0) job = jobs[blockIdx.y];
    __syncthreads();
    // (32-bit index arithmetic - a filter has < 2^31 elements - and a grid wide enough for the largest filter of the step: with 32 workgroups per
    // job and 64-bit divisions the 768 x 384 x 3 x 3 filter alone set the launch's 189 us, r4z)
    const unsigned n = (unsigned)job.C * (unsigned)job.T * (unsigned)job.K, K = (unsigned)job.K, T = (unsigned)job.T;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned k = i % K, r = i / K, t = r % T, c = r / T;
        job.wt[i] = job.w[((long)k * job.RS + job.taps[t]) * job.C + c];
    }
}
extern "C" int32_t sgx_wtrans_batch(const sgx_wtrans_job* jobs_dev, int32_t njobs, void* stream) {
    SGX_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535, "wtrans_batch: bad args (njobs=%d)", njobs);
    SGX_LAUNCH(wtrans_batch_kernel, dim3(SGX_STRIDE_GRID(256), (unsigned)njobs), dim3(256), 0, stream, jobs_dev);
    SGX_CHECK_LAUNCH("wtrans_batch");
    return SGX_OK;
}
// QARepVGG per-step weight preparation, all blocks in one launch: w1p[k][c] = alpha * w1[k][c] + (identity && k == c), and its transpose
// w1pt[c][k] (the data gradient's operand).  Folding the block's identity branch (and the alpha multiplier) into the 1x1 filter removes the
// residual read from the block's forward sweep and the `addend` read from its data gradient.
__global__ void qarep_prep_kernel(const sgx_qarep_prep_job* jobs) {
    __shared__ sgx_qarep_prep_job job;
    if (threadIdx.x == 0) job = jobs[blockIdx.y];
    __syncthreads();
    const long n = (long)job.K * job.C;
    const float a = job.alpha ? job.alpha[0] : 1.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % job.C), k = (int)(i / job.C);
        const float v = a * job.w1[i] + ((job.identity && k == c) ? 1.f : 0.f);
        job.w1p[i] = v;
        job.w1pt[(long)c * job.K + k] = v;
    }
}
extern "C" int32_t sgx_qarep_prep_batch(const sgx_qarep_prep_job* jobs_dev, int32_t njobs, void* stream) {
    SGX_CHECK_ARG(jobs_dev && njobs > 0 && njobs <= 65535, "qarep_prep_batch: bad args (njobs=%d)", njobs);
    SGX_LAUNCH(qarep_prep_kernel, dim3(SGX_STRIDE_GRID(8), (unsigned)njobs), dim3(256), 0, stream, jobs_dev);
    SGX_CHECK_LAUNCH("qarep_prep_batch");
    return SGX_OK;
}
// mode 3 of conv_bwd_data_impl records the transposes instead of launching them
struct WtransRecorder {
    sgx_wtrans_job* jobs;
    int max, n;
};
static thread_local WtransRecorder* g_wt_rec = nullptr;
extern "C" int32_t sgx_conv2d_transpose_jobs(const sgx_conv_desc* d, const float* w, float* wt, int64_t wt_bytes, sgx_wtrans_job* jobs,
                                             int32_t max_jobs, int32_t* njobs) {
    SGX_CHECK_ARG(jobs && njobs && max_jobs > 0, "transpose_jobs: null pointer");
    WtransRecorder rec{jobs, max_jobs, 0};
    g_wt_rec = &rec;
    int32_t rc = conv_bwd_data_impl(d, nullptr, w, nullptr, nullptr, nullptr, 0, wt, wt_bytes, nullptr, 1);
    g_wt_rec = nullptr;
    if (rc) return rc;
    if (rec.n > max_jobs) SGX_FAIL(SGX_ERR_BAD_ARG, "transpose_jobs: %d jobs, room for %d", rec.n, max_jobs);
    *njobs = rec.n;
    return SGX_OK;
}
// dx = conv_transpose RxS(dy, w) + conv_transpose 1x1(ds, w1) [+ addend] [+ dx] as ONE launch per output-parity class: the data gradient of
// a QARepVGG block's two convolution branches (same stride; the 1x1 branch reaches parity class (0, 0) only).  wt: the RxS weights as
// sgx_conv2d_transpose_weights lays them out; w1t: the 1x1 weights transposed, [C][K].
extern "C" int32_t sgx_conv2d_bwd_data_dual(const sgx_conv_desc* d, const float* dy, const float* wt, const float* ds, int64_t ds_ld_pix,
                                            int64_t ds_ld_img, const float* w1t, const float* addend, const float* addend2, int64_t a2_ld_pix,
                                            int64_t a2_ld_img, float a2_scale, const float* a2_scale_dev, float* dx, int32_t accumulate,
                                            void* stream) {
    SGX_CHECK_ARG(ds && w1t && d && d->K >= IG_BK && ds_ld_pix % 4 == 0, "conv bwd_data_dual: bad args (needs K >= 16)");
    SGX_CHECK_ARG(!addend2 || (d->stride == 1 && a2_ld_pix % 4 == 0 && a2_ld_img % 4 == 0 && ((uintptr_t)addend2 % 16) == 0),
                  "conv bwd_data_dual: the scaled second addend needs stride 1 and 16-byte aligned rows");
    DgradSecond sec{ds, (long)ds_ld_pix, (long)ds_ld_img, w1t, addend2, (long)a2_ld_pix, (long)a2_ld_img, a2_scale, a2_scale_dev};
    return conv_bwd_data_impl(d, dy, nullptr, nullptr, addend, dx, accumulate, const_cast<float*>(wt), sgx_conv2d_bwd_data_workspace(d), stream, 2, &sec);
}
extern "C" int32_t sgx_conv2d_bwd_data_dual_req(const sgx_conv_desc* d, const float* dy, const float* wt, const float* ds, int64_t ds_ld_pix,
                                                int64_t ds_ld_img, const float* w1t, const float* addend, const float* addend2, int64_t a2_ld_pix,
                                                int64_t a2_ld_img, float a2_scale, const float* a2_scale_dev, float* dx, int32_t accumulate,
                                                const sgx_bn_reduce_req* reqs, int32_t nreq, void* stream) {
    SGX_CHECK_ARG(ds && w1t && d && d->K >= IG_BK && ds_ld_pix % 4 == 0, "conv bwd_data_dual: bad args (needs K >= 16)");
    SGX_CHECK_ARG(!addend2 || (d->stride == 1 && a2_ld_pix % 4 == 0 && a2_ld_img % 4 == 0 && ((uintptr_t)addend2 % 16) == 0),
                  "conv bwd_data_dual: the scaled second addend needs stride 1 and 16-byte aligned rows");
    DgradSecond sec{ds, (long)ds_ld_pix, (long)ds_ld_img, w1t, addend2, (long)a2_ld_pix, (long)a2_ld_img, a2_scale, a2_scale_dev};
    return conv_bwd_data_impl(d, dy, nullptr, nullptr, addend, dx, accumulate, const_cast<float*>(wt), sgx_conv2d_bwd_data_workspace(d), stream, 2, &sec, reqs, nreq);
}
// mode 3: no launch - counts the statistic rows a launch with requests would write (*rows_out; 0 = cannot carry requests)
static int32_t conv_bwd_data_impl(const sgx_conv_desc* d, const float* dy, const float* w, const float* bias, const float* addend,
                                  float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream, int mode, const DgradSecond* sec,
                                  const sgx_bn_reduce_req* reqs, int nreq, int* rows_out) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    const bool dry = mode == 3;
    SGX_CHECK_ARG(dry || ((mode == 1 || (dy && dx)) && (mode == 2 || w)), "conv bwd_data: null pointer");
    TuneScope tune(1, d);
    SGX_CHECK_ARG(d->K % 4 == 0 && d->y_ld_pix % 4 == 0, "conv bwd_data: K and dy pixel stride must be multiples of 4");
    if (!dry && (ws_bytes < sgx_conv2d_bwd_data_workspace(d) || !ws)) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_data: workspace too small");
    SGX_CHECK_ARG(nreq >= 0 && nreq <= SGX_MAX_BN_REQ && (nreq == 0 || (reqs && mode == 2)), "conv bwd_data: bad BatchNorm-reduce requests");
    for (int r = 0; r < nreq; ++r)
        SGX_CHECK_ARG(reqs[r].t && reqs[r].scale && reqs[r].shift && reqs[r].mean && reqs[r].partials && reqs[r].c_lo >= 0 && reqs[r].c_hi <= d->C &&
                          reqs[r].c_lo < reqs[r].c_hi && reqs[r].c_lo % 4 == 0 && reqs[r].c_hi % 4 == 0 && reqs[r].t_ld_pix % 4 == 0 &&
                          reqs[r].t_ld_img % 4 == 0 && ((uintptr_t)reqs[r].t % 16) == 0 && ((uintptr_t)reqs[r].scale % 16) == 0 &&
                          ((uintptr_t)reqs[r].shift % 16) == 0 && ((uintptr_t)reqs[r].mean % 16) == 0,
                      "conv bwd_data: BatchNorm-reduce request %d: null pointer, unaligned operand or a channel range outside [0, C)", r);
    if (nreq) {  // the rows the launches below will write, before anything is launched
        int rows = 0;
        rc = conv_bwd_data_impl(d, nullptr, nullptr, nullptr, nullptr, sec ? reinterpret_cast<float*>(16) : nullptr, 0, nullptr, 0, nullptr, 3, nullptr, nullptr, 0, &rows);
        if (rc) return rc;
        if (rows <= 0) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_data: this problem cannot carry BatchNorm-reduce requests (sgx_conv2d_bwd_data_stat_blocks = 0)");
        for (int r = 0; r < nreq; ++r)
            SGX_CHECK_ARG(reqs[r].rows == rows, "conv bwd_data: request %d was allocated for %d rows, the launch writes %d (sgx_conv2d_bwd_data_stat_blocks)", r, reqs[r].rows, rows);
    }
    int rows_total = 0;
    bool rows_ok = true;  // every parity class can carry requests
    const int s = d->stride;
    float* wt = (float*)ws;
    for (int ph = 0; ph < s; ++ph)
        for (int pw = 0; pw < s; ++pw) {
            IgemmParams p;
            memset(&p, 0, sizeof(p));
            // taps of this output-parity class: filter rows r with (ph + pad - r) % s == 0, in increasing r; the input (dY)
            // row they read is (ph + pad - r)/s, i.e. it DEcreases by one per tap: dstep = -1
            TapList taps;
            memset(&taps, 0, sizeof(taps));
            int rows[SGX_MAX_TAPS], cols[SGX_MAX_TAPS], Th = 0, Tw = 0;
            for (int r = 0; r < d->R; ++r)
                if ((((ph + d->pad - r) % s) + s) % s == 0) rows[Th++] = r;
            for (int q = 0; q < d->S; ++q)
                if ((((pw + d->pad - q) % s) + s) % s == 0) cols[Tw++] = q;
            const int T = Th * Tw;
            for (int i = 0; i < Th; ++i)
                for (int j = 0; j < Tw; ++j) taps.idx[i * Tw + j] = (unsigned char)(rows[i] * d->S + cols[j]);
            const int Ha = (d->H - ph + s - 1) / s, Wa = (d->W - pw + s - 1) / s;
            if (Ha <= 0 || Wa <= 0) continue;
            p.A = dy; p.Wt = wt; p.bias = bias; p.addend = addend; p.Y = dx; p.stat_partials = nullptr;
            p.M = d->N * Ha * Wa; p.Ha = Ha; p.Wa = Wa; p.Hin = d->Ho; p.Win = d->Wo;
            p.C = d->K; p.Nout = d->C; p.Th = Th; p.Tw = Tw;
            // floor division: (ph + pad - r) is a multiple of s for the selected taps
            p.dh0 = T ? (ph + d->pad - rows[0]) / s : 0;
            p.dw0 = T ? (pw + d->pad - cols[0]) / s : 0;
            p.dstep = -1;
            p.si = 1; p.so = s; p.ph = ph; p.pw = pw; p.Hout = d->H; p.Wout = d->W;
            p.a_ld_pix = d->y_ld_pix; p.a_ld_img = d->y_ld_img; p.y_ld_pix = d->x_ld_pix; p.y_ld_img = d->x_ld_img;
            p.w_ld_n = (long)T * d->K;
            p.a_bytes = view_bytes(d->N, d->Ho, d->Wo, d->K, d->y_ld_pix, d->y_ld_img);
            p.w_bytes = (long)d->C * p.w_ld_n * 4;
            p.act = SGX_ACT_NONE; p.accumulate = accumulate;
            if (T == 0) {
                if (mode == 1) continue;
                if (dry || nreq) {  // a class no filter tap reaches still defines dx there (zeros / the addends): its g is not formed by any launch
                    rows_ok = false;
                    if (nreq) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_data: BatchNorm-reduce requests need a filter that reaches every output-parity class");
                    continue;
                }
                // no filter tap reaches this parity class (e.g. 1x1 stride 2): nothing to add when accumulating
                if (accumulate && !addend && !bias) continue;
                long n = (long)p.M * (p.Nout / 4);
                int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
                SGX_LAUNCH(dgrad_fill_kernel, dim3(grid), dim3(256), 0, stream, p);
                SGX_CHECK_LAUNCH("dgrad_fill");
                continue;
            }
            if (dry || nreq) {
                // the rows of this class: one per pixel tile of the launch, decided exactly as run_igemm decides below
                IgemmParams q = p;
                q.vec = (q.Nout % 4 == 0 && q.y_ld_pix % 4 == 0 && q.y_ld_img % 4 == 0) ? 1 : 0;
                // the launch itself (run_igemm) also looks at the operands' addresses: a real call decides with them BEFORE anything is
                // launched, so a launch that cannot carry the requests is refused here and never writes partial rows
                if (!dry && (((uintptr_t)p.Y % 16) || ((uintptr_t)p.addend % 16) || ((uintptr_t)p.bias % 16))) q.vec = 0;
                const int ph2 = (sec || (dry && dx != nullptr)) && ph == 0 && pw == 0 ? 1 : 0;  // (dry: dx != nullptr marks the two-source form)
                if (ph2) {
                    q.A2 = q.A ? q.A : reinterpret_cast<const float*>(16);
                    q.Th2 = q.Tw2 = 1; q.dh02 = q.dw02 = 0; q.dstep2 = 1;
                }
                const bool flat = q.C < IG_BK && T > 1;
                int mt;
                if (!q.vec || flat || (ph2 && q.C < IG_BK)) rows_ok = false, mt = 0;
                else if (pconv_ok(q, ph2)) mt = pconv_tiles(q.M / (q.Ha * q.Wa), q.Ha, q.Wa);
                else if (ph2) mt = sgx_cdiv(q.M, ph2_tile(q.M, q.Nout).bm);
                else mt = sgx_cdiv(q.M, pick_tile(q.M, q.Nout, conv_math_for(q.Th * q.Tw, q.C)).bm);
                if (nreq) {
                    if (!rows_ok) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_data: this problem cannot carry BatchNorm-reduce requests (sgx_conv2d_bwd_data_stat_blocks = 0)");
                    p.nreq = nreq;
                    p.req_row0 = rows_total;
                    for (int r = 0; r < nreq; ++r) {
                        p.req[r].t = reqs[r].t; p.req[r].scale = reqs[r].scale; p.req[r].shift = reqs[r].shift; p.req[r].mean = reqs[r].mean;
                        p.req[r].parts = reqs[r].partials; p.req[r].t_ld_pix = reqs[r].t_ld_pix; p.req[r].t_ld_img = reqs[r].t_ld_img;
                        p.req[r].c_lo = reqs[r].c_lo; p.req[r].c_hi = reqs[r].c_hi; p.req[r].act = reqs[r].act; p.req[r].rows = reqs[r].rows;
                    }
                }
                rows_total += mt;
                if (dry) continue;
            }
            if (mode == 1 && g_wt_rec) {
                if (g_wt_rec->n < g_wt_rec->max) {
                    sgx_wtrans_job& j = g_wt_rec->jobs[g_wt_rec->n];
                    memset(&j, 0, sizeof(j));
                    j.w = w; j.wt = wt; j.K = d->K; j.C = d->C; j.RS = d->R * d->S; j.T = T;
                    memcpy(j.taps, taps.idx, sizeof(taps.idx) < sizeof(j.taps) ? sizeof(taps.idx) : sizeof(j.taps));
                }
                ++g_wt_rec->n;
            } else if (mode != 2) {
                long n = (long)d->C * T * d->K;
                int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
                SGX_LAUNCH(wtrans_kernel, dim3(grid), dim3(256), 0, stream, w, wt, d->K, d->C, d->R * d->S, T, taps);
                SGX_CHECK_LAUNCH("wtrans");
            }
            if (mode != 1 && sec && ph == 0 && pw == 0) {
                p.A2 = sec->ds; p.Wt2 = sec->w1t;
                p.Hin2 = d->Ho; p.Win2 = d->Wo; p.Th2 = 1; p.Tw2 = 1; p.dh02 = 0; p.dw02 = 0; p.dstep2 = 1;
                p.a2_ld_pix = sec->ld_pix; p.a2_ld_img = sec->ld_img; p.w2_ld_n = d->K;
                p.a2_bytes = view_bytes(d->N, d->Ho, d->Wo, d->K, sec->ld_pix, sec->ld_img);
                p.w2_bytes = (long)d->C * d->K * 4;
                p.addend2 = sec->addend2; p.a2d_ld_pix = sec->a2_ld_pix; p.a2d_ld_img = sec->a2_ld_img;
                p.addend2_scale = sec->a2_scale; p.addend2_scale_dev = sec->a2_scale_dev;
                TileCfg t = ph2_tile(p.M, p.Nout);
                rc = run_igemm(p, t.bm, t.bn, stream, 1);
                if (rc) return rc;
            } else if (mode != 1) {
                TileCfg t = igemm_tile(p);
                TileCfg m = pick_tile(p.M, p.Nout, conv_math_for(p.Th * p.Tw, p.C));
                rc = run_igemm(p, m.bm, t.bn, stream);
                if (rc) return rc;
            }
            if (nreq && p.req_row0 + p.mt != rows_total) SGX_FAIL(SGX_ERR_HIP, "conv bwd_data: internal - statistic rows %d + %d != %d", p.req_row0, p.mt, rows_total);
            wt += (long)d->C * T * d->K;
        }
    if (rows_out) *rows_out = rows_ok ? rows_total : 0;
    return SGX_OK;
}
extern "C" int32_t sgx_conv2d_bwd_data_stat_blocks(const sgx_conv_desc* d, int32_t two_source) {
    int rows = 0;
    // (dry run: the non-null dx pointer only marks the two-source form, nothing is dereferenced)
    if (conv_bwd_data_impl(d, nullptr, nullptr, nullptr, nullptr, two_source ? reinterpret_cast<float*>(16) : nullptr, 0, nullptr, 0, nullptr, 3, nullptr, nullptr, 0, &rows)) return 0;
    return rows;
}
extern "C" int32_t sgx_conv2d_bwd_data_wt_req(const sgx_conv_desc* d, const float* dy, const float* wt, const float* addend, float* dx,
                                              int32_t accumulate, const sgx_bn_reduce_req* reqs, int32_t nreq, void* stream) {
    return conv_bwd_data_impl(d, dy, nullptr, nullptr, addend, dx, accumulate, const_cast<float*>(wt), sgx_conv2d_bwd_data_workspace(d), stream, 2, nullptr, reqs, nreq);
}

// ------------------------------------------------------------------------------------------------
// Weight gradient as ONE GEMM over the flattened filter axis:  dW[k][j] = sum_m dY[m][k] * Xcol[m][j],  j = tap*C + c
// (exactly the OHWI memory order of the weights, so a tile of j may span several taps: narrow-channel layers - the
// 3->48 stem, the 32/48/64-channel CSP blocks - still fill a 96- or 128-wide MFMA tile).  Pixels are the GEMM-K axis:
// a workgroup owns a (k tile, j tile) and a contiguous pixel range [split] and stages 16-pixel slabs of dY and of the
// shifted X in LDS (pixel-major, so MFMA fragments are conflict-free ds_read_b32).
// A lane owns ONE pixel row of the slab and walks it forward 16 pixels per step (no divisions in the loop); its column
// groups have fixed (tap, channel) -> fixed byte deltas, so each load is base + delta with a 4-compare bounds mask.
//
// GROUPED launches (round 3).  The weight gradients of a step are mutually independent, so one launch carries a TABLE of
// them (kernel arguments: up to WG_MAX_JOBS jobs of one tile shape) and the pixel split of every job is sized so that the
// GROUP fills the chip, not each layer on its own: fewer, longer work items -> fewer partial tiles in HBM, ~8x fewer
// launches, no per-layer tail.  Workgroup -> (job, split, tile) is XCD-aware: the (k, j) tiles of one pixel range run
// back to back on ONE XCD, so the range is fetched from HBM once and re-read from that XCD's L2 by the other tiles.
// The split partials are folded INSIDE the launch, deterministically, by a fixed binary tree over the split index that the workgroups walk
// in arrival order (see the kernel's tail): fixed association whatever the arrival order, pairwise summation, no float atomics, no
// separate reduce launch, no workgroup folding more than one partner tile per level.  The hand-over moves with device-scope stores /
// loads (sc1), not with L2 write-back fences.  Tickets are zero on entry and are left zero (the second arriver of a pair resets it).
// ------------------------------------------------------------------------------------------------
#define WG_BKP 16      // pixels per slab (template parameter BKP of the kernel: 16, or WG_BKP_DEEP for the tiles whose two slabs stay <= 32 KB)
#define WG_BKP_DEEP 32
#define WG_MAX_SPLIT 4096
#ifdef SGX_WGRAD_LAB
#define WG_AB(bit) (g.lab & (bit))
#else
#define WG_AB(bit) false
#endif
#define WG_MAX_JOBS 24  // the job table travels as kernel arguments: 24 x (152 + 4) B + 8 B < 4 KB

struct WgJob {
    const float* X;
    const float* DY;
    float* dw;
    float* part;    // [tile][ksplit][BNK * BJ]: node values of the fold tree, in place (a node lives in its leftmost leaf's slot)
    int* tickets;   // [tile][ksplit]: one per sibling pair
    long x_ld_pix, x_ld_img, y_ld_pix, y_ld_img;
    long x_bytes, dy_bytes;
    int H, W, C, K, S, stride, pad, Ho, Wo;
    int M, J, ksplit, mchunk, kt_tiles, jt_tiles;
    int blk0;  // first workgroup of the job in its launch (a multiple of 8: XCD phase 0)
};
struct WgGroupParams {
    int njobs, xcd_order;
    int lab;  // measurement builds (-DSGX_WGRAD_LAB) only: ablation bits - 1 no global loads, 2 no LDS stores (after the first slab), 4 no MFMAs, 8 no fold / dW
    int blk0[WG_MAX_JOBS];  // first workgroup of every job, together: the job lookup is a handful of scalar loads, not one per job
    WgJob jobs[WG_MAX_JOBS];
};

// waves per SIMD the register budget is held to: accumulators (16 per 32x32 block of the wave's sub-tile) + 48 for the loop
// (one-block sub-tiles fit 64 registers unprompted: no request)
// (round 6: two blocks with two slabs in flight spilled one dword under the five waves the formula gives them - four there)
constexpr int wg_min_waves(int acc_regs, int pf) { return acc_regs <= 16 ? 1 : (acc_regs == 32 && pf == 2) ? 4 : 512 / (acc_regs + (pf == 2 ? 64 : 48)); }
// MATH = 1 (the default since round 4 - r3zj / r4a: alone 13.5 -> 12.0 -> with the patch kernel 11.0 ms per step of weight gradients; written
// against the host emulation in round 3 after the transpose read's lane mapping had been probed): the bf16x3 arithmetic of the patch kernel for the weight gradient.  A slab is staged as three bf16 planes
// [plane][pixel][channel] (the split happens once per element, at the LDS store); the MFMA operands - eight consecutive PIXELS of one channel
// per lane - come out of that pixel-major image through ds_read_b64_tr_b16, two reads per plane and 32-row block; six v_mfma_f32_32x32x16_bf16
// per block pair and 16 pixels (hi*hi into the accumulator, the five cross terms into a second one that is added at the end).  Pixel rows are
// pitched at 16 or 48 banks mod 64 so that the four rows a 16-lane group reads fall on distinct banks.
constexpr int wg_bf16_pitch(int row) { return row + (row % 64 == 0 ? 32 : 0); }  // bf16 elements per pixel row
template <int BNK, int BJ, int WK, int WC, int BKP, int PF, int MATH = 0>
__global__ __launch_bounds__(WK * WC * 64, MATH ? 1 : wg_min_waves((BNK / (WK * 32)) * (BJ / (WC * 32)) * 16, PF)) void wgrad_kernel(WgGroupParams g) {
    constexpr int NTH = WK * WC * 64;
    constexpr int TK = BNK / (WK * 32), TC = BJ / (WC * 32);
    static_assert(TK >= 1 && TC >= 1 && TK * WK * 32 == BNK && TC * WC * 32 == BJ, "bad tile");
    static_assert(MATH == 0 || BKP == 16, "bf16x3 loop: 16-pixel slabs (one K step of the bf16 MFMA)");
    constexpr int G = NTH / BKP;  // lanes per pixel row
    constexpr int DJ = (BNK / 4 + G - 1) / G, XJ = (BJ / 4 + G - 1) / G;
    constexpr int DP = wg_bf16_pitch(BNK), XP = wg_bf16_pitch(BJ);
    __shared__ float Ds[MATH ? 4 : 2 * BKP * BNK];
    __shared__ float Xs[MATH ? 4 : 2 * BKP * BJ];
    __shared__ unsigned short Dh[MATH ? 2 * 3 * BKP * DP : 4];  // [buffer][plane][pixel][DP]
    __shared__ unsigned short Xh[MATH ? 2 * 3 * BKP * XP : 4];
    __shared__ int s_last;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / WC, wc = wave % WC;
    // which job: blocks [blk0, next blk0) belong to it
    int ji = 0;
#pragma unroll
    for (int i = 1; i < WG_MAX_JOBS; ++i)
        if (i < g.njobs && (int)blockIdx.x >= g.blk0[i]) ji = i;
    const WgJob& p = g.jobs[ji];
    const int T = p.kt_tiles * p.jt_tiles;
    // (split, tile) of this workgroup.  XCD order: workgroup b runs on XCD b % 8; an XCD walks the tiles of split x, then of x + 8, ...
    const int bl = (int)blockIdx.x - p.blk0;
    int split, tile;
    if (g.xcd_order) {
        const int r = bl >> 3;
        split = (bl & 7) + 8 * (r / T);
        tile = r % T;
    } else {
        split = bl / T;
        tile = bl % T;
    }
    if (split >= p.ksplit) return;  // whole workgroup leaves together (before any barrier)
    const int ktile = tile / p.jt_tiles, jtile = tile - ktile * p.jt_tiles;
    const int k0 = ktile * BNK, j0 = jtile * BJ;
    const int M = p.M, K = p.K, J = p.J, C = p.C, S = p.S, pad = p.pad, stride = p.stride, H = p.H, W = p.W, Ho = p.Ho, Wo = p.Wo;
    const long x_ld_pix = p.x_ld_pix, x_ld_img = p.x_ld_img, y_ld_pix = p.y_ld_pix, y_ld_img = p.y_ld_img;
    const int mbeg = split * p.mchunk;
    const int mend = min(M, mbeg + p.mchunk);
    const int nkt = (mend > mbeg) ? (mend - mbeg + BKP - 1) / BKP : 0;
    const int hw = Ho * Wo;

    // descriptors re-based at the split's first image
    const int img0 = mbeg / hw;
    const sgx_buf bufX = sgx_make_buf(p.X + (long)img0 * x_ld_img, p.x_bytes - (long)img0 * x_ld_img * 4);
    const sgx_buf bufD = sgx_make_buf(p.DY + (long)img0 * y_ld_img, p.dy_bytes - (long)img0 * y_ld_img * 4);

    const int prow = tid / G, cg = tid % G;  // this lane's pixel row of the slab and first column group
    // pixel of slab 0
    int m = mbeg + prow;
    int img = m / hw, rem = m - img * hw;
    int ho = rem / Wo, wo = rem - ho * Wo;
    img -= img0;

    // fixed per column group: dY channel, X (tap, channel) -> byte delta and tap shift
    int dcol[DJ];
    bool dok[DJ];
#pragma unroll
    for (int q = 0; q < DJ; ++q) {
        const int c4 = (cg + q * G) * 4;
        dcol[q] = c4;
        dok[q] = c4 < BNK && k0 + c4 < K;
    }
    int xcol[XJ], xdelta[XJ], xdh[XJ], xdw[XJ];
    bool xok[XJ];
#pragma unroll
    for (int q = 0; q < XJ; ++q) {
        const int c4 = (cg + q * G) * 4;
        const int j = j0 + c4;
        xcol[q] = c4;
        xok[q] = c4 < BJ && j < J;
        const int tap = xok[q] ? j / C : 0;
        const int c = j - tap * C;
        const int tr = tap / S, ts = tap - tr * S;
        xdh[q] = tr - pad;
        xdw[q] = ts - pad;
        xdelta[q] = (int)((((long)xdh[q] * W + xdw[q]) * x_ld_pix + c) * 4);
    }

    // PF register sets of one slab each: PF = 1 loads slab kt+1 under the MFMAs of slab kt (a round trip to HBM has ONE slab's MFMA time);
    // PF = 2 keeps two slabs in flight - the set stored at the top of step kt was requested two steps earlier
    float4 rdA[DJ], rxA[XJ], rdB[PF == 2 ? DJ : 1], rxB[PF == 2 ? XJ : 1];
    // Running byte offsets of this lane's pixel into dY (+ k0) and X (tap (0,0), before the padding shift) and its input coordinates: a slab
    // step is additions - dY is linear in the pixel index inside an image; X moves BKP * stride pixels along the row, one constant more at
    // a row wrap, another at an image wrap (r3o: 5.9 VALU instructions per MFMA with the offsets recomputed from (img, ho, wo) through
    // 64-bit multiplies every slab).  31-bit by the split rule of the plan (every split under 1 GiB of either operand).
    auto d_offset = [&](int im, int h, int w) { return (int)(((long)im * y_ld_img + ((long)h * Wo + w) * y_ld_pix + k0) * 4); };
    auto x_offset = [&](int im, int h, int w) { return (int)(((long)im * x_ld_img + ((long)h * stride * W + w * stride) * x_ld_pix) * 4); };
    int d_off = d_offset(img, ho, wo), x_off = x_offset(img, ho, wo);
    int hi0 = ho * stride, wi0 = wo * stride;  // the lane's state from here on: m, hi0, wi0, d_off, x_off
    const int d_step = (int)(BKP * y_ld_pix * 4), d_img_fix = (int)((y_ld_img - (long)hw * y_ld_pix) * 4);
    const int x_step = (int)((long)BKP * stride * x_ld_pix * 4), x_row_fix = (int)((long)stride * (W - Wo) * x_ld_pix * 4);
    const int x_img_fix = (int)((x_ld_img - (long)Ho * stride * W * x_ld_pix) * 4);
    const int w_span = Wo * stride, h_span = Ho * stride;
    auto load_tile = [&](float4* rd, float4* rx) {
        const bool pok = m < mend;
#pragma unroll
        for (int q = 0; q < DJ; ++q) rd[q] = sgx_buf_ld4(bufD, (pok && dok[q]) ? (unsigned)(d_off + dcol[q] * 4) : SGX_BUF_OOB);
#pragma unroll
        for (int q = 0; q < XJ; ++q) {
            const int hi = hi0 + xdh[q], wi = wi0 + xdw[q];
            const bool ok = pok && xok[q] && hi >= 0 && hi < H && wi >= 0 && wi < W;
            rx[q] = sgx_buf_ld4(bufX, ok ? (unsigned)(x_off + xdelta[q]) : SGX_BUF_OOB);
        }
        // advance this lane's pixel by one slab
        m += BKP;
        if (Wo >= BKP) {
            wi0 += BKP * stride;
            d_off += d_step;
            x_off += x_step;
            if (wi0 >= w_span) {  // next output row
                wi0 -= w_span;
                hi0 += stride;
                x_off += x_row_fix;
                if (hi0 >= h_span) {  // next image
                    hi0 = 0;
                    d_off += d_img_fix;
                    x_off += x_img_fix;
                }
            }
        } else {
            const int mm = m < M ? m : M - 1;
            const int im = mm / hw, rm = mm - im * hw;
            const int h = rm / Wo, w = rm - h * Wo;
            d_off = d_offset(im - img0, h, w);
            x_off = x_offset(im - img0, h, w);
            hi0 = h * stride;
            wi0 = w * stride;
        }
    };
    auto store_tile = [&](int buf, const float4* rd, const float4* rx) {
        if constexpr (MATH == 1) {
#pragma unroll
            for (int q = 0; q < DJ; ++q)
                if (dcol[q] < BNK) {
                    uint2 h, m, l;
                    sgx_split3(rd[q], h, m, l);
                    unsigned short* const b = &Dh[((buf * 3) * BKP + prow) * DP + dcol[q]];
                    *reinterpret_cast<uint2*>(b) = h;
                    *reinterpret_cast<uint2*>(b + BKP * DP) = m;
                    *reinterpret_cast<uint2*>(b + 2 * BKP * DP) = l;
                }
#pragma unroll
            for (int q = 0; q < XJ; ++q)
                if (xcol[q] < BJ) {
                    uint2 h, m, l;
                    sgx_split3(rx[q], h, m, l);
                    unsigned short* const b = &Xh[((buf * 3) * BKP + prow) * XP + xcol[q]];
                    *reinterpret_cast<uint2*>(b) = h;
                    *reinterpret_cast<uint2*>(b + BKP * XP) = m;
                    *reinterpret_cast<uint2*>(b + 2 * BKP * XP) = l;
                }
            return;
        }
#pragma unroll
        for (int q = 0; q < DJ; ++q)
            if (dcol[q] < BNK) sgx_st4(&Ds[buf * BKP * BNK + prow * BNK + dcol[q]], rd[q]);
#pragma unroll
        for (int q = 0; q < XJ; ++q)
            if (xcol[q] < BJ) sgx_st4(&Xs[buf * BKP * BJ + prow * BJ + xcol[q]], rx[q]);
    };

    sgx_f32x16 acc[TK][TC];
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nkt > 0) {
        load_tile(rdA, rxA);
        store_tile(0, rdA, rxA);
        if (PF == 2) {
            load_tile(rdA, rxA);
            load_tile(rdB, rxB);
        }
    }
    __syncthreads();
    const int fcol = lane & 31, fkh = lane >> 5;
    sgx_f32x16 acc2[MATH ? TK : 1][MATH ? TC : 1];  // bf16x3: the five cross terms
    if constexpr (MATH == 1) {
#pragma unroll
        for (int i = 0; i < TK; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    // bf16x3 operand of a 32-row block: lane l ends up with the eight pixels 8 (l / 32) .. + 7 of column blk0 + l % 32 (two transpose reads)
    auto frag = [&](const unsigned short* img, int pitch, int buf, int pl, int blk0) {
        const int gq = lane >> 4, q = lane & 15;
        const unsigned short* const ptr = img + ((buf * 3 + pl) * BKP + 8 * (gq >> 1) + (q >> 2)) * pitch + blk0 + 16 * (gq & 1) + 4 * (q & 3);
        const uint2 lo = sgx_lds_tr_read(ptr), hi = sgx_lds_tr_read(ptr + 4 * pitch);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto mfma_slab = [&](int buf) {
        if constexpr (MATH == 1) {
            uint4 b[TC][3];
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[j][pl] = frag(Xh, XP, buf, pl, wc * TC * 32 + j * 32);
#pragma unroll
            for (int i = 0; i < TK; ++i) {
                uint4 a[3];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[pl] = frag(Dh, DP, buf, pl, wk * TK * 32 + i * 32);
#pragma unroll
                for (int j = 0; j < TC; ++j) {
                    acc[i][j] = sgx_mfma_bf16(a[0], b[j][0], acc[i][j]);
                    acc2[i][j] = sgx_mfma_bf16(a[0], b[j][1], acc2[i][j]);
                    acc2[i][j] = sgx_mfma_bf16(a[1], b[j][0], acc2[i][j]);
                    acc2[i][j] = sgx_mfma_bf16(a[1], b[j][1], acc2[i][j]);
                    acc2[i][j] = sgx_mfma_bf16(a[0], b[j][2], acc2[i][j]);
                    acc2[i][j] = sgx_mfma_bf16(a[2], b[j][0], acc2[i][j]);
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < BKP / 2; ++kk) {
            float af[TK], bf[TC];
#pragma unroll
            for (int i = 0; i < TK; ++i) af[i] = Ds[buf * BKP * BNK + (2 * kk + fkh) * BNK + wk * TK * 32 + i * 32 + fcol];
#pragma unroll
            for (int j = 0; j < TC; ++j) bf[j] = Xs[buf * BKP * BJ + (2 * kk + fkh) * BJ + wc * TC * 32 + j * 32 + fcol];
#pragma unroll
            for (int i = 0; i < TK; ++i)
#pragma unroll
                for (int j = 0; j < TC; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    if constexpr (PF == 1) {
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nkt && !WG_AB(1)) load_tile(rdA, rxA);
            if (!WG_AB(4)) mfma_slab(buf);
            if (kt + 1 < nkt && !WG_AB(2)) store_tile(buf ^ 1, rdA, rxA);
            __syncthreads();
        }
    } else {
        // step kt: the other LDS buffer is free (its readers passed the barrier): store slab kt+1 (requested two steps ago), re-issue the
        // set for slab kt+3, then the MFMAs of slab kt
        // Store and re-issue are UNCONDITIONAL: past the end of the range the lanes' offsets are out of bounds (no memory access, zeros) and
        // the slab stored last is never read.  With `if (kt + 3 < nkt)` around the loads hipcc's wait-count pass has a path on which the
        // younger set was never requested, takes the minimum over paths and waits for vmcnt(0) before the store - i.e. for the set requested
        // ONE step ago as well, which makes the second register set pointless (r3n's build: read off the ISA afterwards).
        auto step = [&](int kt, float4* rd, float4* rx) {
            const int buf = kt & 1;
            if (!WG_AB(2)) store_tile(buf ^ 1, rd, rx);
            if (!WG_AB(1)) load_tile(rd, rx);
            if (!WG_AB(4)) mfma_slab(buf);
            __syncthreads();
        };
        int kt = 0;
        for (; kt + 1 < nkt; kt += 2) {
            step(kt, rdA, rxA);
            step(kt + 1, rdB, rxB);
        }
        if (kt < nkt) step(kt, rdA, rxA);
    }
    if constexpr (MATH == 1) {
#pragma unroll
        for (int i = 0; i < TK; ++i)
#pragma unroll
            for (int j = 0; j < TC; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += acc2[i][j][r];
    }
    if (WG_AB(8)) {  // main loop only (the accumulators stay observable)
        if (acc[0][0][0] == 1.2345e-30f) p.dw[0] = 1.f;
        return;
    }
    // ---- fold the pixel splits of this tile: a fixed binary tree over the split index, walked by arrival -------------------------------------
    // Level L pairs node i = split >> L with its sibling i ^ 1.  A workgroup holding a node's value publishes it (device-scope stores into
    // the slot of the node's leftmost leaf), takes the pair's ticket, and leaves if it is the FIRST of the pair; the second one adds the
    // sibling's value to its registers and carries the parent one level up.  Whoever arrives last, every node is left + right of the same
    // two children (fp32 addition commutes), so the result is bit-reproducible - and it is a pairwise summation: ~log2(ksplit) roundings
    // on top of the chains.  No workgroup ever folds more than one partner tile per level (the serial "last arriver folds them all" form
    // measured r3b: 55 TFLOP/s against 88 for the two-launch kernel it replaced).  Tickets: one per pair, at the right child's
    // leftmost-leaf index (odd x 2^L: unique over the tree), reset by the second arriver.
    const int ksplit = p.ksplit;
    if (ksplit > 1) {
        constexpr int TE = BNK * BJ;
        float* const base = p.part + (long)tile * ksplit * TE;
        int* const tk = p.tickets + (long)tile * ksplit;
        const int foff = (wk * TK * 32 + (lane >> 5) * 4) * BJ + wc * TC * 32 + (lane & 31);  // this lane's corner of the wave's sub-tile
        for (int L = 0; (1 << L) < ksplit; ++L) {
            const int i = split >> L, sib = i ^ 1;
            if (((long)sib << L) >= ksplit) continue;  // no sibling on this level: the value passes up as it is
            float* const mine = base + ((long)i << L) * TE;
#pragma unroll
            for (int a = 0; a < TK; ++a)
#pragma unroll
                for (int b = 0; b < TC; ++b) {
                    float* const q = mine + foff + (a * 32) * BJ + b * 32;
#pragma unroll
                    for (int r = 0; r < 16; ++r) sgx_st_dev(&q[((r & 3) + 8 * (r >> 2)) * BJ], acc[a][b][r]);
                    sgx_sched_fence();
                }
            sgx_wait_stores();
            __syncthreads();
            if (tid == 0) {
                int* const t = &tk[((long)(i | 1)) << L];
                const int old = atomicAdd(t, 1);
                s_last = old;
                if (old) *t = 0;
            }
            __syncthreads();
            const int second = s_last;
            __syncthreads();  // (s_last is rewritten on the next level)
            if (!second) return;
            const float* const other = base + ((long)sib << L) * TE;
#pragma unroll
            for (int a = 0; a < TK; ++a)
#pragma unroll
                for (int b = 0; b < TC; ++b) {  // one 32x32 block at a time: 16 loads in flight per lane, 16 registers - not TK*TC*16
                    const float* const q = other + foff + (a * 32) * BJ + b * 32;
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = sgx_ld_dev(&q[((r & 3) + 8 * (r >> 2)) * BJ]);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] += v[r];
                    sgx_sched_fence();
                }
        }
    }
    // the root of the tile: into dW
    float* const dw = p.dw;
#pragma unroll
    for (int i = 0; i < TK; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
            const int jj = j0 + wc * TC * 32 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * TK * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (k < K && jj < J) dw[(long)k * J + jj] += acc[i][j][r];
            }
        }
}

// ---- host side: tile choice, group plan, launches ------------------------------------------------------------------------------------
static int wg_tile(int n) {  // least padding among {32,64,96,128}, ties to the wider tile
    const int cand[4] = {32, 64, 96, 128};
    int b = 128, best = 1 << 30;
    for (int i = 0; i < 4; ++i) {
        int padded = ((n + cand[i] - 1) / cand[i]) * cand[i];
        if (padded <= best) {
            best = padded;
            b = cand[i];
        }
    }
    return b;
}
static int wg_waves(int bnk, int bj) {  // waves per workgroup of the instantiated (bnk, bj) variant
    if (bnk == 96 && bj == 96) return 3;
    if (bnk == 96 && bj == 64) return 3;
    if (bnk == 96 && bj == 32) return 3;
    if (bnk == 32 && bj == 96) return 3;
    if (bnk == 64 && bj == 96) return 2;
    if (bnk == 64 && bj == 32) return 2;
    if (bnk == 32 && bj == 64) return 2;
    if (bnk == 32 && bj == 32) return 1;
    return 4;
}
extern "C" int32_t sgx_stats_blocks(int64_t M);
extern "C" int64_t sgx_colsum_workspace(int64_t M, int32_t C);
// grouped-launch knobs (measurement: sgx_debug_set_wgrad_group): rounds of work items a large group is cut into, the work of an item
// below which a small group is not cut further (MFLOP), XCD-aware block order
static std::atomic<int> g_wg_rounds{6}, g_wg_item_mflop{8}, g_wg_xcd{1}, g_wg_deep{0};
extern "C" int32_t sgx_debug_set_wgrad_group(int32_t rounds, int32_t item_mflop, int32_t xcd_order) {
    SGX_CHECK_ARG(rounds >= 0 && item_mflop >= 0, "debug_set_wgrad_group: negative value");
    g_wg_rounds = rounds ? rounds : 6;
    g_wg_item_mflop = item_mflop ? item_mflop : 8;
    g_wg_xcd = xcd_order ? 1 : 0;
    return SGX_OK;
}
static std::atomic<int> g_wg_lab{0};
extern "C" int32_t sgx_debug_set_wgrad_loop(int32_t deep_slab, int32_t ablate) {
    g_wg_deep = deep_slab;
    g_wg_lab = ablate;
    return SGX_OK;
}
// Arithmetic of the weight gradient (process-wide, product default 2).  0: the fp32 matrix pipe.  1: the bf16x3 slab loop (round 3: the whole
// GPU parity suite green under it, r4a; 662 -> 686 images/s).  2: mode 1 + the PATCH kernel (wgrad_patch.hip) on the 3x3 pad-1 problems.
static std::atomic<int> g_wg_math{2};
static std::atomic<int> g_wp_item_mflop{48}, g_wp_kb{0}, g_wp_fill{60};
extern "C" int32_t sgx_conv_set_wgrad_math(int32_t mode) {
    SGX_CHECK_ARG(mode >= 0 && mode <= 2, "conv_set_wgrad_math: mode must be 0 (fp32), 1 (bf16x3) or 2 (bf16x3 + patch kernel)");
    g_wg_math = mode;
    return SGX_OK;
}
extern "C" int32_t sgx_conv_get_wgrad_math(void) { return g_wg_math; }
extern "C" int32_t sgx_debug_set_wgrad_patch(int32_t item_mflop, int32_t kb, int32_t min_fill_pct) {
    SGX_CHECK_ARG(item_mflop >= 0 && kb >= 0 && kb <= 3 && min_fill_pct >= 0 && min_fill_pct <= 100, "debug_set_wgrad_patch: item_mflop >= 0, kb in 0..3, fill in 0..100");
    g_wp_item_mflop = item_mflop ? item_mflop : 48;
    g_wp_kb = kb;
    g_wp_fill = min_fill_pct ? min_fill_pct : 60;
    return SGX_OK;
}
struct WgPlan {
    int bnk, bj, waves, kt_tiles, jt_tiles, ksplit, mchunk;
    long part_off, ticket_off;  // floats, ints
    WpPlan wp;                  // wp.cfg != 0: the patch kernel takes this job (the fields above are then unused)
};
// tile choice from the exhaustive search: 64x64 where the channel count allows, 96x128 for 96-wide layers; a tuning-table entry (kind 2)
// or the measurement override replaces it
static void wgrad_tile(const sgx_conv_desc* d, WgPlan& pl) {
    TuneScope tune(2, d);
    const int J = d->R * d->S * d->C;
    int bnk = wg_tile(d->K), bj;
    if (d->K % 128 == 0 && d->K >= 384) bnk = 128;
    else if (d->K % 64 == 0) bnk = 64;
    else if (d->K % 96 == 0) bnk = 96;
    if (bnk == 96) bj = J >= 128 ? 128 : wg_tile(J);
    else if (bnk == 32) bj = J >= 128 ? 128 : wg_tile(J);
    else bj = J >= 64 ? 64 : wg_tile(J);
    if (t_tune && t_tune->bm) bnk = t_tune->bm, bj = t_tune->bn;
    const int owk = g_ovr_wk.load(std::memory_order_relaxed), owj = g_ovr_wj.load(std::memory_order_relaxed);
    pl.bnk = owk ? owk : bnk;
    pl.bj = owj ? owj : bj;
    pl.waves = wg_waves(pl.bnk, pl.bj);
    pl.kt_tiles = sgx_cdiv(d->K, pl.bnk);
    pl.jt_tiles = sgx_cdiv(J, pl.bj);
}
#define WG_SLOTS 1536  // workgroups the chip holds at once (256 CUs x ~6): the unit a group's work is cut against
static int32_t wgrad_group_plan(const sgx_wgrad_job* jobs, int n, std::vector<WgPlan>& plans, long* part_floats, long* ticket_ints) {
    plans.resize(n);
    double work = 0.0;
    // the patch kernel: product mode 2, no measurement override of the slab loop's tiles / loop in force
    const bool patch_on = g_wg_math.load(std::memory_order_relaxed) == 2 && !g_ovr_wk.load(std::memory_order_relaxed) &&
                          !g_ovr_wj.load(std::memory_order_relaxed) && !g_ovr_split.load(std::memory_order_relaxed) &&
                          !(g_wg_deep.load(std::memory_order_relaxed) & 16) && !g_wg_lab.load(std::memory_order_relaxed);
    for (int i = 0; i < n; ++i) {
        const sgx_conv_desc* d = &jobs[i].d;
        int32_t rc = check_desc(d);
        if (rc) return rc;
        SGX_CHECK_ARG(d->K % 4 == 0 && d->y_ld_pix % 4 == 0, "conv bwd_weight: K and dy pixel stride must be multiples of 4");
        plans[i].wp.cfg = 0;
        if (patch_on && wpatch_plan_job(d, plans[i].wp, g_wp_kb.load(std::memory_order_relaxed), g_wp_fill.load(std::memory_order_relaxed))) {
            continue;
        }
        wgrad_tile(d, plans[i]);
        work += 2.0 * d->N * d->Ho * d->Wo * (double)plans[i].kt_tiles * plans[i].bnk * (double)plans[i].jt_tiles * plans[i].bj;
    }
    // work of one item: a large group is cut into `rounds` rounds of WG_SLOTS items; a small one into items of >= item_mflop (~100 us of
    // one workgroup beside its co-residents) as long as that still leaves ~1.3 rounds
    // ... and never more than 2x that: an item is ONE sequential fp32 accumulation chain on the matrix pipe; short chains folded pairwise
    // by the tree keep the sum a blocked summation, and the partial tile an item adds is ~6 % of the operand bytes it reads.
    const double lo = 1e6 * g_wg_item_mflop.load(std::memory_order_relaxed);
    double item = work / ((double)WG_SLOTS * g_wg_rounds.load(std::memory_order_relaxed));
    if (item < lo) item = fmin(lo, work / (WG_SLOTS * 1.3));
    if (item > 2.0 * lo) item = 2.0 * lo;
    const int osp = g_ovr_split.load(std::memory_order_relaxed);
    long poff = 0, toff = 0;
    // patch jobs: the jobs of one kernel form (stride, tile columns, filter blocks) run as ONE launch, so the items are sized per form: ONE
    // round of the ~1024 workgroups the chip holds (four per CU) - all of a launch resident together - never more than g_wp_item_mflop MFLOP
    // and never fewer than 256 pixels (wpatch_plan_split).  r4g (lab, ms per step of all weight gradients alone): 1 round 11.22, 2 rounds
    // 11.40, 3 rounds 11.61, 6 rounds 13.14, 12 rounds 16.72 - an item pays its prologue, its first tile's latency and a 37 KB hand-over
    // of its partial tile, and short items pay them too often; the tail of a one-round launch is filled by the main stream's kernels.
    {
        const double phi = 1e6 * g_wp_item_mflop.load(std::memory_order_relaxed);
        std::vector<char> sized(n, 0);
        for (int first = 0; first < n; ++first) {
            if (sized[first] || !plans[first].wp.cfg) continue;
            const WpPlan& f = plans[first].wp;
            double fw = 0.0;
            for (int i = first; i < n; ++i) {
                const WpPlan& w = plans[i].wp;
                if (!sized[i] && w.cfg && jobs[i].d.stride == jobs[first].d.stride && w.pc == f.pc && w.kb == f.kb)
                    fw += 2.0 * 16.0 * w.nks * w.ntiles * w.kt_tiles * 32.0 * w.kb * w.ct_tiles * 32.0 * 9.0;
            }
            double pitem = fw / 1024.0;
            if (pitem > phi) pitem = phi;
            for (int i = first; i < n; ++i) {
                WpPlan& w = plans[i].wp;
                if (!sized[i] && w.cfg && jobs[i].d.stride == jobs[first].d.stride && w.pc == f.pc && w.kb == f.kb) {
                    sized[i] = 1;
                    wpatch_plan_split(&jobs[i].d, w, pitem, &poff, &toff);
                }
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        const sgx_conv_desc* d = &jobs[i].d;
        WgPlan& pl = plans[i];
        if (pl.wp.cfg) continue;
        const long M = (long)d->N * d->Ho * d->Wo;
        const long tiles = (long)pl.kt_tiles * pl.jt_tiles;
        long mchunk = (long)(item / (2.0 * pl.bnk * pl.bj));
        if (osp) mchunk = M / sgx_cdiv(sgx_cdiv(osp, pl.waves), tiles);  // measurement: that many waves per job
        if (mchunk < 256) mchunk = 256;  // at least 16 slabs per item
        if (mchunk > M) mchunk = M;
        long ks = (M + mchunk - 1) / mchunk;
        if (ks > WG_MAX_SPLIT) ks = WG_MAX_SPLIT;
        // a split's lane offsets are 31-bit: keep every split under 1 GiB of either operand
        const long big = (long)d->N * (d->x_ld_img > d->y_ld_img ? d->x_ld_img : d->y_ld_img) * 4;
        const long need = big / (1L << 30) + 1;
        if (ks < need) ks = need;
        if (ks > WG_MAX_SPLIT) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_weight: operand too large (%ld pixel ranges of 1 GiB)", need);
        mchunk = (M + ks - 1) / ks;
        mchunk = ((mchunk + WG_BKP_DEEP - 1) / WG_BKP_DEEP) * WG_BKP_DEEP;
        ks = (M + mchunk - 1) / mchunk;
        pl.ksplit = (int)ks;
        pl.mchunk = (int)mchunk;
        const long te = (long)pl.bnk * pl.bj;
        pl.part_off = poff;
        poff += ks > 1 ? tiles * ks * te : 0;
        pl.ticket_off = toff;
        toff += ks > 1 ? tiles * ks : 0;
    }
    *part_floats = poff;
    *ticket_ints = toff;
    return SGX_OK;
}
extern "C" int32_t sgx_conv2d_bwd_weight_group_sizes(const sgx_wgrad_job* jobs, int32_t njobs, int64_t* ws_bytes, int64_t* ticket_ints) {
    SGX_CHECK_ARG(jobs && njobs > 0 && ws_bytes && ticket_ints, "conv bwd_weight_group_sizes: bad args");
    std::vector<WgPlan> plans;
    long pf = 0, ti = 0;
    int32_t rc = wgrad_group_plan(jobs, njobs, plans, &pf, &ti);
    if (rc) return rc;
    *ws_bytes = pf * 4 + 256;
    *ticket_ints = ti + 1;
    return SGX_OK;
}
template <int BNK, int BJ, int WK, int WC>
static void launch_wgrad(const WgGroupParams& g, int nblk, void* stream) {
    const int loop = g_wg_deep.load(std::memory_order_relaxed);  // bit 0: 32-pixel slabs, bit 1: ONE slab in flight (default: two)
    const dim3 grid((unsigned)nblk), block(WK * WC * 64);
    if constexpr (BNK == 64 && BJ == 64 && WK == 2 && WC == 2) {
        // bit 2 (measurement, not yet measured): the 64x64 tile on TWO waves - two 32x32 blocks per wave, half the address arithmetic /
        // LDS traffic per MFMA (r3o: 5.9 VALU + 3.7 SALU instructions per MFMA with one block per wave)
        if (loop & 4) {
            launch_wgrad<64, 64, 1, 2>(g, nblk, stream);
            return;
        }
    }
    if constexpr (BNK + BJ <= 128) {  // 32-pixel slabs (half the barriers, twice the bytes in flight per lane) where two of them fit 32 KB
        if (loop & 1) {
            SGX_LAUNCH((wgrad_kernel<BNK, BJ, WK, WC, WG_BKP_DEEP, 1>), grid, block, wg_lds_pad(wgrad_kernel<BNK, BJ, WK, WC, WG_BKP_DEEP, 1>), stream, g);
            return;
        }
    }
    // the bf16x3 loop (r3zj: 13.50 -> 11.99 ms alone with four of the tile shapes; every shape since): product modes 1 and 2
    // (sgx_conv_set_wgrad_math; r4a: the whole GPU suite green under it, 662 -> 686 images/s), or measurement bit 3; bit 5 forces the fp32 loop
    if constexpr (!(BNK == 64 && BJ == 64 && WK == 1)) {
        if ((loop & 8) || (g_wg_math.load(std::memory_order_relaxed) >= 1 && !(loop & 32))) {
            SGX_LAUNCH((wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 2, 1>), grid, block, wg_lds_pad(wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 2, 1>), stream, g);
            return;
        }
    }
    // two register sets of loads in flight: measured r3n on YOLO-NAS-S, 14.29 -> 13.92 ms of weight-gradient time alone, +0.6 % on the step
    if (loop & 2) SGX_LAUNCH((wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 1>), grid, block, wg_lds_pad(wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 1>), stream, g);
    else SGX_LAUNCH((wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 2>), grid, block, wg_lds_pad(wgrad_kernel<BNK, BJ, WK, WC, WG_BKP, 2>), stream, g);
}
extern "C" int32_t sgx_conv2d_bwd_weight_group(const sgx_wgrad_job* jobs, int32_t njobs, void* ws, int64_t ws_bytes, int32_t* tickets,
                                               int64_t ticket_ints, void* stream) {
    SGX_CHECK_ARG(jobs && njobs > 0, "conv bwd_weight_group: bad args");
    // every job adds into its dw without atomics (one fold tree per job): two jobs of one group must not share a filter gradient
    for (int i = 1; i < njobs; ++i)
        for (int j = 0; j < i; ++j)
            SGX_CHECK_ARG(jobs[i].dw != jobs[j].dw, "conv bwd_weight_group: jobs %d and %d write the same dw (shared weights: launch them in separate calls)", j, i);
    std::vector<WgPlan> plans;
    long pf = 0, ti = 0;
    int32_t rc = wgrad_group_plan(jobs, njobs, plans, &pf, &ti);
    if (rc) return rc;
    if (!ws || ws_bytes < pf * 4 + 256 || ((uintptr_t)ws % 16) != 0) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_weight_group: workspace too small / unaligned");
    if (!tickets || ticket_ints < ti + 1) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_weight_group: ticket buffer too small");
    // The fold trees leave their tickets zero (the second arriver of a pair resets it), but that invariant would not survive an aborted
    // launch or a plan that changed between two calls on one buffer: the used range is cleared on every call (a few KB, stream-ordered).
    if (ti > 0) SGX_MEMSET_ASYNC(tickets, 0, ti * 4, stream);
    std::vector<char> done(njobs, 0);
    // ---- the patch kernel's jobs: one launch per kernel form (stride, tile columns, filter blocks) and WP_MAX_JOBS jobs of it
    for (int first = 0; first < njobs; ++first) {
        if (done[first] || !plans[first].wp.cfg) continue;
        WpGroupParams g;
        memset(&g, 0, sizeof(g));
        g.xcd_order = g_wg_xcd.load(std::memory_order_relaxed);
        const WpPlan& form = plans[first].wp;
        const int stride = jobs[first].d.stride;
        int nblk = 0;
        double flops = 0.0, bytes = 0.0;
        for (int i = first; i < njobs && g.njobs < WP_MAX_JOBS; ++i) {
            const WpPlan& w = plans[i].wp;
            if (done[i] || !w.cfg || jobs[i].d.stride != stride || w.pc != form.pc || w.kb != form.kb) continue;
            done[i] = 1;
            const sgx_conv_desc* d = &jobs[i].d;
            SGX_CHECK_ARG(jobs[i].x && jobs[i].dy && jobs[i].dw, "conv bwd_weight: null pointer");
            WpJob& p = g.jobs[g.njobs++];
            p.X = jobs[i].x; p.DY = jobs[i].dy; p.dw = jobs[i].dw;
            p.part = (float*)ws + w.part_off; p.tickets = tickets + w.ticket_off;
            p.x_ld_pix = d->x_ld_pix; p.x_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
            p.x_bytes = view_bytes(d->N, d->H, d->W, d->C, d->x_ld_pix, d->x_ld_img);
            p.dy_bytes = view_bytes(d->N, d->Ho, d->Wo, d->K, d->y_ld_pix, d->y_ld_img);
            p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
            p.tiles_h = w.tiles_h; p.tiles_w = w.tiles_w; p.ntiles = (int)w.ntiles;
            p.ksplit = w.ksplit; p.tchunk = w.tchunk; p.kt_tiles = w.kt_tiles; p.ct_tiles = w.ct_tiles;
            p.blk0 = nblk;
            g.blk0[g.njobs - 1] = nblk;
            // range-major XCD order idles XCDs when the last round of eight ranges is short (r4b: 12 ranges x 144 tiles, +18 % against the
            // plain order): only with a multiple of eight ranges or enough rounds for the remainder not to matter
            p.xcd_ranges = (w.ksplit % 8 == 0 || w.ksplit >= 40) ? 1 : 0;
            nblk += p.xcd_ranges ? 8 * sgx_cdiv(w.ksplit, 8) * w.kt_tiles * w.ct_tiles : 8 * sgx_cdiv((long)w.ksplit * w.kt_tiles * w.ct_tiles, 8);
            flops += 2.0 * (double)d->N * d->Ho * d->Wo * (double)d->K * 9.0 * d->C;
            bytes += 4.0 * ((double)d->N * d->H * d->W * d->C + (double)d->N * d->Ho * d->Wo * d->K + (double)d->K * 9.0 * d->C);
        }
        {
            SGX_PROF(1, flops, bytes, stream);
            rc = wpatch_launch(stride, form, g, nblk, stream);
            if (rc) return rc;
        }
    }
    for (int first = 0; first < njobs; ++first) {
        if (done[first]) continue;
        // one launch per tile shape (and per WG_MAX_JOBS jobs of it)
        WgGroupParams g;
        memset(&g, 0, sizeof(g));
        g.xcd_order = g_wg_xcd.load(std::memory_order_relaxed);
        g.lab = g_wg_lab.load(std::memory_order_relaxed);
        int nblk = 0;
        double flops = 0.0, bytes = 0.0;
        for (int i = first; i < njobs && g.njobs < WG_MAX_JOBS; ++i) {
            if (done[i] || plans[i].bnk != plans[first].bnk || plans[i].bj != plans[first].bj) continue;
            done[i] = 1;
            const sgx_conv_desc* d = &jobs[i].d;
            const WgPlan& pl = plans[i];
            SGX_CHECK_ARG(jobs[i].x && jobs[i].dy && jobs[i].dw, "conv bwd_weight: null pointer");
            WgJob& p = g.jobs[g.njobs++];
            p.X = jobs[i].x; p.DY = jobs[i].dy; p.dw = jobs[i].dw;
            p.part = (float*)ws + pl.part_off; p.tickets = tickets + pl.ticket_off;
            p.x_ld_pix = d->x_ld_pix; p.x_ld_img = d->x_ld_img; p.y_ld_pix = d->y_ld_pix; p.y_ld_img = d->y_ld_img;
            p.x_bytes = view_bytes(d->N, d->H, d->W, d->C, d->x_ld_pix, d->x_ld_img);
            p.dy_bytes = view_bytes(d->N, d->Ho, d->Wo, d->K, d->y_ld_pix, d->y_ld_img);
            p.H = d->H; p.W = d->W; p.C = d->C; p.K = d->K; p.S = d->S; p.stride = d->stride; p.pad = d->pad; p.Ho = d->Ho; p.Wo = d->Wo;
            p.M = d->N * d->Ho * d->Wo; p.J = d->R * d->S * d->C;
            p.ksplit = pl.ksplit; p.mchunk = pl.mchunk; p.kt_tiles = pl.kt_tiles; p.jt_tiles = pl.jt_tiles;
            p.blk0 = nblk;
            g.blk0[g.njobs - 1] = nblk;
            nblk += 8 * sgx_cdiv(pl.ksplit, 8) * pl.kt_tiles * pl.jt_tiles;
            flops += 2.0 * (double)p.M * (double)d->K * (double)p.J;
            bytes += 4.0 * ((double)d->N * d->H * d->W * d->C + (double)p.M * d->K + (double)d->K * p.J);
        }
        {
            SGX_PROF(1, flops, bytes, stream);
            const WgPlan& pl = plans[first];
            bool launched = false;
#define WG_CASE(BK_, BJ_, WK_, WC_)                              \
    if (!launched && pl.bnk == BK_ && pl.bj == BJ_) {            \
        launch_wgrad<BK_, BJ_, WK_, WC_>(g, nblk, stream);       \
        launched = true;                                         \
    }
            WG_CASE(128, 128, 2, 2)
            WG_CASE(128, 96, 4, 1)
            WG_CASE(128, 64, 2, 2)
            WG_CASE(128, 32, 4, 1)
            WG_CASE(96, 128, 1, 4)
            WG_CASE(96, 96, 3, 1)
            WG_CASE(96, 64, 3, 1)
            WG_CASE(96, 32, 3, 1)
            WG_CASE(64, 128, 2, 2)
            WG_CASE(64, 96, 2, 1)
            WG_CASE(64, 64, 2, 2)
            WG_CASE(64, 32, 2, 1)
            WG_CASE(32, 128, 1, 4)
            WG_CASE(32, 96, 1, 3)
            WG_CASE(32, 64, 1, 2)
            WG_CASE(32, 32, 1, 1)
#undef WG_CASE
            if (!launched) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_weight: no tile %dx%d", pl.bnk, pl.bj);
        }
        SGX_CHECK_LAUNCH("wgrad");
    }
    return SGX_OK;
}

// One weight gradient (a group of one).  ws holds the partial tiles, then the tickets (cleared here: the caller's workspace is scratch).
static int64_t wgrad_single_sizes(const sgx_conv_desc* d, int64_t* part_bytes, int64_t* ticket_ints) {
    sgx_wgrad_job job;
    memset(&job, 0, sizeof(job));
    job.d = *d;
    return sgx_conv2d_bwd_weight_group_sizes(&job, 1, part_bytes, ticket_ints);
}
extern "C" int64_t sgx_conv2d_bwd_weight_workspace(const sgx_conv_desc* d) {
    int64_t pb = 0, ti = 0;
    if (wgrad_single_sizes(d, &pb, &ti)) return -1;
    int64_t slabs = ((pb + 255) / 256) * 256 + ti * 4;
    int64_t bias = sgx_colsum_workspace((int64_t)d->N * d->Ho * d->Wo, d->K);
    return (slabs > bias ? slabs : bias) + 256;
}

extern "C" int32_t sgx_conv2d_bwd_weight(const sgx_conv_desc* d, const float* x, const float* dy, float* dw, float* dbias,
                                         void* ws, int64_t ws_bytes, void* stream) {
    int32_t rc = check_desc(d);
    if (rc) return rc;
    SGX_CHECK_ARG(x && dy && dw, "conv bwd_weight: null pointer");
    int64_t pb = 0, ti = 0;
    rc = wgrad_single_sizes(d, &pb, &ti);
    if (rc) return rc;
    if (!ws || ws_bytes < sgx_conv2d_bwd_weight_workspace(d)) SGX_FAIL(SGX_ERR_WORKSPACE, "conv bwd_weight: workspace too small");
    const int64_t toff = ((pb + 255) / 256) * 256;
    int32_t* tickets = (int32_t*)((char*)ws + toff);
    SGX_MEMSET_ASYNC(tickets, 0, ti * 4, stream);
    sgx_wgrad_job job;
    job.d = *d; job.x = x; job.dy = dy; job.dw = dw;
    rc = sgx_conv2d_bwd_weight_group(&job, 1, ws, toff, tickets, ti, stream);
    if (rc) return rc;
    if (dbias) {
        // column sum of dy; the workspace is free again once the launch above has run (stream order)
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
    return conv_bwd_data_impl(&d, x, wt, bias, nullptr, y, 0, ws, ws_bytes, stream, 0);
}
// wtt: the filter as sgx_conv2d_transpose_weights leaves it for the adjoint convolution (K = C filters of K channels, 2x2, stride 2, no padding) -
// the host mirror keeps it in the network's per-step transpose batch, so the four parity launches go out without a transpose launch each
extern "C" int32_t sgx_convT2x2_fwd_wt(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix, int64_t x_ld_img,
                                       const float* wtt, const float* bias, float* y, int64_t y_ld_pix, int64_t y_ld_img, void* stream) {
    SGX_CHECK_ARG(wtt, "convT2x2_fwd_wt: null pointer");
    sgx_conv_desc d = convT_adjoint_desc(N, H, W, C, K, x_ld_pix, x_ld_img, y_ld_pix, y_ld_img);
    return conv_bwd_data_impl(&d, x, nullptr, bias, nullptr, y, 0, const_cast<float*>(wtt), sgx_conv2d_bwd_data_workspace(&d), stream, 2);
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

// ---- a HIP stream confined to part of the chip ---------------------------------------------------------------------------------------
// The weight gradients run on a side stream underneath the backward pass.  Their workgroups live for hundreds of microseconds and take every
// CU; the short, dependent kernels of the main stream (the critical path of the step) then wait for slots between them - r4t: the
// BatchNorm-backward sweeps run 7x longer while a weight-gradient kernel is resident, 6.8 ms per step over all main-stream kernels
// (profiles/r4t_*).  A stream created here dispatches to `cus` of the device's CUs only (spread evenly: every `keep`-th ... CU index is
// left out), so the rest of the chip always has room for the main stream.
extern "C" int32_t sgx_conv_set_wgrad_lds_reserve(int32_t kb) {
    SGX_CHECK_ARG(kb >= 0 && kb <= 120, "wgrad LDS reserve: 0..120 KB");
    g_wg_lds_reserve = kb * 1024;
    return SGX_OK;
}
extern "C" int32_t sgx_conv_get_wgrad_lds_reserve(void) { return g_wg_lds_reserve.load() / 1024; }
extern "C" int32_t sgx_stream_create_partial(int32_t percent, void** stream) {
    SGX_CHECK_ARG(stream && percent >= 10 && percent <= 100, "stream_create_partial: percent of the CUs in 10..100");
#ifdef SGX_EMU
    SGX_FAIL(SGX_ERR_UNSUPPORTED, "stream_create_partial: no CU masks on the host emulation");
#else
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0)
        SGX_FAIL(SGX_ERR_HIP, "stream_create_partial: cannot query the device");
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    // bit i set <=> floor((i + 1) * percent / 100) > floor(i * percent / 100): `percent` of every run of 100 consecutive CU indices, evenly
    // spaced - whatever the driver's mapping of mask bits to XCDs is (round-robin or blocked), every XCD keeps the same share
    int on = 0;
    for (int i = 0; i < ncu; ++i)
        if ((long)(i + 1) * percent / 100 > (long)i * percent / 100) {
            mask[i / 32] |= 1u << (i % 32);
            ++on;
        }
    hipStream_t st = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "hipExtStreamCreateWithCUMask(%d of %d CUs): %s", on, ncu, hipGetErrorString(e));
    *stream = (void*)st;
    return SGX_OK;
#endif
}
extern "C" int32_t sgx_stream_destroy(void* stream) {
#ifndef SGX_EMU
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "hipStreamDestroy failed");
#endif
    return SGX_OK;
}
