// Weight gradient of a 3x3 convolution (stride 1 or 2, pad 1) from an LDS-resident input PATCH, bf16x3 arithmetic ("wpatch").
// Replaces, for these layers, the slab loop of conv.hip's wgrad_kernel; reached through sgx_conv2d_bwd_weight_group like it.
// Reference semantics: the weight half of aten::convolution_backward for every nn.Conv2d(k=3) of the train step
// (sg_trainer.py:611-647 -> modules/qarepvgg_block.py:184-204, modules/conv_bn_act_block.py:88-93).
//
// Why a second kernel (round 4).  In the slab loop the GEMM-N axis is the flattened (tap, channel) axis: every input element is loaded,
// split into three bf16 pieces (11 VALU instructions per pair of floats) and stored to LDS once PER TAP, and per slab every lane redoes
// its pixel / bounds arithmetic.  Read off the ISA (r4): wgrad_kernel<128,128,...,bf16x3> issues 7.2 VALU + 1 LDS instructions per MFMA,
// the 64x64 tile 15 - on a pipe that hides about five; the split alone is 7.33 x (32 / BNK + 32 / BJ) VALU per MFMA, i.e. the narrow
// layers of YOLO-NAS-S (K = 32 ... 96) are bound by the vector unit at a quarter of the matrix rate whatever the tile.
// Here a workgroup owns a (filter tile BNK) x (channel chunk CT) x ALL NINE TAPS block of dW and walks a range of pixel TILES of 64 (stride 2: 32)
// output pixels (PR rows x PC columns, PC in {16, 8, 4} so that 160/80-, 40- and 20-wide maps all tile exactly).  Per tile it stages
//   * the dY tile   [64 | 32 pixels][BNK]                               and
//   * the X patch   [(PR-1) S + 3 rows][(PC-1) S + 3 columns][CT]       ONCE (split once per element: 1.3 - 3 VALU per MFMA),
// and the nine taps read their MFMA operands straight out of the patch: the GEMM-K axis is the pixel axis, so an operand is eight
// consecutive PIXELS of one channel per lane - ds_read_b64_tr_b16 (the LDS transpose read) delivers exactly that from the pixel-major
// image, and a tap is nothing but an LDS address offset (an instruction immediate).  The MFMA phase of a tile has no vector-ALU work at
// all: per 16 pixels and tap six reads and six MFMAs.  Stride 2: the patch columns are stored de-interleaved by parity, so that the
// pixels of a K step (input columns 2 col + dw) stay consecutive LDS rows.
// One LDS buffer, two barriers per tile: the next tile's global loads travel in registers under this tile's MFMAs; the split + store
// phase of one workgroup runs under the MFMA phase of the other workgroup(s) of the CU (>= 2 per CU by construction).
// Arithmetic: the six-product bf16x3 scheme of conv_mma.h, ONE accumulator per 32x32 block (six roundings per 16 pixels where the fp32
// pipe has sixteen; the pixel splits are summed pairwise by the fold tree below).
// The pixel splits of a (job, tile) are folded inside the launch by the arrival-walked binary tree of wgrad_kernel (fixed association,
// no float atomics, tickets left zero); a node travels as 16-byte device-scope stores in accumulator order (fully coalesced).
#include "conv_mma.h"
#include "wgrad_patch.h"

#define WP_TAPS 9
constexpr int wp_pitch(int n) { return n + (n % 64 == 0 ? 32 : 0); }  // bf16 elements per LDS pixel row: 16 or 48 banks mod 64

// S: stride; PC: tile columns (tile = 16 NKS / PC rows x PC columns); KB: 32-row filter blocks of the workgroup's dW tile (its channel chunk is
// 32 wide).  A wave owns ONE filter block and ONE tap row (three 32x32 accumulators, 48 registers): 3 KB waves per workgroup, ~110
// registers per lane, 20 - 51 KB of LDS - several workgroups per CU, three to four waves per SIMD.  (The first form of this kernel gave a
// wave all nine taps of a block - 144 accumulator registers: every 256-thread variant sat on the 256-register line, three of them
// spilling.)
// NKS: K steps (16 pixels each) per tile: a tile is 16 NKS / PC rows x PC columns (four at stride 1 - 64 pixels, half the barriers and 1.7
// instead of 2.25 patch pixels staged per output pixel; two at stride 2, whose 9 x 33 patch of a 64-pixel tile would take 57 KB of LDS)
// (round 6: the stride-2 four-column form spilled one dword under the three-wave bound - two there; tools/kernel_regs.py --check)
template <int S, int PC, int KB, int NKS>
__global__ __launch_bounds__(192 * KB, (S == 2 && PC == 4) ? 2 : 3) void wpatch_kernel(WpGroupParams g) {
    constexpr int CB = 1, WT = 3;
    constexpr int NW = KB * CB * WT, NTH = 64 * NW;
    constexpr int PIX = 16 * NKS;
    constexpr int PR = PIX / PC, RW = 16 / PC;  // tile rows; tile rows per K step of 16 pixels
    constexpr int PRin = (PR - 1) * S + 3, PCin = (PC - 1) * S + 3;
    constexpr int PCH = (PCin + 1) / 2, PSLOTS = S == 1 ? PCin : 2 * PCH;  // stride 2: even columns first, then the odd ones
    constexpr int BNK = 32 * KB, CT = 32 * CB;
    constexpr int CTP = wp_pitch(CT), DP = wp_pitch(BNK);
    constexpr int TAPW = WP_TAPS / WT;  // taps per wave
    constexpr int XPL = PRin * PSLOTS * CTP, DPL = PIX * DP;  // elements per plane
    constexpr int XG = CT / 4, DG = BNK / 4;                 // 16-byte groups per pixel
    // Staging enumeration of the patch.  One 16-lane store group covers two consecutive staging pixels; their de-interleaved stride-2 slots
    // must lie on disjoint halves of the 32 store banks: an (even, odd) column pair does (slots PCH - odd - pixel rows of 16 banks apart), an
    // (odd, even) pair does not (slots ONE row apart: the same 16 banks).  The patch is an odd number of columns wide, so a plain row-major
    // enumeration pairs (odd, even) columns in every second patch row - a two-way conflict on half of the patch stores (r4final counters:
    // 15.8 % of the stride-2 forms' LDS cycles).  Odd patch rows are therefore enumerated rotated by one column (one (even, even) pair
    // per two rows is left).  (Padding the rows to an even width instead cost the 16-column form an eighth staging item per lane: its
    // launches 0.50 -> 0.66 ms, r5e.)
    static_assert(S == 1 || (PCH & 1) == 1, "stride 2: an odd half-width keeps (even, odd) column pairs on disjoint bank halves");
    constexpr int NXE = PRin * PCin * XG, NDE = PIX * DG;    // staging items of a tile
    constexpr int NXI = (NXE + NTH - 1) / NTH, NDI = (NDE + NTH - 1) / NTH;
    static_assert(PC == 16 || PC == 8 || PC == 4, "tile columns");
    __shared__ __attribute__((aligned(16))) unsigned short Xs[3 * XPL];
    __shared__ __attribute__((aligned(16))) unsigned short Ds[3 * DPL];
    __shared__ int s_last;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / (KB * CB), kb = (wave / CB) % KB, cb = wave % CB;
    int ji = 0;
#pragma unroll
    for (int i = 1; i < WP_MAX_JOBS; ++i)
        if (i < g.njobs && (int)blockIdx.x >= g.blk0[i]) ji = i;
    const WpJob& p = g.jobs[ji];
    const int T = p.kt_tiles * p.ct_tiles;
    const int bl = (int)blockIdx.x - p.blk0;
    int split, tile;
    if (g.xcd_order && p.xcd_ranges) {  // an XCD walks the (filter tile, channel chunk) tiles of ONE pixel range back to back: the range stays in its L2
        const int r = bl >> 3;
        split = (bl & 7) + 8 * (r / T);
        tile = r % T;
    } else {
        split = bl / T;
        tile = bl % T;
    }
    if (split >= p.ksplit) return;  // whole workgroup leaves together (before any barrier)
    const int ktile = tile / p.ct_tiles, ctile = tile - ktile * p.ct_tiles;
    const int k0 = ktile * BNK, c0 = ctile * CT;
    const int H = p.H, W = p.W, C = p.C, K = p.K, Ho = p.Ho, Wo = p.Wo, pad = p.pad;
    const int t0 = split * p.tchunk, t1 = min(p.ntiles, t0 + p.tchunk);
    const int tpi = p.tiles_h * p.tiles_w;  // tiles per image
    int img = t0 / tpi;
    const int img0 = img;
    int th = (t0 - img * tpi) / p.tiles_w, tw = t0 - img * tpi - th * p.tiles_w;
    const sgx_buf bufX = sgx_make_buf(p.X + (long)img0 * p.x_ld_img, p.x_bytes - (long)img0 * p.x_ld_img * 4);
    const sgx_buf bufD = sgx_make_buf(p.DY + (long)img0 * p.y_ld_img, p.dy_bytes - (long)img0 * p.y_ld_img * 4);

    // ---- this lane's staging items: fixed (patch pixel, channel group) / (tile pixel, filter group) -> byte delta from the tile's base,
    // LDS offset, coordinates inside the patch / tile for the border test
    int xdelta[NXI], xlds[NXI], xrc[NXI];
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int e = tid + i * NTH;
        const int pix = e / XG, cg = e - pix * XG;
        const int pr = pix / PCin, pj = pix - pr * PCin;
        const int pc = S == 2 ? (pj + (pr & 1)) % PCin : pj;
        const bool ok = e < NXE && c0 + 4 * cg < C;
        xdelta[i] = (int)((((long)pr * W + pc) * p.x_ld_pix + 4 * cg) * 4);
        const int slot = S == 1 ? pc : (pc & 1) * PCH + (pc >> 1);
        xlds[i] = (pr * PSLOTS + slot) * CTP + 4 * cg;
        xrc[i] = ok ? (pr | (pc << 8)) : -1;
    }
    int ddelta[NDI], dlds[NDI], drc[NDI];
#pragma unroll
    for (int i = 0; i < NDI; ++i) {
        const int e = tid + i * NTH;
        const int pix = e / DG, kg = e - pix * DG;
        const int tr = pix / PC, tc = pix - tr * PC;
        const bool ok = e < NDE && k0 + 4 * kg < K;
        ddelta[i] = (int)((((long)tr * Wo + tc) * p.y_ld_pix + 4 * kg) * 4);
        dlds[i] = pix * DP + 4 * kg;
        drc[i] = ok ? (tr | (tc << 8)) : -1;
    }
    float4 rx[NXI], rd[NDI];
    // global loads of the tile (img, th, tw); the tile walk (scalar) moves on afterwards
    auto load_tile = [&]() {
        const int ho0 = th * PR, wo0 = tw * PC;
        const int hs = ho0 * S - pad, ws = wo0 * S - pad;
        const int xbase = (int)(((long)(img - img0) * p.x_ld_img + ((long)hs * W + ws) * p.x_ld_pix + c0) * 4);
        const int dbase = (int)(((long)(img - img0) * p.y_ld_img + ((long)ho0 * Wo + wo0) * p.y_ld_pix + k0) * 4);
        const bool inside = hs >= 0 && hs + PRin <= H && ws >= 0 && ws + PCin <= W && ho0 + PR <= Ho && wo0 + PC <= Wo;
        if (inside) {  // uniform: no per-lane border arithmetic for the tiles in the middle of the image
#pragma unroll
            for (int i = 0; i < NXI; ++i) rx[i] = sgx_buf_ld4(bufX, xrc[i] >= 0 ? (unsigned)(xbase + xdelta[i]) : SGX_BUF_OOB);
#pragma unroll
            for (int i = 0; i < NDI; ++i) rd[i] = sgx_buf_ld4(bufD, drc[i] >= 0 ? (unsigned)(dbase + ddelta[i]) : SGX_BUF_OOB);
        } else {
#pragma unroll
            for (int i = 0; i < NXI; ++i) {
                const int hi = hs + (xrc[i] & 0xff), wi = ws + ((xrc[i] >> 8) & 0xff);
                const bool ok = xrc[i] >= 0 && hi >= 0 && hi < H && wi >= 0 && wi < W;
                rx[i] = sgx_buf_ld4(bufX, ok ? (unsigned)(xbase + xdelta[i]) : SGX_BUF_OOB);
            }
#pragma unroll
            for (int i = 0; i < NDI; ++i) {
                const bool ok = drc[i] >= 0 && ho0 + (drc[i] & 0xff) < Ho && wo0 + ((drc[i] >> 8) & 0xff) < Wo;
                rd[i] = sgx_buf_ld4(bufD, ok ? (unsigned)(dbase + ddelta[i]) : SGX_BUF_OOB);
            }
        }
        if (++tw == p.tiles_w) {
            tw = 0;
            if (++th == p.tiles_h) {
                th = 0;
                ++img;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NXI; ++i)
            if (NXE % NTH == 0 || tid + i * NTH < NXE) {
                uint2 h, m, l;
                sgx_split3(rx[i], h, m, l);
                unsigned short* const b = &Xs[xlds[i]];
                *reinterpret_cast<uint2*>(b) = h;
                *reinterpret_cast<uint2*>(b + XPL) = m;
                *reinterpret_cast<uint2*>(b + 2 * XPL) = l;
            }
#pragma unroll
        for (int i = 0; i < NDI; ++i)
            if (NDE % NTH == 0 || tid + i * NTH < NDE) {
                uint2 h, m, l;
                sgx_split3(rd[i], h, m, l);
                unsigned short* const b = &Ds[dlds[i]];
                *reinterpret_cast<uint2*>(b) = h;
                *reinterpret_cast<uint2*>(b + DPL) = m;
                *reinterpret_cast<uint2*>(b + 2 * DPL) = l;
            }
    };

    sgx_f32x16 acc[TAPW];
#pragma unroll
    for (int t = 0; t < TAPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- MFMA operand addressing.  Lane l = 16 gq + q of a wave reads, per transpose read, the 4 bf16 at [pixel row (q >> 2)][columns
    // 16 (gq & 1) + 4 (q & 3) ..] of a [4 pixels][32 columns] block and receives column l % 32 of those four pixels; K half l / 32.
    const int gq = lane >> 4, q = lane & 15, kh = lane >> 5;
    const int klo = 8 * kh + (q >> 2);  // first pixel (of the K step) of this lane's first read
    const int colsel = 16 * (gq & 1) + 4 * (q & 3);
    const unsigned short* const aL = Ds + klo * DP + kb * 32 + colsel;
    const int pr_l = klo / PC, pc_l = klo - pr_l * PC;
    constexpr int XHI = PC >= 8 ? 4 * CTP : S * PSLOTS * CTP;  // pixels klo + 4 ..: same tile row (PC >= 8) or the next one (PC = 4)
    const unsigned short* const bL = Xs + ((pr_l * S + (WT == 3 ? wt : 0)) * PSLOTS + pc_l) * CTP + cb * 32 + colsel;
    auto afrag = [&](int ks, int pl) {
        const unsigned short* const s = aL + pl * DPL + 16 * ks * DP;
        const uint2 lo = sgx_lds_tr_read(s), hi = sgx_lds_tr_read(s + 4 * DP);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    auto bfrag = [&](int ks, int dh, int dw, int pl) {  // (dh: relative to the wave's first tap row)
        const int slot = S == 1 ? dw : (dw & 1) * PCH + (dw >> 1);
        const unsigned short* const s = bL + pl * XPL + ((ks * RW * S + dh) * PSLOTS + slot) * CTP;
        const uint2 lo = sgx_lds_tr_read(s), hi = sgx_lds_tr_read(s + XHI);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    // The MFMA phase of a tile: two K steps x TAPW taps x six products.  Left to itself hipcc schedules "read - wait - multiply" per product
    // pair (r4b: three exposed LDS round trips per tap, waves parked in s_waitcnt for half their cycles, matrix pipe busy 23 %).  The order
    // is therefore fixed by hand between scheduling fences: all operands of K step 0 are requested up front (one exposed round trip per
    // tile); while tap t multiplies, the reads of one operand set of K step 1 are in flight into the registers tap t - 1 has released.
    auto six = [&](sgx_f32x16& c, const uint4 (&a)[3], const uint4 (&b)[3]) {  // smallest terms first
        c = sgx_mfma_bf16(a[2], b[0], c);
        c = sgx_mfma_bf16(a[0], b[2], c);
        c = sgx_mfma_bf16(a[1], b[1], c);
        c = sgx_mfma_bf16(a[1], b[0], c);
        c = sgx_mfma_bf16(a[0], b[1], c);
        c = sgx_mfma_bf16(a[0], b[0], c);
    };
    auto mfma_tile = [&]() {
        static_assert(TAPW == 3, "the hand-placed order below is written for one tap row per wave");
        // steps s = 0 .. 3 NKS - 1 = (K step s / 3, tap s % 3).  Operand sets: A of a K step in a[ks & 1] (requested during the first tap of
        // the K step before), B of step s in b[s % 3] (requested during step s - 2, into the registers step s - 3 released).
        uint4 a[2][3], b[3][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) a[0][pl] = afrag(0, pl);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) b[t][pl] = bfrag(0, 0, t, pl);
        sgx_sched_fence();
#pragma unroll
        for (int st = 0; st < 3 * NKS; ++st) {
            const int ks = st / 3, t = st % 3;
            if (t == 0 && ks + 1 < NKS) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) a[(ks + 1) & 1][pl] = afrag(ks + 1, pl);
            }
            if (st >= 1 && st + 2 < 3 * NKS) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) b[(st + 2) % 3][pl] = bfrag((st + 2) / 3, 0, (st + 2) % 3, pl);
            }
            six(acc[t], a[ks & 1], b[st % 3]);
            sgx_sched_fence();
        }
    };

    const int ntl = t1 - t0;
    if (ntl > 0) {
        load_tile();
        store_tile();
    }
    __syncthreads();
    for (int t = 0; t < ntl; ++t) {
        const bool more = t + 1 < ntl;
        if (more) load_tile();  // in flight under the MFMAs
        mfma_tile();
        __syncthreads();        // every wave is done reading this tile
        if (more) store_tile();
        __syncthreads();
    }

    // ---- fold the pixel splits of this (job, tile): the arrival-walked binary tree of wgrad_kernel (conv.hip) --------------------------
    const int ksplit = p.ksplit;
    constexpr int TE = NW * TAPW * 16 * 64;  // floats of a node: [wave][tap][register quad][lane][4]
    if (ksplit > 1) {
        float* const base = p.part + (long)tile * ksplit * TE;
        int* const tk = p.tickets + (long)tile * ksplit;
        const int foff = wave * (TAPW * 16 * 64) + lane * 4;
        for (int L = 0; (1 << L) < ksplit; ++L) {
            const int i = split >> L, sib = i ^ 1;
            if (((long)sib << L) >= ksplit) continue;  // no sibling on this level: the value passes up as it is
            float* const mine = base + ((long)i << L) * TE + foff;
#pragma unroll
            for (int t = 0; t < TAPW; ++t) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    sgx_st4_dev(mine + (t * 4 + r4) * 256, make_float4(acc[t][4 * r4], acc[t][4 * r4 + 1], acc[t][4 * r4 + 2], acc[t][4 * r4 + 3]));
                sgx_sched_fence();
            }
            sgx_wait_stores();
            __syncthreads();
            if (tid == 0) {
                int* const tp = &tk[((long)(i | 1)) << L];
                const int old = atomicAdd(tp, 1);
                s_last = old;
                if (old) *tp = 0;
            }
            __syncthreads();
            const int second = s_last;
            __syncthreads();  // (s_last is rewritten on the next level)
            if (!second) return;
            const float* const other = base + ((long)sib << L) * TE + foff;
#pragma unroll
            for (int t = 0; t < TAPW; ++t) {
                float4 v[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) v[r4] = sgx_ld4_dev(other + (t * 4 + r4) * 256);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    acc[t][4 * r4] += v[r4].x;
                    acc[t][4 * r4 + 1] += v[r4].y;
                    acc[t][4 * r4 + 2] += v[r4].z;
                    acc[t][4 * r4 + 3] += v[r4].w;
                }
                sgx_sched_fence();
            }
        }
    }
    // ---- the root: dW[k][tap][c] += acc (OHWI)
    float* const dw = p.dw;
    const int c = c0 + cb * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < TAPW; ++t) {
        const int tap = (WT == 3 ? 3 * wt : 0) + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (k < K && c < C) dw[((long)k * WP_TAPS + tap) * C + c] += acc[t][r];
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------
// choice of the kernel form for a job (0 = not a patch problem: the slab loop takes it)
bool wpatch_plan_job(const sgx_conv_desc* d, WpPlan& pl, int kb_override, int min_fill_pct) {
    pl.cfg = 0;
    if (d->R != 3 || d->S != 3 || d->pad != 1 || (d->stride != 1 && d->stride != 2)) return false;
    if (d->C % 4 != 0 || d->K % 4 != 0 || d->x_ld_pix % 4 != 0 || d->y_ld_pix % 4 != 0) return false;
    if (d->Ho < 1 || d->Wo < 1) return false;
    // filter blocks per workgroup: ONE (three waves).  r4b / r4c (lab, every 3x3 layer of a YOLO-NAS-S step, alone on the chip): one block
    // 10.98 ms per step, "two from 192 filters on" 12.37, two everywhere 14.24, three 15.7 - although a one-block workgroup re-stages the
    // patch once per 32 filters (K = 768: 24 times).  Many small workgroups keep the matrix pipe fed while others stage or wait at their
    // barriers; the vector work per MFMA (3.05 instructions) is not what bounds the kernel.  (kb_override: measurement.)
    int kb = 1;
    if (kb_override >= 1 && kb_override <= 3) kb = kb_override;  // measurement
    const int cb = 1, wt = 3;
    const int bnk = 32 * kb, ct = 32 * cb;
    pl.kt_tiles = sgx_cdiv(d->K, bnk);
    pl.ct_tiles = sgx_cdiv(d->C, ct);
    // tile columns: the exact fit with the widest rows
    const int nks = d->stride == 1 ? 4 : 2, pix = 16 * nks;
    int best = 16;
    double bu = -1.0;
    const int cand[3] = {16, 8, 4};
    for (int i = 0; i < 3; ++i) {
        const int pc = cand[i], pr = pix / pc;
        const double u = ((double)d->Wo / (sgx_cdiv(d->Wo, pc) * pc)) * ((double)d->Ho / (sgx_cdiv(d->Ho, pr) * pr));
        if (u > bu + 1e-9) bu = u, best = pc;
    }
    const double fill = ((double)d->K / (pl.kt_tiles * bnk)) * ((double)d->C / (pl.ct_tiles * ct)) * bu;
    if (fill * 100.0 < (double)min_fill_pct) return false;  // padded matrix work the slab loop (flattened tap x channel axis) would not do
    pl.pc = best;
    pl.nks = nks;
    pl.kb = kb; pl.cb = cb; pl.wt = wt;
    pl.tiles_h = sgx_cdiv(d->Ho, pix / best);
    pl.tiles_w = sgx_cdiv(d->Wo, best);
    pl.ntiles = (long)d->N * pl.tiles_h * pl.tiles_w;
    pl.cfg = 1;
    return true;
}

// pixel-tile ranges of the jobs of one group: items of ~item_flops (the caller sizes them for the whole group)
void wpatch_plan_split(const sgx_conv_desc* d, WpPlan& pl, double item_flops, long* part_floats, long* ticket_ints) {
    const int bnk = 32 * pl.kb, ct = 32 * pl.cb, nw = pl.kb * pl.cb * pl.wt;
    const double tile_flops = 2.0 * 16 * pl.nks * bnk * ct * WP_TAPS;
    long tchunk = (long)(item_flops / tile_flops);
    if (tchunk < 16 / pl.nks) tchunk = 16 / pl.nks;  // at least 256 pixels per item
    if (tchunk > pl.ntiles) tchunk = pl.ntiles;
    long ks = (pl.ntiles + tchunk - 1) / tchunk;
    if (ks > WP_MAX_SPLIT) ks = WP_MAX_SPLIT;
    // a split's lane offsets are 31-bit: every split under 1 GiB of either operand
    const long big = (long)d->N * (d->x_ld_img > d->y_ld_img ? d->x_ld_img : d->y_ld_img) * 4;
    const long need = big / (1L << 30) + 1;
    if (ks < need) ks = need;
    tchunk = (pl.ntiles + ks - 1) / ks;
    ks = (pl.ntiles + tchunk - 1) / tchunk;
    pl.ksplit = (int)ks;
    pl.tchunk = (int)tchunk;
    const long te = (long)nw * (WP_TAPS / pl.wt) * 16 * 64;
    const long tiles = (long)pl.kt_tiles * pl.ct_tiles;
    pl.part_off = *part_floats;
    pl.ticket_off = *ticket_ints;
    if (ks > 1) {
        *part_floats += tiles * ks * te;
        *ticket_ints += tiles * ks;
    }
}

template <int S, int PC, int KB>
static void wpatch_launch_t(const WpGroupParams& g, int nblk, void* stream) {
    SGX_LAUNCH((wpatch_kernel<S, PC, KB, S == 1 ? 4 : 2>), dim3((unsigned)nblk), dim3(192 * KB), wg_lds_pad(wpatch_kernel<S, PC, KB, S == 1 ? 4 : 2>), stream, g);
}
template <int S, int KB>
static void wpatch_launch_pc(int pc, const WpGroupParams& g, int nblk, void* stream) {
    if (pc == 16) wpatch_launch_t<S, 16, KB>(g, nblk, stream);
    else if (pc == 8) wpatch_launch_t<S, 8, KB>(g, nblk, stream);
    else wpatch_launch_t<S, 4, KB>(g, nblk, stream);
}
template <int S>
static void wpatch_launch_kb(const WpPlan& form, const WpGroupParams& g, int nblk, void* stream) {
    if (form.kb == 1) wpatch_launch_pc<S, 1>(form.pc, g, nblk, stream);
    else if (form.kb == 2) wpatch_launch_pc<S, 2>(form.pc, g, nblk, stream);
    else wpatch_launch_pc<S, 3>(form.pc, g, nblk, stream);
}
// one launch: jobs of ONE kernel form (stride, tile columns, filter blocks)
int32_t wpatch_launch(int stride, const WpPlan& form, const WpGroupParams& g, int nblk, void* stream) {
    if (form.kb < 1 || form.kb > 3 || form.cb != 1 || form.wt != 3) SGX_FAIL(SGX_ERR_UNSUPPORTED, "conv bwd_weight (patch): no wave grid %dx%dx%d", form.kb, form.cb, form.wt);
    if (stride == 1) wpatch_launch_kb<1>(form, g, nblk, stream);
    else wpatch_launch_kb<2>(form, g, nblk, stream);
    SGX_CHECK_LAUNCH("wgrad (patch)");
    return SGX_OK;
}
