// PPYoloELoss on gfx950: DFL decode, TaskAligned / ATSS assignment, varifocal|focal + GIoU + DFL with a
// hand-written backward.  No MFMA here: these are wavefront-reduction / streaming kernels bound by HBM.
//
// What the reference does (training/losses/ppyolo_loss.py) vs. what runs here:
//  * reference materialises ~10 [B,n,L] fp32 temporaries and a [B,n,13,L] int64 one-hot (:229), syncs the
//    host three times (:533, :758, :1033).  Here a (image, GT) workgroup streams the L anchors once, keeps the
//    alignment metric in LDS, picks the top-k with k block-wide arg-max rounds, and only scatters k hits;
//    per-anchor claim counts replace the [B,n,L] masks; nothing returns to the host.
//  * the loss is linear in its four sums, so the forward also writes d(sum)/d(logits|distri) once and the
//    backward is a single scale by upstream/max(score_sum,1).
// Compile with -ffp-contract=off: decisions (top-k, arg-max, thresholds) follow the CPU op-by-op rounding.
#include "sgx_common.h"

#define LOSS_THREADS 256
#define TAL_EPS 1e-9f

struct LossWs {
    float* pbox;    // [B,L,4] decoded boxes, grid units
    int* cnt;       // [B,L] number of GTs claiming the anchor
    int* sel;       // [B,L] a claiming GT (exact when cnt == 1)
    int* agt;       // [B,L] resolved GT slot or -1
    float* am;      // [B,L] alignment metric at the resolved pair
    float* aiou;    // [B,L] IoU(pred, gt) at the resolved pair
    unsigned* maxm; // [B,nmax] per-GT max metric (float bits, values >= 0)
    unsigned* maxi; // [B,nmax] per-GT max IoU
    float* partials;// [nblk][4]
    long bytes;
};

static long align_up(long v) { return (v + 255) & ~255L; }

static LossWs carve(const sgx_loss_desc* d, void* ws) {
    LossWs w;
    char* p = (char*)ws;
    long BL = (long)d->B * d->L, Bn = (long)d->B * (d->nmax > 0 ? d->nmax : 1);
    long off = 0;
    w.pbox = (float*)(p + off); off += align_up(BL * 4 * 4);
    w.cnt = (int*)(p + off); off += align_up(BL * 4);
    w.sel = (int*)(p + off); off += align_up(BL * 4);
    w.agt = (int*)(p + off); off += align_up(BL * 4);
    w.am = (float*)(p + off); off += align_up(BL * 4);
    w.aiou = (float*)(p + off); off += align_up(BL * 4);
    w.maxm = (unsigned*)(p + off); off += align_up(Bn * 4);
    w.maxi = (unsigned*)(p + off); off += align_up(Bn * 4);
    const long cls_items = (long)d->B * d->L * (d->C % 4 == 0 ? d->C / 4 : d->C);  // float4 groups, or single classes when C % 4 != 0
    long nblk = (BL + LOSS_THREADS - 1) / LOSS_THREADS + (cls_items + LOSS_THREADS - 1) / LOSS_THREADS;
    w.partials = (float*)(p + off); off += align_up((nblk + 8) * 4 * 4);
    w.bytes = off;
    return w;
}

extern "C" int64_t sgx_ppyoloe_loss_workspace(const sgx_loss_desc* d) { return carve(d, nullptr).bytes + 256; }

// ---------------------------------------------------------------------------------------------
// geometry helpers (operation order follows ppyolo_loss.py:17-57, 178-211)
// ---------------------------------------------------------------------------------------------
struct Box {
    float x1, y1, x2, y2;
};
__device__ __forceinline__ float clip0(float v) { return v > 0.f ? v : 0.f; }
__device__ __forceinline__ float pair_iou(const Box& g, const Box& p, float eps) {
    float lx = fmaxf(p.x1, g.x1), ly = fmaxf(p.y1, g.y1);
    float rx = fminf(p.x2, g.x2), ry = fminf(p.y2, g.y2);
    float ov = clip0(rx - lx) * clip0(ry - ly);
    float a1 = clip0(g.x2 - g.x1) * clip0(g.y2 - g.y1);
    float a2 = clip0(p.x2 - p.x1) * clip0(p.y2 - p.y1);
    float un = a1 + a2 - ov + eps;
    return ov / un;
}
__device__ __forceinline__ bool point_in_box(float px, float py, const Box& g) {
    float m = fminf(fminf(px - g.x1, py - g.y1), fminf(g.x2 - px, g.y2 - py));
    return m > 1e-9f;
}
// targets row -> class, xyxy (cxcywh.py:36-56: x2 = x1 + w)
__device__ __forceinline__ Box target_box(const float* t) {
    float x1 = t[2] - 0.5f * t[4], y1 = t[3] - 0.5f * t[5];
    Box b{x1, y1, x1 + t[4], y1 + t[5]};
    return b;
}
__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// ---------------------------------------------------------------------------------------------
// 0. targets [T,6] -> per-image slot lists (order of appearance preserved, ppyolo_loss.py:747-769)
// ---------------------------------------------------------------------------------------------
// One wave per image: the wave walks the target rows 64 at a time, a ballot marks the rows of this image, a row's slot is the image's
// running count plus the matches in the lanes below it - order of appearance preserved.  (Rounds 1-4: one THREAD per image walked all T
// rows, a chain of T dependent loads - 95 us at T = 640 on the critical path of every step.)
__global__ __launch_bounds__(64) void targets_index_kernel(const float* targets, int T, int B, int nmax, int* gt_count, int* gt_index, int* overflow) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int n = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        float v = -1.f;
        if (t < T) v = targets[(long)t * 6];
        const bool hit = t < T && (int)v == b && v == (float)b;
        const unsigned long long bits = __ballot(hit);
        if (hit) {
            const int slot = n + __popcll(bits & ((1ull << lane) - 1ull));
            if (slot < nmax) gt_index[(long)b * nmax + slot] = t;
        }
        n += __popcll(bits);
    }
    if (lane == 0) {
        gt_count[b] = n < nmax ? n : nmax;
        if (n > nmax) atomicAdd(overflow, n - nmax);
    }
    for (int i = n + lane; i < nmax; i += 64) gt_index[(long)b * nmax + i] = -1;
}
extern "C" int32_t sgx_targets_index(const float* targets, int32_t T, int32_t B, int32_t nmax, int32_t* gt_count, int32_t* gt_index,
                                     int32_t* overflow, void* stream) {
    SGX_CHECK_ARG(gt_count && overflow && B > 0 && T >= 0 && nmax >= 0, "targets_index: bad args");
    SGX_CHECK_ARG(T == 0 || targets, "targets_index: null targets");
    SGX_CHECK_ARG(nmax == 0 || gt_index, "targets_index: null gt_index");
    SGX_MEMSET_ASYNC(overflow, 0, sizeof(int), stream);
    SGX_LAUNCH(targets_index_kernel, dim3(B), dim3(64), 0, stream, targets, T, B, nmax, gt_count, gt_index, overflow);
    SGX_CHECK_LAUNCH("targets_index");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
// 1. decode: softmax over R+1 bins x [0..R] -> ltrb -> xyxy (grid units).  One thread per (b,l).
//    Also used (with scores/boxes outputs) as the head decode of dfl_heads.py:207-235.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void softmax_expect(const float* d, int R1, float& e) {
    float m = d[0];
    for (int j = 1; j < R1; ++j) m = fmaxf(m, d[j]);
    float s = 0.f, acc = 0.f;
    for (int j = 0; j < R1; ++j) {
        float ex = expf(d[j] - m);
        s += ex;
        acc += ex * (float)j;
    }
    e = acc / s;
}
// One thread per (anchor, side).  A workgroup's 256 rows of (R+1) floats are one contiguous run of memory: it is staged through LDS with
// coalesced 4-byte loads (thread t takes elements t, t + 256, ...) and every thread then reads ITS row out of LDS (row pitch R+1 - odd for
// the networks' reg_max = 16: conflict-free) - the per-side arithmetic and its order are unchanged.  (Rounds 1-4: every thread read its
// 68-byte row from global memory twice, element by element - 64 lanes x 68 B strides per instruction: 82 us for 73 MB.)
#define DEC_MAXR1 33
template <bool PX>
__global__ __launch_bounds__(256) void decode_rows_kernel(int B, int L, int R1, const float* distri, const float* points, const float* strides, float mul_stride,
                                                          float* boxes) {
    __shared__ float rows[256 * DEC_MAXR1];
    const long n = (long)B * L * 4;
    for (long i0 = (long)blockIdx.x * 256; i0 < n; i0 += (long)gridDim.x * 256) {
        const long cnt = (n - i0 < 256 ? n - i0 : 256) * R1;
        const float* src = distri + i0 * R1;
        for (long q = threadIdx.x; q < cnt; q += 256) rows[q] = src[q];
        __syncthreads();
        const long i = i0 + threadIdx.x;
        if (i < n) {
            const int k = (int)(i & 3);
            const int l = (int)((i >> 2) % L);
            float e;
            softmax_expect(rows + threadIdx.x * R1, R1, e);
            if (PX) {  // decode from PIXEL anchor points: grid point = point / stride (ppyolo_loss.py:803-804)
                const float pc = points[2 * l + (k & 1)] / strides[l];
                boxes[i] = k < 2 ? pc - e : pc + e;
            } else {
                const float pc = points[2 * l + (k & 1)];
                const float sc = mul_stride != 0.f ? strides[l] : 1.f;
                boxes[i] = (k < 2 ? pc - e : pc + e) * sc;
            }
        }
        __syncthreads();
    }
}
__global__ void decode_kernel(int B, int L, int R1, const float* distri, const float* points_grid, const float* strides, float mul_stride,
                              float* boxes) {  // (reg_max > 32: the direct form)
    long n = (long)B * L * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 3);
        const int l = (int)((i >> 2) % L);
        float e;
        softmax_expect(distri + i * R1, R1, e);
        const float pc = points_grid[2 * l + (k & 1)];
        const float s = mul_stride != 0.f ? strides[l] : 1.f;
        boxes[i] = (k < 2 ? pc - e : pc + e) * s;
    }
}
__global__ void sigmoid_kernel(const float* x, float* y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = sigmoidf(x[i]);
}
extern "C" int32_t sgx_dfl_decode(int32_t B, int32_t L, int32_t C, int32_t reg_max, const float* logits, const float* distri,
                                  const float* points_grid, const float* strides, float* boxes, float* scores, void* stream) {
    SGX_CHECK_ARG(distri && points_grid && strides && boxes, "dfl_decode: null pointer");
    long n = (long)B * L, blocks = (4 * n + 255) / 256;
    if (reg_max + 1 <= DEC_MAXR1)
        SGX_LAUNCH(decode_rows_kernel<false>, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, B, L, reg_max + 1, distri, points_grid,
                   strides, 1.f, boxes);
    else
        SGX_LAUNCH(decode_kernel, dim3((unsigned)(blocks > 16384 ? 16384 : blocks)), dim3(256), 0, stream, B, L, reg_max + 1, distri, points_grid,
                   strides, 1.f, boxes);
    SGX_CHECK_LAUNCH("dfl_decode");
    if (scores) {
        SGX_CHECK_ARG(logits, "dfl_decode: scores requested without logits");
        long m = n * C, b2 = (m + 255) / 256;
        SGX_LAUNCH(sigmoid_kernel, dim3((unsigned)(b2 > 8192 ? 8192 : b2)), dim3(256), 0, stream, logits, scores, m);
        SGX_CHECK_LAUNCH("sigmoid");
    }
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
// block-wide arg-max / arg-min over LDS values; ties -> smallest index.  All threads must call.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_argbest(float v, int idx, bool want_max, float* sv, int* si, float& out_v, int& out_i) {
    // wave reduce
    for (int off = 32; off > 0; off >>= 1) {
        float ov = __shfl_down(v, off);
        int oi = __shfl_down(idx, off);
        bool better = want_max ? (ov > v || (ov == v && oi < idx)) : (ov < v || (ov == v && oi < idx));
        if (oi >= 0 && (idx < 0 || better)) {
            v = ov;
            idx = oi;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) {
        sv[wave] = v;
        si[wave] = idx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bv = sv[0];
        int bi = si[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) {
            float ov = sv[w];
            int oi = si[w];
            bool better = want_max ? (ov > bv || (ov == bv && oi < bi)) : (ov < bv || (ov == bv && oi < bi));
            if (oi >= 0 && (bi < 0 || better)) {
                bv = ov;
                bi = oi;
            }
        }
        sv[0] = bv;
        si[0] = bi;
    }
    __syncthreads();
    out_v = sv[0];
    out_i = si[0];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// 2a. TaskAligned candidates: one workgroup per (GT slot, image).  ppyolo_loss.py:506-524, 214-230
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LOSS_THREADS) void tal_candidates_kernel(sgx_loss_desc d, const float* logits, const float* points,
                                                                      const float* strides, const float* targets, const int* gt_count,
                                                                      const int* gt_index, LossWs w, int topk, float alpha, float beta) {
    SGX_DYN_SMEM(float, metric);  // [L]
    __shared__ float sv[LOSS_THREADS / 64];
    __shared__ int si[LOSS_THREADS / 64];
    const int g = blockIdx.x, b = blockIdx.y;
    if (g >= gt_count[b]) return;  // uniform per workgroup
    const float* t = targets + (long)gt_index[(long)b * d.nmax + g] * 6;
    const Box gt = target_box(t);
    const int cls = (int)t[1];
    bool valid = (gt.x1 + gt.y1 + gt.x2 + gt.y2) > 0.f;  // pad_gt_mask, ppyolo_loss.py:754
    const int L = d.L;
    for (int l = threadIdx.x; l < L; l += LOSS_THREADS) {
        const float* pb = w.pbox + ((long)b * L + l) * 4;
        const float s = strides[l];
        Box p{pb[0] * s, pb[1] * s, pb[2] * s, pb[3] * s};
        float iou = pair_iou(gt, p, 1e-9f);
        float sc = sigmoidf(logits[((long)b * L + l) * d.C + cls]);
        float m = powf(sc, alpha) * powf(iou, beta);
        bool in = point_in_box(points[2 * l], points[2 * l + 1], gt);
        metric[l] = in ? m : 0.f;
    }
    __syncthreads();
    const int k = topk < L ? topk : L;
    for (int round = 0; round < k; ++round) {
        float bv = -1.f;
        int bi = -1;
        for (int l = threadIdx.x; l < L; l += LOSS_THREADS) {
            float v = metric[l];
            if (v > bv) {  // strict: keeps the smallest index among equal values within a thread
                bv = v;
                bi = l;
            }
        }
        float rv;
        int ri;
        block_argbest(bv, bi, true, sv, si, rv, ri);
        if (ri < 0) break;  // uniform
        // sequential assignment (pad_gt_mask = None): the GT is kept iff its best candidate metric exceeds eps
        // (gather_topk_anchors, ppyolo_loss.py:224-226); round 0 delivers that maximum.
        if (round == 0 && d.sequential_assignment) valid = rv > 1e-9f;
        if (threadIdx.x == 0) {
            metric[ri] = -2.f;  // taken
            if (valid && point_in_box(points[2 * ri], points[2 * ri + 1], gt)) {
                atomicAdd(&w.cnt[(long)b * L + ri], 1);
                w.sel[(long)b * L + ri] = g;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// 2b. ATSS candidates: per (GT slot, image); top-k nearest anchor centres per level, mean+std threshold.
//     ppyolo_loss.py:355-386, 281-299
// ---------------------------------------------------------------------------------------------
#define ATSS_MAXCAND 128
__global__ __launch_bounds__(LOSS_THREADS) void atss_candidates_kernel(sgx_loss_desc d, const float* anchors, const float* targets,
                                                                       const int* gt_count, const int* gt_index, LossWs w, int topk) {
    SGX_DYN_SMEM(float, dist);  // [L]
    __shared__ float sv[LOSS_THREADS / 64];
    __shared__ int si[LOSS_THREADS / 64];
    __shared__ int cand[ATSS_MAXCAND];
    __shared__ float ciou[ATSS_MAXCAND];
    const int g = blockIdx.x, b = blockIdx.y;
    if (g >= gt_count[b]) return;
    const float* t = targets + (long)gt_index[(long)b * d.nmax + g] * 6;
    const Box gt = target_box(t);
    const bool valid = d.sequential_assignment ? true : (gt.x1 + gt.y1 + gt.x2 + gt.y2) > 0.f;  // pad_gt_mask (None when sequential)
    const int L = d.L;
    const float gcx = (gt.x1 + gt.x2) / 2.f, gcy = (gt.y1 + gt.y2) / 2.f;
    for (int l = threadIdx.x; l < L; l += LOSS_THREADS) {
        const float* a = anchors + (long)l * 4;
        float acx = (a[0] + a[2]) / 2.f, acy = (a[1] + a[3]) / 2.f;
        float dx = gcx - acx, dy = gcy - acy;
        dist[l] = sqrtf(dx * dx + dy * dy);
    }
    __syncthreads();
    int ncand = 0, off = 0;
    for (int lev = 0; lev < d.num_levels; ++lev) {
        const int n = d.level_count[lev];
        const int k = topk < n ? topk : n;
        for (int round = 0; round < k; ++round) {
            float bv = 3.0e38f;
            int bi = -1;
            for (int l = off + (int)threadIdx.x; l < off + n; l += LOSS_THREADS) {
                float v = dist[l];
                if (v < bv) {
                    bv = v;
                    bi = l;
                }
            }
            float rv;
            int ri;
            block_argbest(bv, bi, false, sv, si, rv, ri);
            if (ri < 0) break;
            if (threadIdx.x == 0) {
                dist[ri] = 3.3e38f;
                if (ncand + round < ATSS_MAXCAND) cand[ncand + round] = ri;
            }
            __syncthreads();
        }
        ncand += k;
        off += n;
    }
    if (ncand > ATSS_MAXCAND) ncand = ATSS_MAXCAND;
    // candidate IoUs with the ANCHOR boxes (eps 1e-10), zero for padded GTs (is_in_topk * pad_gt_mask)
    for (int i = threadIdx.x; i < ncand; i += LOSS_THREADS) {
        const float* a = anchors + (long)cand[i] * 4;
        Box ab{a[0], a[1], a[2], a[3]};
        ciou[i] = valid ? pair_iou(gt, ab, 1e-10f) : 0.f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mean = 0.f;
        for (int i = 0; i < ncand; ++i) mean += ciou[i];
        mean /= (float)ncand;
        float var = 0.f;
        for (int i = 0; i < ncand; ++i) var += (ciou[i] - mean) * (ciou[i] - mean);
        float sd = ncand > 1 ? sqrtf(var / (float)(ncand - 1)) : 0.f;  // torch.std: unbiased
        float thr = mean + sd;
        for (int i = 0; i < ncand; ++i) {
            int l = cand[i];
            const float* a = anchors + (long)l * 4;
            float acx = (a[0] + a[2]) / 2.f, acy = (a[1] + a[3]) / 2.f;
            if (valid && ciou[i] > thr && point_in_box(acx, acy, gt)) {
                atomicAdd(&w.cnt[(long)b * L + l], 1);
                w.sel[(long)b * L + l] = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3. resolve: anchors claimed by several GTs go to the arg-max-IoU GT over ALL slots of the image
//    (ppyolo_loss.py:527-538 / 389-406); per-GT maxima for the TAL rescale (:553-556).
// ---------------------------------------------------------------------------------------------
__global__ void resolve_kernel(sgx_loss_desc d, const float* logits, const float* anchors, const float* strides, const float* targets,
                               const int* gt_count, const int* gt_index, LossWs w, float alpha, float beta) {
    long n = (long)d.B * d.L;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / d.L), l = (int)(i % d.L);
        const int c = w.cnt[i];
        int g = -1;
        const float s = strides[l];
        const float* pb = w.pbox + i * 4;
        Box p{pb[0] * s, pb[1] * s, pb[2] * s, pb[3] * s};
        if (c == 1) g = w.sel[i];
        else if (c > 1) {
            float best = -1.f;
            Box q = p;
            if (d.use_static_assigner) {
                const float* a = anchors + (long)l * 4;
                q = Box{a[0], a[1], a[2], a[3]};
            }
            const int ng = gt_count[b];
            for (int k = 0; k < ng; ++k) {
                Box gt = target_box(targets + (long)gt_index[(long)b * d.nmax + k] * 6);
                float v = pair_iou(gt, q, d.use_static_assigner ? 1e-10f : 1e-9f);
                if (v > best) {
                    best = v;
                    g = k;
                }
            }
        }
        w.agt[i] = g;
        float m = 0.f, iou = 0.f;
        if (g >= 0) {
            const float* t = targets + (long)gt_index[(long)b * d.nmax + g] * 6;
            Box gt = target_box(t);
            iou = pair_iou(gt, p, 1e-9f);
            if (!d.use_static_assigner) {
                float sc = sigmoidf(logits[i * d.C + (int)t[1]]);
                m = powf(sc, alpha) * powf(iou, beta);
                atomicMax((int*)&w.maxm[(long)b * d.nmax + g], __float_as_int(m));
                atomicMax((int*)&w.maxi[(long)b * d.nmax + g], __float_as_int(iou));
            }
        }
        w.am[i] = m;
        w.aiou[i] = iou;
    }
}

// ---------------------------------------------------------------------------------------------
// 4a. per-anchor targets + box losses (GIoU, DFL) and d/d distri.  ppyolo_loss.py:540-559, 1008-1052
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LOSS_THREADS) void box_loss_kernel(sgx_loss_desc d, const float* distri, const float* points, const float* strides,
                                                                const float* targets, const int* gt_index, LossWs w, int* assigned_label,
                                                                float* assigned_box, float* assigned_score, float* g_distri, float* partials) {
    __shared__ float red[3][LOSS_THREADS];
    const int R1 = d.reg_max + 1;
    const long n = (long)d.B * d.L;
    const long i = (long)blockIdx.x * LOSS_THREADS + threadIdx.x;
    float l_iou = 0.f, l_dfl = 0.f, s_sum = 0.f;
    if (i < n) {
        const int b = (int)(i / d.L), l = (int)(i % d.L);
        const int g = w.agt[i];
        float* gd = g_distri + i * 4 * R1;
        if (g < 0) {
            assigned_label[i] = d.C;
            // reference gathers gt slot 0 for background anchors (assigned_gt_index = argmax of an all-zero column)
            float bx[4] = {0.f, 0.f, 0.f, 0.f};
            if (d.nmax > 0 && gt_index[(long)b * d.nmax] >= 0) {
                Box g0 = target_box(targets + (long)gt_index[(long)b * d.nmax] * 6);
                bx[0] = g0.x1; bx[1] = g0.y1; bx[2] = g0.x2; bx[3] = g0.y2;
            }
            for (int k = 0; k < 4; ++k) assigned_box[i * 4 + k] = bx[k];
            assigned_score[i] = 0.f;
            for (int k = 0; k < 4 * R1; ++k) gd[k] = 0.f;
        } else {
            const float* t = targets + (long)gt_index[(long)b * d.nmax + g] * 6;
            const Box gt = target_box(t);
            float score;
            if (d.use_static_assigner) score = w.aiou[i];
            else {
                float mm = __int_as_float((int)w.maxm[(long)b * d.nmax + g]);
                float mi = __int_as_float((int)w.maxi[(long)b * d.nmax + g]);
                score = w.am[i] / (mm + TAL_EPS) * mi;
            }
            assigned_label[i] = (int)t[1];
            assigned_box[i * 4 + 0] = gt.x1; assigned_box[i * 4 + 1] = gt.y1;
            assigned_box[i * 4 + 2] = gt.x2; assigned_box[i * 4 + 3] = gt.y2;
            assigned_score[i] = score;
            s_sum = score;
            // ---- GIoU in grid units (ppyolo_loss.py:579-638) ----
            const float s = strides[l];
            const float* pb = w.pbox + i * 4;
            const float x1 = pb[0], y1 = pb[1], x2 = pb[2], y2 = pb[3];
            const float x1g = gt.x1 / s, y1g = gt.y1 / s, x2g = gt.x2 / s, y2g = gt.y2 / s;
            const float eps = 1e-10f;
            float xk1 = fmaxf(x1, x1g), yk1 = fmaxf(y1, y1g), xk2 = fminf(x2, x2g), yk2 = fminf(y2, y2g);
            float wi = xk2 - xk1, hi = yk2 - yk1;
            float wic = clip0(wi), hic = clip0(hi);
            float ov = wic * hic;
            float a1 = (x2 - x1) * (y2 - y1), a2 = (x2g - x1g) * (y2g - y1g);
            float un = a1 + a2 - ov + eps;
            float iou = ov / un;
            float xc1 = fminf(x1, x1g), yc1 = fminf(y1, y1g), xc2 = fmaxf(x2, x2g), yc2 = fmaxf(y2, y2g);
            float cw = xc2 - xc1, ch = yc2 - yc1;
            float hull = cw * ch + eps;
            float miou = iou - ((hull - un) / hull);
            l_iou = (1.f - miou) * score;
            // adjoints of (1 - miou)
            float g_un = -1.f / hull + ov / (un * un);
            float g_hull = un / (hull * hull);
            float g_ov = -1.f / un - g_un;
            float g_wi = wi > 0.f ? g_ov * hic : 0.f, g_hi = hi > 0.f ? g_ov * wic : 0.f;
            float gx1 = 0.f, gy1 = 0.f, gx2 = 0.f, gy2 = 0.f;
            // xk1 = max(x1,x1g), xk2 = min(x2,x2g)  (ties split like torch.maximum's backward)
            gx1 += -g_wi * (x1 > x1g ? 1.f : (x1 == x1g ? 0.5f : 0.f));
            gx2 += g_wi * (x2 < x2g ? 1.f : (x2 == x2g ? 0.5f : 0.f));
            gy1 += -g_hi * (y1 > y1g ? 1.f : (y1 == y1g ? 0.5f : 0.f));
            gy2 += g_hi * (y2 < y2g ? 1.f : (y2 == y2g ? 0.5f : 0.f));
            // a1
            gx2 += g_un * (y2 - y1); gx1 -= g_un * (y2 - y1);
            gy2 += g_un * (x2 - x1); gy1 -= g_un * (x2 - x1);
            // hull
            float g_cw = g_hull * ch, g_ch = g_hull * cw;
            gx1 += -g_cw * (x1 < x1g ? 1.f : (x1 == x1g ? 0.5f : 0.f));
            gx2 += g_cw * (x2 > x2g ? 1.f : (x2 == x2g ? 0.5f : 0.f));
            gy1 += -g_ch * (y1 < y1g ? 1.f : (y1 == y1g ? 0.5f : 0.f));
            gy2 += g_ch * (y2 > y2g ? 1.f : (y2 == y2g ? 0.5f : 0.f));
            const float wiou = d.w_iou * score;
            // box = (px - l, py - t, px + r, py + b)
            float g_ltrb[4] = {-gx1 * wiou, -gy1 * wiou, gx2 * wiou, gy2 * wiou};
            // ---- DFL (ppyolo_loss.py:994-1006, 1063-1067) ----
            const float px = points[2 * l] / s, py = points[2 * l + 1] / s;
            const float hi_clip = (float)d.reg_max - 0.01f;
            float tgt[4] = {px - x1g, py - y1g, x2g - px, y2g - py};
            const float* dd = distri + i * 4 * R1;
            const float wdfl = d.w_dfl * score * 0.25f;
            float dfl = 0.f;
            for (int k = 0; k < 4; ++k) {
                float tk = fminf(fmaxf(tgt[k], 0.f), hi_clip);
                int tl = (int)tk;
                float wl = (float)(tl + 1) - tk, wr = 1.f - wl;
                const float* dk = dd + k * R1;
                float m = dk[0];
                for (int j = 1; j < R1; ++j) m = fmaxf(m, dk[j]);
                float se = 0.f, ex = 0.f;
                for (int j = 0; j < R1; ++j) {
                    float e = expf(dk[j] - m);
                    se += e;
                    ex += e * (float)j;
                }
                float lse = logf(se) + m;
                float expect = ex / se;
                dfl += (lse - dk[tl]) * wl + (lse - dk[tl + 1]) * wr;
                for (int j = 0; j < R1; ++j) {
                    float pj = expf(dk[j] - m) / se;
                    float gj = wdfl * (pj - (j == tl ? wl : 0.f) - (j == tl + 1 ? wr : 0.f));
                    gj += g_ltrb[k] * pj * ((float)j - expect);
                    gd[k * R1 + j] = gj;
                }
            }
            l_dfl = dfl * 0.25f * score;
        }
    }
    red[0][threadIdx.x] = l_iou;
    red[1][threadIdx.x] = l_dfl;
    red[2][threadIdx.x] = s_sum;
    __syncthreads();
    for (int wdt = LOSS_THREADS / 2; wdt > 0; wdt >>= 1) {
        if ((int)threadIdx.x < wdt) {
            red[0][threadIdx.x] += red[0][threadIdx.x + wdt];
            red[1][threadIdx.x] += red[1][threadIdx.x + wdt];
            red[2][threadIdx.x] += red[2][threadIdx.x + wdt];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = partials + (long)blockIdx.x * 4;
        o[0] = 0.f; o[1] = red[0][0]; o[2] = red[1][0]; o[3] = red[2][0];
    }
}

// ---------------------------------------------------------------------------------------------
// 4b. classification loss over [B,L,C] and d/d logits.  ppyolo_loss.py:1069-1084
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cls_elem(float x, float t, bool is_label, int vfl, int static_alpha, float& loss, float& grad) {
    float p = sigmoidf(x);
    // ATen binary_cross_entropy_with_logits: (1-t)*x + max(-x,0) + log(exp(-max(-x,0)) + exp(-x-max(-x,0)))
    float mv = fmaxf(-x, 0.f);
    float bce = (1.f - t) * x + mv + logf(expf(-mv) + expf(-x - mv));
    float wgt, dw;
    if (vfl) {
        float lab = is_label ? 1.f : 0.f;
        wgt = 0.75f * (p * p) * (1.f - lab) + t * lab;
        dw = 0.75f * 2.f * p * (p * (1.f - p)) * (1.f - lab);
    } else {
        float df = p - t;
        wgt = df * df;
        dw = 2.f * df * (p * (1.f - p));
        if (static_alpha) {
            float at = 0.25f * t + 0.75f * (1.f - t);
            wgt *= at;
            dw *= at;
        }
    }
    loss = wgt * bce;
    grad = dw * bce + wgt * (p - t);
}
__global__ __launch_bounds__(LOSS_THREADS) void cls_loss_kernel(sgx_loss_desc d, const float* logits, const int* assigned_label,
                                                                const float* assigned_score, float* g_logits, float* partials) {
    __shared__ float red[LOSS_THREADS];
    const int C4 = d.C / 4;
    const long n = (long)d.B * d.L * C4;
    const long i = (long)blockIdx.x * LOSS_THREADS + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        const long a = i / C4;
        const int c = (int)(i % C4) * 4;
        const int lab = assigned_label[a];
        const float sc = assigned_score[a];
        float4 x = sgx_ld4(logits + a * d.C + c);
        float xs[4] = {x.x, x.y, x.z, x.w}, gs[4];
        for (int k = 0; k < 4; ++k) {
            bool is_lab = (c + k) == lab;
            float ls, gr;
            cls_elem(xs[k], is_lab ? sc : 0.f, is_lab, d.use_varifocal, d.use_static_assigner, ls, gr);
            acc += ls;
            gs[k] = gr * d.w_cls;
        }
        sgx_st4(g_logits + a * d.C + c, make_float4(gs[0], gs[1], gs[2], gs[3]));
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int wdt = LOSS_THREADS / 2; wdt > 0; wdt >>= 1) {
        if ((int)threadIdx.x < wdt) red[threadIdx.x] += red[threadIdx.x + wdt];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = partials + (long)blockIdx.x * 4;
        o[0] = red[0]; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
    }
}

// class counts that are not a multiple of 4 (fine-tuning on custom datasets): one thread per (anchor, class), 4-byte accesses
__global__ __launch_bounds__(LOSS_THREADS) void cls_loss_scalar_kernel(sgx_loss_desc d, const float* logits, const int* assigned_label,
                                                                       const float* assigned_score, float* g_logits, float* partials) {
    __shared__ float red[LOSS_THREADS];
    const long n = (long)d.B * d.L * d.C;
    const long i = (long)blockIdx.x * LOSS_THREADS + threadIdx.x;
    float acc = 0.f;
    if (i < n) {
        const long a = i / d.C;
        const int c = (int)(i - a * d.C);
        const bool is_lab = c == assigned_label[a];
        float ls, gr;
        cls_elem(logits[i], is_lab ? assigned_score[a] : 0.f, is_lab, d.use_varifocal, d.use_static_assigner, ls, gr);
        acc = ls;
        g_logits[i] = gr * d.w_cls;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int wdt = LOSS_THREADS / 2; wdt > 0; wdt >>= 1) {
        if ((int)threadIdx.x < wdt) red[threadIdx.x] += red[threadIdx.x + wdt];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float* o = partials + (long)blockIdx.x * 4;
        o[0] = red[0]; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
    }
}

__global__ __launch_bounds__(256) void loss_sums_kernel(const float* partials, int nblk, float* sums) {
    __shared__ double red[4][256];
    double s[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int k = 0; k < 4; ++k) s[k] += (double)partials[(long)i * 4 + k];
    for (int k = 0; k < 4; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 4; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = (float)red[threadIdx.x][0];
}

// decode from PIXEL anchor points: grid point = point / stride (ppyolo_loss.py:803-804)
__global__ void decode_px_kernel(int B, int L, int R1, const float* distri, const float* points, const float* strides, float* boxes) {
    long n = (long)B * L * 4;  // one thread per (anchor, side), see decode_kernel
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i & 3);
        const int l = (int)((i >> 2) % L);
        float e;
        softmax_expect(distri + i * R1, R1, e);
        const float pc = points[2 * l + (k & 1)] / strides[l];
        boxes[i] = k < 2 ? pc - e : pc + e;
    }
}

extern "C" int32_t sgx_ppyoloe_loss_fwd(const sgx_loss_desc* d, const float* logits, const float* distri, const float* anchors,
                                        const float* points, const float* strides, const float* targets, const int32_t* gt_count,
                                        const int32_t* gt_index, float* sums, int32_t* assigned_label, float* assigned_box,
                                        float* assigned_score, float* g_logits, float* g_distri, void* ws, int64_t ws_bytes, void* stream) {
    SGX_CHECK_ARG(d && logits && distri && anchors && points && strides && sums && assigned_label && assigned_box && assigned_score &&
                      g_logits && g_distri && gt_count,
                  "ppyoloe_loss: null pointer");
    SGX_CHECK_ARG(d->B > 0 && d->L > 0 && d->C > 0 && d->reg_max > 0 && d->reg_max < 64, "ppyoloe_loss: bad dims");
    SGX_CHECK_ARG(d->nmax == 0 || (targets && gt_index), "ppyoloe_loss: null targets");
    SGX_CHECK_ARG((long)d->L * 4 <= 160 * 1024 - 4096, "ppyoloe_loss: L=%d does not fit the LDS metric buffer", d->L);
    if (!ws || ws_bytes < sgx_ppyoloe_loss_workspace(d)) SGX_FAIL(SGX_ERR_WORKSPACE, "ppyoloe_loss: workspace too small");
    if (d->use_static_assigner) {
        long tot = 0;
        SGX_CHECK_ARG(d->num_levels > 0 && d->num_levels <= 8, "ppyoloe_loss: bad num_levels");
        for (int i = 0; i < d->num_levels; ++i) tot += d->level_count[i];
        SGX_CHECK_ARG(tot == d->L, "ppyoloe_loss: level counts do not sum to L");
        SGX_CHECK_ARG(d->num_levels * 9 <= ATSS_MAXCAND, "ppyoloe_loss: too many ATSS candidates");
    }
    LossWs w = carve(d, ws);
    const long BL = (long)d->B * d->L;
    hipStream_t st = (hipStream_t)stream;
    SGX_MEMSET_ASYNC(w.cnt, 0, BL * 4, st);
    SGX_MEMSET_ASYNC(w.maxm, 0, (long)d->B * (d->nmax > 0 ? d->nmax : 1) * 4, st);
    SGX_MEMSET_ASYNC(w.maxi, 0, (long)d->B * (d->nmax > 0 ? d->nmax : 1) * 4, st);
    long blocks = (BL + 255) / 256;
    // 1. decode pred boxes (grid units)
    if (d->reg_max + 1 <= DEC_MAXR1)
        SGX_LAUNCH(decode_rows_kernel<true>, dim3((unsigned)(4 * blocks > 16384 ? 16384 : 4 * blocks)), dim3(256), 0, stream, d->B, d->L, d->reg_max + 1, distri,
                   points, strides, 0.f, w.pbox);
    else
        SGX_LAUNCH(decode_px_kernel, dim3((unsigned)(4 * blocks > 16384 ? 16384 : 4 * blocks)), dim3(256), 0, stream, d->B, d->L, d->reg_max + 1, distri, points,
                   strides, w.pbox);
    SGX_CHECK_LAUNCH("decode_px");
    if (d->nmax > 0) {
        size_t smem = (size_t)d->L * sizeof(float);
        if (d->use_static_assigner)
            SGX_LAUNCH(atss_candidates_kernel, dim3(d->nmax, d->B), dim3(LOSS_THREADS), smem, stream, *d, anchors, targets, gt_count, gt_index, w, 9);
        else
            SGX_LAUNCH(tal_candidates_kernel, dim3(d->nmax, d->B), dim3(LOSS_THREADS), smem, stream, *d, logits, points, strides, targets, gt_count,
                       gt_index, w, 13, 1.0f, 6.0f);
        SGX_CHECK_LAUNCH("candidates");
    }
    SGX_LAUNCH(resolve_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, stream, *d, logits, anchors, strides, targets, gt_count,
               gt_index, w, 1.0f, 6.0f);
    SGX_CHECK_LAUNCH("resolve");
    const int nb_box = (int)((BL + LOSS_THREADS - 1) / LOSS_THREADS);
    SGX_LAUNCH(box_loss_kernel, dim3(nb_box), dim3(LOSS_THREADS), 0, stream, *d, distri, points, strides, targets, gt_index, w, assigned_label,
               assigned_box, assigned_score, g_distri, w.partials);
    SGX_CHECK_LAUNCH("box_loss");
    const bool vec_cls = d->C % 4 == 0;
    const int nb_cls = (int)((BL * (vec_cls ? d->C / 4 : d->C) + LOSS_THREADS - 1) / LOSS_THREADS);
    if (vec_cls)
        SGX_LAUNCH(cls_loss_kernel, dim3(nb_cls), dim3(LOSS_THREADS), 0, stream, *d, logits, (const int*)assigned_label,
                   (const float*)assigned_score, g_logits, w.partials + (long)nb_box * 4);
    else
        SGX_LAUNCH(cls_loss_scalar_kernel, dim3(nb_cls), dim3(LOSS_THREADS), 0, stream, *d, logits, (const int*)assigned_label,
                   (const float*)assigned_score, g_logits, w.partials + (long)nb_box * 4);
    SGX_CHECK_LAUNCH("cls_loss");
    SGX_LAUNCH(loss_sums_kernel, dim3(1), dim3(256), 0, stream, (const float*)w.partials, nb_box + nb_cls, sums);
    SGX_CHECK_LAUNCH("loss_sums");
    return SGX_OK;
}

__global__ void loss_finalize_kernel(const float* sums, float w_cls, float w_iou, float w_dfl, float score_div, float* items, float* inv_norm) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = sums[3] / score_div;
    s = s < 1.f ? 1.f : s;  // torch.clip(min=1)
    float lc = w_cls * sums[0] / s, li = w_iou * sums[1] / s, ld = w_dfl * sums[2] / s;
    items[0] = lc; items[1] = li; items[2] = ld; items[3] = lc + li + ld;
    inv_norm[0] = 1.f / s;
}
extern "C" int32_t sgx_ppyoloe_loss_finalize(const float* sums, float w_cls, float w_iou, float w_dfl, float score_div, float* items,
                                             float* inv_norm, void* stream) {
    SGX_CHECK_ARG(sums && items && inv_norm && score_div > 0.f, "ppyoloe_loss_finalize: bad args");
    SGX_LAUNCH(loss_finalize_kernel, dim3(1), dim3(64), 0, stream, sums, w_cls, w_iou, w_dfl, score_div, items, inv_norm);
    SGX_CHECK_LAUNCH("loss_finalize");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------
// softmax cross-entropy: fwd + bwd in one pass, one wave per row.  training/losses/label_smoothing_cross_entropy_loss.py:32-83:
//   smoothing == 0 : F.cross_entropy(logits, labels, weight, ignore_index, reduction)
//                    loss = sum_i w[y_i] * (lse_i - x_i[y_i]) / sum_i w[y_i]   over the rows whose label is not ignore_index ("mean")
//   smoothing  > 0 : lw = weight * log_softmax;  loss_i = -((1 - eps) * lw_i[y_i] + eps * mean_j lw_i[j]);  rows with label == ignore_index
//                    (>= 0) contribute 0;  loss = sum_i loss_i / (B - masked rows)   ("mean": NOT weight-normalised, as in the reference)
//   reduction "sum": the same sums without the division.
// dlogits is written WITHOUT the 1 / denominator factor; it comes out as loss[1] (the backward multiplies it in with the upstream gradient).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void softmax_ce_kernel(int B, int K, const float* logits, const int64_t* labels, float smoothing, const float* weight,
                                                        int ignore_index, float* row_loss, float* row_den, float* dlogits) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* x = logits + (long)b * K;
    const int y = (int)labels[b];
    const bool ignored = y == ignore_index && (smoothing == 0.f || ignore_index >= 0);
    if (ignored || y < 0 || y >= K) {  // (an out-of-range label that is not the ignore index is a caller error: treated as ignored)
        if (lane == 0) {
            row_loss[b] = 0.f;
            row_den[b] = 0.f;
        }
        for (int j = lane; j < K; j += 64) dlogits[(long)b * K + j] = 0.f;
        return;
    }
    float m = -INFINITY;
    for (int j = lane; j < K; j += 64) m = fmaxf(m, x[j]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    float s = 0.f, swx = 0.f, sw = 0.f;  // sum exp, sum_j w_j * x_j, sum_j w_j
    for (int j = lane; j < K; j += 64) {
        const float wj = weight ? weight[j] : 1.f;
        s += expf(x[j] - m);
        swx += wj * x[j];
        sw += wj;
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_xor(s, off);
        swx += __shfl_xor(swx, off);
        sw += __shfl_xor(sw, off);
    }
    const float lse = logf(s) + m;
    const float wy = weight ? weight[y] : 1.f;
    // -lw[y] = wy * (lse - x_y);   -mean_j lw[j] = (sw * lse - swx) / K
    if (lane == 0) {
        row_loss[b] = (1.f - smoothing) * wy * (lse - x[y]) + smoothing / (float)K * (sw * lse - swx);
        row_den[b] = smoothing == 0.f ? wy : 1.f;
    }
    const float pk = (1.f - smoothing) * wy + smoothing / (float)K * sw;  // coefficient of the softmax probability
    for (int j = lane; j < K; j += 64) {
        const float p = expf(x[j] - m) / s, wj = weight ? weight[j] : 1.f;
        dlogits[(long)b * K + j] = p * pk - ((j == y ? (1.f - smoothing) * wy : 0.f) + smoothing / (float)K * wj);
    }
}
// loss[0] = sum of the row losses / denominator, loss[1] = 1 / denominator (fp64 folds, fixed order); denominator = sum of row_den, or 1 for "sum"
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* row_loss, const float* row_den, int n, int reduction_sum, float* out) {
    __shared__ double red[2][256];
    double s = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        s += (double)row_loss[i];
        d += (double)row_den[i];
    }
    red[0][threadIdx.x] = s;
    red[1][threadIdx.x] = d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            red[0][threadIdx.x] += red[0][threadIdx.x + w];
            red[1][threadIdx.x] += red[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double den = reduction_sum ? 1.0 : red[1][0];
        out[0] = (float)(red[0][0] / den);  // every row ignored: 0 / 0 = nan, like torch
        out[1] = (float)(1.0 / den);
    }
}
// loss: 2 * B + 2 floats of scratch; loss[0] = the loss, loss[1] = 1 / denominator (multiply dlogits by it and by the upstream gradient)
extern "C" int32_t sgx_softmax_ce_fwd_bwd(int32_t B, int32_t K, const float* logits, const int64_t* labels, float smoothing, const float* weight,
                                          int32_t ignore_index, int32_t reduction_sum, float* loss, float* dlogits, void* stream) {
    SGX_CHECK_ARG(logits && labels && loss && dlogits && B > 0 && K > 0, "softmax_ce: bad args");
    SGX_CHECK_ARG(smoothing >= 0.f && smoothing <= 1.f, "softmax_ce: smoothing must be in [0, 1]");
    SGX_LAUNCH(softmax_ce_kernel, dim3(B), dim3(64), 0, stream, B, K, logits, labels, smoothing, weight, ignore_index, loss + 2, loss + 2 + B, dlogits);
    SGX_CHECK_LAUNCH("softmax_ce");
    SGX_LAUNCH(ce_reduce_kernel, dim3(1), dim3(256), 0, stream, (const float*)(loss + 2), (const float*)(loss + 2 + B), B, reduction_sum, loss);
    SGX_CHECK_LAUNCH("softmax_ce reduce");
    return SGX_OK;
}
