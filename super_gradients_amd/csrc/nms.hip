// Detection post-processing on gfx950: score filter -> exact top-k -> greedy NMS, one 1024-thread workgroup
// per image, everything after the score stream stays in LDS.  No MFMA: the score stream is HBM/L2-bound
// (B*L*C*4 bytes per pass), the suppression loop is latency-bound (one barrier per KEPT box).
//
// Replaces PPYoloEPostPredictionCallback.forward (pp_yolo_e/post_prediction_callback.py:42-123) and the
// torchvision.ops.nms / batched_nms it calls (:85,87).  Ordering rule (matches a stable descending sort):
// candidates are ranked by (score desc, candidate index asc), candidate index = anchor*C + class
// (the row-major order of `(scores > thr).nonzero()`), or the anchor index in single-label mode.
// The exact k-th composite key is found by a 5-digit radix select over (score bits, ~index) - no sort of
// the 672k scores - then <=1024 survivors are bitonic-sorted in LDS.
// Compile with -ffp-contract=off: the IoU test must round exactly like the CPU restatement.
#include "sgx_common.h"

#define NMS_THREADS 1024
#define NMS_MAXK_LIMIT 4096  // largest instantiated top-k capacity (LDS: 37 B per candidate + 8 KB histogram <= 160 KB)
#define NMS_IDXBITS 22
#define NMS_HBINS 2048

typedef unsigned long long u64;

extern "C" int64_t sgx_nms_workspace(const sgx_nms_desc* d) {
    (void)d;
    return 256;  // everything lives in LDS; kept for ABI stability
}

__device__ __forceinline__ bool nms_candidate(const sgx_nms_desc& d, const float* sc, long e, float& score, int& cls) {
    if (d.multi_label) {
        score = sc[e];
        cls = (int)(e % d.C);
        return score > d.score_threshold;
    }
    const float* row = sc + e * d.C;
    float m = row[0];
    int mi = 0;
    for (int c = 1; c < d.C; ++c)
        if (row[c] > m) {
            m = row[c];
            mi = c;
        }
    score = m;
    cls = mi;
    return m >= d.score_threshold;
}
__device__ __forceinline__ u64 nms_key(float score, long e) {
    return ((u64)__float_as_uint(score) << NMS_IDXBITS) | (u64)(((1u << NMS_IDXBITS) - 1u) - (unsigned)e);
}

template <int NMS_MAXK>
__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(sgx_nms_desc d, const float* boxes, const float* scores, float* out, int* out_count,
                                                          int* out_index, int* num_candidates) {
    __shared__ int hist[NMS_HBINS];
    __shared__ u64 keys[NMS_MAXK];
    __shared__ float bx[NMS_MAXK][4];
    __shared__ float area[NMS_MAXK];
    __shared__ int cls_s[NMS_MAXK];
    __shared__ unsigned char sup[NMS_MAXK];
    __shared__ int s_count, s_n, s_kept, s_remaining;
    __shared__ u64 s_prefix;
    __shared__ float s_maxc;
    __shared__ int keep_list[NMS_MAXK];
    __shared__ float wmax[NMS_THREADS / 64];
    __shared__ float cur[5];

    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sc = scores + (long)b * d.L * d.C;
    const float* bxs = boxes + (long)b * d.L * 4;
    const long E = d.multi_label ? (long)d.L * d.C : (long)d.L;
    const int K = d.nms_top_k < NMS_MAXK ? d.nms_top_k : NMS_MAXK;

    // ---- pass 0: count candidates ----
    if (tid == 0) s_count = 0;
    __syncthreads();
    {
        int local = 0;
        for (long e = tid; e < E; e += NMS_THREADS) {
            float s;
            int c;
            if (nms_candidate(d, sc, e, s, c)) ++local;
        }
        if (local) atomicAdd(&s_count, local);
    }
    __syncthreads();
    const int count = s_count;
    const int n = count < K ? count : K;
    // ---- radix select of the n-th largest composite key (only when count > K) ----
    u64 thr_key = 0;  // select keys >= thr_key
    if (count > K) {
        const int total_bits = 31 + NMS_IDXBITS;  // 53
        const int widths[5] = {11, 11, 11, 11, total_bits - 44};
        if (tid == 0) {
            s_prefix = 0;
            s_remaining = K;
        }
        int shift = total_bits;
        for (int pass = 0; pass < 5; ++pass) {
            const int wbits = widths[pass];
            shift -= wbits;
            for (int i = tid; i < NMS_HBINS; i += NMS_THREADS) hist[i] = 0;
            __syncthreads();
            const u64 prefix = s_prefix;
            for (long e = tid; e < E; e += NMS_THREADS) {
                float s;
                int c;
                if (nms_candidate(d, sc, e, s, c)) {
                    u64 k = nms_key(s, e);
                    if ((k >> (shift + wbits)) == prefix) atomicAdd(&hist[(int)((k >> shift) & ((1u << wbits) - 1u))], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining, cum = 0, dsel = 0;
                for (int bin = (1 << wbits) - 1; bin >= 0; --bin) {
                    if (cum + hist[bin] >= rem) {
                        dsel = bin;
                        break;
                    }
                    cum += hist[bin];
                }
                s_remaining = rem - cum;
                s_prefix = (prefix << wbits) | (u64)dsel;
            }
            __syncthreads();
        }
        thr_key = s_prefix;
    }
    // ---- gather survivors (unordered), then bitonic sort descending ----
    if (tid == 0) s_n = 0;
    for (int i = tid; i < NMS_MAXK; i += NMS_THREADS) keys[i] = 0;  // 0 sorts last (real keys have score bits > 0)
    __syncthreads();
    for (long e = tid; e < E; e += NMS_THREADS) {
        float s;
        int c;
        if (nms_candidate(d, sc, e, s, c)) {
            u64 k = nms_key(s, e);
            if (k >= thr_key) {
                int slot = atomicAdd(&s_n, 1);
                if (slot < NMS_MAXK) keys[slot] = k;
            }
        }
    }
    __syncthreads();
    for (int size = 2; size <= NMS_MAXK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < NMS_MAXK; i += NMS_THREADS) {
                int j = i ^ stride;
                if (j > i) {
                    u64 a = keys[i], c = keys[j];
                    bool desc = (i & size) == 0;
                    if (desc ? (a < c) : (a > c)) {
                        keys[i] = c;
                        keys[j] = a;
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- load boxes / classes of the n survivors (candidate t is owned by thread t % NMS_THREADS) ----
    if (tid == 0) s_maxc = -INFINITY;
    __syncthreads();
    float mymax = -INFINITY;
    for (int t = tid; t < NMS_MAXK; t += NMS_THREADS) {
        if (t < n) {
            u64 k = keys[t];
            long e = (long)(((1u << NMS_IDXBITS) - 1u) - (unsigned)(k & ((1u << NMS_IDXBITS) - 1u)));
            long anchor = d.multi_label ? e / d.C : e;
            int c;
            if (d.multi_label) c = (int)(e % d.C);
            else {
                float s;
                nms_candidate(d, sc, e, s, c);
            }
            cls_s[t] = c;
            for (int q = 0; q < 4; ++q) {
                float v = bxs[anchor * 4 + q];
                bx[t][q] = v;
                mymax = fmaxf(mymax, v);
            }
        }
        sup[t] = 0;
    }
    // class_mode 3 = torchvision's own CPU dispatch (ops/boxes.py batched_nms): coordinate trick while boxes.numel() <= 4000
    const int class_mode = d.class_mode == 3 ? (4 * n > 4000 ? 2 : 1) : d.class_mode;
    if (class_mode == 1) {
        // torchvision batched_nms coordinate trick: boxes + cls * (max_coordinate + 1)
        for (int off = 32; off > 0; off >>= 1) mymax = fmaxf(mymax, __shfl_xor(mymax, off));
        if ((tid & 63) == 0) wmax[tid >> 6] = mymax;
        __syncthreads();
        if (tid == 0) {
            float m = wmax[0];
            for (int w = 1; w < NMS_THREADS / 64; ++w) m = fmaxf(m, wmax[w]);
            s_maxc = m;
        }
        __syncthreads();
    }
    __syncthreads();
    // boxes as the suppression test sees them: original coordinates (+ the class offset in mode 1), recomputed from LDS where needed
    const float offscale = class_mode == 1 ? s_maxc + 1.f : 0.f;
    auto nbox = [&](int t, float* o) {
        const float offv = class_mode == 1 ? (float)cls_s[t] * offscale : 0.f;
        for (int q = 0; q < 4; ++q) o[q] = bx[t][q] + offv;
    };
    for (int t = tid; t < n; t += NMS_THREADS) {
        float nb[4];
        nbox(t, nb);
        area[t] = (nb[2] - nb[0]) * (nb[3] - nb[1]);
    }
    if (tid == 0) s_kept = 0;
    __syncthreads();
    // An IoU never exceeds 1, so with a threshold >= 2 nothing can be suppressed and the scan degenerates to "the first max_predictions
    // of the sorted candidates": the pre-NMS top-k of the decoding modules (kernels.decode_topk) takes this exit instead of n barrier pairs.
    const bool no_suppression = d.iou_threshold >= 2.0f;
    if (no_suppression) {
        const int kn = n < d.max_predictions ? n : d.max_predictions;
        for (int t = tid; t < kn; t += NMS_THREADS) keep_list[t] = t;
        if (tid == 0) s_kept = kn;
        __syncthreads();
    }
    // ---- greedy scan: one barrier pair per kept box; box i is broadcast through LDS (offset form) ----
    for (int i = 0; i < (no_suppression ? 0 : n); ++i) {
        if (sup[i]) continue;  // uniform: flags only change before a barrier
        if (tid == 0) {
            float nb[4];
            nbox(i, nb);
            cur[0] = nb[0]; cur[1] = nb[1]; cur[2] = nb[2]; cur[3] = nb[3]; cur[4] = area[i];
            keep_list[s_kept] = i;
            s_kept = s_kept + 1;
        }
        __syncthreads();
        for (int t = tid; t < n; t += NMS_THREADS) {
            if (t > i && !sup[t] && (class_mode != 2 || cls_s[t] == cls_s[i])) {
                float nb[4];
                nbox(t, nb);
                float xx1 = fmaxf(cur[0], nb[0]), yy1 = fmaxf(cur[1], nb[1]);
                float xx2 = fminf(cur[2], nb[2]), yy2 = fminf(cur[3], nb[3]);
                float w = xx2 - xx1; w = w < 0.f ? 0.f : w;
                float h = yy2 - yy1; h = h < 0.f ? 0.f : h;
                float inter = w * h;
                float ovr = inter / (cur[4] + area[t] - inter);
                if (ovr > d.iou_threshold) sup[t] = 1;
            }
        }
        __syncthreads();
        if (s_kept >= d.max_predictions) break;  // uniform
    }
    __syncthreads();
    const int kept = s_kept < d.max_predictions ? s_kept : d.max_predictions;
    if (tid == 0) {
        out_count[b] = kept;
        if (num_candidates) num_candidates[b] = n;
    }
    for (int r = tid; r < d.max_predictions; r += NMS_THREADS) {
        float* o = out + ((long)b * d.max_predictions + r) * 6;
        if (r < kept) {
            int i = keep_list[r];
            u64 k = keys[i];
            unsigned e = ((1u << NMS_IDXBITS) - 1u) - (unsigned)(k & ((1u << NMS_IDXBITS) - 1u));
            o[0] = bx[i][0]; o[1] = bx[i][1]; o[2] = bx[i][2]; o[3] = bx[i][3];
            o[4] = __uint_as_float((unsigned)(k >> NMS_IDXBITS));
            o[5] = (float)cls_s[i];
            if (out_index) out_index[(long)b * d.max_predictions + r] = (int)e;
        } else {
            for (int q = 0; q < 6; ++q) o[q] = 0.f;
            if (out_index) out_index[(long)b * d.max_predictions + r] = -1;
        }
    }
}

extern "C" int32_t sgx_nms(const sgx_nms_desc* d, const float* boxes, const float* scores, float* out, int32_t* out_count, int32_t* out_index,
                           int32_t* num_candidates, void* ws, int64_t ws_bytes, void* stream) {
    (void)ws;
    (void)ws_bytes;
    SGX_CHECK_ARG(d && boxes && scores && out && out_count, "nms: null pointer");
    SGX_CHECK_ARG(d->B > 0 && d->L > 0 && d->C > 0, "nms: bad dims");
    SGX_CHECK_ARG(d->nms_top_k > 0 && d->nms_top_k <= NMS_MAXK_LIMIT, "nms: nms_top_k=%d unsupported (max %d)", d->nms_top_k, NMS_MAXK_LIMIT);
    SGX_CHECK_ARG(d->max_predictions > 0 && d->max_predictions <= NMS_MAXK_LIMIT, "nms: bad max_predictions");
    SGX_CHECK_ARG((long)d->L * (d->multi_label ? d->C : 1) <= (1L << NMS_IDXBITS), "nms: too many candidates for the %d-bit index field", NMS_IDXBITS);
    SGX_CHECK_ARG(d->class_mode >= 0 && d->class_mode <= 3, "nms: bad class_mode");
    SGX_CHECK_ARG(d->score_threshold >= 0.f, "nms: negative score threshold unsupported (keys assume non-negative scores)");
    const int need = d->nms_top_k > d->max_predictions ? d->nms_top_k : d->max_predictions;
    if (need <= 1024) SGX_LAUNCH(nms_kernel<1024>, dim3(d->B), dim3(NMS_THREADS), 0, stream, *d, boxes, scores, out, out_count, out_index, num_candidates);
    else if (need <= 2048) SGX_LAUNCH(nms_kernel<2048>, dim3(d->B), dim3(NMS_THREADS), 0, stream, *d, boxes, scores, out, out_count, out_index, num_candidates);
    else SGX_LAUNCH(nms_kernel<4096>, dim3(d->B), dim3(NMS_THREADS), 0, stream, *d, boxes, scores, out, out_count, out_index, num_candidates);
    SGX_CHECK_LAUNCH("nms");
    return SGX_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Detection matching for the validation metrics (mAP): predictions (NMS rows) vs ground truth, per image and IoU threshold.
// Replaces compute_detection_matching / compute_img_detection_matching / IoUMatching.compute_targets / compute_crowd_targets /
// get_top_k_idx_per_cls (training/utils/detection_utils.py:1120-1290, 880-1005, 1342-1358), which the reference runs as a Python
// loop over images with a Python loop over candidate (prediction, target) pairs inside.
// One workgroup per image.  Per IoU threshold j the reference's rule reduces to an independent greedy pass: predictions in
// (score desc, index asc) order, restricted to the top_k of their class; each takes the free same-class target of highest IoU
// (first index among equals: stable descending sort) and is a true positive iff that IoU > thr[j] (and > thr[0], the reference's
// candidate filter).  Crowd targets: a used prediction is ignored at threshold j iff its best same-class IoA (intersection over
// detection area) > thr[j]; predictions outside the per-class top_k are ignored at every threshold.
// Arithmetic order follows box_iou / crowd_ioa (:257-276, :797-810) and cxcywh2xyxy (:725-736); compiled with -ffp-contract=off.
// ---------------------------------------------------------------------------------------------------------------------
#define MATCH_THREADS 256
__device__ __forceinline__ float match_iou(const float* p, const float* t) {
    const float area1 = (p[2] - p[0]) * (p[3] - p[1]);
    const float area2 = (t[2] - t[0]) * (t[3] - t[1]);
    float w = fminf(p[2], t[2]) - fmaxf(p[0], t[0]);
    float h = fminf(p[3], t[3]) - fmaxf(p[1], t[1]);
    w = w < 0.f ? 0.f : w;
    h = h < 0.f ? 0.f : h;
    const float inter = w * h;
    return inter / (area1 + area2 - inter);
}
__device__ __forceinline__ void match_load_target(const float* t, int denorm, float W, float H, float* box, float* cls) {
    // row = (img, cls, cx, cy, w, h);  y1 = cy - h*0.5; x1 = cx - w*0.5; y2 = h + y1; x2 = w + x1  (then optional de-normalisation)
    *cls = t[1];
    const float y1 = t[3] - t[5] * 0.5f, x1 = t[2] - t[4] * 0.5f;
    float b[4] = {x1, y1, t[4] + x1, t[5] + y1};
    if (denorm) {
        b[0] *= W; b[2] *= W; b[1] *= H; b[3] *= H;
    }
    box[0] = b[0]; box[1] = b[1]; box[2] = b[2]; box[3] = b[3];
}
__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(sgx_match_desc d, const float* preds, const int* pred_count, const float* targets,
                                                              const int* gt_count, const int* gt_index, const float* crowd, const int* crowd_count,
                                                              const int* crowd_index, const float* thr, unsigned char* matched,
                                                              unsigned char* ignore) {
    SGX_DYN_SMEM(float, sm);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int P = d.P, nthr = d.nthr;
    int n = pred_count[b];
    n = n < P ? n : P;
    const int nt = d.nmax > 0 ? min(gt_count[b], d.nmax) : 0;
    const int nc = d.cmax > 0 ? min(crowd_count[b], d.cmax) : 0;
    float* pbox = sm;                        // [P][4]
    float* pscore = pbox + (size_t)P * 4;    // [P]
    float* pcls = pscore + P;                // [P]
    float* tbox = pcls + P;                  // [nmax][4]
    float* tcls = tbox + (size_t)d.nmax * 4; // [nmax]
    float* cbox = tcls + d.nmax;             // [cmax][4]
    float* ccls = cbox + (size_t)d.cmax * 4; // [cmax]
    int* order = (int*)(ccls + d.cmax);      // [P]
    unsigned char* used = (unsigned char*)(order + P);      // [P]
    unsigned char* tmat = used + P;                          // [nthr][nmax]
    const float Wf = (float)d.W, Hf = (float)d.H;
    const bool clip = nt > 0 || nc > 0;  // the reference clips the predictions only when there is something to match (:1262-1264)
    for (int i = tid; i < n; i += MATCH_THREADS) {
        const float* r = preds + ((size_t)b * P + i) * 6;
        float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
        if (clip) {
            x1 = fminf(fmaxf(x1, 0.f), Wf); x2 = fminf(fmaxf(x2, 0.f), Wf);
            y1 = fminf(fmaxf(y1, 0.f), Hf); y2 = fminf(fmaxf(y2, 0.f), Hf);
        }
        pbox[i * 4 + 0] = x1; pbox[i * 4 + 1] = y1; pbox[i * 4 + 2] = x2; pbox[i * 4 + 3] = y2;
        pscore[i] = r[4];
        pcls[i] = r[5];
    }
    for (int t = tid; t < nt; t += MATCH_THREADS) match_load_target(targets + (size_t)gt_index[(size_t)b * d.nmax + t] * 6, d.denormalize, Wf, Hf, tbox + t * 4, tcls + t);
    for (int t = tid; t < nc; t += MATCH_THREADS) match_load_target(crowd + (size_t)crowd_index[(size_t)b * d.cmax + t] * 6, d.denormalize, Wf, Hf, cbox + t * 4, ccls + t);
    for (int i = tid; i < nthr * d.nmax; i += MATCH_THREADS) tmat[i] = 0;
    __syncthreads();
    // global order (score desc, index asc) and membership in the per-class top_k
    for (int i = tid; i < n; i += MATCH_THREADS) {
        const float s = pscore[i], c = pcls[i];
        int grank = 0, crank = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = pscore[j];
            const bool before = sj > s || (sj == s && j < i);
            grank += before ? 1 : 0;
            crank += (before && pcls[j] == c) ? 1 : 0;
        }
        order[grank] = i;
        used[i] = (crank < d.top_k && s != 0.f) ? 1 : 0;  // zero scores drop out of the reference's nonzero() selection
    }
    __syncthreads();
    // one greedy pass per threshold
    if (tid < nthr) {
        const int j = tid;
        const float tj = thr[j], t0 = thr[0];
        unsigned char* free_t = tmat + (size_t)j * d.nmax;
        for (int r = 0; r < n; ++r) {
            const int i = order[r];
            unsigned char m = 0;
            if (used[i] && nt > 0) {
                float best = -1.f;
                int bi = -1;
                const float c = pcls[i];
                for (int t = 0; t < nt; ++t) {
                    if (tcls[t] != c || free_t[t]) continue;
                    const float v = match_iou(pbox + i * 4, tbox + t * 4);
                    if (v > best) {
                        best = v;
                        bi = t;
                    }
                }
                if (bi >= 0 && best > t0 && best > tj) {
                    free_t[bi] = 1;
                    m = 1;
                }
            }
            matched[((size_t)b * P + i) * nthr + j] = m;
        }
    }
    // crowd targets and the top_k rule -> ignore flags
    for (int i = tid; i < n; i += MATCH_THREADS) {
        float best = 0.f;
        bool any = false;
        if (used[i] && nc > 0) {
            const float* pb = pbox + i * 4;
            const float det_area = (pb[2] - pb[0]) * (pb[3] - pb[1]);
            for (int t = 0; t < nc; ++t) {
                float v = 0.f;
                if (ccls[t] == pcls[i]) {
                    float w = fminf(pb[2], cbox[t * 4 + 2]) - fmaxf(pb[0], cbox[t * 4 + 0]);
                    float h = fminf(pb[3], cbox[t * 4 + 3]) - fmaxf(pb[1], cbox[t * 4 + 1]);
                    w = w < 0.f ? 0.f : w;
                    h = h < 0.f ? 0.f : h;
                    v = (w * h) / det_area;
                }
                if (!any || v > best) best = v;
                any = true;
            }
        }
        for (int j = 0; j < nthr; ++j) ignore[((size_t)b * P + i) * nthr + j] = (!used[i] || (any && best > thr[j])) ? 1 : 0;
    }
    for (int i = n + tid; i < P; i += MATCH_THREADS)
        for (int j = 0; j < nthr; ++j) {
            matched[((size_t)b * P + i) * nthr + j] = 0;
            ignore[((size_t)b * P + i) * nthr + j] = 1;
        }
}
static size_t match_smem(const sgx_match_desc* d) {
    return ((size_t)d->P * 6 + (size_t)d->nmax * 5 + (size_t)d->cmax * 5) * 4 + (size_t)d->P * 4 + (size_t)d->P + (size_t)d->nthr * d->nmax + 64;
}
extern "C" int32_t sgx_detection_match(const sgx_match_desc* d, const float* preds, const int32_t* pred_count, const float* targets,
                                       const int32_t* gt_count, const int32_t* gt_index, const float* crowd, const int32_t* crowd_count,
                                       const int32_t* crowd_index, const float* thresholds, uint8_t* matched, uint8_t* ignore, void* stream) {
    SGX_CHECK_ARG(d && preds && pred_count && thresholds && matched && ignore, "detection_match: null pointer");
    SGX_CHECK_ARG(d->B > 0 && d->P > 0 && d->nthr > 0 && d->nthr <= MATCH_THREADS && d->top_k > 0, "detection_match: bad dims");
    SGX_CHECK_ARG(d->nmax == 0 || (targets && gt_count && gt_index), "detection_match: null targets");
    SGX_CHECK_ARG(d->cmax == 0 || (crowd && crowd_count && crowd_index), "detection_match: null crowd targets");
    const size_t smem = match_smem(d);
    if (smem > 160 * 1024) SGX_FAIL(SGX_ERR_UNSUPPORTED, "detection_match: %zu bytes of LDS needed (P=%d, targets %d, crowd %d)", smem, d->P, d->nmax, d->cmax);
#ifndef SGX_EMU
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)match_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) SGX_FAIL(SGX_ERR_HIP, "detection_match: cannot reserve %zu bytes of LDS: %s", smem, hipGetErrorString(e));
    }
#endif
    SGX_LAUNCH(match_kernel, dim3(d->B), dim3(MATCH_THREADS), smem, stream, *d, preds, pred_count, targets, gt_count, gt_index, crowd, crowd_count,
               crowd_index, thresholds, matched, ignore);
    SGX_CHECK_LAUNCH("detection_match");
    return SGX_OK;
}
