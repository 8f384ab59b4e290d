/* TEST INFRASTRUCTURE ONLY (the parity oracle) - never linked into the product library.
 *
 * CPU restatement of torchvision.ops.nms, the routine the reference calls at
 *   /root/reference/src/super_gradients/training/models/detection_models/pp_yolo_e/post_prediction_callback.py:85,87
 * torchvision is a third-party dependency pinned only as `torchvision>=0.10.0`
 * (/root/reference/requirements.txt:12); its source is not vendored under /root/reference and
 * the package is not installed in the build container, so this restates the PUBLISHED algorithm
 * of torchvision/csrc/ops/cpu/nms_kernel.cpp (v0.10 .. v0.20 are identical in arithmetic):
 *   areas = (x2-x1)*(y2-y1); order = stable argsort(scores, descending);
 *   greedy scan; suppress j iff inter/(area_i+area_j-inter) > thr  (strict >), fp32 arithmetic.
 * PARITY UNPINNED: no reference test holds NMS golden vectors; the fixtures in
 * tests/golden/nms_*.npz are generated from THIS file.
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -shared -fPIC nms.c -o _build/liboracle_nms.so
 */
#include <stdint.h>
#include <stdlib.h>

/* order: indices sorted by score descending, ties by ascending index (stable). */
static const float* g_scores;
static int cmp_desc(const void* a, const void* b) {
    int64_t ia = *(const int64_t*)a, ib = *(const int64_t*)b;
    float sa = g_scores[ia], sb = g_scores[ib];
    if (sa > sb) return -1;
    if (sa < sb) return 1;
    return (ia > ib) - (ia < ib);
}

int64_t oracle_nms(const float* boxes, const float* scores, int64_t n, float thr, int64_t* keep) {
    if (n <= 0) return 0;
    int64_t* order = (int64_t*)malloc(sizeof(int64_t) * n);
    uint8_t* sup = (uint8_t*)calloc(n, 1);
    float* area = (float*)malloc(sizeof(float) * n);
    for (int64_t i = 0; i < n; ++i) {
        order[i] = i;
        area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    g_scores = scores;
    qsort(order, n, sizeof(int64_t), cmp_desc);
    int64_t nk = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3], ia = area[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; if (w < 0.f) w = 0.f;
            float h = yy2 - yy1; if (h < 0.f) h = 0.f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > thr) sup[j] = 1;
        }
    }
    free(order); free(sup); free(area);
    return nk;
}
