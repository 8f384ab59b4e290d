// Internal interface between the grouped weight-gradient entry points (conv.hip) and the patch kernel (wgrad_patch.hip).  Not part of the C ABI.
#pragma once
#include "sgx_common.h"

#define WP_MAX_JOBS 20   // the job table travels as kernel arguments: 20 x 152 B + 88 B < 4 KB
#define WP_MAX_SPLIT 4096

struct WpJob {
    const float* X;
    const float* DY;
    float* dw;
    float* part;   // [tile][ksplit][node]: node values of the fold tree, in place (a node lives in its leftmost leaf's slot)
    int* tickets;  // [tile][ksplit]: one per sibling pair
    long x_ld_pix, x_ld_img, y_ld_pix, y_ld_img;
    long x_bytes, dy_bytes;
    int H, W, C, K, pad, Ho, Wo;
    int tiles_h, tiles_w, ntiles;  // pixel tiles of (32 / PC) x PC output pixels
    int ksplit, tchunk;            // pixel-tile ranges: count, tiles per range
    int kt_tiles, ct_tiles;        // filter tiles x channel chunks
    int blk0;                      // first workgroup of the job in its launch (a multiple of 8: XCD phase 0)
    int xcd_ranges;                // 1: workgroup b works on pixel range b % 8 + ... (enough ranges to keep the eight XCDs level)
};
struct WpGroupParams {
    int njobs, xcd_order;
    int blk0[WP_MAX_JOBS];
    WpJob jobs[WP_MAX_JOBS];
};
struct WpPlan {
    int cfg;              // 0: not a patch problem
    int pc, kb, cb, wt;   // kernel form: tile columns, wave grid (filter blocks x channel blocks x tap groups)
    int nks;              // K steps of 16 pixels per tile (by stride)
    int kt_tiles, ct_tiles, tiles_h, tiles_w;
    long ntiles;
    int ksplit, tchunk;
    long part_off, ticket_off;  // floats, ints (inside the group's workspace / ticket buffer)
};
bool wpatch_plan_job(const sgx_conv_desc* d, WpPlan& pl, int kb_override, int min_fill_pct);
void wpatch_plan_split(const sgx_conv_desc* d, WpPlan& pl, double item_flops, long* part_floats, long* ticket_ints);
int32_t wpatch_launch(int stride, const WpPlan& form, const WpGroupParams& g, int nblk, void* stream);
