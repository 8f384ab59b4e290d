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

// ---- LDS the weight-gradient kernels leave to the other streams ------------------------------------------------------------------------
// The weight gradients run on a side stream underneath the dependent kernels of the backward pass.  Their workgroups live for hundreds of
// microseconds and, four to a CU, hold 150 of its 160 KB of LDS: a data-gradient workgroup of the main stream (21 - 78 KB) then waits for
// one of them to END before it can start at all (r4t: 77 us kernels taking 500 - 700 us; 6.8 ms per step over the main stream).  With a
// reserve set (sgx_conv_set_wgrad_lds_reserve), the weight-gradient launches ask for enough dynamic LDS on top of their static
// allocation that one workgroup fewer fits a CU and at least the reserve stays free.
#include <atomic>
#include <map>
#include <mutex>
inline std::atomic<int> g_wg_lds_reserve{0};  // bytes
template <typename KernelFn>
static unsigned wg_lds_pad(KernelFn kernel) {
#ifdef SGX_EMU
    (void)kernel;
    return 0;
#else
    const long reserve = g_wg_lds_reserve.load(std::memory_order_relaxed);
    if (reserve <= 0) return 0;
    // keyed by the kernel's ADDRESS: every wgrad_kernel<...> instantiation has the same function-pointer type, so a function-local static
    // of this template would be shared by all tile shapes
    static std::mutex mu;
    static std::map<const void*, long> sizes;
    long s;
    {
        std::lock_guard<std::mutex> g(mu);
        auto it = sizes.find((const void*)kernel);
        if (it == sizes.end()) {
            hipFuncAttributes a;
            it = sizes.emplace((const void*)kernel, hipFuncGetAttributes(&a, (const void*)kernel) == hipSuccess ? (long)a.sharedSizeBytes : 0L).first;
        }
        s = it->second;
    }
    if (s <= 0) return 0;
    const long cu = 160 * 1024;
    const long now = cu / s;
    long want = (cu - reserve) / s;
    if (want < 1) want = 1;
    if (want >= now) return 0;
    long per = cu / (want + 1) + 2048;  // one more workgroup than `want` must not fit
    if (per * want > cu) per = cu / want;
    return per > s ? (unsigned)(per - s) : 0u;
#endif
}
