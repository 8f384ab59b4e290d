/* libsgx_hip.so - flat C ABI of the MI355X (gfx950) train-step hot path.
 *
 * The reference (Deci-AI/super-gradients 3.7.1) owns NO native code and has NO FFI: every FLOP of
 * its train step is an ATen / torchvision call issued from Python.  This
 * header is therefore the boundary a maintainer would bind (ctypes stub in INTEGRATION.md) to replace
 * those call sites; each entry point cites the reference call site(s) it replaces.  Paths are
 * relative to /root/reference/src/super_gradients/.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no torch types.  All tensors are fp32 unless noted.
 *   - Activations are NHWC in HBM with explicit element strides: element (n,h,w,c) of a tensor lives
 *     at  base + n*ld_img + (h*W + w)*ld_pix + c   (ld_pix >= C lets a producer write straight into a
 *     channel slice of a concat buffer - the reference's torch.cat copies disappear).
 *     Channel counts, ld_pix and channel offsets must be multiples of 4 floats (16-byte vector access).
 *   - Convolution weights are OHWI: w[k][r][s][c]  (the GEMM-K axis, c, is contiguous).
 *   - Every call is asynchronous on the given hipStream_t (passed as void*), never allocates, never
 *     synchronises, keeps no pointer after return.  Scratch comes from the caller (workspace queries).
 *   - Return value: 0 = OK, negative = error (see codes); sgx_last_error() gives a thread-local message.
 *   - Re-entrant: safe from the Python main thread and the autograd engine thread concurrently.  The few process-wide settings
 *     (sgx_conv_set_math, sgx_conv_tuning_load, sgx_bn_set_fused_finalize, the sgx_debug_* measurement aids) are atomics / lock-protected:
 *     changing one while another thread is inside a call affects the calls that follow.
 */
#ifndef SGX_HIP_H
#define SGX_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGX_OK 0
#define SGX_ERR_BAD_ARG (-1)
#define SGX_ERR_UNSUPPORTED (-2)
#define SGX_ERR_HIP (-3)
#define SGX_ERR_WORKSPACE (-4)

#define SGX_ACT_NONE 0
#define SGX_ACT_RELU 1
#define SGX_ACT_SILU 2

int32_t sgx_version(void);
const char* sgx_last_error(void);

/* Measurement aid (bench.py roofline leg).  While enabled, every launch of the two MFMA kernel classes is bracketed by
 * HIP events on its own launch stream and its algorithmic FLOPs (2*M*N*K of the real, unpadded problem) are tallied.
 * cls 0 = implicit-GEMM kernel (conv forward and data gradient, fp32 matrix pipe), cls 1 = weight-gradient kernel (one record per
 * grouped launch), cls 2 = the patch kernel (conv forward and data gradient of 3x3 problems on the bf16 matrix pipe, conv math modes 3+),
 * cls 3 = implicit-GEMM launches in bf16x3 arithmetic (bf16 matrix pipe; conv math modes 1, 2, 4, 5).
 * sgx_prof_summary synchronises on the recorded events and returns the sums since the last sgx_prof_enable().      */
int32_t sgx_prof_enable(int32_t on);
int32_t sgx_prof_summary(int32_t cls, double* ms, double* flops, int64_t* launches);
/* algorithmic HBM bytes of the same launches: every input element, weight and output element moved once (fp32)           */
int32_t sgx_prof_bytes(int32_t cls, double* bytes);
/* roofline time of the same launches: sum over launches of max(FLOPs / peak_flops, bytes / hbm_bytes_per_s), in ms - the time a
 * launch mix would take if every launch ran at whichever of the two bounds (matrix pipe, HBM) is the tighter one for ITS shape      */
int32_t sgx_prof_bound_ms(int32_t cls, double peak_flops, double hbm_bytes_per_s, double* ms);

/* Per-problem tuning table: n entries of 12 int32 {kind (0 = forward, 1 = data gradient), N, H, W, C, K, R, stride, pad, BM, BN, variant}
 * (BM / BN = 0: keep the heuristic's), or {2 = weight gradient, N, ..., pad, filter tile, (tap, channel) tile, split target in waves} - a
 * different split regroups the pixel sum of dW (fp32 rounding level, deterministic for a given table).  A convolution call whose descriptor matches an entry uses that tile / kernel variant instead of the
 * built-in heuristic - what tools/conv_tune.py measured as the fastest for that problem on this chip; outputs are bit-identical (every tile / variant
 * reduces in the same order; only the statistic partial rows regroup with BM).  Load before the first launch (not synchronised with running calls); n = 0 clears the table.             */
int32_t sgx_conv_tuning_load(const int32_t* entries, int32_t n);
int32_t sgx_conv_tuning_size(void);

/* Measurement aid (tools/conv_tune.py): force the conv tile shapes (0 = built-in heuristic).  Not thread-safe; never set by the product. */
int32_t sgx_debug_set_tiles(int32_t bm, int32_t bn, int32_t wgrad_bnk, int32_t wgrad_bj, int32_t wgrad_split_target);
/* Arithmetic of the forward / data-gradient GEMMs.  0: fp32 matrix pipe, exact fp32 FMA chains.  1: "bf16x3" - every fp32
 * operand is split into three bf16 pieces (24 mantissa bits) and the six significant cross products run on the bf16 matrix pipe with
 * fp32 accumulation: fp32-accurate results (dropped terms <= 2^-24 of a product) at 2.7x fewer matrix-pipe cycles.  2: per problem -
 * bf16x3 where the reduction depth (taps x channels) is >= 192, fp32 MFMA for shallow ones.  3 (the default of round 3): the 3x3 stride-1 pad-1
 * problems with C % 16 == 0 on output maps of 40 x 40 and larger (forward, the QARepVGG two-branch forward, data gradient, two-source
 * data gradient) run the PATCH kernel: a workgroup owns 8 x 16 output pixels of one image, stages their 10 x 18 input patch and the
 * filter slabs of all nine taps in LDS once per 16-channel chunk (split into three bf16 planes) and reads the taps from there - bf16x3
 * arithmetic as in mode 1; statistics rows are then one per tile (sgx_conv2d_fwd_stat_blocks follows); every other problem stays on the
 * fp32 pipe.  4: mode 3's patch kernel on its problems, mode 2's per-problem rule for the rest.  5 (DEFAULT since round 4): mode 4, and the
 * two-branch / two-source (QARepVGG) launches follow the per-problem rule too - the five correction products of the bf16x3 scheme keep their
 * own accumulator in every form (three accumulators per block in the two-output launch): a single shared accumulator leaves a same-signed
 * offset of 1e-8 .. 8e-8 of the output's rms per convolution (the bf16 MFMA's accumulate floors what it shifts out).  Process-wide.           */
int32_t sgx_conv_set_math(int32_t mode);
int32_t sgx_conv_get_math(void);
int32_t sgx_debug_set_variant(int32_t wave_layout_variant);

/* ---------------------------------------------------------------------------------------------
 * Convolution (implicit GEMM on fp32 MFMA, v_mfma_f32_32x32x2_f32).
 * Replaces torch.nn.functional.conv2d fwd/bwd issued by nn.Conv2d inside
 *   modules/qarepvgg_block.py:105-128,197-198   (branch_3x3 / branch_1x1)
 *   modules/conv_bn_act_block.py:88,93          (Conv)      modules/conv_bn_relu_block.py:8-60
 *   training/models/detection_models/yolo_nas/dfl_heads.py:57-66 (stem / cls / reg / pred convs)
 *   training/models/classification_models/resnet.py:29-31,57-61,162 (ResNet convs)
 * ------------------------------------------------------------------------------------------- */
typedef struct sgx_conv_desc {
    int32_t N, H, W, C;       /* input: images, height, width, channels (C % 4 == 0)            */
    int32_t K;                /* output channels                                                */
    int32_t R, S;             /* filter height / width                                          */
    int32_t stride, pad;      /* same in both spatial dims; groups = 1, dilation = 1            */
    int32_t Ho, Wo;           /* output spatial = (H + 2*pad - R)/stride + 1                    */
    int64_t x_ld_pix, x_ld_img; /* input strides (elements)                                     */
    int64_t y_ld_pix, y_ld_img; /* output strides (elements)                                    */
} sgx_conv_desc;

/* y = act(conv(x, w) + bias [+ addend]);  bias/addend may be NULL; addend has y's strides.
 * If stat_partials != NULL the epilogue also emits per-CTA-row-block per-channel partial sums of the
 * PRE-activation output: stat_partials[2][n_row_blocks][K] (sum, sum of squares), where
 * n_row_blocks = sgx_conv2d_fwd_stat_blocks(d).  (BatchNorm statistics without re-reading y.)  */
int32_t sgx_conv2d_fwd(const sgx_conv_desc* d, const float* x, const float* w, const float* bias,
                       const float* addend, float* y, int32_t act, float* stat_partials, void* stream);
int32_t sgx_conv2d_fwd_stat_blocks(const sgx_conv_desc* d);

/* dx = conv_transpose(dy, w) [+ addend] [+ dx if accumulate].  d describes the FORWARD conv; dy uses
 * the y strides, dx/addend the x strides.  ws: sgx_conv2d_bwd_data_workspace(d) bytes.            */
int64_t sgx_conv2d_bwd_data_workspace(const sgx_conv_desc* d);
int32_t sgx_conv2d_bwd_data(const sgx_conv_desc* d, const float* dy, const float* w, const float* addend,
                            float* dx, int32_t accumulate, void* ws, int64_t ws_bytes, void* stream);

/* The data gradient reads the weights transposed per output-parity class ([c][tap][k]).  sgx_conv2d_bwd_data does that transpose
 * into its workspace on every call; a caller that knows the weights are fixed between forward and backward can do it ahead of
 * time (e.g. on a side stream under the forward pass) into a buffer of sgx_conv2d_bwd_data_workspace(d) bytes and call the _wt form. */
int32_t sgx_conv2d_transpose_weights(const sgx_conv_desc* d, const float* w, float* wt, int64_t wt_bytes, void* stream);
int32_t sgx_conv2d_bwd_data_wt(const sgx_conv_desc* d, const float* dy, const float* wt, const float* addend, float* dx,
                               int32_t accumulate, void* stream);
/* BatchNorm-backward REDUCE inside the data gradient that finalises a layer's output gradient (round 4).  The reference runs
 * aten::native_batch_norm_backward per BatchNorm2d (modules/conv_bn_act_block.py:88-93 under sg_trainer.py:611-647): a reduction over dy and
 * the saved conv output, then the input gradient.  When the LAST writer of a layer's dy is a data-gradient launch, its epilogue holds every
 * dy value in registers: given the layer's saved pre-BatchNorm conv output t and scale / shift / mean it leaves the per-channel partial sums
 * sum g, sum g (t - mean) with g = dy * act'(scale t + shift) - exactly the rows sgx_bn_bwd_reduce produces with its own pass over dy and t
 * (same fp32 arithmetic per element; the rows regroup with the launch's tile rows, the fp64 finalize is unchanged).
 * A request covers a channel range [c_lo, c_hi) of dx (multiples of 4): a data gradient that writes a concat gradient can serve the
 * layers behind two of its slices.  partials: [2][rows][c_hi - c_lo] with rows = sgx_conv2d_bwd_data_stat_blocks(d, two_source) (0: this
 * problem cannot carry requests - unaligned rows, fewer than 16 filters ...).  t is indexed like dx: [N, H, W] pixels, its own strides.    */
#define SGX_MAX_BN_REQ 2
typedef struct sgx_bn_reduce_req {
    int32_t c_lo, c_hi;
    const float* t;
    int64_t t_ld_pix, t_ld_img;
    const float* scale;
    const float* shift;
    const float* mean;
    int32_t act;
    int32_t rows;     /* rows the partials were allocated for: must equal sgx_conv2d_bwd_data_stat_blocks */
    float* partials;
} sgx_bn_reduce_req;
int32_t sgx_conv2d_bwd_data_stat_blocks(const sgx_conv_desc* d, int32_t two_source);
int32_t sgx_conv2d_bwd_data_wt_req(const sgx_conv_desc* d, const float* dy, const float* wt, const float* addend, float* dx,
                                   int32_t accumulate, const sgx_bn_reduce_req* reqs, int32_t nreq, void* stream);
/* (the two-source form, sgx_conv2d_bwd_data_dual below, with requests) */
int32_t sgx_conv2d_bwd_data_dual_req(const sgx_conv_desc* d, const float* dy, const float* wt, const float* ds, int64_t ds_ld_pix,
                                     int64_t ds_ld_img, const float* w1t, const float* addend, const float* addend2, int64_t a2_ld_pix,
                                     int64_t a2_ld_img, float a2_scale, const float* a2_scale_dev, float* dx, int32_t accumulate,
                                     const sgx_bn_reduce_req* reqs, int32_t nreq, void* stream);
/* All of a network's data-gradient weight transposes as ONE launch per step instead of one per convolution and parity class
 * (YOLO-NAS-S: 165 launches of ~10 us): sgx_conv2d_transpose_jobs appends the jobs of convolution d (weights w -> buffer wt of
 * sgx_conv2d_bwd_data_workspace(d) bytes; pointers must stay valid - they are arena views) to a HOST array; the caller uploads the
 * array once and runs it with sgx_wtrans_batch before each backward pass.                                                        */
typedef struct sgx_wtrans_job {
    const float* w;      /* [K][RS][C]                                     */
    float* wt;           /* [C][T][K] of this parity class                  */
    int32_t K, C, RS, T; /* T taps of the class, listed in taps[]            */
    uint8_t taps[64];
} sgx_wtrans_job;
int32_t sgx_conv2d_transpose_jobs(const sgx_conv_desc* d, const float* w, float* wt, int64_t wt_bytes, sgx_wtrans_job* jobs,
                                  int32_t max_jobs, int32_t* njobs);
int32_t sgx_wtrans_batch(const sgx_wtrans_job* jobs_dev, int32_t njobs, void* stream);

/* QARepVGG block, both convolution branches per launch (modules/qarepvgg_block.py:184-204: branch_3x3 conv and branch_1x1 read the same x).
 * Forward: y = convRxS(x, w) (no bias; d describes it, pad = R / 2), u = conv1x1(x, w1) + bias1 with the same stride - the 1x1 filter reads the
 * RxS filter's centre tap, so one workgroup produces both tiles; stat5 = [5][sgx_conv2d_fwd_dual_stat_blocks(d)][K] per-row-block sums of
 * y, y^2, u0, u0^2, y*u0 with u0 = u - bias1 (every moment both BatchNorms of the block need; sgx_qarep_fwd_finalize adds the bias terms in
 * fp64).  u has y's strides.  Needs C >= 16.
 * Backward: dx = convT RxS(dy, wt) + convT 1x1(ds, w1t) [+ addend] [+ dx]; wt as sgx_conv2d_transpose_weights writes it, w1t = w1 transposed
 * [C][K]; ds has its own strides.  Needs K >= 16.  addend2 (optional, stride-1 blocks): dx += a2_scale * a2_scale_dev[0] * addend2 with its own
 * strides - the residual branch of a YOLO-NAS bottleneck (yolo_stages.py:61-63: d(alpha * x) = alpha * dz) folded into the launch.
 * sgx_qarep_prep_batch: per optimizer step, for every block at once: w1p = alpha * w1 + I (identity branch and alpha folded into the 1x1
 * filter) and its transpose w1pt.                                                                                                          */
int32_t sgx_conv2d_fwd_dual_stat_blocks(const sgx_conv_desc* d);
int32_t sgx_conv2d_fwd_dual(const sgx_conv_desc* d, const float* x, const float* w, const float* w1, const float* bias1, float* y, float* u,
                            float* stat5, void* stream);
int32_t sgx_conv2d_bwd_data_dual(const sgx_conv_desc* d, const float* dy, const float* wt, const float* ds, int64_t ds_ld_pix, int64_t ds_ld_img,
                                 const float* w1t, const float* addend, const float* addend2, int64_t a2_ld_pix, int64_t a2_ld_img, float a2_scale,
                                 const float* a2_scale_dev, float* dx, int32_t accumulate, void* stream);
typedef struct sgx_qarep_prep_job {
    const float* w1;    /* [K][C] the block's 1x1 filter (OHWI, C padded)   */
    float* w1p;         /* [K][C] alpha * w1 + identity                      */
    float* w1pt;        /* [C][K] its transpose                              */
    const float* alpha; /* device scalar or NULL (= 1)                       */
    int32_t K, C, identity, pad_;
} sgx_qarep_prep_job;
int32_t sgx_qarep_prep_batch(const sgx_qarep_prep_job* jobs_dev, int32_t njobs, void* stream);

/* Pre-split filter planes of the bf16x3 convolutions (round 5).  The reference keeps one fp32 weight tensor per convolution
 * (modules/qarepvgg_block.py:184-204, conv_bn_act_block.py:68-93) and leaves its representation inside a launch to cuDNN; here a bf16x3
 * launch stages its filter as three bf16 pieces per element, and a filter changes once per optimizer step while every pixel tile of every
 * launch would split it again.  sgx_filter_planes_batch splits each job's filter src[rows][taps][ch] (dense, ch a multiple of 16) once into
 * planes[ch / 16][hi | mid | lo][tap][row][16 bf16] (sgx_filter_planes_bytes(rows, taps, ch) bytes) and registers src -> planes; forward /
 * data-gradient launches whose filter pointer and shape match a VALID entry, made while a scope is open, copy the planes instead of
 * splitting - bit-identical results.  Contract of the caller (modules/engine.py): open the scope only around a training step's forward /
 * backward, run the batch after the last weight update of the step, invalidate (NULL: everything) before weights change or the
 * addresses are released.  jobs_host and jobs_dev hold the same records (host copy for the registry, device copy for the kernel).
 * Ordering: the registry entries become valid when the batch is ENQUEUED; the launches that read the planes must be ordered behind it on
 * the device (the same stream, or an event) - as every consumer of `stream`'s earlier work must.  The scope is a DEPTH COUNTER OF THE
 * CALLING THREAD (open / close nest; a launch issued by another thread, or outside every scope, never looks planes up); the registry is
 * process-wide and keyed by the filter's device address (one-process-per-GPU model of the data-parallel path).                        */
typedef struct sgx_fplanes_job {
    const float* src; /* [rows][taps][ch] fp32                                         */
    void* planes;     /* sgx_filter_planes_bytes(rows, taps, ch), 16-byte aligned      */
    int32_t rows, taps, ch, pad_;
} sgx_fplanes_job;
int64_t sgx_filter_planes_bytes(int32_t rows, int32_t taps, int32_t ch);
int32_t sgx_filter_planes_batch(const sgx_fplanes_job* jobs_host, const sgx_fplanes_job* jobs_dev, int32_t njobs, void* stream);
int32_t sgx_filter_planes_invalidate(const sgx_fplanes_job* jobs_host, int32_t njobs);
int32_t sgx_filter_planes_scope(int32_t open);
int32_t sgx_debug_set_filter_planes(int32_t mode); /* 0 = every launch splits its filter while staging; 1 (default) = planes copied into the
                                                    * LDS slabs; 2 = 1, and every GEMM-loop launch on a tile of one 32-filter block per wave
                                                    * reads its filter fragments straight from the planes into registers (in mode 1: the
                                                    * problems whose tuning-table variant is 12, and the two-output forward pair)          */
int64_t sgx_debug_filter_planes_hits(void);      /* launches that read planes so far (tests)                         */

/* dw[k][r][s][c] += sum_pixels dy * x   (accumulates into dw: callers zero the gradient arena once per
 * optimizer step).  dbias[k] += sum dy if dbias != NULL.  ws: sgx_conv2d_bwd_weight_workspace(d).  */
int64_t sgx_conv2d_bwd_weight_workspace(const sgx_conv_desc* d);
int32_t sgx_conv2d_bwd_weight(const sgx_conv_desc* d, const float* x, const float* dy, float* dw,
                              float* dbias, void* ws, int64_t ws_bytes, void* stream);
/* The weight gradients of a training step are mutually independent (the reference's autograd issues one aten::convolution_backward per
 * nn.Conv2d, sg_trainer.py:611-647 -> qarepvgg_block.py:105-128): a GROUP of them runs as one launch per tile shape, the pixel split of
 * every job sized so that the group - not each layer alone - fills the chip, and the split partials are folded inside the launch by
 * arrival tickets in a fixed order (deterministic; no separate reduce launch).  jobs: HOST array (read during the call only); ws:
 * sgx_conv2d_bwd_weight_group_sizes bytes of scratch; tickets: that many int32, ZERO on entry, owned by this stream between calls (the
 * kernels leave them zero, so a caller clears the buffer once).  dw += gradient, as above.                                            */
typedef struct sgx_wgrad_job {
    sgx_conv_desc d;
    const float* x;
    const float* dy;
    float* dw;
} sgx_wgrad_job;
int32_t sgx_conv2d_bwd_weight_group_sizes(const sgx_wgrad_job* jobs, int32_t njobs, int64_t* ws_bytes, int64_t* ticket_ints);
int32_t sgx_conv2d_bwd_weight_group(const sgx_wgrad_job* jobs, int32_t njobs, void* ws, int64_t ws_bytes, int32_t* tickets,
                                    int64_t ticket_ints, void* stream);
/* Measurement aid for the grouped weight gradient: rounds of work items a large group is cut into (0 = default 6), work of an item below
 * which a small group is not cut further (MFLOP, 0 = default 8; an item is never larger than twice that), XCD-aware workgroup order (default 1).  Never set by the product.     */
int32_t sgx_debug_set_wgrad_group(int32_t rounds, int32_t item_mflop, int32_t xcd_order);
/* Measurement aid: deep_slab bit 0 = 32-pixel slabs in the weight-gradient loop for the tiles whose two slabs fit 32 KB of LDS, bit 1 = ONE
 * slab of global loads in flight instead of two (one register set), bit 2 = the 64x64 tile on two waves instead of four, bit 3 = the
 * bf16x3 loop (three bf16 planes per slab, operands through the LDS transpose read, six bf16 MFMAs per product; the main tile shapes); `ablate` is
 * honoured only by a library built with -DSGX_WGRAD_LAB (tools/wgrad_lab.py: loop ablations, results are wrong by design).          */
int32_t sgx_debug_set_wgrad_loop(int32_t deep_slab, int32_t ablate);
/* Arithmetic of the weight gradient (aten::convolution_backward's weight half under sg_trainer.py:611-647), process-wide.
 * 0: the fp32 matrix pipe.  1: the bf16x3 slab loop (three bf16 planes per slab, operands through the LDS transpose read, six bf16 MFMAs per
 * product: fp32-accurate).  2 (DEFAULT): mode 1, and the 3x3 pad-1 problems (stride 1 and 2) run the PATCH kernel - a workgroup owns a
 * (32 / 64 / 96 filters) x (32 channels) x (nine taps) block of dW, walks tiles of 32 output pixels, stages the dY tile and the input patch
 * of a tile ONCE (one bf16x3 split per element instead of one per tap) and the taps read their operands from the patch (a tap is an LDS
 * address offset).  sgx_debug_set_wgrad_loop bit 4 keeps the patch kernel out, bit 5 forces the fp32 loop (measurement / parity tests).   */
int32_t sgx_conv_set_wgrad_math(int32_t mode);
int32_t sgx_conv_get_wgrad_math(void);
/* Measurement aid for the patch kernel: largest work item (MFLOP, 0 = default 48), filter blocks per workgroup (1..3, 0 = by padding), least
 * share of useful matrix work (filters x channels x pixels over their padded tiles, percent, 0 = default 60) for a job to take the kernel. */
int32_t sgx_debug_set_wgrad_patch(int32_t item_mflop, int32_t kb, int32_t min_fill_pct);
/* Measurement aid: the reduction depth (taps x channels) from which conv math modes 2 / 4 / 5 run a problem in bf16x3 arithmetic
 * (0 = the default, 192).                                                                                                      */
int32_t sgx_debug_set_bf3_min_depth(int32_t depth);
/* measurement switch (round 5): the patch conv kernel's 32-filter tiles request the next tap's fragments ahead of this tap's MFMAs (1, default) or not (0) */
int32_t sgx_debug_set_pconv_pipe(int32_t on);
/* LDS (KB per CU, 0..120; 0 = off) the weight-gradient kernels leave free for kernels of other streams: their launches then request
 * dynamic LDS on top of their static allocation so that fewer of their workgroups fit a CU.  The weight gradients run on a side stream
 * under the backward pass; four of their workgroups hold 150 of a CU's 160 KB, and a data-gradient workgroup of the main stream (the
 * step's critical path) cannot start before one of them ends.                                                                */
int32_t sgx_conv_set_wgrad_lds_reserve(int32_t kb);
int32_t sgx_conv_get_wgrad_lds_reserve(void);
/* A HIP stream whose kernels are dispatched to `percent` (10..100) of the device's CUs only, evenly spread over the chip
 * (hipExtStreamCreateWithCUMask).  For the weight gradients' side stream: their long-lived workgroups otherwise take every CU and the short
 * dependent kernels of the main stream - the step's critical path - queue between them.  *stream is a hipStream_t; release it with
 * sgx_stream_destroy.  SGX_ERR_UNSUPPORTED on the host emulation.                                                          */
int32_t sgx_stream_create_partial(int32_t percent, void** stream);
int32_t sgx_stream_destroy(void* stream);

/* ConvTranspose2d kernel 2, stride 2 (+bias): modules/sampling.py:72-73 via yolo_stages.py:292-294.
 * x [N,H,W,C] -> y [N,2H,2W,K].  It is the adjoint of a 2x2 stride-2 convolution, so it runs on the same
 * kernels: forward = data-gradient kernel of that conv, backward-data = its forward, backward-weight =
 * its weight-gradient.  Weight layout wt[c][r][s][k] (OHWI of the adjoint conv; nn.ConvTranspose2d's
 * logical [C][K][2][2] permuted - the host mirror keeps the logical shape as a strided view).        */
int64_t sgx_convT2x2_workspace(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K);
int32_t sgx_convT2x2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix,
                         int64_t x_ld_img, const float* wt, const float* bias, float* y, int64_t y_ld_pix,
                         int64_t y_ld_img, void* ws, int64_t ws_bytes, void* stream);
/* the same with the filter already in the data-gradient order of the adjoint convolution (sgx_conv2d_transpose_weights / _jobs on a
 * descriptor with K = C filters of K channels, R = S = 2, stride 2, pad 0): no transpose launches inside the call */
int32_t sgx_convT2x2_fwd_wt(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix,
                            int64_t x_ld_img, const float* wtt, const float* bias, float* y, int64_t y_ld_pix,
                            int64_t y_ld_img, void* stream);
int32_t sgx_convT2x2_bwd_data(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* dy, int64_t dy_ld_pix,
                              int64_t dy_ld_img, const float* wt, float* dx, int64_t dx_ld_pix, int64_t dx_ld_img,
                              void* stream);
int32_t sgx_convT2x2_bwd_weight(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const float* x, int64_t x_ld_pix,
                                int64_t x_ld_img, const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img, float* dwt,
                                float* dbias, void* ws, int64_t ws_bytes, void* stream);

/* Layout changes at the model boundary (reference keeps NCHW end to end).
 * nchw_to_nhwc pads channels with zeros up to Cpad (Cpad % 4 == 0).                               */
int32_t sgx_nchw_to_nhwc(int32_t N, int32_t C, int32_t H, int32_t W, int32_t Cpad, const float* x, float* y, void* stream);
int32_t sgx_nhwc_to_nchw(int32_t N, int32_t C, int32_t H, int32_t W, const float* x, int64_t x_ld_pix, int64_t x_ld_img,
                         float* y, void* stream);
/* Device side of the input pipeline: the uint8 HWC images the dataset yields, stacked [N,H,W,C], become the
 * standardized fp32 NHWC batch (channels zero-padded to Cpad) the first convolution reads - replacing, on the host, DetectionStandardize
 * (transforms.py:490-510: image / max_value), the optional (x - mean) / std normalisation, DetectionCollateFN's stack + moveaxis + float()
 * (collate_fn/detection_collate_fn.py:27-32) and this library's own nchw_to_nhwc.  y = (x / max_value - mean[c]) / std[c]; mean/std NULL:
 * y = x / max_value (a true division: bit-identical to the reference's numpy arithmetic for every uint8 value).                         */
int32_t sgx_standardize_u8_hwc(int32_t N, int32_t H, int32_t W, int32_t C, int32_t Cpad, const uint8_t* x, float max_value,
                               const float* mean, const float* std, float* y, void* stream);

/* The same for ragged batches: ONE image [h,w,C] uint8 into its slot y [H,W,Cpad] of the padded batch at offset (top, left), the standardized
 * pad_value[C] (uint8 scale, e.g. 114) everywhere else - DetectionPadIfNeeded / DetectionPadToSize (transforms.py:846-941; "center" or
 * "bottom_right" offsets, transforms/utils.py:79-106) + DetectionStandardize + the collate on the device, one launch per image.            */
int32_t sgx_pad_standardize_u8_hwc(int32_t h, int32_t w, int32_t C, const uint8_t* x, int32_t H, int32_t W, int32_t Cpad, int32_t top,
                                   int32_t left, float max_value, const float* mean, const float* std, const float* pad_value, float* y,
                                   void* stream);

/* predict(): a whole batch of ragged uint8 HWC images -> the standardized fp32 NHWC batch y [N][H][W][Cpad], ONE launch.  Replaces the
 * reference's per-image host passes (training/processing/processing.py): ReverseImageChannels :230-257, Detection[LongestMaxSize]Rescale
 * :510-589 (cv2.resize INTER_LINEAR, transforms/utils.py:17-25), Detection{Center,BottomRight,Auto}Padding :326-471 (np.pad of the uint8
 * image), StandardizeImage :260-295 (float64 division, cast to float32), NormalizeImage :298-323 ((x - mean) / std in float32) and the
 * batching of pipelines.py:241-247.  Image n: source [h0][w0][C] is rescaled to [h][w] (h == h0 and w == w0: copied), placed at (top, left)
 * of its H x W slot, pad_value[C] (uint8, in output channel order) elsewhere; caller guarantees top + h <= H, left + w <= W.  Rescale =
 * OpenCV's 8-bit INTER_LINEAR fixed-point arithmetic restated (exact 2x reductions: its INTER_AREA shortcut) - cv2 is absent from the
 * reference tree and from this image, so that one stage is not pinned against cv2 output (csrc/image.hip header).  standardize == 0:
 * y = (float)pixel; mean/std NULL: no normalisation.                                                                                      */
typedef struct sgx_image_job {
    const uint8_t* src; /* [h0][w0][C] uint8, device memory */
    int32_t h0, w0;     /* source size                      */
    int32_t h, w;       /* size after the rescale           */
    int32_t top, left;  /* position inside the H x W slot   */
} sgx_image_job;
int32_t sgx_preprocess_u8_hwc(const sgx_image_job* jobs_dev, int32_t N, int32_t C, int32_t Cpad, int32_t H, int32_t W, int32_t reverse_channels,
                              int32_t standardize, double max_value, const float* mean, const float* std, const uint8_t* pad_value, float* y,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm (training mode) and the fused elementwise stages around it.
 * Replaces F.batch_norm + ReLU/SiLU + the branch adds at
 *   modules/qarepvgg_block.py:118,162,184-204   modules/conv_bn_act_block.py:89-93
 *   yolo_nas/yolo_stages.py:61-63 (alpha*x + y)   classification_models/resnet.py:43-50,72-84
 * A "row" is one pixel (M = N*H*W rows), channels are columns.
 * ------------------------------------------------------------------------------------------- */
/* Two-stage deterministic per-channel reduction.  partials: [2][nblk][C], nblk = sgx_stats_blocks(M). */
int32_t sgx_stats_blocks(int64_t M);
/* Scratch (bytes) the *_finalize entry points need to fold nblk partial rows of C channels: when nblk is large they run a
 * wide fp64 pre-reduction into this workspace before the per-channel finalisation.                                        */
int64_t sgx_reduce_workspace(int32_t nblk, int32_t C);
int32_t sgx_channel_stats_partial(const float* x, int64_t M, int32_t C, int64_t ld, float* partials, void* stream);
/* From partial sums: batch mean / biased var -> scale = gamma*invstd, shift = beta - mean*scale;
 * saves mean & invstd; updates running stats in place (momentum, unbiased var) like nn.BatchNorm2d.  */
int32_t sgx_bn_finalize(const float* partials, int32_t nblk, int64_t M, int32_t C, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        float* save_mean, float* save_invstd, float* scale, float* shift, void* ws, int64_t ws_bytes, void* stream);
/* Synchronised BatchNorm across data-parallel ranks (reference: nn.SyncBatchNorm conversion at sg_trainer.py:1344-1350, recipe
 * `sync_bn: True`): the per-channel sums leave as fp64 [2][C] (sum, sum of squares / sum g, sum g*(x-mean)), the host all-reduces
 * them over RCCL, and the *_sums forms finish with the GLOBAL sums and element count.                               */
int32_t sgx_bn_reduce_sums(const float* partials, int32_t nblk, int32_t C, double* sums, void* ws, int64_t ws_bytes, void* stream);
int32_t sgx_bn_finalize_sums(const double* sums, int64_t M, int32_t C, const float* gamma, const float* beta, float eps, float momentum,
                             float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* scale, float* shift,
                             void* stream);
/* dgamma / dbeta accumulate THIS rank's sums (the gradient exchange adds the others); dx coefficients use the global ones.    */
int32_t sgx_bn_bwd_finalize_sums(const double* local_sums, const double* global_sums, int64_t M_total, int32_t C, const float* gamma,
                                 const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta, float* coef, void* stream);
/* eval-mode BatchNorm folded to an affine map from the running statistics.                          */
int32_t sgx_bn_eval_scale_shift(int32_t C, const float* gamma, const float* beta, const float* running_mean,
                                const float* running_var, float eps, float* scale, float* shift, void* stream);
/* y = act(scale[c]*x + shift[c] + a1*r1 + a2*r2); scale/shift/r1/r2 may be NULL (identity / absent).
 * a1_dev (device scalar, e.g. the learnable bottleneck alpha) overrides a1 when not NULL.
 * If partials != NULL also emits per-channel partial stats of the PRE-activation value.            */
int32_t sgx_affine_act_fwd(const float* x, int64_t x_ld, const float* scale, const float* shift, const float* r1,
                           int64_t r1_ld, float a1, const float* a1_dev, const float* r2, int64_t r2_ld, float a2,
                           float* y, int64_t y_ld, int64_t M, int32_t C, int32_t act, float* partials, void* stream);
/* BN backward, stage 1: g = dy * act'(scale*x+shift) ; partial sums of g and g*(x - mean) per channel
 * (partials [2][nblk][C]).  act mask is recomputed from x, scale, shift (nothing else is stored).  */
int32_t sgx_bn_bwd_reduce(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* scale,
                          const float* shift, const float* save_mean, int64_t M, int32_t C, int32_t act, float* partials,
                          void* stream);
/* stage 2 (tiny): dgamma += sum g*xhat, dbeta += sum g; coefficients for stage 3:
 * dx = c1[c] * ((g - mg[c] - mg_lo[c]) - (x - mean[c]) * k[c]).  coef: [5][C] = c1, mg, k, mean, mg_lo (the mean of g travels as hi + lo
 * floats: its rounding would otherwise be a constant per-channel offset in every dx element, see csrc/bn.hip).                          */
int32_t sgx_bn_bwd_finalize(const float* partials, int32_t nblk, int64_t M, int32_t C, const float* gamma,
                            const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                            float* coef, void* ws, int64_t ws_bytes, void* stream);
/* stage 3: dx from the stage-2 coefficients with g recomputed as in stage 1.  Optionally also writes g itself
 * (g_out != NULL) for consumers that need the masked upstream gradient.                            */
int32_t sgx_bn_bwd_apply(const float* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* scale,
                         const float* shift, const float* coef, float* dx, int64_t dx_ld, float* g_out, int64_t g_ld,
                         int64_t M, int32_t C, int32_t act, void* stream);
/* QARepVGG block, training form, on sgx_conv2d_fwd_dual's outputs (y = conv3x3(x), u = conv1x1(x; alpha*W1 + I) + b1, five moment planes):
 * replaces, per block, F.batch_norm x2 + the two branch adds + the activation (modules/qarepvgg_block.py:184-204) and their backward.
 *   sgx_qarep_fwd_finalize  both BatchNorms' statistics from the moments (s = bn3(y) + u is affine in (y, u) per channel; fp64): running
 *                           statistics updated like nn.BatchNorm2d; cf[4][C] = operand rows of the forward sweep
 *                           out = act(cf0*y + cf1 + cf2*u + cf3)  (sgx_dual_affine_act_fwd(y, cf0, cf1, u, cf2, cf3));
 *                           sv[8][C] = mean3, invstd3, scale3, shift3, mean_s, invstd_p, scale_p, shift_p for the backward.
 *                           ws: sgx_qarep_workspace(nblk, C) bytes (also covers the backward finalize).
 *   sgx_qarep_bwd_reduce    ONE sweep over (dout, y, u): partials4[4][sgx_stats_blocks(M)][C] = sum g, g*(s-mean_s), g*(y-mean3),
 *                           (s-mean_s)*(y-mean3), g = dout * act'(pre-activation recomputed with the forward's roundings).
 *   sgx_qarep_bwd_finalize  d gamma / d beta of post_bn and d gamma of bn3 accumulated in place (d beta3 is analytically zero: a BatchNorm's
 *                           input gradient sums to zero per channel), cb[6][C] = coefficients of the apply sweep.
 *   sgx_qarep_bwd_apply     ONE sweep over (dout, y, u) writing ds (gradient of u; may alias u) and dy (gradient of y; may alias y).     */
int64_t sgx_qarep_workspace(int32_t nblk, int32_t C);
int32_t sgx_qarep_fwd_finalize(const float* stat5, int32_t nblk, int64_t M, int32_t C, const float* bias1, const float* gamma3, const float* beta3,
                               float eps3, float mom3, float* rmean3, float* rvar3, const float* gammap, const float* betap, float epsp, float momp,
                               float* rmeanp, float* rvarp, float* cf, float* sv, void* ws, int64_t ws_bytes, void* stream);
int32_t sgx_qarep_bwd_reduce(const float* dout, int64_t d_ld, const float* y, int64_t y_ld, const float* u, int64_t u_ld, const float* cf,
                             const float* sv, int64_t M, int32_t C, int32_t act, float* partials4, void* stream);
int32_t sgx_qarep_bwd_finalize(const float* partials4, int32_t nblk, int64_t M, int32_t C, const float* gamma3, const float* gammap, const float* sv,
                               float* dgamma3, float* dgammap, float* dbetap, float* cb, void* ws, int64_t ws_bytes, void* stream);
int32_t sgx_qarep_bwd_apply(const float* dout, int64_t d_ld, const float* y, int64_t y_ld, const float* u, int64_t u_ld, const float* cf,
                            const float* sv, const float* cb, float* ds, int64_t ds_ld, float* dy, int64_t dy_ld, int64_t M, int32_t C, int32_t act,
                            void* stream);
/* Default ON (0 restores the two-launch form): the finalize stages above (sgx_bn_finalize, sgx_bn_bwd_finalize, sgx_bn_reduce_sums,
 * sgx_colsum, sgx_qarep_*_finalize) as ONE cooperative launch (32 channels x 16 row lanes per workgroup fold the fp32 partial rows in fp64, fixed order) instead
 * of a pre-reduction launch + a finalize launch, for up to 4096 partial rows.  Same sums up to fp64 regrouping.  Not thread-safe.   */
int32_t sgx_bn_set_fused_finalize(int32_t on);
int32_t sgx_bn_get_fused_finalize(void);
/* z = a*x + y with a device-resident scalar a (yolo_stages.py:61-63) and its backward pieces:
 * sgx_dot: out[0] (+)= scale * sum(a*b) over two [M,C] views (d a = <x, dz>): per-lane error-free accumulation (those sums cancel ~1e3x),
 * fp64 folds, two launches, deterministic.  ws: sgx_dot_workspace(M, C) bytes.                                                        */
int64_t sgx_dot_workspace(int64_t M, int32_t C);
int32_t sgx_dot(const float* a, int64_t a_ld, const float* b, int64_t b_ld, int64_t M, int32_t C, float scale, float* out,
                int32_t accumulate, void* ws, int64_t ws_bytes, void* stream);
/* y = a*x (+ y if accumulate) elementwise over [M,C] with strides; a_dev overrides a if not NULL.   */
int32_t sgx_axpy(const float* x, int64_t x_ld, float a, const float* a_dev, float* y, int64_t y_ld, int64_t M,
                 int32_t C, int32_t accumulate, void* stream);
/* g = dy where y > 0, else 0: backward of a ReLU applied after a residual add (classification_models/resnet.py:43-50,72-84). */
int32_t sgx_relu_bwd(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, float* g, int64_t g_ld, int64_t M, int32_t C,
                     void* stream);
/* The same g AND the BatchNorm-backward reduce of the layer under the add (out = relu(bn_n(conv_n) + shortcut), resnet.py:72-84) in one pass:
 * partials [2][sgx_stats_blocks(M)][C] = per row block sum g, sum g * (x - save_mean) - what sgx_bn_bwd_reduce(g, x, act = none) would
 * leave, bit for bit (same row blocks, same order), so sgx_bn_bwd_finalize takes them as they are.  x: the saved conv output of bn_n. */
int32_t sgx_relu_bwd_bn_reduce(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, const float* x, int64_t x_ld,
                               const float* save_mean, float* g, int64_t g_ld, int64_t M, int32_t C, float* partials, void* stream);
/* RepVGGBlock training forward (modules/repvgg_block.py:98-107: act(bn3(conv3x3 x) + bn1(conv1x1 x))) and the post-activation
 * residuals around it (csp_resnet.py:43-49 `x + y`, pp_yolo_head.py:205 `stem_cls(feat) + feat`) as ONE sweep:
 *   y = act(s1[c]*x1 + t1[c] [+ s2[c]*x2 + t2[c]]) [+ r_scale * r_scale_dev[0] * r]        x2 / r / r_scale_dev may be NULL.
 * The scaled form is the YOLO-NAS bottleneck's `alpha * x + cv2(cv1(x))` (yolo_stages.py:61-63) written by cv2's own sweep.            */
int32_t sgx_dual_affine_act_fwd(const float* x1, int64_t x1_ld, const float* s1, const float* t1, const float* x2, int64_t x2_ld,
                                const float* s2, const float* t2, const float* r, int64_t r_ld, float r_scale, const float* r_scale_dev,
                                float* y, int64_t y_ld, int64_t M, int32_t C, int32_t act, void* stream);
/* its backward through the activation: g = dy * act'(s1*x1 + t1 [+ s2*x2 + t2]) - the upstream gradient both BatchNorm backward
 * passes (sgx_bn_bwd_*, act = none) then consume.                                                                            */
int32_t sgx_dual_affine_act_bwd(const float* dy, int64_t dy_ld, const float* x1, int64_t x1_ld, const float* s1, const float* t1,
                                const float* x2, int64_t x2_ld, const float* s2, const float* t2, float* g, int64_t g_ld, int64_t M,
                                int32_t C, int32_t act, void* stream);
/* the same sweep also leaving the reduce rows of BOTH BatchNorm backward passes (round 5): partials4 = [4][sgx_stats_blocks(M)][C] =
 * sum g, sum g (x1 - mean1), sum g, sum g (x2 - mean2) - rows 0-1 are what sgx_bn_bwd_reduce(g, x1) would produce, rows 2-3 what
 * sgx_bn_bwd_reduce(g, x2) would (act = none): two passes over g and the saved conv outputs less per RepVGG block.                   */
int32_t sgx_dual_affine_act_bwd_reduce(const float* dy, int64_t dy_ld, const float* x1, int64_t x1_ld, const float* s1, const float* t1,
                                       const float* mean1, const float* x2, int64_t x2_ld, const float* s2, const float* t2, const float* mean2,
                                       float* g, int64_t g_ld, int64_t M, int32_t C, int32_t act, float* partials4, void* stream);
/* per-channel column sum: out[c] (+)= sum_rows x[row][c]  (conv bias gradients).                     */
/* rows_per_img/ld_img: rows are grouped in images of rows_per_img rows, image i starts at x + i*ld_img
 * (pass rows_per_img = M, ld_img = 0 for a plain [M,C] matrix).  ws: sgx_colsum_workspace(M, C) bytes.  */
int64_t sgx_colsum_workspace(int64_t M, int32_t C);
int32_t sgx_colsum(const float* x, int64_t ld, int64_t M, int32_t C, int64_t rows_per_img, int64_t ld_img, float* out,
                   int32_t accumulate, float* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pooling.  SPP max-pool k in {5,9,13} stride 1 (detection_models/csp_darknet53.py:146-150),
 * ResNet stem max-pool 3x3 s2 p1 and global average pool (classification_models/resnet.py:164,205).
 * ------------------------------------------------------------------------------------------- */
/* argmax (optional, int32 [N,Ho,Wo,C] contiguous) = flat input pixel index h*W+w of the first maximum in
 * row-major window order (ATen's tie rule); it is what the backward consumes.                        */
int32_t sgx_maxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                        const float* x, int64_t x_ld_pix, int64_t x_ld_img, float* y, int64_t y_ld_pix,
                        int64_t y_ld_img, int32_t* argmax, void* stream);
/* dx (+)= sum of dy over the windows whose arg-max is this pixel (gather form: deterministic, no atomics). */
int32_t sgx_maxpool_bwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad,
                        const int32_t* argmax, const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img, float* dx,
                        int64_t dx_ld_pix, int64_t dx_ld_img, int32_t accumulate, void* stream);
int32_t sgx_avgpool_fwd(int32_t N, int32_t HW, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img,
                        float* y, void* stream);
int32_t sgx_avgpool_bwd(int32_t N, int32_t HW, int32_t C, const float* dy, float* dx, int64_t dx_ld_pix,
                        int64_t dx_ld_img, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Squeeze-excitation gates and nearest up-sampling of PP-YOLOE:
 *   EffectiveSEBlock.forward  modules/se_blocks.py:39-42     x * hardsigmoid(project(mean_hw(x)))
 *   ESEAttn.forward           pp_yolo_e/pp_yolo_head.py:90-92  conv(feat * sigmoid(fc(avg_feat)))
 *   F.interpolate(scale_factor=2, mode="nearest")  pp_yolo_e/pan.py:170
 * ------------------------------------------------------------------------------------------- */
#define SGX_GATE_NONE 0        /* f(p) = p (plain per-image channel scale) */
#define SGX_GATE_HARDSIGMOID 1 /* f(p) = min(max(p/6 + 1/2, 0), 1); f' = 1/6 on -3 < p < 3 (ATen) */
#define SGX_GATE_SIGMOID 2
/* Per-image column reduction, deterministic two-stage (fp32 chunk partials, fp64 finalize in fixed order):
 *   out[n][c] = scale * f'(pre[n][c]) * sum_p u[n][p][c] * v[n][p][c]      v NULL -> 1 ; pre NULL -> no f' factor.
 * mean over H*W: u = x, scale = 1/HW.  Gate gradient: u = dy, v = x, pre = the gate's pre-activation.                          */
int64_t sgx_image_colsum_workspace(int32_t N, int32_t HW, int32_t C);
int32_t sgx_image_colsum(int32_t N, int32_t HW, int32_t C, const float* u, int64_t u_ld_pix, int64_t u_ld_img, const float* v,
                         int64_t v_ld_pix, int64_t v_ld_img, float scale, const float* pre, int32_t gate, float* out, void* ws,
                         int64_t ws_bytes, void* stream);
/* y[n][p][c] (+)= x[n][p][c] * f(pre[n][c]) + bias_scale * bias[n][c]     bias NULL -> 0.  Forward of both gates; with
 * x = dy and bias = d(mean) it is also their backward to the gated tensor.  In-place (y == x) is allowed when !accumulate.      */
int32_t sgx_channel_gate(int32_t N, int32_t HW, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img, const float* pre,
                         int32_t gate, const float* bias, float bias_scale, float* y, int64_t y_ld_pix, int64_t y_ld_img,
                         int32_t accumulate, void* stream);
/* nearest-neighbour x2: y[n][2h+i][2w+j][c] = x[n][h][w][c]; backward dx (+)= the sum of the four.                                */
int32_t sgx_upsample2x_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* x, int64_t x_ld_pix, int64_t x_ld_img, float* y,
                           int64_t y_ld_pix, int64_t y_ld_img, void* stream);
int32_t sgx_upsample2x_bwd(int32_t N, int32_t H, int32_t W, int32_t C, const float* dy, int64_t dy_ld_pix, int64_t dy_ld_img,
                           float* dx, int64_t dx_ld_pix, int64_t dx_ld_img, int32_t accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Detection head decode, PPYoloELoss (assigner + VFL/GIoU/DFL with hand-written backward), NMS.
 * ------------------------------------------------------------------------------------------- */
/* dfl_heads.py:207-235 + bbox_utils.py:9-29: boxes[B,L,4] = dist2bbox(softmax(distri)·[0..R]) * stride,
 * scores = sigmoid(logits).  points are in grid units (anchor_points / stride).                    */
int32_t sgx_dfl_decode(int32_t B, int32_t L, int32_t C, int32_t reg_max, const float* logits, const float* distri,
                       const float* points_grid, const float* strides, float* boxes, float* scores, void* stream);

/* training/losses/ppyolo_loss.py:726-775: flat targets [T,6] -> padded per-image lists.
 * gt_count[B] (int32), gt_index[B][nmax] (row index into targets or -1).  Built on device, no host sync;
 * nmax is chosen by the caller (>= max boxes per image; rows beyond it are dropped and counted in
 * overflow[0]).                                                                                    */
int32_t sgx_targets_index(const float* targets, int32_t T, int32_t B, int32_t nmax, int32_t* gt_count,
                          int32_t* gt_index, int32_t* overflow, void* stream);

typedef struct sgx_loss_desc {
    int32_t B, L, C, reg_max;   /* batch, anchors, classes, DFL bins-1 (16)                      */
    int32_t nmax;               /* padded GT slots per image                                      */
    int32_t use_static_assigner;/* 1 = ATSS (ppyolo_loss.py:258-434), 0 = TAL (:437-561)         */
    int32_t use_varifocal;      /* 1 = varifocal (:1079-1084), 0 = focal (:1069-1077)            */
    int32_t num_levels;         /* ATSS: pyramid levels                                           */
    int32_t level_count[8];     /* ATSS: anchors per level                                        */
    float w_cls, w_iou, w_dfl;  /* 1.0, 2.5, 0.5                                                  */
    int32_t sequential_assignment; /* 1 = use_batched_assignment=False semantics (ppyolo_loss.py:854-942): the
                                    * assigners run with pad_gt_mask=None, so TAL drops a GT whose best candidate metric
                                    * is <= 1e-9 (gather_topk_anchors :224-226) and no GT row is masked by sum(coords)>0 */
} sgx_loss_desc;

int64_t sgx_ppyoloe_loss_workspace(const sgx_loss_desc* d);
/* Forward: assignment (no grad) + the four sums.  Outputs:
 *   sums[4]      = cls_sum, iou_sum, dfl_sum, assigned_scores_sum          (ppyolo_loss.py:834-852)
 *   assigned_label[B,L] int32 (bg = C), assigned_box[B,L,4] (pixels), assigned_score[B,L]
 *   g_logits[B,L,C], g_distri[B,L,4*(R+1)]: d(w_cls*cls_sum)/dlogits and d(w_iou*iou_sum+w_dfl*dfl_sum)/ddistri
 * The caller finishes  loss_k = w_k*sum_k / max(score_sum_allreduced,1)  and scales the stored
 * gradients by upstream/max(...) in sgx_scale_by_device_scalar (no host sync anywhere).            */
int32_t sgx_ppyoloe_loss_fwd(const sgx_loss_desc* d, const float* logits, const float* distri, const float* anchors,
                             const float* points, const float* strides, const float* targets, const int32_t* gt_count,
                             const int32_t* gt_index, float* sums, int32_t* assigned_label, float* assigned_box,
                             float* assigned_score, float* g_logits, float* g_distri, void* ws, int64_t ws_bytes,
                             void* stream);
/* items[4] = (w_cls*cls, w_iou*iou, w_dfl*dfl)/max(score_sum,1) and their sum; inv_norm[0] = 1/max(..). */
int32_t sgx_ppyoloe_loss_finalize(const float* sums, float w_cls, float w_iou, float w_dfl, float score_div,
                                  float* items, float* inv_norm, void* stream);
/* y[i] = x[i] * s[0] * t[0]  (t may be NULL)                                                       */
int32_t sgx_scale_by_device_scalar(const float* x, const float* s, const float* t, float* y, int64_t n, void* stream);

/* pp_yolo_e/post_prediction_callback.py:42-123 + torchvision.ops.nms/batched_nms (requirements.txt:12).
 * Per image: score filter -> top-k (score desc, candidate index asc) -> greedy NMS (IoU > thr suppresses)
 * -> at most max_predictions rows [x1,y1,x2,y2,score,class].  out [B][max_predictions][6], out_count[B],
 * out_index[B][max_predictions] = candidate index (anchor*C + class for multi-label, anchor otherwise).
 * class_mode: 0 = class-agnostic (nms), 1 = per-class via coordinate offsets (batched_nms, numel<=4000),
 *             2 = per-class exact (batched_nms vanilla loop),
 *             3 = batched_nms as torchvision dispatches it on CPU: mode 1 while 4*candidates <= 4000, else mode 2.  */
typedef struct sgx_nms_desc {
    int32_t B, L, C;
    int32_t multi_label, class_mode;
    int32_t nms_top_k, max_predictions;
    float score_threshold, iou_threshold;
} sgx_nms_desc;
/* Measurement aid: 0 keeps the suppression stage inside the per-image kernel (the round-2 form); default 1 = the bit matrix is built by
 * a chip-wide launch and walked by one wave per image (top-k <= 1024, with a workspace).  Same rows either way.                       */
int32_t sgx_debug_set_nms_split(int32_t on);
/* Candidate selection of the multi-label path: 1 (default) = one pass over the scores behind a threshold estimated from a 1/32 sample (exact:
 * stage 2 falls back to streaming an image whose list came out short or overflowed), 0 = the exact three-pass histogram selection. */
int32_t sgx_debug_set_nms_selection(int32_t sampled);
/* measurement: bytes of dynamic LDS (0 .. 32768) added to every implicit-GEMM launch - an occupancy cap without another kernel build */
int32_t sgx_debug_set_igemm_lds_pad(int32_t bytes);
/* Where a multi-label sgx_nms call with a workspace records, per image, that stage 2 fell back from stage 1's candidate list to streaming
 * the image's raw scores (exact rows either way, many times slower): int32 index offset_ints + b * stride_ints of the call's workspace
 * holds 1 / 0 for image b once the call's stream work is complete.  Benches and tests assert it stays 0 on their inputs.     */
int32_t sgx_debug_nms_fallback_slot(const sgx_nms_desc* d, int64_t* offset_ints, int32_t* stride_ints);
int64_t sgx_nms_workspace(const sgx_nms_desc* d);
int32_t sgx_nms(const sgx_nms_desc* d, const float* boxes, const float* scores, float* out, int32_t* out_count,
                int32_t* out_index, int32_t* num_candidates, void* ws, int64_t ws_bytes, void* stream);

/* Validation metrics: match NMS rows to ground truth per image and IoU threshold (training/utils/detection_utils.py:1120-1290,
 * IoUMatching :880-1005, get_top_k_idx_per_cls :1342-1358).  preds [B][P][6] = x1,y1,x2,y2,score,class with pred_count[B] valid rows
 * (the layout sgx_nms writes); targets / crowd targets flat [T,6] = (img, class, cx, cy, w, h) indexed per image as by
 * sgx_targets_index; thresholds[nthr].  Outputs uint8 [B][P][nthr]: matched (true positive) and ignore (outside the per-class
 * top_k, or matched to a crowd target).                                                                                     */
typedef struct sgx_match_desc {
    int32_t B, P, nthr, top_k;
    int32_t H, W, denormalize;  /* image size for clipping; denormalize: targets are in [0,1] and get multiplied by W / H */
    int32_t nmax, cmax;         /* padded target / crowd-target slots per image                                             */
} sgx_match_desc;
int32_t sgx_detection_match(const sgx_match_desc* d, const float* preds, const int32_t* pred_count, const float* targets,
                            const int32_t* gt_count, const int32_t* gt_index, const float* crowd, const int32_t* crowd_count,
                            const int32_t* crowd_index, const float* thresholds, uint8_t* matched, uint8_t* ignore, void* stream);

/* predict(): inverse box maps of the image processing + packing of a batch's detections for ONE device-to-host copy.  The reference maps
 * each image's boxes back through its processing stages on the host (training/processing/processing.py:361-364 shift by the padding,
 * :401-403 multiply by 1 / scale factor; training/pipelines/pipelines.py:222-247 per image).  rows [B][P][6] / counts [B]: the layout
 * sgx_nms writes; steps [B][nsteps][3] = (kind, a_x, a_y) applied in order to (x1, x2) / (y1, y2) in fp32, one rounding per step as
 * numpy's float32 arithmetic: kind 0 += a, kind 1 *= a, kind 2 nothing; out: B * P * 6 floats (rows beyond an image's count zeroed)
 * followed by the B clamped counts as int32 bit patterns.                                                                          */
int32_t sgx_detection_unmap(const float* rows, const int32_t* counts, int32_t B, int32_t P, const float* steps, int32_t nsteps, float* out,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * Half-precision INFERENCE (csrc/half.hip): what `predict(fp16=True)` runs on the fused deployment form of the model
 * (training/pipelines/pipelines.py:76,223,375 wraps the reference's forward in torch.autocast; the fused form is
 * modules/qarepvgg_block.py:255-321 + conv/BatchNorm folding).  Activations are NHWC **bf16** (void* = bf16 elements; strides in
 * elements; channel counts, strides and addresses multiples of 8 elements = 16 bytes), filters OHWI bf16, bias fp32, accumulation
 * fp32 on v_mfma_f32_32x32x16_bf16.  Training never calls these.
 * ------------------------------------------------------------------------------------------- */
/* y = act(conv(x, w) + bias) [+ post_scale * (*post_scale_dev) * post_add, AFTER the activation]; y is bf16, or fp32 when y_is_f32 (the
 * prediction convs, whose outputs feed the fp32 decode / NMS kernels); post_add: bf16, y's logical shape, its own strides (the YOLO-NAS
 * bottleneck's shortcut, yolo_stages.py:61-63); bias / post_add / post_scale_dev may be NULL.  d: the fp32 path's descriptor, strides in
 * the operands' own elements.                                                                                                      */
int32_t sgx_hconv2d_fwd(const sgx_conv_desc* d, const void* x, const void* w, const float* bias, void* y, int32_t y_is_f32, int32_t act,
                        const void* post_add, int64_t post_ld_pix, int64_t post_ld_img, float post_scale, const float* post_scale_dev,
                        void* stream);
/* ConvTranspose2d(kernel 2, stride 2) + bias on bf16 (modules/sampling.py:72-73): w4 = [2][2][K][C] bf16, parity-major          */
int32_t sgx_hconvT2x2_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t K, const void* x, int64_t x_ld_pix, int64_t x_ld_img,
                          const void* w4, const float* bias, void* y, int64_t y_ld_pix, int64_t y_ld_img, void* stream);
/* F.max_pool2d on bf16 (the SPP of csp_darknet53.py:136-157; no arg-max: inference only)                                        */
int32_t sgx_hmaxpool_fwd(int32_t N, int32_t H, int32_t W, int32_t C, int32_t k, int32_t stride, int32_t pad, const void* x, int64_t x_ld_pix,
                         int64_t x_ld_img, void* y, int64_t y_ld_pix, int64_t y_ld_img, void* stream);
/* M rows of C bf16 elements from one row-strided view into another (a skip tensor into its concat slice)                        */
int32_t sgx_hcopy(const void* x, int64_t x_ld, int64_t M, int32_t C, void* y, int64_t y_ld, void* stream);
/* PP-YOLOE's deployment form on bf16 (round 6): per-image channel means (adaptive_avg_pool2d of EffectiveSEBlock modules/se_blocks.py:39-42 and
 * of the head, pp_yolo_head.py:203) as fp32 sums of bf16 activations in a fixed order; the channel gate x * f(pre[n][c]) (f: SGX_GATE_*) with
 * the product in fp32 and one rounding to bf16; nearest x2 up-sampling (pp_yolo_e/pan.py:170) into a (slice of a) bf16 tensor.  Channel
 * counts, strides (elements) and addresses: multiples of 8 elements.                                                                   */
int32_t sgx_himage_colsum(int32_t N, int32_t HW, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, float scale, float* out, void* stream);
int32_t sgx_hchannel_gate(int32_t N, int32_t HW, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, const float* pre, int32_t gate, void* y,
                          int64_t y_ld_pix, int64_t y_ld_img, void* stream);
int32_t sgx_hupsample2x_fwd(int32_t N, int32_t H, int32_t W, int32_t C, const void* x, int64_t x_ld_pix, int64_t x_ld_img, void* y, int64_t y_ld_pix,
                            int64_t y_ld_img, void* stream);
/* fp32 rows [M][Cs] -> bf16 rows [M][Cd], Cd >= Cs, extra channels zero, round-to-nearest-even (the image batch at the entrance)  */
int32_t sgx_cast_f32_bf16(const float* x, int64_t x_ld, int64_t M, int32_t Cs, void* y, int64_t y_ld, int32_t Cd, void* stream);
/* Measurement aid: force the tile / slab depth of sgx_hconv2d_fwd (0 = heuristic)                                               */
int32_t sgx_hconv_debug_set_tile(int32_t bm, int32_t bn, int32_t kd);

/* ---------------------------------------------------------------------------------------------
 * Classification loss (training/losses/label_smoothing_cross_entropy_loss.py:32-111).
 * ------------------------------------------------------------------------------------------- */
/* F.cross_entropy semantics (per-class weight, ignore_index, "mean" = sum w[y] * nll / sum w[y] over the rows that are not ignored) and,
 * with smoothing > 0, the reference's smoothed form (cross_entropy :32-83: weight multiplies the log-softmax, ignored rows - ignore_index
 * >= 0 - contribute 0, "mean" divides by the number of rows that are not ignored).  weight may be NULL; reduction_sum = 1 skips the division.
 * loss: 2*B + 2 floats; loss[0] = the loss, loss[1] = 1 / denominator.  dlogits = d(numerator) / d logits: multiply by loss[1] and the
 * upstream gradient (sgx_scale_by_device_scalar).                                                                                   */
int32_t sgx_softmax_ce_fwd_bwd(int32_t B, int32_t K, const float* logits, const int64_t* labels, float smoothing, const float* weight,
                               int32_t ignore_index, int32_t reduction_sum, float* loss, float* dlogits, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Optimizer / EMA over flat fp32 arenas (one launch per arena instead of ~500 tiny ATen kernels):
 * sg_trainer.py:639-644, training/utils/ema.py:126-141, optimizer_utils.py:32-59.
 * seg_end[nseg] (int64, ascending, device) splits the arena into segments with per-segment weight
 * decay seg_wd[nseg] (zero-WD groups for BN/bias).                                                  */
int32_t sgx_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, int32_t step, const int64_t* seg_end, const float* seg_wd, int32_t nseg,
                       const float* grad_scale, void* stream);
int32_t sgx_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, float dampening,
                     int32_t nesterov, int32_t first_step, const int64_t* seg_end, const float* seg_wd, int32_t nseg,
                     void* stream);
int32_t sgx_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream);
int32_t sgx_fill(float* p, int64_t n, float v, void* stream);

#ifdef __cplusplus
}
#endif
#endif
