"""Tensor-level wrappers over the C ABI (include/sgx_hip.h): shape/stride marshalling only, no math.

Tensor convention on the hot path: activations are torch tensors of logical shape [N, H, W, C]
(NHWC), last dim contiguous, possibly a channel slice of a wider buffer (stride(2) = ld_pix >= C).
Everything here enqueues on the current torch stream and returns immediately.
"""
import ctypes
import functools
import os
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import ConvDesc, LossDesc, MatchDesc, NmsDesc, check, lib, ptr, stream

ACT = {None: 0, "none": 0, "relu": 1, "silu": 2}


# --------------------------------------------------------------------------------------------- workspace
class _Workspace:
    """One grow-only scratch buffer per (device, stream).  Kernels that share one are ordered on that stream; work forked
    onto a side stream (weight gradients, modules/engine.py) gets its own."""

    def __init__(self):
        self.buf = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = (device, stream() or 0)  # (host emulation: stream() is None)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


WORKSPACE = _Workspace()


# --------------------------------------------------------------------------------------------- layout helpers
def nhwc_strides(t: torch.Tensor) -> Tuple[int, int]:
    shp, st = t.shape, t.stride()  # one call each: this runs ~1300 times per train step
    if len(shp) != 4 or (shp[3] > 1 and st[3] != 1):
        raise _lib.SgxError(f"tensor is not an NHWC view with contiguous channels: shape {tuple(shp)} strides {st}")
    n, h, w, c = shp
    # strides of size-1 dims carry no information (torch leaves arbitrary values there, e.g. after permute on a 1x1 map): derive the
    # pixel stride from the first dimension that has one
    if w > 1:
        ld_pix = st[2]
        if h > 1 and st[1] != w * ld_pix:
            raise _lib.SgxError(f"tensor is not an NHWC view with contiguous channels: shape {tuple(shp)} strides {st}")
    elif h > 1:
        ld_pix = st[1]
    else:  # a single pixel per image: keep a plausible recorded stride (a channel slice of a wider one-pixel buffer), else fall back to C
        ld_pix = st[2] if st[2] >= c else (st[1] if st[1] >= c else c)
    return ld_pix, (st[0] if n > 1 else h * w * ld_pix)


def rows(t: torch.Tensor) -> Tuple[int, int]:
    """(M, ld) of an NHWC view whose pixels are uniformly strided across images (sweep kernels)."""
    ld_pix, ld_img = nhwc_strides(t)
    n, h, w, _ = t.shape
    if n > 1 and ld_img != h * w * ld_pix:
        raise _lib.SgxError("tensor rows are not uniformly strided across images")
    return n * h * w, ld_pix


_DESC_CACHE = {}


def clear_desc_cache():
    """Forget cached descriptors and their size queries: call after changing anything those queries depend on (sgx_debug_set_tiles /
    sgx_debug_set_variant / sgx_conv_tuning_load - the measurement tools do).  A stale size can only fail loudly: the C side checks every
    workspace against its own requirement."""
    _DESC_CACHE.clear()


def clear_caches():
    """Everything this module remembers about the bound library (descriptors, size queries): for code that re-binds the library (the tests'
    host emulation / no-op builds)."""
    clear_desc_cache()
    _TICKETS.clear()
    for f in (stats_blocks, _reduce_workspace, _qarep_workspace, _dot_workspace):
        f.cache_clear()


def conv_desc(x: torch.Tensor, K: int, R: int, S: int, stride: int, pad: int, y: Optional[torch.Tensor] = None) -> ConvDesc:
    """The sgx_conv_desc of (x [, y]) - built and validated once per distinct (shapes, strides, filter) and shared afterwards (a training
    loop presents the same few hundred problems every step; the C side only reads the descriptor during the call)."""
    key = (x.shape, x.stride(), K, R, S, stride, pad, None if y is None else (y.shape, y.stride()))
    d = _DESC_CACHE.get(key)
    if d is None:
        if len(_DESC_CACHE) > 65536:
            _DESC_CACHE.clear()
        d = _DESC_CACHE[key] = _conv_desc_build(x, K, R, S, stride, pad, y)
    return d


def _conv_desc_build(x, K, R, S, stride, pad, y):
    n, h, w, c = x.shape
    ho = (h + 2 * pad - R) // stride + 1
    wo = (w + 2 * pad - S) // stride + 1
    d = ConvDesc()
    d.N, d.H, d.W, d.C, d.K, d.R, d.S, d.stride, d.pad, d.Ho, d.Wo = n, h, w, c, K, R, S, stride, pad, ho, wo
    d.x_ld_pix, d.x_ld_img = nhwc_strides(x)
    if y is not None:
        if tuple(y.shape) != (n, ho, wo, K):
            raise _lib.SgxError(f"conv output shape {tuple(y.shape)} != {(n, ho, wo, K)}")
        d.y_ld_pix, d.y_ld_img = nhwc_strides(y)
    else:
        d.y_ld_pix, d.y_ld_img = K, ho * wo * K
    d.ref = ctypes.byref(d)  # (plain Python attributes of the ctypes object: the by-reference handle and lazily cached size queries)
    d.wgrad_ws = d.dgrad_ws = None
    return d


def conv_out_shape(x, K, R, S, stride, pad):
    n, h, w, _ = x.shape
    return (n, (h + 2 * pad - R) // stride + 1, (w + 2 * pad - S) // stride + 1, K)


def _chk_w(w: torch.Tensor, K, R, S, C):
    """weights are logical [K, C, R, S] with physical OHWI layout (strides (R*S*C, 1, S*C, C))."""
    if tuple(w.shape) != (K, C, R, S) or w.stride() != (R * S * C, 1, S * C, C):
        if not (tuple(w.shape) == (K, C, R, S) and R == 1 and S == 1 and w.stride(0) == C and w.stride(1) == 1):
            raise _lib.SgxError(f"weight must be logical [K,C,R,S] in OHWI memory order; got shape {tuple(w.shape)} strides {w.stride()}")


def ohwi_empty(K, C, R, S, device) -> torch.Tensor:
    return torch.empty(K, R, S, C, device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def to_ohwi(w: torch.Tensor) -> torch.Tensor:
    """logical [K,C,R,S] tensor (any strides) -> same logical tensor stored OHWI."""
    out = ohwi_empty(*[w.shape[i] for i in (0, 1, 2, 3)], device=w.device)
    out.copy_(w)
    return out


# --------------------------------------------------------------------------------------------- convolution
def conv2d_fwd(x, w, bias=None, addend=None, out=None, act=None, stride=1, pad=0, stat_partials=False, post_add=None, post_scale=None):
    """post_add / post_scale: out = act(...) + post_scale * post_add - half-precision inference only (the bf16 conv epilogue carries it)."""
    if x.dtype == torch.bfloat16:
        if addend is not None or stat_partials:
            raise _lib.SgxError("half-precision convolution: inference form only (no pre-activation addend, no BatchNorm statistics)")
        return hconv2d_fwd(x, w, bias=bias, out=out, act=act, stride=stride, pad=pad, post_add=post_add, post_scale=post_scale)
    if post_add is not None:
        raise _lib.SgxError("conv2d_fwd: post_add rides in the half-precision convolution's epilogue only")
    K, C, R, S = w.shape
    _chk_w(w, K, R, S, x.shape[3])
    if out is None:
        out = torch.empty(conv_out_shape(x, K, R, S, stride, pad), device=x.device, dtype=torch.float32)
    d = conv_desc(x, K, R, S, stride, pad, out)
    parts = None
    if stat_partials:
        nblk = lib().sgx_conv2d_fwd_stat_blocks(d.ref)  # (not cached: it follows the tile choice, and the kernel trusts the buffer's size)
        parts = torch.empty(2, nblk, K, device=x.device, dtype=torch.float32)
    if addend is not None and nhwc_strides(addend) != nhwc_strides(out):
        raise _lib.SgxError("conv addend must share the output's strides")
    check(lib().sgx_conv2d_fwd(d.ref, ptr(x), ptr(w), ptr(bias), ptr(addend), ptr(out), ACT[act], ptr(parts), stream()), "sgx_conv2d_fwd")
    return (out, parts) if stat_partials else out


def conv2d_bwd_data(dy, w, x_shape, stride=1, pad=0, addend=None, out=None, accumulate=False):
    K, C, R, S = w.shape
    n, h, wd, c = x_shape
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    d = conv_desc(out, K, R, S, stride, pad, dy)
    if addend is not None and nhwc_strides(addend) != nhwc_strides(out):
        raise _lib.SgxError("bwd_data addend must share dx's strides")
    if d.dgrad_ws is None:
        d.dgrad_ws = lib().sgx_conv2d_bwd_data_workspace(d.ref)
    ws = WORKSPACE.get(d.dgrad_ws, dy.device)
    check(lib().sgx_conv2d_bwd_data(d.ref, ptr(dy), ptr(w), ptr(addend), ptr(out), int(accumulate), ptr(ws), ws.numel(), stream()),
          "sgx_conv2d_bwd_data")
    return out


def conv2d_wt_buffer(w, device):
    """Persistent buffer for the pre-transposed weights of a conv (see conv2d_transpose_weights)."""
    K, C, R, S = w.shape
    return torch.empty(K * C * R * S + 64, device=device, dtype=torch.float32)


def _wt_desc(w, stride, pad):
    # the transpose depends only on (K, C, R, S, stride, pad); spatial sizes are placeholders that satisfy the descriptor checks
    K, C, R, S = w.shape
    h = max(R, stride) + 2
    x = torch.empty(0, device=w.device).new_empty((1, h, h, C))
    return conv_desc(x, K, R, S, stride, pad)


def conv2d_transpose_weights(w, wt, stride=1, pad=0):
    d = _wt_desc(w, stride, pad)
    check(lib().sgx_conv2d_transpose_weights(d.ref, ptr(w), ptr(wt), wt.numel() * 4, stream()), "sgx_conv2d_transpose_weights")


def conv2d_transpose_jobs(w, wt, stride=1, pad=0) -> bytes:
    """The transposes conv2d_transpose_weights(w, wt, stride, pad) would launch, as packed sgx_wtrans_job records (host bytes)."""
    d = _wt_desc(w, stride, pad)
    jobs = (_lib.WtransJob * 16)()
    n = ctypes.c_int32()
    check(lib().sgx_conv2d_transpose_jobs(d.ref, ptr(w), ptr(wt), wt.numel() * 4, jobs, 16, ctypes.byref(n)), "sgx_conv2d_transpose_jobs")
    return bytes(jobs)[: n.value * ctypes.sizeof(_lib.WtransJob)]


def wtrans_batch(jobs_dev, njobs):
    """Run a table of transposes (uint8 tensor on the device holding njobs sgx_wtrans_job records) as one launch."""
    check(lib().sgx_wtrans_batch(ptr(jobs_dev), int(njobs), stream()), "sgx_wtrans_batch")


DEFAULT_FILTER_PLANES = 1  # the library's default mode of sgx_debug_set_filter_planes (tests restore it)


def filter_planes_plan(filters):
    """filters: iterable of (fp32 filter tensor viewed [rows, taps, ch] - its data_ptr is what the conv launches will see).  Returns
    (records (src, byte offset, rows, taps, ch), total bytes) for the filters a bf16x3 launch can take planes of (ch a multiple of 16 - also the
    shallow 1x1 filters, which ride in the QARepVGG two-output / two-source launches whatever their depth); duplicates (one filter, several launches) are planned once."""
    seen, recs, total = set(), [], 0
    for src, rows, taps, ch in filters:
        if src in seen or ch % 16 or rows <= 0:
            continue
        n = int(lib().sgx_filter_planes_bytes(rows, taps, ch))
        if n <= 0 or n >= (1 << 30):
            continue
        seen.add(src)
        recs.append((src, total, rows, taps, ch))
        total += (n + 255) // 256 * 256
    return recs, total


def filter_planes_table(recs, planes_buf):
    """The sgx_fplanes_job records of a plan over the planes buffer (uint8 tensor): -> (host ctypes array, device uint8 tensor)."""
    jobs = (_lib.FplanesJob * len(recs))()
    base = planes_buf.data_ptr()
    for j, (src, off, rows, taps, ch) in zip(jobs, recs):
        j.src, j.planes, j.rows, j.taps, j.ch, j.pad_ = src, base + off, rows, taps, ch, 0
    dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(planes_buf.device)
    return jobs, dev


def filter_planes_batch(jobs_host, jobs_dev):
    """Split every planned filter into its bf16x3 planes (one launch) and mark the registry entries valid."""
    check(lib().sgx_filter_planes_batch(ctypes.cast(jobs_host, ctypes.c_void_p), ptr(jobs_dev), len(jobs_host), stream()), "sgx_filter_planes_batch")


def filter_planes_invalidate(jobs_host=None):
    if jobs_host is None:
        check(lib().sgx_filter_planes_invalidate(None, 0), "sgx_filter_planes_invalidate")
    else:
        check(lib().sgx_filter_planes_invalidate(ctypes.cast(jobs_host, ctypes.c_void_p), len(jobs_host)), "sgx_filter_planes_invalidate")


def filter_planes_scope(open_: bool):
    lib().sgx_filter_planes_scope(1 if open_ else 0)


class BnReduceRequest:
    """A BatchNorm layer's backward reduce, handed to the data gradient that finalises the layer's output gradient (sgx_bn_reduce_req):
    the layer's saved conv output t and its scale / shift / mean, and the channel range [c_lo, c_hi) of that data gradient's dx which IS the
    layer's dy.  After the launch `parts` holds the partial rows [2, rows, C] sgx_bn_bwd_reduce would have produced (None: the launch could
    not carry the request - the layer then runs its own reduce sweep)."""

    __slots__ = ("t", "scale", "shift", "mean", "act", "c_lo", "c_hi", "parts")

    def __init__(self, t, scale, shift, mean, act, c_lo=0, c_hi=None):
        self.t, self.scale, self.shift, self.mean, self.act = t, scale, shift, mean, act
        self.c_lo, self.c_hi = c_lo, (t.shape[3] + c_lo) if c_hi is None else c_hi
        self.parts = None

    def at(self, c_lo):
        """the same request, for a dx whose channel c_lo is this layer's channel 0 (a concat-slice gradient)"""
        self.c_lo, self.c_hi = c_lo, c_lo + self.t.shape[3]
        return self


BN_REQ_STATS = {"taken": 0, "declined": 0}  # requests carried by a data gradient / left to the layer's own sweep (tests, tools)


def _bn_reqs(reqs, d, two_source, out, addend=None):
    """-> (ctypes array, n) for the requests this launch can carry, partial rows allocated; (None, 0) when it cannot (requests keep parts = None)"""
    reqs = [r for r in (reqs or ()) if r is not None]
    if not reqs:
        return None, 0
    # the epilogue that carries requests is the 16-byte one: operands it cannot address that way decline here, before any launch
    if out.data_ptr() % 16 or (addend is not None and addend.data_ptr() % 16) or any(r.t.data_ptr() % 16 for r in reqs):
        BN_REQ_STATS["declined"] += len(reqs)
        return None, 0
    arr, n = _bn_reqs_build(reqs, d, two_source, out)
    BN_REQ_STATS["taken" if n else "declined"] += len(reqs)
    return arr, n


def _bn_reqs_build(reqs, d, two_source, out):
    if len(reqs) > 2:
        return None, 0
    n_, h_, w_, _ = out.shape
    for r in reqs:
        if tuple(r.t.shape[:3]) != (n_, h_, w_) or r.c_hi > out.shape[3] or r.c_lo % 4 or r.c_hi % 4:
            return None, 0
    rows = lib().sgx_conv2d_bwd_data_stat_blocks(d.ref, int(two_source))
    if rows <= 0:
        return None, 0
    arr = (_lib.BnReduceReq * len(reqs))()
    for a, r in zip(arr, reqs):
        r.parts = torch.empty(2, rows, r.c_hi - r.c_lo, device=out.device, dtype=torch.float32)
        tl, ti = nhwc_strides(r.t)
        a.c_lo, a.c_hi, a.t, a.t_ld_pix, a.t_ld_img = r.c_lo, r.c_hi, ptr(r.t), tl, ti
        a.scale, a.shift, a.mean, a.act, a.rows, a.partials = ptr(r.scale), ptr(r.shift), ptr(r.mean), ACT[r.act], rows, ptr(r.parts)
    return arr, len(reqs)


def conv2d_bwd_data_wt(dy, w, wt, x_shape, stride=1, pad=0, addend=None, out=None, accumulate=False, reqs=None):
    """conv2d_bwd_data with weights already transposed into `wt` by conv2d_transpose_weights (same stride / pad).
    reqs: BnReduceRequest objects of the layer(s) whose dy this launch finalises (<= 2 channel ranges of dx)."""
    K, C, R, S = w.shape
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    d = conv_desc(out, K, R, S, stride, pad, dy)
    if addend is not None and nhwc_strides(addend) != nhwc_strides(out):
        raise _lib.SgxError("bwd_data addend must share dx's strides")
    arr, n = _bn_reqs(reqs, d, False, out, addend)
    if n:
        check(lib().sgx_conv2d_bwd_data_wt_req(d.ref, ptr(dy), ptr(wt), ptr(addend), ptr(out), int(accumulate), arr, n, stream()), "sgx_conv2d_bwd_data_wt_req")
        return out
    check(lib().sgx_conv2d_bwd_data_wt(d.ref, ptr(dy), ptr(wt), ptr(addend), ptr(out), int(accumulate), stream()), "sgx_conv2d_bwd_data_wt")
    return out


# ---- half-precision inference (csrc/half.hip): bf16 activations, one bf16 MFMA product, fp32 accumulation ---------------------------------
HALF = torch.bfloat16
_HALF_W = {}
_WEIGHTS_GEN = [0]


def weights_written():
    """Every writer that changes parameters THROUGH RAW POINTERS (sgx_adamw_step / sgx_sgd_step / sgx_ema_update on the arenas, and
    SgxNetwork.weights_changed() for load_state_dict / EMA swaps / broadcasts) calls this: tensor `_version` counters do not see those writes,
    so the bf16 operands cached below are tied to this generation instead (a model that ran half-precision inference and was then
    trained further served the old bf16 prediction / transposed-conv filters)."""
    _WEIGHTS_GEN[0] += 1


def _half_cached(src, tag, build):
    """A bf16 operand derived from the fp32 tensor `src` (a folded filter, a transposed-conv weight, a LIVE arena view): valid for this
    tensor object, its in-place version AND the current weights generation; an entry dies with its source tensor (weakref.finalize), so
    the bf16 filters of a discarded fused copy leave HBM with it."""
    import weakref

    key = (id(src), tag)
    hit = _HALF_W.get(key)
    if hit is not None and hit[0]() is src and hit[1] == (src._version, _WEIGHTS_GEN[0]):
        return hit[2]
    h = build()
    if hit is None or hit[0]() is not src:
        weakref.finalize(src, _HALF_W.pop, key, None)
    _HALF_W[key] = (weakref.ref(src), (src._version, _WEIGHTS_GEN[0]), h)
    return h


def half_filter(w, cin):
    """logical [K,C,R,S] fp32 (any strides) -> contiguous [K,R,S,cin] bf16 (OHWI; channels zero-padded to cin, round-to-nearest-even)."""
    def build():
        K, C, R, S = w.shape
        o = torch.zeros(K, R, S, cin, device=w.device, dtype=HALF)
        o[..., :C] = w.detach().permute(0, 2, 3, 1)
        return o
    return _half_cached(w, ("f", cin), build)


def hconv2d_fwd(x, w, bias=None, out=None, act=None, stride=1, pad=0, post_add=None, post_scale=None):
    """bf16 NHWC x; w: the fp32 filter of the fp32 path (converted once, cached) -> out bf16 (default) or the fp32 `out` given."""
    K, C, R, S = w.shape
    cin = x.shape[3]
    if C > cin:
        raise _lib.SgxError(f"hconv2d_fwd: the filter has {C} input channels, the activation {cin}")
    wh = half_filter(w, cin)
    if out is None:
        out = torch.empty(conv_out_shape(x, K, R, S, stride, pad), device=x.device, dtype=HALF)
    elif out.dtype not in (HALF, torch.float32):
        raise _lib.SgxError(f"hconv2d_fwd: output dtype {out.dtype}")
    d = conv_desc(x, K, R, S, stride, pad, out)
    pl, pi = nhwc_strides(post_add) if post_add is not None else (0, 0)
    if post_add is not None and (post_add.dtype != HALF or tuple(post_add.shape) != tuple(out.shape)):
        raise _lib.SgxError("hconv2d_fwd: post_add must be a bf16 tensor of the output's shape")
    ps_dev = post_scale if torch.is_tensor(post_scale) else None
    ps = 1.0 if post_scale is None or ps_dev is not None else float(post_scale)
    check(lib().sgx_hconv2d_fwd(d.ref, ptr(x), ptr(wh), ptr(bias), ptr(out), int(out.dtype == torch.float32), ACT[act], ptr(post_add), pl, pi, ps,
                                ptr(ps_dev), stream()), "sgx_hconv2d_fwd")
    return out


def cast_bf16(x, cpad=None):
    """fp32 NHWC [N,H,W,C] (uniform rows) -> bf16 [N,H,W,cpad] (default: C rounded up to 8), extra channels zero."""
    n, h, w, c = x.shape
    cpad = cpad or ((c + 7) // 8) * 8
    M, ld = rows(x)
    y = torch.empty(n, h, w, cpad, device=x.device, dtype=HALF)
    check(lib().sgx_cast_f32_bf16(ptr(x), ld, M, c, ptr(y), cpad, cpad, stream()), "sgx_cast_f32_bf16")
    return y


def hcopy(x, out):
    M, ld = rows(x)
    check(lib().sgx_hcopy(ptr(x), ld, M, x.shape[3], ptr(out), rows(out)[1], stream()), "sgx_hcopy")
    return out


# ---- QARepVGG block: both convolution branches per launch (csrc/conv.hip, PH2 kernels) ---------------------------------------------
def conv2d_fwd_dual(x, w, w1p, bias1, stride=1):
    """y = conv RxS(x, w) (pad R // 2, no bias), u = conv1x1(x, w1p) + bias1 in ONE launch -> (y, u, stat5 [5, nblk, K])."""
    K, C, R, S = w.shape
    _chk_w(w, K, R, S, x.shape[3])
    _chk_w(w1p, K, 1, 1, x.shape[3])
    y = torch.empty(conv_out_shape(x, K, R, S, stride, R // 2), device=x.device, dtype=torch.float32)
    u = torch.empty_like(y)
    d = conv_desc(x, K, R, S, stride, R // 2, y)
    nblk = lib().sgx_conv2d_fwd_dual_stat_blocks(d.ref)
    stat5 = torch.empty(5, nblk, K, device=x.device, dtype=torch.float32)
    check(lib().sgx_conv2d_fwd_dual(d.ref, ptr(x), ptr(w), ptr(w1p), ptr(bias1), ptr(y), ptr(u), ptr(stat5), stream()), "sgx_conv2d_fwd_dual")
    return y, u, stat5


def conv2d_bwd_data_dual(dy, w, wt, ds, w1pt, x_shape, stride=1, addend=None, out=None, accumulate=False, addend2=None, addend2_scale=None, reqs=None):
    """dx = convT RxS(dy) + convT 1x1(ds) [+ addend] [+ addend2_scale * addend2] [+ dx] in one launch per parity class; wt:
    conv2d_transpose_weights(w), w1pt: [C, K]; addend2 has its own strides, its scale is a float or a one-element device tensor."""
    K, C, R, S = w.shape
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    d = conv_desc(out, K, R, S, stride, R // 2, dy)
    if addend is not None and nhwc_strides(addend) != nhwc_strides(out):
        raise _lib.SgxError("bwd_data addend must share dx's strides")
    sl, si = nhwc_strides(ds)
    a2l, a2i = nhwc_strides(addend2) if addend2 is not None else (0, 0)
    a2_dev = addend2_scale if torch.is_tensor(addend2_scale) else None
    a2s = 1.0 if addend2_scale is None or a2_dev is not None else float(addend2_scale)
    arr, n = _bn_reqs(reqs, d, True, out, addend)
    if n:
        check(lib().sgx_conv2d_bwd_data_dual_req(d.ref, ptr(dy), ptr(wt), ptr(ds), sl, si, ptr(w1pt), ptr(addend), ptr(addend2), a2l, a2i, a2s,
                                                 ptr(a2_dev), ptr(out), int(accumulate), arr, n, stream()), "sgx_conv2d_bwd_data_dual_req")
        return out
    check(lib().sgx_conv2d_bwd_data_dual(d.ref, ptr(dy), ptr(wt), ptr(ds), sl, si, ptr(w1pt), ptr(addend), ptr(addend2), a2l, a2i, a2s,
                                         ptr(a2_dev), ptr(out), int(accumulate), stream()), "sgx_conv2d_bwd_data_dual")
    return out


def qarep_prep_job(w1, w1p, w1pt, identity, alpha=None) -> bytes:
    """One sgx_qarep_prep_job record (host bytes): w1p = alpha * w1 + I, w1pt = its transpose; operands are arena views / persistent buffers."""
    K, C = w1.shape[0], w1.shape[1]
    j = _lib.QarepPrepJob()
    j.w1, j.w1p, j.w1pt, j.alpha = ptr(w1), ptr(w1p), ptr(w1pt), ptr(alpha)
    j.K, j.C, j.identity, j.pad_ = K, C, int(bool(identity)), 0
    return bytes(j)


def qarep_prep_batch(jobs_dev, njobs):
    check(lib().sgx_qarep_prep_batch(ptr(jobs_dev), int(njobs), stream()), "sgx_qarep_prep_batch")


def qarep_fwd_finalize(stat5, M, bias1, bn3, pbn):
    """-> (cf [4, C], sv [8, C]); bn3 / pbn: BatchNorm layers (weight, bias, eps, momentum, running stats updated in place)."""
    nblk, C = stat5.shape[1], stat5.shape[2]
    cf = torch.empty(4, C, device=stat5.device, dtype=torch.float32)
    sv = torch.empty(8, C, device=stat5.device, dtype=torch.float32)
    ws = WORKSPACE.get(_qarep_workspace(nblk, C), stat5.device)
    check(lib().sgx_qarep_fwd_finalize(ptr(stat5), nblk, M, C, ptr(bias1), ptr(bn3.weight), ptr(bn3.bias), bn3.eps, bn3.momentum, ptr(bn3.running_mean),
                                       ptr(bn3.running_var), ptr(pbn.weight), ptr(pbn.bias), pbn.eps, pbn.momentum, ptr(pbn.running_mean),
                                       ptr(pbn.running_var), ptr(cf), ptr(sv), ptr(ws), ws.numel(), stream()), "sgx_qarep_fwd_finalize")
    return cf, sv


def qarep_bwd(dout, y, u, cf, sv, bn3, pbn, act, chunks=1, after_chunk=None):
    """BatchNorm x2 + activation backward of the block in two sweeps: -> (ds written over u, dy written over y); d gamma / d beta accumulate.
    chunks > 1: the apply sweep runs as that many launches over consecutive runs of images and after_chunk(i, n0, n1) is called behind
    each - a caller that needs no data gradient (the stem) sends the weight gradients of images n0:n1 out while the next run is swept."""
    M, yl = rows(y)
    C = y.shape[3]
    dl, ul = rows(dout)[1], rows(u)[1]
    a = ACT[act]
    nblk = stats_blocks(M)
    parts = torch.empty(4, nblk, C, device=y.device, dtype=torch.float32)
    check(lib().sgx_qarep_bwd_reduce(ptr(dout), dl, ptr(y), yl, ptr(u), ul, ptr(cf), ptr(sv), M, C, a, ptr(parts), stream()), "sgx_qarep_bwd_reduce")
    cb = torch.empty(6, C, device=y.device, dtype=torch.float32)
    ws = WORKSPACE.get(_qarep_workspace(nblk, C), y.device)
    check(lib().sgx_qarep_bwd_finalize(ptr(parts), nblk, M, C, ptr(bn3.weight), ptr(pbn.weight), ptr(sv), ptr(bn3.weight.grad), ptr(pbn.weight.grad),
                                       ptr(pbn.bias.grad), ptr(cb), ptr(ws), ws.numel(), stream()), "sgx_qarep_bwd_finalize")
    N = y.shape[0]
    if chunks > 1 and N % chunks == 0 and M % N == 0:
        per = N // chunks
        for i in range(chunks):
            d_, y_, u_ = dout[i * per:(i + 1) * per], y[i * per:(i + 1) * per], u[i * per:(i + 1) * per]
            check(lib().sgx_qarep_bwd_apply(ptr(d_), dl, ptr(y_), yl, ptr(u_), ul, ptr(cf), ptr(sv), ptr(cb), ptr(u_), ul, ptr(y_), yl, M // chunks, C, a,
                                            stream()), "sgx_qarep_bwd_apply")
            if after_chunk is not None:
                after_chunk(i, i * per, (i + 1) * per)
        return u, y
    check(lib().sgx_qarep_bwd_apply(ptr(dout), dl, ptr(y), yl, ptr(u), ul, ptr(cf), ptr(sv), ptr(cb), ptr(u), ul, ptr(y), yl, M, C, a, stream()),
          "sgx_qarep_bwd_apply")
    if after_chunk is not None:
        after_chunk(0, 0, N)
    return u, y


def conv2d_bwd_weight(x, dy, dw, dbias=None, stride=1, pad=0):
    """dw (logical [K,C,R,S], OHWI memory) += grad; dbias += column sums."""
    K, C, R, S = dw.shape
    _chk_w(dw, K, R, S, x.shape[3])
    d = conv_desc(x, K, R, S, stride, pad, dy)
    if d.wgrad_ws is None:
        d.wgrad_ws = lib().sgx_conv2d_bwd_weight_workspace(d.ref)
    ws = WORKSPACE.get(d.wgrad_ws, x.device)
    check(lib().sgx_conv2d_bwd_weight(d.ref, ptr(x), ptr(dy), ptr(dw), ptr(dbias), ptr(ws), ws.numel(), stream()), "sgx_conv2d_bwd_weight")


_TICKETS = {}


def _ticket_buffer(n_ints: int, device) -> torch.Tensor:
    """The arrival-ticket buffer of the grouped weight gradient on the current stream: int32, zero when handed to a launch and left zero by
    it (include/sgx_hip.h: sgx_conv2d_bwd_weight_group), so it is cleared exactly once - when it is allocated."""
    key = (device, stream() or 0)
    b = _TICKETS.get(key)
    if b is None or b.numel() < n_ints:
        b = _TICKETS[key] = torch.zeros(max(2 * int(n_ints), 1 << 16), dtype=torch.int32, device=device)
    return b


def conv2d_bwd_weight_group(entries):
    """entries: [(x, dy, dw, stride, pad), ...] - every dw (logical [K,C,R,S], OHWI memory) += its weight gradient, as ONE launch per tile
    shape for the whole list (pixel splits sized for the group, partials folded inside the launch)."""
    n = len(entries)
    if n == 0:
        return
    jobs = (_lib.WgradJob * n)()
    for j, (x, dy, dw, stride, pad) in zip(jobs, entries):
        K, C, R, S = dw.shape
        _chk_w(dw, K, R, S, x.shape[3])
        j.d = conv_desc(x, K, R, S, stride, pad, dy)
        j.x, j.dy, j.dw = ptr(x), ptr(dy), ptr(dw)
    ws_bytes, t_ints = ctypes.c_int64(), ctypes.c_int64()
    check(lib().sgx_conv2d_bwd_weight_group_sizes(jobs, n, ctypes.byref(ws_bytes), ctypes.byref(t_ints)), "sgx_conv2d_bwd_weight_group_sizes")
    dev = entries[0][0].device
    ws = WORKSPACE.get(ws_bytes.value, dev)
    tk = _ticket_buffer(t_ints.value, dev)
    check(lib().sgx_conv2d_bwd_weight_group(jobs, n, ptr(ws), ws.numel(), ptr(tk), tk.numel(), stream()), "sgx_conv2d_bwd_weight_group")


def _chk_wt(wt, C, K):
    """ConvTranspose2d weight: logical [C, K, 2, 2], memory [C][2][2][K]."""
    if tuple(wt.shape) != (C, K, 2, 2) or wt.stride() != (4 * K, 1, 2 * K, K):
        raise _lib.SgxError(f"convT weight must be logical [C,K,2,2] stored [C][2][2][K]; got {tuple(wt.shape)} {wt.stride()}")


def convT_empty(C, K, device):
    return torch.empty(C, 2, 2, K, device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def convT2x2_fwd(x, wt, bias=None, out=None, wtt=None):
    """wtt: the filter in data-gradient order (conv2d_transpose_weights of the adjoint 2x2 stride-2 convolution - the network's per-step
    transpose batch keeps it current): the call then launches no transposes."""
    n, h, w, c = x.shape
    K = wt.shape[1]
    _chk_wt(wt, c, K)
    if x.dtype == HALF:  # half-precision inference: four 1x1 launches, one per output parity; w4 = [2][2][K][C] bf16, built once
        w4 = _half_cached(wt, "T", lambda: wt.detach().permute(2, 3, 1, 0).contiguous().to(HALF))
        if out is None:
            out = torch.empty(n, 2 * h, 2 * w, K, device=x.device, dtype=HALF)
        xl, xi = nhwc_strides(x)
        yl, yi = nhwc_strides(out)
        check(lib().sgx_hconvT2x2_fwd(n, h, w, c, K, ptr(x), xl, xi, ptr(w4), ptr(bias), ptr(out), yl, yi, stream()), "sgx_hconvT2x2_fwd")
        return out
    if out is None:
        out = torch.empty(n, 2 * h, 2 * w, K, device=x.device, dtype=torch.float32)
    xl, xi = nhwc_strides(x)
    yl, yi = nhwc_strides(out)
    if wtt is not None:
        check(lib().sgx_convT2x2_fwd_wt(n, h, w, c, K, ptr(x), xl, xi, ptr(wtt), ptr(bias), ptr(out), yl, yi, stream()), "sgx_convT2x2_fwd_wt")
        return out
    nbytes = lib().sgx_convT2x2_workspace(n, h, w, c, K)
    ws = WORKSPACE.get(nbytes, x.device)
    check(lib().sgx_convT2x2_fwd(n, h, w, c, K, ptr(x), xl, xi, ptr(wt), ptr(bias), ptr(out), yl, yi, ptr(ws), ws.numel(), stream()), "sgx_convT2x2_fwd")
    return out


def convT2x2_bwd_data(dy, wt, out=None):
    n, h2, w2, K = dy.shape
    h, w, c = h2 // 2, w2 // 2, wt.shape[0]
    if out is None:
        out = torch.empty(n, h, w, c, device=dy.device, dtype=torch.float32)
    dl, di = nhwc_strides(dy)
    xl, xi = nhwc_strides(out)
    check(lib().sgx_convT2x2_bwd_data(n, h, w, c, K, ptr(dy), dl, di, ptr(wt), ptr(out), xl, xi, stream()), "sgx_convT2x2_bwd_data")
    return out


def convT2x2_bwd_weight(x, dy, dwt, dbias=None):
    n, h, w, c = x.shape
    K = dwt.shape[1]
    _chk_wt(dwt, c, K)
    xl, xi = nhwc_strides(x)
    dl, di = nhwc_strides(dy)
    nbytes = lib().sgx_convT2x2_workspace(n, h, w, c, K)
    ws = WORKSPACE.get(nbytes, x.device)
    check(lib().sgx_convT2x2_bwd_weight(n, h, w, c, K, ptr(x), xl, xi, ptr(dy), dl, di, ptr(dwt), ptr(dbias), ptr(ws), ws.numel(), stream()),
          "sgx_convT2x2_bwd_weight")


def nchw_to_nhwc(x, cpad=None):
    n, c, h, w = x.shape
    cpad = cpad or ((c + 3) // 4) * 4
    x = x.contiguous()
    y = torch.empty(n, h, w, cpad, device=x.device, dtype=torch.float32)
    check(lib().sgx_nchw_to_nhwc(n, c, h, w, cpad, ptr(x), ptr(y), stream()), "sgx_nchw_to_nhwc")
    return y


def standardize_u8(x_u8, max_value=255.0, mean=None, std=None, cpad=None):
    """uint8 [N,H,W,C] (the dataset's HWC images, stacked) -> fp32 NHWC [N,H,W,Cpad]: (x / max_value - mean) / std, zero pad channels."""
    n, h, w, c = x_u8.shape
    if x_u8.dtype != torch.uint8:
        raise _lib.SgxError(f"standardize_u8 needs a uint8 batch, got {x_u8.dtype}")
    cpad = cpad or ((c + 3) // 4) * 4
    x_u8 = x_u8.contiguous()
    y = torch.empty(n, h, w, cpad, device=x_u8.device, dtype=torch.float32)
    check(lib().sgx_standardize_u8_hwc(n, h, w, c, cpad, ptr(x_u8), float(max_value), ptr(mean), ptr(std), ptr(y), stream()), "sgx_standardize_u8_hwc")
    return y


def pad_standardize_u8(img_u8, out_slot, top, left, pad_value, max_value=255.0, mean=None, std=None):
    """One uint8 [h,w,C] image into `out_slot` (a [H,W,Cpad] fp32 view of the padded batch) at (top, left); pad_value: fp32 [C] on the uint8 scale."""
    h, w, c = img_u8.shape
    H, W, cpad = out_slot.shape
    if img_u8.dtype != torch.uint8 or not out_slot.is_contiguous():
        raise _lib.SgxError("pad_standardize_u8 needs a uint8 HWC image and a contiguous [H,W,Cpad] slot")
    img_u8 = img_u8.contiguous()
    check(lib().sgx_pad_standardize_u8_hwc(h, w, c, ptr(img_u8), H, W, cpad, int(top), int(left), float(max_value), ptr(mean), ptr(std), ptr(pad_value),
                                           ptr(out_slot), stream()), "sgx_pad_standardize_u8_hwc")


def preprocess_u8(images, geometry, H, W, pad_value, reverse_channels=False, max_value=None, mean=None, std=None):
    """A batch of ragged uint8 HWC device images -> the fp32 NHWC batch [N, H, W, Cpad] in one launch (sgx_preprocess_u8_hwc).
    geometry[n] = (h, w, top, left): the size image n is rescaled to and where it sits in its H x W slot; pad_value: uint8 [C] device
    tensor; max_value None: no standardisation; mean / std: fp32 [C] device tensors or None."""
    n, c = len(images), images[0].shape[2]
    cpad = ((c + 3) // 4) * 4
    jobs = (_lib.ImageJob * n)()
    keep = []
    for i, (img, (h, w, top, left)) in enumerate(zip(images, geometry)):
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != c:
            raise _lib.SgxError("preprocess_u8 needs uint8 [h, w, C] images with one channel count")
        if min(h, w) <= 0 or top < 0 or left < 0 or top + h > H or left + w > W:
            raise _lib.SgxError(f"preprocess_u8: a {h}x{w} image at ({top}, {left}) does not fit the {H}x{W} batch")
        img = img.contiguous()
        keep.append(img)
        jobs[i].src, jobs[i].h0, jobs[i].w0, jobs[i].h, jobs[i].w, jobs[i].top, jobs[i].left = ptr(img), img.shape[0], img.shape[1], h, w, top, left
    dev = images[0].device
    jobs_dev = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
    y = torch.empty(n, H, W, cpad, device=dev, dtype=torch.float32)
    check(lib().sgx_preprocess_u8_hwc(ptr(jobs_dev), n, c, cpad, H, W, int(bool(reverse_channels)), int(max_value is not None),
                                      float(max_value if max_value is not None else 1.0), ptr(mean), ptr(std), ptr(pad_value), ptr(y), stream()),
          "sgx_preprocess_u8_hwc")
    return y


def nhwc_as_nchw_view(y, channels):
    """Logical NCHW [N,C,H,W] view of an NHWC buffer (what model(x) receives from DeviceDetectionCollateFN): no copy."""
    return y.permute(0, 3, 1, 2)[:, :channels]


def input_to_nhwc(x):
    """The NHWC fp32 (channels padded to 4) tensor the first convolution reads, from whatever the loader delivered: a logical NCHW
    view of an NHWC buffer (nhwc_as_nchw_view: used as is, zero copies) or a plain NCHW batch (one re-layout kernel)."""
    n, c, h, w = x.shape
    cp = ((c + 3) // 4) * 4
    if x.dtype == torch.float32 and x.stride() == (h * w * cp, 1, w * cp, cp) and x.storage_offset() % 4 == 0:
        return torch.as_strided(x, (n, h, w, cp), (h * w * cp, w * cp, cp, 1), x.storage_offset())
    return nchw_to_nhwc(x.float())


def nhwc_to_nchw(x):
    n, h, w, c = x.shape
    ld_pix, ld_img = nhwc_strides(x)
    y = torch.empty(n, c, h, w, device=x.device, dtype=torch.float32)
    check(lib().sgx_nhwc_to_nchw(n, c, h, w, ptr(x), ld_pix, ld_img, ptr(y), stream()), "sgx_nhwc_to_nchw")
    return y


# --------------------------------------------------------------------------------------------- batch norm & sweeps
@functools.lru_cache(maxsize=None)
def stats_blocks(M: int) -> int:
    return lib().sgx_stats_blocks(M)


@functools.lru_cache(maxsize=None)
def _reduce_workspace(nblk: int, C: int) -> int:
    return lib().sgx_reduce_workspace(nblk, C)


@functools.lru_cache(maxsize=None)
def _qarep_workspace(nblk: int, C: int) -> int:
    return lib().sgx_qarep_workspace(nblk, C)


@functools.lru_cache(maxsize=None)
def _dot_workspace(M: int, C: int) -> int:
    return lib().sgx_dot_workspace(M, C)


def channel_stats_partial(x):
    M, ld = rows(x)
    C = x.shape[3]
    parts = torch.empty(2, stats_blocks(M), C, device=x.device, dtype=torch.float32)
    check(lib().sgx_channel_stats_partial(ptr(x), M, C, ld, ptr(parts), stream()), "sgx_channel_stats_partial")
    return parts


def bn_finalize(parts, M, gamma, beta, eps, momentum, running_mean, running_var):
    """-> scale, shift, save_mean, save_invstd (each [C]); running stats updated in place."""
    C = parts.shape[2]
    st = torch.empty(4, C, device=parts.device, dtype=torch.float32)
    ws = WORKSPACE.get(_reduce_workspace(parts.shape[1], C), parts.device)
    p0, row = ptr(st), 4 * C  # rows of st: scale, shift, save_mean, save_invstd
    check(lib().sgx_bn_finalize(ptr(parts), parts.shape[1], M, C, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var),
                                p0 + 2 * row, p0 + 3 * row, p0, p0 + row, ptr(ws), ws.numel(), stream()), "sgx_bn_finalize")
    return st.unbind(0)


def _allreduce_sums(sums):
    import torch.distributed as dist

    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return dist.get_world_size()


def bn_finalize_sync(parts, M, gamma, beta, eps, momentum, running_mean, running_var):
    """bn_finalize with the statistics summed over all data-parallel ranks (equal per-rank element counts, as DistributedSampler
    guarantees): one 2*C fp64 all-reduce."""
    C = parts.shape[2]
    ws = WORKSPACE.get(_reduce_workspace(parts.shape[1], C), parts.device)
    sums = torch.empty(2, C, device=parts.device, dtype=torch.float64)
    check(lib().sgx_bn_reduce_sums(ptr(parts), parts.shape[1], C, ptr(sums), ptr(ws), ws.numel(), stream()), "sgx_bn_reduce_sums")
    world = _allreduce_sums(sums)
    st = torch.empty(4, C, device=parts.device, dtype=torch.float32)
    check(lib().sgx_bn_finalize_sums(ptr(sums), M * world, C, ptr(gamma), ptr(beta), eps, momentum, ptr(running_mean), ptr(running_var), ptr(st[2]),
                                     ptr(st[3]), ptr(st[0]), ptr(st[1]), stream()), "sgx_bn_finalize_sums")
    return st[0], st[1], st[2], st[3]


def bn_eval_scale_shift(gamma, beta, running_mean, running_var, eps):
    C = running_mean.numel()
    st = torch.empty(2, C, device=running_mean.device, dtype=torch.float32)
    check(lib().sgx_bn_eval_scale_shift(C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), eps, ptr(st[0]), ptr(st[1]), stream()),
          "sgx_bn_eval_scale_shift")
    return st[0], st[1]


def affine_act(x, scale=None, shift=None, r1=None, a1=1.0, a1_dev=None, r2=None, a2=1.0, out=None, act=None, want_stats=False):
    M, ld = rows(x)
    C = x.shape[3]
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    Mo, ldo = rows(out)
    parts = torch.empty(2, stats_blocks(M), C, device=x.device, dtype=torch.float32) if want_stats else None
    r1_ld = rows(r1)[1] if r1 is not None else 0
    r2_ld = rows(r2)[1] if r2 is not None else 0
    check(lib().sgx_affine_act_fwd(ptr(x), ld, ptr(scale), ptr(shift), ptr(r1), r1_ld, float(a1), ptr(a1_dev), ptr(r2), r2_ld, float(a2), ptr(out), ldo,
                                   M, C, ACT[act], ptr(parts), stream()), "sgx_affine_act_fwd")
    return (out, parts) if want_stats else out


def bn_bwd(dy, x, scale, shift, gamma, save_mean, save_invstd, dgamma, dbeta, act=None, dx_out=None, want_g=False, sync=False, parts=None):
    """Full BN(+activation) backward: returns dx (and the masked upstream gradient g if want_g).
    dgamma/dbeta (views into the gradient arena) are accumulated in place.
    parts: this layer's reduce partials when an earlier kernel already produced them (skips the reduce sweep)."""
    M, ld = rows(x)
    C = x.shape[3]
    dl = rows(dy)[1]
    a = ACT[act]
    if parts is None:
        parts = torch.empty(2, stats_blocks(M), C, device=x.device, dtype=torch.float32)
        check(lib().sgx_bn_bwd_reduce(ptr(dy), dl, ptr(x), ld, ptr(scale), ptr(shift), ptr(save_mean), M, C, a, ptr(parts), stream()), "sgx_bn_bwd_reduce")
    coef = torch.empty(5, C, device=x.device, dtype=torch.float32)
    ws = WORKSPACE.get(_reduce_workspace(parts.shape[1], C), x.device)
    if sync:
        local = torch.empty(2, C, device=x.device, dtype=torch.float64)
        check(lib().sgx_bn_reduce_sums(ptr(parts), parts.shape[1], C, ptr(local), ptr(ws), ws.numel(), stream()), "sgx_bn_reduce_sums")
        glob = local.clone()
        world = _allreduce_sums(glob)
        check(lib().sgx_bn_bwd_finalize_sums(ptr(local), ptr(glob), M * world, C, ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(dgamma), ptr(dbeta),
                                             ptr(coef), stream()), "sgx_bn_bwd_finalize_sums")
    else:
        check(lib().sgx_bn_bwd_finalize(ptr(parts), parts.shape[1], M, C, ptr(gamma), ptr(save_mean), ptr(save_invstd), ptr(dgamma), ptr(dbeta),
                                        ptr(coef), ptr(ws), ws.numel(), stream()), "sgx_bn_bwd_finalize")
    dx = dx_out if dx_out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float32)
    g = torch.empty(x.shape, device=x.device, dtype=torch.float32) if want_g else None
    check(lib().sgx_bn_bwd_apply(ptr(dy), dl, ptr(x), ld, ptr(scale), ptr(shift), ptr(coef), ptr(dx), rows(dx)[1], ptr(g), rows(g)[1] if want_g else 0,
                                 M, C, a, stream()), "sgx_bn_bwd_apply")
    return (dx, g) if want_g else dx


def dot_sum(a, b, out, accumulate=True, scale=1.0):
    """out[0] (+)= scale * sum(a*b) over NHWC views a, b."""
    M, la = rows(a)
    C = a.shape[3]
    ws = WORKSPACE.get(_dot_workspace(M, C), a.device)
    check(lib().sgx_dot(ptr(a), la, ptr(b), rows(b)[1], M, C, float(scale), ptr(out), int(accumulate), ptr(ws), ws.numel(), stream()), "sgx_dot")


def axpy(x, a=1.0, a_dev=None, out=None, accumulate=False):
    if x.dtype == HALF:
        if a != 1.0 or a_dev is not None or accumulate or out is None:
            raise _lib.SgxError("axpy on bf16: the half-precision inference path only copies a view into a concat slice")
        return hcopy(x, out)
    M, ld = rows(x)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(lib().sgx_axpy(ptr(x), ld, float(a), ptr(a_dev), ptr(out), rows(out)[1], M, x.shape[3], int(accumulate), stream()), "sgx_axpy")
    return out


def relu_bwd(dy, y, out=None):
    """g = dy * (y > 0) over NHWC views."""
    M, ld = rows(y)
    if out is None:
        out = torch.empty(y.shape, device=y.device, dtype=torch.float32)
    check(lib().sgx_relu_bwd(ptr(dy), rows(dy)[1], ptr(y), ld, ptr(out), rows(out)[1], M, y.shape[3], stream()), "sgx_relu_bwd")
    return out


def relu_bwd_bn_reduce(dy, y, x, save_mean):
    """(g, parts): g = dy * (y > 0) and the BatchNorm-backward reduce partials of (g, x) - bn_bwd(g, x, ..., act=None, parts=parts) skips
    its reduce sweep."""
    M, ld = rows(y)
    C = y.shape[3]
    g = torch.empty(y.shape, device=y.device, dtype=torch.float32)
    parts = torch.empty(2, stats_blocks(M), C, device=y.device, dtype=torch.float32)
    check(lib().sgx_relu_bwd_bn_reduce(ptr(dy), rows(dy)[1], ptr(y), ld, ptr(x), rows(x)[1], ptr(save_mean), ptr(g), rows(g)[1], M, C, ptr(parts),
                                       stream()), "sgx_relu_bwd_bn_reduce")
    return g, parts


def dual_affine_act(x1, s1, t1, x2=None, s2=None, t2=None, post_add=None, act=None, out=None, post_scale=None):
    """y = act(s1*x1 + t1 [+ s2*x2 + t2]) [+ post_scale * post_add]   (RepVGG two-branch BatchNorm sum; post-activation residual;
    post_scale: a float or a one-element device tensor, default 1)."""
    ps_dev = post_scale if torch.is_tensor(post_scale) else None
    ps = 1.0 if post_scale is None or ps_dev is not None else float(post_scale)
    M, ld1 = rows(x1)
    C = x1.shape[3]
    if out is None:
        out = torch.empty(x1.shape, device=x1.device, dtype=torch.float32)
    check(lib().sgx_dual_affine_act_fwd(ptr(x1), ld1, ptr(s1), ptr(t1), ptr(x2), rows(x2)[1] if x2 is not None else 0, ptr(s2), ptr(t2), ptr(post_add),
                                        rows(post_add)[1] if post_add is not None else 0, ps, ptr(ps_dev), ptr(out), rows(out)[1], M, C, ACT[act],
                                        stream()), "sgx_dual_affine_act_fwd")
    return out


def dual_affine_act_bwd(dy, x1, s1, t1, x2=None, s2=None, t2=None, act=None, out=None):
    """g = dy * act'(s1*x1 + t1 [+ s2*x2 + t2])."""
    M, ld1 = rows(x1)
    C = x1.shape[3]
    if out is None:
        out = torch.empty(x1.shape, device=x1.device, dtype=torch.float32)
    check(lib().sgx_dual_affine_act_bwd(ptr(dy), rows(dy)[1], ptr(x1), ld1, ptr(s1), ptr(t1), ptr(x2), rows(x2)[1] if x2 is not None else 0, ptr(s2), ptr(t2),
                                        ptr(out), rows(out)[1], M, C, ACT[act], stream()), "sgx_dual_affine_act_bwd")
    return out


def dual_affine_act_bwd_reduce(dy, x1, s1, t1, mean1, x2, s2, t2, mean2, act=None, out=None):
    """g = dy * act'(s1*x1 + t1 + s2*x2 + t2) AND the reduce rows of both BatchNorm backward passes -> (g, parts1, parts2), each parts [2, blocks, C]
    as bn_bwd(parts=...) takes them (sum g, sum g (x - mean))."""
    M, ld1 = rows(x1)
    C = x1.shape[3]
    if out is None:
        out = torch.empty(x1.shape, device=x1.device, dtype=torch.float32)
    parts = torch.empty(4, stats_blocks(M), C, device=x1.device, dtype=torch.float32)
    check(lib().sgx_dual_affine_act_bwd_reduce(ptr(dy), rows(dy)[1], ptr(x1), ld1, ptr(s1), ptr(t1), ptr(mean1), ptr(x2), rows(x2)[1], ptr(s2), ptr(t2), ptr(mean2),
                                               ptr(out), rows(out)[1], M, C, ACT[act], ptr(parts), stream()), "sgx_dual_affine_act_bwd_reduce")
    return out, parts[0:2], parts[2:4]


GATE = {None: 0, "none": 0, "hardsigmoid": 1, "sigmoid": 2}


def image_colsum(u, v=None, scale=1.0, pre=None, gate=None):
    """out[n,c] = scale * f'(pre[n,c]) * sum_pixels u*v   ([N,C]; v / pre optional)."""
    n, h, w, c = u.shape
    ul, ui = nhwc_strides(u)
    if u.dtype == HALF:  # half-precision inference: the per-image channel means of a bf16 activation, fp32 sums in a fixed order
        if v is not None or pre is not None:
            raise _lib.SgxError("image_colsum on bf16: inference form only (plain per-image channel sums)")
        out = torch.empty(n, c, device=u.device, dtype=torch.float32)
        check(lib().sgx_himage_colsum(n, h * w, c, ptr(u), ul, ui, float(scale), ptr(out), stream()), "sgx_himage_colsum")
        return out
    vl, vi = nhwc_strides(v) if v is not None else (0, 0)
    out = torch.empty(n, c, device=u.device, dtype=torch.float32)
    ws = WORKSPACE.get(lib().sgx_image_colsum_workspace(n, h * w, c), u.device)
    check(lib().sgx_image_colsum(n, h * w, c, ptr(u), ul, ui, ptr(v), vl, vi, float(scale), ptr(pre), GATE[gate], ptr(out), ptr(ws), ws.numel(), stream()),
          "sgx_image_colsum")
    return out


def channel_gate(x, pre, gate, bias=None, bias_scale=1.0, out=None, accumulate=False):
    """y (+)= x * f(pre[n,c]) + bias_scale * bias[n,c]."""
    n, h, w, c = x.shape
    if x.dtype == HALF:  # half-precision inference: product in fp32, one rounding to bf16
        if bias is not None or accumulate:
            raise _lib.SgxError("channel_gate on bf16: inference form only (no bias, no accumulation)")
        if out is None:
            out = torch.empty(x.shape, device=x.device, dtype=HALF)
        xl, xi = nhwc_strides(x)
        yl, yi = nhwc_strides(out)
        check(lib().sgx_hchannel_gate(n, h * w, c, ptr(x), xl, xi, ptr(pre), GATE[gate], ptr(out), yl, yi, stream()), "sgx_hchannel_gate")
        return out
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    xl, xi = nhwc_strides(x)
    yl, yi = nhwc_strides(out)
    check(lib().sgx_channel_gate(n, h * w, c, ptr(x), xl, xi, ptr(pre), GATE[gate], ptr(bias), float(bias_scale), ptr(out), yl, yi, int(accumulate), stream()),
          "sgx_channel_gate")
    return out


def upsample2x_fwd(x, out=None):
    n, h, w, c = x.shape
    if out is None:
        out = torch.empty(n, 2 * h, 2 * w, c, device=x.device, dtype=x.dtype)
    xl, xi = nhwc_strides(x)
    yl, yi = nhwc_strides(out)
    if x.dtype == HALF:
        check(lib().sgx_hupsample2x_fwd(n, h, w, c, ptr(x), xl, xi, ptr(out), yl, yi, stream()), "sgx_hupsample2x_fwd")
        return out
    check(lib().sgx_upsample2x_fwd(n, h, w, c, ptr(x), xl, xi, ptr(out), yl, yi, stream()), "sgx_upsample2x_fwd")
    return out


def upsample2x_bwd(dy, out=None, accumulate=False):
    n, h2, w2, c = dy.shape
    h, w = h2 // 2, w2 // 2
    if out is None:
        out = torch.empty(n, h, w, c, device=dy.device, dtype=torch.float32)
    dl, di = nhwc_strides(dy)
    xl, xi = nhwc_strides(out)
    check(lib().sgx_upsample2x_bwd(n, h, w, c, ptr(dy), dl, di, ptr(out), xl, xi, int(accumulate), stream()), "sgx_upsample2x_bwd")
    return out


def colsum(x, out, accumulate=True):
    ld_pix, ld_img = nhwc_strides(x)
    n, h, w, C = x.shape
    M = n * h * w
    ws = WORKSPACE.get(lib().sgx_colsum_workspace(M, C), x.device)
    check(lib().sgx_colsum(ptr(x), ld_pix, M, C, h * w, ld_img, ptr(out), int(accumulate), ptr(ws), stream()), "sgx_colsum")


def fill(t, v=0.0):
    check(lib().sgx_fill(ptr(t), t.numel(), float(v), stream()), "sgx_fill")


# --------------------------------------------------------------------------------------------- pooling
def maxpool_fwd(x, k, stride, pad, out=None, want_argmax=True):
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    if x.dtype == HALF:
        if want_argmax:
            raise _lib.SgxError("maxpool_fwd on bf16: inference only (no arg-max)")
        if out is None:
            out = torch.empty(n, ho, wo, c, device=x.device, dtype=HALF)
        xl, xi = nhwc_strides(x)
        yl, yi = nhwc_strides(out)
        check(lib().sgx_hmaxpool_fwd(n, h, w, c, k, stride, pad, ptr(x), xl, xi, ptr(out), yl, yi, stream()), "sgx_hmaxpool_fwd")
        return out, None
    if out is None:
        out = torch.empty(n, ho, wo, c, device=x.device, dtype=torch.float32)
    am = torch.empty(n, ho, wo, c, device=x.device, dtype=torch.int32) if want_argmax else None
    xl, xi = nhwc_strides(x)
    yl, yi = nhwc_strides(out)
    check(lib().sgx_maxpool_fwd(n, h, w, c, k, stride, pad, ptr(x), xl, xi, ptr(out), yl, yi, ptr(am), stream()), "sgx_maxpool_fwd")
    return out, am


def maxpool_bwd(dy, argmax, x_shape, k, stride, pad, out=None, accumulate=False):
    n, h, w, c = x_shape
    if out is None:
        out = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    dl, di = nhwc_strides(dy)
    xl, xi = nhwc_strides(out)
    check(lib().sgx_maxpool_bwd(n, h, w, c, k, stride, pad, ptr(argmax), ptr(dy), dl, di, ptr(out), xl, xi, int(accumulate), stream()), "sgx_maxpool_bwd")
    return out


def avgpool_fwd(x):
    n, h, w, c = x.shape
    xl, xi = nhwc_strides(x)
    y = torch.empty(n, c, device=x.device, dtype=torch.float32)
    check(lib().sgx_avgpool_fwd(n, h * w, c, ptr(x), xl, xi, ptr(y), stream()), "sgx_avgpool_fwd")
    return y


def avgpool_bwd(dy, x_shape):
    n, h, w, c = x_shape
    dx = torch.empty(x_shape, device=dy.device, dtype=torch.float32)
    dy = dy.contiguous()
    check(lib().sgx_avgpool_bwd(n, h * w, c, ptr(dy), ptr(dx), c, h * w * c, stream()), "sgx_avgpool_bwd")
    return dx


# --------------------------------------------------------------------------------------------- detection head / loss / nms
def dfl_decode(logits, distri, points_grid, strides, reg_max):
    B, L, C = logits.shape
    boxes = torch.empty(B, L, 4, device=logits.device, dtype=torch.float32)
    scores = torch.empty(B, L, C, device=logits.device, dtype=torch.float32)
    check(lib().sgx_dfl_decode(B, L, C, reg_max, ptr(logits), ptr(distri), ptr(points_grid), ptr(strides), ptr(boxes), ptr(scores), stream()),
          "sgx_dfl_decode")
    return boxes, scores


def loss_desc(B, L, C, reg_max, nmax, static, vfl, counts, weights, sequential=False) -> LossDesc:
    d = LossDesc()
    d.B, d.L, d.C, d.reg_max, d.nmax = B, L, C, reg_max, nmax
    d.use_static_assigner, d.use_varifocal = int(static), int(vfl)
    d.num_levels = len(counts)
    for i, c in enumerate(counts):
        d.level_count[i] = int(c)
    d.w_cls, d.w_iou, d.w_dfl = [float(w) for w in weights]
    d.sequential_assignment = int(sequential)
    return d


def ppyoloe_loss_fwd(logits, distri, anchors, points, strides, targets, counts, static, vfl, weights, sequential=False):
    """-> dict(sums[4], label[B,L] int32, box[B,L,4], score[B,L], g_logits, g_distri)"""
    B, L, C = logits.shape
    reg_max = distri.shape[2] // 4 - 1
    dev = logits.device
    T = int(targets.shape[0])
    nmax = T
    targets = targets.contiguous().float()
    gt_count = torch.empty(B, device=dev, dtype=torch.int32)
    gt_index = torch.empty(B, max(nmax, 1), device=dev, dtype=torch.int32)
    overflow = torch.empty(1, device=dev, dtype=torch.int32)
    check(lib().sgx_targets_index(ptr(targets) if T else None, T, B, nmax, ptr(gt_count), ptr(gt_index) if nmax else None, ptr(overflow), stream()),
          "sgx_targets_index")
    d = loss_desc(B, L, C, reg_max, nmax, static, vfl, counts, weights, sequential)
    out = dict(
        sums=torch.empty(4, device=dev, dtype=torch.float32),
        label=torch.empty(B, L, device=dev, dtype=torch.int32),
        box=torch.empty(B, L, 4, device=dev, dtype=torch.float32),
        score=torch.empty(B, L, device=dev, dtype=torch.float32),
        g_logits=torch.empty(B, L, C, device=dev, dtype=torch.float32),
        g_distri=torch.empty(B, L, 4 * (reg_max + 1), device=dev, dtype=torch.float32),
    )
    nbytes = lib().sgx_ppyoloe_loss_workspace(ctypes.byref(d))
    ws = WORKSPACE.get(nbytes, dev)
    # converted operands stay bound until the launch is enqueued: a temporary freed earlier could be handed out again by the caching allocator
    logits, distri, anchors, points, strides = (t.contiguous() for t in (logits, distri, anchors, points, strides))
    check(lib().sgx_ppyoloe_loss_fwd(ctypes.byref(d), ptr(logits), ptr(distri), ptr(anchors), ptr(points),
                                     ptr(strides), ptr(targets) if T else None, ptr(gt_count), ptr(gt_index) if nmax else None,
                                     ptr(out["sums"]), ptr(out["label"]), ptr(out["box"]), ptr(out["score"]), ptr(out["g_logits"]), ptr(out["g_distri"]),
                                     ptr(ws), ws.numel(), stream()), "sgx_ppyoloe_loss_fwd")
    return out


def ppyoloe_loss_finalize(sums, weights, score_div=1.0):
    items = torch.empty(4, device=sums.device, dtype=torch.float32)
    inv = torch.empty(1, device=sums.device, dtype=torch.float32)
    check(lib().sgx_ppyoloe_loss_finalize(ptr(sums), float(weights[0]), float(weights[1]), float(weights[2]), float(score_div), ptr(items), ptr(inv),
                                          stream()), "sgx_ppyoloe_loss_finalize")
    return items, inv


def scale_by_device_scalar(x, s, t=None):
    y = torch.empty_like(x)
    check(lib().sgx_scale_by_device_scalar(ptr(x), ptr(s), ptr(t), ptr(y), x.numel(), stream()), "sgx_scale_by_device_scalar")
    return y


def nms(boxes, scores, score_threshold, iou_threshold, nms_top_k, max_predictions, multi_label=True, class_mode=0):
    B, L, C = scores.shape
    d = NmsDesc()
    d.B, d.L, d.C, d.multi_label, d.class_mode = B, L, C, int(multi_label), int(class_mode)
    d.nms_top_k, d.max_predictions = int(nms_top_k), int(max_predictions)
    d.score_threshold, d.iou_threshold = float(score_threshold), float(iou_threshold)
    dev = scores.device
    out = torch.empty(B, max_predictions, 6, device=dev, dtype=torch.float32)
    cnt = torch.empty(B, device=dev, dtype=torch.int32)
    idx = torch.empty(B, max_predictions, device=dev, dtype=torch.int32)
    ncand = torch.empty(B, device=dev, dtype=torch.int32)
    boxes, scores = boxes.contiguous().float(), scores.contiguous().float()  # bound to names: alive until the launch is enqueued
    ws = WORKSPACE.get(lib().sgx_nms_workspace(ctypes.byref(d)), dev)
    check(lib().sgx_nms(ctypes.byref(d), ptr(boxes), ptr(scores), ptr(out), ptr(cnt), ptr(idx), ptr(ncand), ptr(ws), ws.numel(), stream()), "sgx_nms")
    _NMS_LAST[:] = [d, ws]
    return out, cnt, idx, ncand


_NMS_LAST = [None, None]


def nms_fallbacks() -> int:
    """Images of the LAST `nms` call (multi-label) whose stage 2 streamed the raw scores instead of stage 1's candidate list - exact rows,
    many times slower (csrc/nms.hip).  Synchronises; benches and tests assert 0 on their inputs."""
    d, ws = _NMS_LAST
    if d is None or not d.multi_label:
        return 0
    off, stride = ctypes.c_int64(0), ctypes.c_int32(0)
    check(lib().sgx_debug_nms_fallback_slot(ctypes.byref(d), ctypes.byref(off), ctypes.byref(stride)), "sgx_debug_nms_fallback_slot")
    flags = ws[: ws.numel() // 4 * 4].view(torch.int32)[off.value: off.value + d.B * stride.value: stride.value]
    return int(flags.sum())


def decode_topk(boxes, scores, k: int):
    """Pre-NMS top-k of the decoding modules (reference: yolo_nas_variants.py:53-72, pp_yolo_e.py:57-84): per image the k anchors with the
    largest class confidence max_c scores[b, l, c], sorted by confidence descending (equal confidences: lower anchor index first - torch.topk
    leaves that order unspecified), with their boxes and full score rows.  One launch of the post-prediction kernel in single-label mode
    with no threshold and no suppression (its threshold -> exact radix select -> LDS sort stages), then a row gather.
    -> (boxes [B, k, 4], scores [B, k, C], anchor index [B, k] int64).  Scores must be >= 0 (sigmoid outputs)."""
    B, L, C = scores.shape
    if not 0 < k <= L:
        raise ValueError(f"decode_topk: k={k} must be in [1, {L}] (torch.topk raises likewise)")
    out, cnt, idx, _ = nms(boxes, scores, 0.0, 2.0, k, k, multi_label=False, class_mode=0)
    if int(cnt.min()) != k:
        raise ValueError("decode_topk: negative or NaN class scores (the decoding modules expect sigmoid outputs)")
    idx = idx.long()
    return out[:, :, :4].contiguous(), torch.gather(scores, 1, idx[:, :, None].expand(B, k, C)), idx


def _index_targets(t, B):
    """flat [T,6] targets -> (contiguous float tensor or None, gt_count[B], gt_index[B][nmax], nmax) on t's device."""
    T = int(t.shape[0])
    dev = t.device
    cnt = torch.zeros(B, device=dev, dtype=torch.int32)
    if T == 0:
        return None, cnt, None, 0
    t = t.contiguous().float()
    idx = torch.empty(B, T, device=dev, dtype=torch.int32)
    ovf = torch.empty(1, device=dev, dtype=torch.int32)
    check(lib().sgx_targets_index(ptr(t), T, B, T, ptr(cnt), ptr(idx), ptr(ovf), stream()), "sgx_targets_index")
    return t, cnt, idx, T


def detection_unmap(rows, counts, steps):
    """rows [B,P,6] + counts [B] (the NMS output layout), steps [B,S,3] fp32 (kind 0 add / 1 multiply / 2 none, a_x, a_y; S may be 0) -> ONE
    flat fp32 tensor of B*P*6 + B elements: every image's rows mapped back through its processing stages (one fp32 rounding per step, as
    the reference's numpy passes round), rows beyond the count zeroed, then the B clamped counts as int32 bit patterns - what predict()
    copies to the host in one transfer."""
    B, P, _ = rows.shape
    rows, counts = rows.contiguous().float(), counts.contiguous().int()
    S = 0 if steps is None else int(steps.shape[1])
    if S:
        steps = steps.contiguous().float()
    out = torch.empty(B * P * 6 + B, device=rows.device, dtype=torch.float32)
    check(lib().sgx_detection_unmap(ptr(rows), ptr(counts), B, P, ptr(steps) if S else None, S, ptr(out), stream()), "sgx_detection_unmap")
    return out


def detection_match(rows, counts, targets, crowd_targets, thresholds, height, width, top_k, denormalize):
    """rows [B,P,6] + counts [B] (the NMS output layout), targets / crowd_targets flat [T,6] -> (matched, ignore) uint8 [B,P,nthr]."""
    B, P, _ = rows.shape
    dev = rows.device
    thr = thresholds.to(dev).float().contiguous()
    t, tc, ti, nmax = _index_targets(targets.to(dev), B)
    c, cc, ci, cmax = _index_targets(crowd_targets.to(dev), B)
    d = MatchDesc(B, P, int(thr.numel()), int(top_k), int(height), int(width), int(bool(denormalize)), nmax, cmax)
    matched = torch.empty(B, P, thr.numel(), device=dev, dtype=torch.uint8)
    ignore = torch.empty(B, P, thr.numel(), device=dev, dtype=torch.uint8)
    rows, counts = rows.contiguous().float(), counts.contiguous().int()  # bound to names: alive until the launch is enqueued
    check(lib().sgx_detection_match(ctypes.byref(d), ptr(rows), ptr(counts), ptr(t), ptr(tc), ptr(ti), ptr(c), ptr(cc), ptr(ci), ptr(thr), ptr(matched),
                                    ptr(ignore), stream()), "sgx_detection_match")
    return matched, ignore


def softmax_ce(logits, labels, smoothing=0.0, weight=None, ignore_index=-100, reduction="mean"):
    """-> (loss scalar, dlogits without the 1 / denominator factor, the device scalar 1 / denominator)."""
    B, K = logits.shape
    loss = torch.empty(2 * B + 2, device=logits.device, dtype=torch.float32)
    dlogits = torch.empty(B, K, device=logits.device, dtype=torch.float32)
    logits, labels = logits.contiguous(), labels.contiguous().long()  # bound to names: alive until the launch is enqueued
    weight = weight.contiguous().float() if weight is not None else None
    check(lib().sgx_softmax_ce_fwd_bwd(B, K, ptr(logits), ptr(labels), float(smoothing), ptr(weight), int(ignore_index), int(reduction == "sum"), ptr(loss),
                                       ptr(dlogits), stream()), "sgx_softmax_ce_fwd_bwd")
    return loss[0], dlogits, loss[1:2]


# --------------------------------------------------------------------------------------------- optimizer
def adamw_step(p, g, m, v, lr, beta1, beta2, eps, step, seg_end, seg_wd, grad_scale=None):
    check(lib().sgx_adamw_step(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, step, ptr(seg_end), ptr(seg_wd), seg_end.numel(),
                               ptr(grad_scale), stream()), "sgx_adamw_step")
    weights_written()


def sgd_step(p, g, mom, lr, momentum, dampening, nesterov, first_step, seg_end, seg_wd):
    check(lib().sgx_sgd_step(ptr(p), ptr(g), ptr(mom), p.numel(), lr, momentum, dampening, int(nesterov), int(first_step), ptr(seg_end), ptr(seg_wd),
                             seg_end.numel(), stream()), "sgx_sgd_step")
    weights_written()


def ema_update(ema, p, decay):
    check(lib().sgx_ema_update(ptr(ema), ptr(p), ema.numel(), float(decay), stream()), "sgx_ema_update")
    weights_written()


# --------------------------------------------------------------------------------------------- conv arithmetic
CONV_MATH = {"fp32": 0, "bf16x3": 1, "auto": 2, "patch": 3, "patch_auto": 4, "patch_bf3": 5}
DEFAULT_CONV_MATH = "patch_bf3"  # what the library starts with (csrc/conv.hip g_conv_math): the patch kernel on its problems + bf16x3 per problem elsewhere
DEFAULT_WGRAD_MATH = "patch"  # (csrc/conv.hip g_wg_math): bf16x3 slab loop + the weight-gradient patch kernel


def set_conv_math(mode: str):
    """"fp32": fp32 matrix pipe (exact fp32 FMA chains).  "bf16x3": fp32 operands split into three bf16 pieces, six cross products on the
    bf16 matrix pipe with fp32 accumulation - fp32-accurate, 2.7x fewer matrix-pipe cycles.  "auto": bf16x3 for reductions of depth
    (taps x channels) >= 192, fp32 MFMA for shallow ones.  "patch": 3x3 stride-1 forward / data-gradient problems (channel counts in 16s) run
    the patch kernel - bf16x3 arithmetic with the input patch of an 8 x 16 pixel tile staged in LDS once for all nine taps - everything
    else stays on the fp32 pipe (include/sgx_hip.h: sgx_conv_set_math)."""
    check(lib().sgx_conv_set_math(CONV_MATH[mode]), "sgx_conv_set_math")
    clear_desc_cache()


def get_conv_math() -> str:
    return {v: k for k, v in CONV_MATH.items()}[lib().sgx_conv_get_math()]


# --------------------------------------------------------------------------------------------- measurement aid
def prof_enable(on: bool):
    check(lib().sgx_prof_enable(int(on)), "sgx_prof_enable")


def prof_summary(cls: int):
    """-> (total ms, algorithmic FLOPs, launches) of kernel class cls (0 = fp32-MFMA implicit GEMM fwd/dgrad, 1 = weight gradient, 2 = the bf16x3 patch kernel fwd/dgrad)."""
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    check(lib().sgx_prof_summary(cls, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), "sgx_prof_summary")
    return ms.value, fl.value, n.value


def prof_bytes(cls: int) -> float:
    """Algorithmic HBM bytes (inputs + weights + outputs once, fp32) of the launches prof_summary(cls) covers."""
    b = ctypes.c_double()
    check(lib().sgx_prof_bytes(cls, ctypes.byref(b)), "sgx_prof_bytes")
    return b.value


def prof_bound_ms(cls: int, peak_flops: float, hbm_bytes_per_s: float) -> float:
    """Roofline time (ms) of the launches prof_summary(cls) covers: sum of max(FLOPs / peak, algorithmic bytes / HBM rate) per launch."""
    t = ctypes.c_double()
    check(lib().sgx_prof_bound_ms(cls, float(peak_flops), float(hbm_bytes_per_s), ctypes.byref(t)), "sgx_prof_bound_ms")
    return t.value
