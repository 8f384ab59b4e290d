"""Parity of every C-ABI kernel (include/sgx_hip.h) against the CPU path the reference executes: ATen CPU ops
(F.conv2d, F.batch_norm, ...) for the conv stacks, oracle/ for the loss and NMS.

Each test runs twice through the `backend` fixture:
  gpu  (marked gpu)  the product library libsgx_hip.so on cuda:0;
  emu                the same kernel sources compiled against tests/emu (one host fiber per HIP thread) - logic check, small shapes.
Tolerances: fp32 with a different summation order -> 2e-5 of the tensor's max-abs (north star: 1e-4 rel);
integer/index outputs bit-exact.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import assert_close, empty_nhwc, rel_err, to_nchw_cpu, to_nhwc

from super_gradients_amd import _lib
from super_gradients_amd import kernels as K

TOL = 2e-5


def _sizes(backend, gpu, emu):
    return gpu if backend.type == "cuda" else emu


# (N, H, W, C, K, R, stride, pad)
CONV_GPU = [
    (2, 40, 40, 64, 64, 3, 1, 1),
    (2, 33, 31, 96, 48, 3, 1, 1),      # ragged spatial, N tile 96->48
    (4, 64, 64, 4, 48, 3, 2, 1),       # stem (3 channels padded to 4), stride 2
    (2, 40, 40, 192, 384, 3, 2, 1),    # downsample
    (2, 20, 20, 1536, 768, 1, 1, 0),   # SPP cv2: long K
    (3, 40, 40, 128, 80, 1, 1, 0),     # cls_pred
    (2, 23, 17, 64, 68, 1, 1, 0),      # reg_pred, ragged
    (2, 56, 56, 4, 64, 7, 2, 3),       # ResNet stem
    (2, 28, 28, 64, 128, 1, 2, 0),     # ResNet 1x1 s2 shortcut
    (1, 160, 160, 32, 32, 3, 1, 1),    # many row tiles
    (2, 20, 20, 32, 50, 1, 1, 0),      # K % 4 != 0: scalar epilogue path (forward only)
]
CONV_EMU = [
    (1, 9, 7, 8, 36, 3, 1, 1),
    (1, 8, 8, 4, 32, 3, 2, 1),
    (2, 5, 6, 20, 8, 1, 1, 0),
    (1, 9, 9, 4, 8, 7, 2, 3),
    (1, 6, 6, 8, 4, 1, 2, 0),
    (1, 48, 48, 4, 8, 1, 1, 0),        # 2304 pixels: weight gradient split over 9 slabs (parallel slab reduce)
    (1, 5, 5, 8, 6, 3, 1, 1),          # K % 4 != 0: scalar epilogue path (forward only)
    # (round 5: the emulation runs a workgroup's lanes as fibers - these no longer cost minutes)
    (1, 40, 40, 32, 32, 3, 1, 1),      # 1600 pixels: the patch kernel under the DEFAULT arithmetic (forward, data gradient), patch weight gradient
    (1, 16, 16, 192, 96, 1, 1, 0),     # deep 1x1: the 32-deep bf16x3 GEMM loop, 96 filters (128x32 / 96-wide tiles)
    (2, 20, 20, 64, 128, 3, 2, 1),     # 3x3 stride 2, depth 576: bf16x3 GEMM, four output-parity classes in the data gradient
    (1, 24, 20, 32, 48, 3, 1, 1),      # 3x3 stride 1 below the patch kernel's 40 x 40 floor: GEMM loop, ragged 48 filters
]


def _conv_case(shape, seed=0):
    n, h, w, c, k, r, s, p = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5
    b = torch.randn(k, generator=g)
    return x, wt, b


@pytest.mark.parametrize("idx", range(max(len(CONV_GPU), len(CONV_EMU))))
def test_conv_fwd(backend, idx):
    shapes = _sizes(backend, CONV_GPU, CONV_EMU)
    if idx >= len(shapes):
        pytest.skip("no such case")
    shape = shapes[idx]
    n, h, w, c, k, r, s, p = shape
    x, wt, b = _conv_case(shape)
    ref = F.conv2d(x, wt, b, stride=s, padding=p)
    xd = to_nhwc(x, backend)
    wd = K.to_ohwi(wt.to(backend))
    y = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p)
    assert_close(to_nchw_cpu(y), ref, TOL, f"conv fwd {shape}")
    # fused epilogue: bias + addend + relu, output written into a channel slice of a wider (concat) buffer,
    # input read from a channel slice; BN statistics partials of the pre-activation value
    add = torch.randn(ref.shape, generator=torch.Generator().manual_seed(1))
    xs = to_nhwc(x, backend, ld_pix=c + 8, c_off=4)
    out = empty_nhwc(n, ref.shape[2], ref.shape[3], k, backend, ld_pix=k + 12, c_off=8)
    addd = to_nhwc(add, backend, ld_pix=k + 12, c_off=8)
    y2, parts = K.conv2d_fwd(xs, wd, bias=b.to(backend), addend=addd, out=out, act="relu", stride=s, pad=p, stat_partials=True)
    pre = ref + add
    assert_close(to_nchw_cpu(y2), F.relu(pre), TOL, f"conv fwd fused {shape}")
    M = pre.numel() // k
    s1 = parts[0].sum(0).cpu()
    s2 = parts[1].sum(0).cpu()
    assert_close(s1 / M, pre.mean((0, 2, 3)), 1e-4, "stat sum")
    assert_close(s2 / M, (pre * pre).mean((0, 2, 3)), 1e-4, "stat sumsq")


@pytest.mark.parametrize("idx", range(max(len(CONV_GPU), len(CONV_EMU))))
def test_conv_bwd(backend, idx):
    shapes = _sizes(backend, CONV_GPU, CONV_EMU)
    if idx >= len(shapes):
        pytest.skip("no such case")
    shape = shapes[idx]
    n, h, w, c, k, r, s, p = shape
    if k % 4:
        pytest.skip("bwd needs K % 4 == 0")
    x, wt, b = _conv_case(shape)
    x.requires_grad_(True)
    wt.requires_grad_(True)
    b.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dy)
    xd = to_nhwc(x.detach(), backend)
    wd = K.to_ohwi(wt.detach().to(backend))
    dyd = to_nhwc(dy, backend)
    dx = K.conv2d_bwd_data(dyd, wd, tuple(xd.shape), stride=s, pad=p)
    assert_close(to_nchw_cpu(dx), x.grad, TOL, f"dgrad {shape}")
    # weights transposed ahead of time (what the engine does under the forward pass): identical result
    wtb = K.conv2d_wt_buffer(wd, backend)
    K.conv2d_transpose_weights(wd, wtb, stride=s, pad=p)
    dx3 = K.conv2d_bwd_data_wt(dyd, wd, wtb, tuple(xd.shape), stride=s, pad=p)
    assert torch.equal(dx3.cpu(), dx.cpu()), f"dgrad with pre-transposed weights {shape}"
    # accumulate + addend form
    add = torch.randn(x.shape, generator=torch.Generator().manual_seed(3))
    dx2 = to_nhwc(torch.ones_like(x.detach()), backend)
    K.conv2d_bwd_data(dyd, wd, tuple(xd.shape), stride=s, pad=p, addend=to_nhwc(add, backend), out=dx2, accumulate=True)
    assert_close(to_nchw_cpu(dx2), x.grad + add + 1.0, TOL, f"dgrad acc {shape}")
    dw = K.ohwi_empty(k, c, r, r, backend)
    dw.fill_(0.5)
    db = torch.zeros(k, device=backend)
    K.conv2d_bwd_weight(xd, dyd, dw, db, stride=s, pad=p)
    assert_close(dw.cpu(), wt.grad + 0.5, TOL, f"wgrad {shape}")
    assert_close(db.cpu(), b.grad, TOL, f"dbias {shape}")


@pytest.mark.parametrize("idx", range(max(len(CONV_GPU), len(CONV_EMU))))
def test_conv_bf16x3(backend, idx):
    """The bf16x3 arithmetic of the forward / data-gradient GEMMs (sgx_conv_set_math(1)): fp32 operands split into three bf16 pieces,
    six cross products on the bf16 matrix pipe, fp32 accumulate - held to the SAME tolerance as the fp32-MFMA path."""
    shapes = _sizes(backend, CONV_GPU, CONV_EMU)
    if idx >= len(shapes):
        pytest.skip("no such case")
    shape = shapes[idx]
    n, h, w, c, k, r, s, p = shape
    x, wt, b = _conv_case(shape)
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dy)
    add = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    K.set_conv_math("bf16x3")
    try:
        assert K.get_conv_math() == "bf16x3"
        xd = to_nhwc(x.detach(), backend, ld_pix=c + 8, c_off=4)
        wd = K.to_ohwi(wt.to(backend))
        yd = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p)
        assert_close(to_nchw_cpu(yd), y.detach(), TOL, f"bf16x3 conv fwd {shape}")
        # the error of the split arithmetic itself, against the fp64 truth: no worse than the fp32 matrix pipe's (same inputs, same
        # summation order inside a slab), i.e. the split is NOT a precision trade
        y64 = F.conv2d(x.detach().double(), wt.double(), b.double(), stride=s, padding=p)
        K.set_conv_math("fp32")
        y32 = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p)
        K.set_conv_math("bf16x3")
        e_bf3, e_f32 = rel_err(to_nchw_cpu(yd).double(), y64), rel_err(to_nchw_cpu(y32).double(), y64)
        assert e_bf3 <= 1.5 * e_f32 + 5e-8, f"bf16x3 error vs fp64 {e_bf3:.2e}, fp32 MFMA {e_f32:.2e}"
        if k % 4 == 0:
            out = empty_nhwc(n, y.shape[2], y.shape[3], k, backend, ld_pix=k + 12, c_off=8)
            y2, parts = K.conv2d_fwd(xd, wd, bias=b.to(backend), addend=to_nhwc(add, backend, ld_pix=k + 12, c_off=8), out=out, act="relu", stride=s, pad=p,
                                     stat_partials=True)
            pre = y.detach() + add
            assert_close(to_nchw_cpu(y2), F.relu(pre), TOL, f"bf16x3 conv fwd fused {shape}")
            assert_close(parts[0].sum(0).cpu() / (pre.numel() // k), pre.mean((0, 2, 3)), 1e-4, "stat sum")
            dx = K.conv2d_bwd_data(to_nhwc(dy, backend), wd, (n, h, w, c), stride=s, pad=p)
            assert_close(to_nchw_cpu(dx), x.grad, TOL, f"bf16x3 dgrad {shape}")
    finally:
        K.set_conv_math(K.DEFAULT_CONV_MATH)
    assert K.get_conv_math() == K.DEFAULT_CONV_MATH


def test_conv_transpose(backend):
    n, h, w, c, k = _sizes(backend, (2, 20, 20, 192, 192), (1, 3, 4, 8, 4))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(c, k, 2, 2, generator=g) / c ** 0.5).requires_grad_(True)
    b = torch.randn(k, generator=g, requires_grad=True)
    y = F.conv_transpose2d(x, wt, b, stride=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    xd = to_nhwc(x.detach(), backend)
    wd = K.convT_empty(c, k, backend)
    wd.copy_(wt.detach().to(backend))
    yd = K.convT2x2_fwd(xd, wd, b.detach().to(backend))
    assert_close(to_nchw_cpu(yd), y, TOL, "convT fwd")
    # the pre-transposed form (round 6: the filter rides in the network's per-step transpose batch) launches the same four parity GEMMs
    wtt = K.conv2d_wt_buffer(wd, wd.device)
    K.conv2d_transpose_weights(wd, wtt, stride=2, pad=0)
    assert torch.equal(K.convT2x2_fwd(xd, wd, b.detach().to(backend), wtt=wtt), yd), "convT fwd from the pre-transposed filter"
    if backend == "cuda":  # the batched job table writes the buffer conv2d_transpose_weights writes
        wtt2 = torch.zeros_like(wtt)
        table = K.conv2d_transpose_jobs(wd, wtt2, stride=2, pad=0)
        K.wtrans_batch(torch.frombuffer(bytearray(table), dtype=torch.uint8).to(backend), len(table) // ctypes.sizeof(_lib.WtransJob))
        n_w = wd.numel()
        assert torch.equal(wtt2[:n_w], wtt[:n_w])
    dyd = to_nhwc(dy, backend)
    assert_close(to_nchw_cpu(K.convT2x2_bwd_data(dyd, wd)), x.grad, TOL, "convT dgrad")
    dw = K.convT_empty(c, k, backend)
    dw.zero_()
    db = torch.zeros(k, device=backend)
    K.convT2x2_bwd_weight(xd, dyd, dw, db)
    assert_close(dw.cpu(), wt.grad, TOL, "convT wgrad")
    assert_close(db.cpu(), b.grad, TOL, "convT dbias")


def test_layout(backend):
    n, c, h, w = _sizes(backend, (3, 3, 37, 41), (2, 3, 5, 4))
    x = torch.randn(n, c, h, w)
    y = K.nchw_to_nhwc(x.to(backend))
    assert y.shape == (n, h, w, 4)
    assert torch.equal(y[..., :3].cpu(), x.permute(0, 2, 3, 1)) and float(y[..., 3].abs().max()) == 0.0
    z = K.nhwc_to_nchw(y[..., :3] if False else y)
    assert torch.equal(z[:, :3].cpu(), x)


@pytest.mark.parametrize("act", ["relu", "silu", None, "relu-many-rows"])
def test_batchnorm_train(backend, act):
    n, h, w, c = _sizes(backend, (4, 40, 40, 96), (2, 5, 6, 8))
    if act == "relu-many-rows":  # > 32 partial rows: the finalize kernels go through the wide fp64 pre-reduction
        act = "relu"
        n, h, w, c = _sizes(backend, (8, 80, 80, 48), (1, 96, 96, 4))
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    eps, mom = 1e-3, 0.03
    rm_ref, rv_ref = rm.clone(), rv.clone()
    pre = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, mom, eps)
    y = {"relu": F.relu, "silu": F.silu, None: lambda t: t}[act](pre)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)

    xd = to_nhwc(x.detach(), backend)
    parts = K.channel_stats_partial(xd)
    rmd, rvd = rm.to(backend), rv.to(backend)
    M = n * h * w
    scale, shift, mean, invstd = K.bn_finalize(parts, M, gamma.detach().to(backend), beta.detach().to(backend), eps, mom, rmd, rvd)
    yd = K.affine_act(xd, scale, shift, act=act)
    assert_close(to_nchw_cpu(yd), y, TOL, "bn fwd")
    assert_close(rmd.cpu(), rm_ref, TOL, "running_mean")
    assert_close(rvd.cpu(), rv_ref, TOL, "running_var")
    dgamma = torch.zeros(c, device=backend)
    dbeta = torch.zeros(c, device=backend)
    dx = K.bn_bwd(to_nhwc(dy, backend), xd, scale, shift, gamma.detach().to(backend), mean, invstd, dgamma, dbeta, act=act)
    assert_close(to_nchw_cpu(dx), x.grad, 5e-5, "bn dx")
    assert_close(dgamma.cpu(), gamma.grad, 5e-5, "bn dgamma")
    assert_close(dbeta.cpu(), beta.grad, 5e-5, "bn dbeta")
    # eval-mode affine
    es, eh = K.bn_eval_scale_shift(gamma.detach().to(backend), beta.detach().to(backend), rmd, rvd, eps)
    ye = K.affine_act(xd, es, eh)
    assert_close(to_nchw_cpu(ye), F.batch_norm(x.detach(), rm_ref, rv_ref, gamma.detach(), beta.detach(), False, mom, eps), TOL, "bn eval")


def test_relu_bwd_bn_reduce_is_the_two_passes(backend):
    """sgx_relu_bwd_bn_reduce (round 6; the ResNet blocks' out = relu(bn(conv) + shortcut)): g and the reduce partials out of ONE sweep are, bit
    for bit, what sgx_relu_bwd followed by sgx_bn_bwd_reduce(act = none) leave - so the BatchNorm backward that takes the partials is the
    unfused one; strided views (a channel slice of a wider buffer) included."""
    from super_gradients_amd._lib import check, lib
    from super_gradients_amd.kernels import ptr, rows, stats_blocks, stream

    for (n, h, w, c), pad in ((_sizes(backend, (4, 28, 28, 256), (2, 5, 6, 8)), 0), (_sizes(backend, (8, 56, 56, 64), (1, 40, 40, 4)), 4)):
        g0 = torch.Generator().manual_seed(5)
        dy, y, x = [torch.randn(n, c, h, w, generator=g0) for _ in range(3)]
        y = F.relu(y)
        mean = torch.randn(c, generator=g0).to(backend)
        dyd, yd, xd = [to_nhwc(t, backend, ld_pix=c + pad) for t in (dy, y, x)]
        g, parts = K.relu_bwd_bn_reduce(dyd, yd, xd, mean)
        g_ref = K.relu_bwd(dyd, yd)
        M = n * h * w
        parts_ref = torch.empty(2, stats_blocks(M), c, device=backend, dtype=torch.float32)
        one, zero = torch.ones(c, device=backend), torch.zeros(c, device=backend)
        check(lib().sgx_bn_bwd_reduce(ptr(g_ref), rows(g_ref)[1], ptr(xd), rows(xd)[1], ptr(one), ptr(zero), ptr(mean), M, c, K.ACT[None], ptr(parts_ref),
                                      stream()), "sgx_bn_bwd_reduce")
        assert torch.equal(g.cpu(), g_ref.cpu()), "g differs from sgx_relu_bwd's"
        assert torch.equal(parts.cpu(), parts_ref.cpu()), "reduce partials differ from sgx_bn_bwd_reduce's"
        assert_close(to_nchw_cpu(g), dy * (y > 0), 0.0, "relu mask")
        gx = (dy * (y > 0)).double()
        assert_close(parts[0].sum(0).cpu(), gx.sum((0, 2, 3)).float(), 1e-4, "sum g")
        assert_close(parts[1].sum(0).cpu(), (gx * (x.double() - mean.cpu().double().view(1, c, 1, 1))).sum((0, 2, 3)).float(), 1e-4, "sum g (x - mean)")


def test_affine_residuals_and_sweeps(backend):
    n, h, w, c = _sizes(backend, (2, 20, 20, 192), (1, 4, 5, 8))
    g = torch.Generator().manual_seed(0)
    x, r1, r2 = [torch.randn(n, c, h, w, generator=g) for _ in range(3)]
    sc, sh = torch.randn(c, generator=g), torch.randn(c, generator=g)
    alpha = torch.tensor([0.7])
    xd, r1d, r2d = [to_nhwc(t, backend, ld_pix=c + 4) for t in (x, r1, r2)]
    y, parts = K.affine_act(xd, sc.to(backend), sh.to(backend), r1=r1d, a1_dev=alpha.to(backend), r2=r2d, a2=1.5, act="relu", want_stats=True)
    pre = x * sc.view(1, c, 1, 1) + sh.view(1, c, 1, 1) + 0.7 * r1 + 1.5 * r2
    assert_close(to_nchw_cpu(y), F.relu(pre), TOL, "affine_act")
    assert_close(parts[0].sum(0).cpu(), pre.sum((0, 2, 3)), 1e-4, "affine stats")
    out = torch.zeros(1, device=backend)
    K.dot_sum(xd, r1d, out, accumulate=False)
    assert_close(out.cpu(), (x * r1).sum().view(1), 1e-4, "dot")
    z = K.axpy(xd, a_dev=alpha.to(backend))
    K.axpy(r1d, a=2.0, out=z, accumulate=True)
    assert_close(to_nchw_cpu(z), 0.7 * x + 2 * r1, TOL, "axpy")
    cs = torch.ones(c, device=backend)
    K.colsum(xd, cs, accumulate=True)
    assert_close(cs.cpu(), x.sum((0, 2, 3)) + 1, 1e-4, "colsum")
    s = K.scale_by_device_scalar(z, alpha.to(backend), torch.tensor([2.0], device=backend))
    assert_close(s.cpu(), z.cpu() * 1.4, TOL, "scale")
    t = torch.empty(1000, device=backend)
    K.fill(t, 3.0)
    assert float(t.min()) == 3.0 and float(t.max()) == 3.0


@pytest.mark.parametrize("k,stride,pad", [(5, 1, 2), (9, 1, 4), (13, 1, 6), (3, 2, 1)])
def test_maxpool(backend, k, stride, pad):
    """Forward values and arg-max routing against ATen (ties on purpose: the first maximum in row-major window order wins), through every
    kernel form: the direct kernels (C = 8 at stride 2; large maps), the LDS-tile forms (stride 1, 8-channel groups) and the scatter-form
    backward (64-channel groups of a map whose gradient slice fits LDS), plain and accumulating."""
    shapes = [(2, 20, 20, 384), (2, 56, 56, 64), (2, 20, 20, 72)] if backend.type == "cuda" else [(1, 7, 6, 8), (1, 6, 5, 64)]
    for n, h, w, c in shapes:
        g = torch.Generator().manual_seed(0)
        x = torch.randn(n, c, h, w, generator=g).round(decimals=1).requires_grad_(True)  # ties on purpose
        y = F.max_pool2d(x, k, stride, pad)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        xd = to_nhwc(x.detach(), backend)
        yd, am = K.maxpool_fwd(xd, k, stride, pad)
        assert torch.equal(to_nchw_cpu(yd), y.detach())
        dx = K.maxpool_bwd(to_nhwc(dy, backend), am, tuple(xd.shape), k, stride, pad)
        assert_close(to_nchw_cpu(dx), x.grad, TOL, f"maxpool bwd {(n, h, w, c)}")
        base = torch.randn(n, c, h, w, generator=g)
        acc = to_nhwc(base, backend)
        K.maxpool_bwd(to_nhwc(dy, backend), am, tuple(xd.shape), k, stride, pad, out=acc, accumulate=True)
        assert_close(to_nchw_cpu(acc), base + x.grad, TOL, f"maxpool bwd accumulate {(n, h, w, c)}")


def test_avgpool(backend):
    n, h, w, c = _sizes(backend, (4, 7, 7, 2048), (2, 3, 3, 8))
    x = torch.randn(n, c, h, w, requires_grad=True)
    y = F.adaptive_avg_pool2d(x, 1).flatten(1)
    dy = torch.randn(y.shape)
    y.backward(dy)
    xd = to_nhwc(x.detach(), backend)
    assert_close(K.avgpool_fwd(xd).cpu(), y, TOL, "avgpool")
    assert_close(to_nchw_cpu(K.avgpool_bwd(dy.to(backend), tuple(xd.shape))), x.grad, TOL, "avgpool bwd")


def _reference_cross_entropy(inputs, target, weight, ignore_index, reduction, smooth_eps):
    """training/losses/label_smoothing_cross_entropy_loss.py:32-83, integer targets on logits (restated; checked live against the reference's
    own function by test_cross_entropy_restatement_live where /root/reference exists)."""
    if not smooth_eps:
        return F.cross_entropy(inputs, target, weight, ignore_index=ignore_index, reduction=reduction)
    lsm = F.log_softmax(inputs, dim=-1)
    masked = target.eq(ignore_index) if ignore_index >= 0 else None
    if weight is not None:
        lsm = lsm * weight.unsqueeze(0)
    likelihood = lsm.gather(dim=-1, index=target.unsqueeze(-1)).squeeze(-1)
    loss = -((1.0 - smooth_eps) * likelihood + smooth_eps * lsm.mean(-1))
    if masked is not None:
        loss = loss.masked_fill(masked, 0)
    if reduction == "sum":
        return loss.sum()
    return loss.mean() if masked is None else loss.sum() / float(loss.size(0) - masked.sum())


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="/root/reference not present (GPU box)")
def test_cross_entropy_restatement_live():
    from oracle import ref_shim

    ref_shim.install()
    from super_gradients.training.losses.label_smoothing_cross_entropy_loss import cross_entropy

    g = torch.Generator().manual_seed(0)
    x, y, w = torch.randn(12, 7, generator=g) * 2, torch.randint(0, 7, (12,), generator=g), torch.rand(7, generator=g) + 0.5
    for eps in (0.0, 0.1):
        for weight in (None, w):
            for ig in (-100, 3):
                for red in ("mean", "sum"):
                    assert torch.equal(cross_entropy(x, y, weight=weight, ignore_index=ig, reduction=red, smooth_eps=eps),
                                       _reference_cross_entropy(x, y, weight, ig, red, eps))


def test_softmax_ce(backend):
    B, Kc = _sizes(backend, (64, 1000), (6, 10))
    g = torch.Generator().manual_seed(1)
    w = torch.rand(Kc, generator=g) + 0.5
    for smoothing in (0.0, 0.1):
        for weight in (None, w):
            for ig in (-100, 2):
                for red in ("mean", "sum"):
                    logits = (torch.randn(B, Kc, generator=g) * 3).requires_grad_(True)
                    labels = torch.randint(0, Kc, (B,), generator=g)
                    labels[1] = 2  # one row that ignore_index = 2 masks
                    if smoothing == 0.0 and ig == -100:
                        labels[3] = -100  # F.cross_entropy's default ignore index
                    loss = _reference_cross_entropy(logits, labels, weight, ig, red, smoothing)
                    loss.backward()
                    l, dl, inv = K.softmax_ce(logits.detach().to(backend), labels.to(backend), smoothing, weight.to(backend) if weight is not None else None, ig, red)
                    what = f"eps {smoothing} weight {weight is not None} ignore {ig} {red}"
                    assert_close(l.cpu().view(1), loss.detach().view(1), TOL, "ce " + what)
                    assert_close((dl * inv).cpu(), logits.grad, TOL, "ce grad " + what)


def test_optimizers(backend):
    n = _sizes(backend, 1_000_003, 1003)
    g = torch.Generator().manual_seed(0)
    p0, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    seg = [n // 3, n]
    wds = [1e-2, 0.0]
    # AdamW: two param groups with different weight decay, 3 steps
    pa, pb = p0[: seg[0]].clone().requires_grad_(True), p0[seg[0]:].clone().requires_grad_(True)
    opt = torch.optim.AdamW([{"params": [pa], "weight_decay": wds[0]}, {"params": [pb], "weight_decay": wds[1]}], lr=2e-3, betas=(0.9, 0.999), eps=1e-8)
    pd, m, v = p0.clone().to(backend), torch.zeros(n, device=backend), torch.zeros(n, device=backend)
    se = torch.tensor(seg, dtype=torch.int64, device=backend)
    sw = torch.tensor(wds, dtype=torch.float32, device=backend)
    for step in range(1, 4):
        gi = gr * step
        pa.grad, pb.grad = gi[: seg[0]].clone(), gi[seg[0]:].clone()
        opt.step()
        K.adamw_step(pd, gi.to(backend), m, v, 2e-3, 0.9, 0.999, 1e-8, step, se, sw)
    assert_close(pd.cpu(), torch.cat([pa.detach(), pb.detach()]), 1e-5, "adamw")
    # SGD momentum
    pa, pb = p0[: seg[0]].clone().requires_grad_(True), p0[seg[0]:].clone().requires_grad_(True)
    opt = torch.optim.SGD([{"params": [pa], "weight_decay": wds[0]}, {"params": [pb], "weight_decay": wds[1]}], lr=0.1, momentum=0.9)
    pd, mom = p0.clone().to(backend), torch.zeros(n, device=backend)
    for step in range(1, 4):
        gi = gr * step
        pa.grad, pb.grad = gi[: seg[0]].clone(), gi[seg[0]:].clone()
        opt.step()
        K.sgd_step(pd, gi.to(backend), mom, 0.1, 0.9, 0.0, False, step == 1, se, sw)
    assert_close(pd.cpu(), torch.cat([pa.detach(), pb.detach()]), 1e-5, "sgd")
    e = p0.clone().to(backend)
    K.ema_update(e, gr.to(backend), 0.9)
    assert_close(e.cpu(), p0 * 0.9 + gr * (1 - 0.9), 1e-6, "ema")


# ------------------------------------------------------------------------------------------------ loss / head / nms
def _head_case(B, hw, C, seed=0, size=None, kmax=6):
    """Synthetic raw head outputs on the anchor grid of a (hw[0][0]*8)-pixel image + targets."""
    from oracle.yolo_nas import make_anchors
    from util import synthetic_targets

    g = torch.Generator().manual_seed(seed)
    anchors, pts, pts_grid, counts, strides = make_anchors(hw, [8, 16, 32])
    L = anchors.shape[0]
    logits = torch.randn(B, L, C, generator=g) * 1.5 - 2.0
    distri = torch.randn(B, L, 68, generator=g) * 1.2
    size = size or hw[0][0] * 8
    targets = synthetic_targets(B, seed=seed, kmax=kmax, size=size, num_classes=C)
    return logits, distri, anchors, pts, pts_grid, counts, strides, targets


@pytest.mark.parametrize("static", [False, True])
@pytest.mark.parametrize("vfl", [True, False])
def test_ppyoloe_loss(backend, static, vfl):
    from oracle.ppyolo_loss import PPYoloELossOracle

    B, hw, C = _sizes(backend, (4, [(40, 40), (20, 20), (10, 10)], 80), (2, [(12, 12), (6, 6), (3, 3)], 8))
    logits, distri, anchors, pts, pts_grid, counts, strides, targets = _head_case(B, hw, C, seed=3)
    if B > 2:
        targets = targets[targets[:, 0] != 1]  # image 1 has no boxes
    logits.requires_grad_(True)
    distri.requires_grad_(True)
    w = (1.0, 2.5, 0.5)
    orc = PPYoloELossOracle(C, use_varifocal_loss=vfl, use_static_assigner=static)
    cls_sum, iou_sum, dfl_sum, score_sum = orc.sums((logits, distri, anchors, pts, counts, strides), targets)
    (w[0] * cls_sum + w[1] * iou_sum + w[2] * dfl_sum).backward()
    _, a_label, a_box, a_score = orc.assign((logits.detach(), distri.detach(), anchors, pts, counts, strides), targets)

    dev = backend
    out = K.ppyoloe_loss_fwd(logits.detach().to(dev), distri.detach().to(dev), anchors.to(dev), pts.to(dev), strides.to(dev), targets.to(dev), counts,
                             static, vfl, w)
    assert torch.equal(out["label"].cpu().long(), a_label), "assigned labels differ"
    pos = a_label != C
    assert int(pos.sum()) > 0
    assert_close(out["box"].cpu()[pos], a_box[pos], 1e-6, "assigned boxes")
    assert_close(out["score"].cpu(), a_score, 2e-5, "assigned scores")
    ref = torch.stack([cls_sum, iou_sum, dfl_sum, score_sum]).detach()
    for i, name in enumerate(["cls", "iou", "dfl", "score"]):
        assert_close(out["sums"].cpu()[i:i + 1], ref[i:i + 1], 2e-5, f"sum {name}")
    assert_close(out["g_logits"].cpu(), logits.grad, 5e-5, "g_logits")
    assert_close(out["g_distri"].cpu(), distri.grad, 5e-5, "g_distri")
    items, inv = K.ppyoloe_loss_finalize(out["sums"], w, 1.0)
    loss, log_items = orc((None, (logits, distri, anchors, pts, counts, strides)), targets)
    assert_close(items.cpu(), log_items, 2e-5, "loss items")


@pytest.mark.parametrize("C", [5, 6, 17])
def test_ppyoloe_loss_any_class_count(backend, C):
    """Class counts that are not a multiple of 4 (scalar classification-loss kernel): same parity bar as the vector path."""
    from oracle.ppyolo_loss import PPYoloELossOracle

    B, hw = 2, [(12, 12), (6, 6), (3, 3)]
    logits, distri, anchors, pts, pts_grid, counts, strides, targets = _head_case(B, hw, C, seed=4)
    logits.requires_grad_(True)
    distri.requires_grad_(True)
    w = (1.0, 2.5, 0.5)
    for static in (False, True):
        logits.grad = distri.grad = None
        orc = PPYoloELossOracle(C, use_varifocal_loss=True, use_static_assigner=static)
        cls_sum, iou_sum, dfl_sum, score_sum = orc.sums((logits, distri, anchors, pts, counts, strides), targets)
        (w[0] * cls_sum + w[1] * iou_sum + w[2] * dfl_sum).backward()
        _, a_label, _, _ = orc.assign((logits.detach(), distri.detach(), anchors, pts, counts, strides), targets)
        out = K.ppyoloe_loss_fwd(logits.detach().to(backend), distri.detach().to(backend), anchors.to(backend), pts.to(backend), strides.to(backend),
                                 targets.to(backend), counts, static, True, w)
        assert torch.equal(out["label"].cpu().long(), a_label)
        ref = torch.stack([cls_sum, iou_sum, dfl_sum, score_sum]).detach()
        assert_close(out["sums"].cpu(), ref, 2e-5, f"sums C={C}")
        assert_close(out["g_logits"].cpu(), logits.grad, 5e-5, "g_logits")
        assert_close(out["g_distri"].cpu(), distri.grad, 5e-5, "g_distri")
    scores = K.dfl_decode(logits.detach().to(backend), distri.detach().to(backend), pts_grid.to(backend), strides.to(backend), 16)[1]
    assert_close(scores.cpu(), logits.detach().sigmoid(), 2e-5, "scores")


def test_ppyoloe_loss_empty_targets(backend):
    from oracle.ppyolo_loss import PPYoloELossOracle

    B, hw, C = 2, [(12, 12), (6, 6), (3, 3)], 8
    logits, distri, anchors, pts, pts_grid, counts, strides, _ = _head_case(B, hw, C)
    targets = torch.zeros(0, 6)
    for static in (False, True):
        orc = PPYoloELossOracle(C, use_static_assigner=static)
        loss, items = orc((None, (logits, distri, anchors, pts, counts, strides)), targets)
        out = K.ppyoloe_loss_fwd(logits.to(backend), distri.to(backend), anchors.to(backend), pts.to(backend), strides.to(backend), targets.to(backend),
                                 counts, static, True, (1.0, 2.5, 0.5))
        it, _ = K.ppyoloe_loss_finalize(out["sums"], (1.0, 2.5, 0.5), 1.0)
        assert_close(it.cpu(), items, 2e-5, "empty-target loss items")
        assert float(it[1]) == 0.0 and float(it[2]) == 0.0
        assert int((out["label"] != C).sum()) == 0


def test_dfl_decode(backend):
    B, hw, C = _sizes(backend, (3, [(40, 40), (20, 20), (10, 10)], 80), (2, [(4, 4), (2, 2), (1, 1)], 8))
    logits, distri, anchors, pts, pts_grid, counts, strides, _ = _head_case(B, hw, C)
    from oracle.ppyolo_loss import decode_distribution

    boxes_ref = decode_distribution(pts_grid, distri) * strides
    boxes, scores = K.dfl_decode(logits.to(backend), distri.to(backend), pts_grid.to(backend), strides.to(backend), 16)
    assert_close(boxes.cpu(), boxes_ref, 2e-5, "decoded boxes")
    assert_close(scores.cpu(), logits.sigmoid(), 2e-5, "scores")


def _nms_case(B, L, C, seed, clusters=12, size=640.0):
    g = np.random.RandomState(seed)
    cen = g.uniform(0.15 * size, 0.85 * size, (B, clusters, 2))
    which = g.randint(0, clusters, (B, L))
    c = np.take_along_axis(cen, which[..., None].repeat(2, -1), 1) + g.normal(0, 6, (B, L, 2))
    wh = g.uniform(20, 120, (B, L, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32)
    scores = g.beta(0.5, 0.5, (B, L, C)).astype(np.float32) ** 4
    # engineered exact ties in score and duplicated boxes
    scores[:, 1::7] = scores[:, 0:-1:7][:, : scores[:, 1::7].shape[1]]
    boxes[:, 2::11] = boxes[:, 0:-2:11][:, : boxes[:, 2::11].shape[1]]
    return torch.from_numpy(boxes), torch.from_numpy(scores)


def test_nms_against_torchvision_fixture(backend):
    """The pin of the NMS arithmetic to REAL torchvision (pp_yolo_e/post_prediction_callback.py:85,87 call torchvision.ops.nms /
    batched_nms; requirements.txt:12): tests/golden/nms_torchvision.pt is written by oracle/make_nms_golden.py wherever torchvision is
    importable.  With the fixture present, both the C restatement (oracle/nms.c) and the HIP kernel must reproduce torchvision's kept
    indices bit for bit.  Without it this test SKIPS and the NMS parity stays "unpinned" (DESIGN.md section 4)."""
    import os

    from oracle import nms as onms

    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nms_torchvision.pt")
    if not os.path.exists(f):
        pytest.skip("NMS parity UNPINNED: tests/golden/nms_torchvision.pt absent (torchvision is not installed here; run oracle/make_nms_golden.py where it is)")
    fx = torch.load(f)
    for c in fx["cases"]:
        boxes, scores, classes, thr = c["boxes"], c["scores"], c["classes"], c["iou"]
        n = boxes.shape[0]
        assert torch.equal(onms.nms(boxes, scores, thr), c["keep"]), "oracle/nms.c vs torchvision.ops.nms"
        assert torch.equal(onms.batched_nms(boxes, scores, classes, thr), c["keep_batched"]), "oracle batched_nms vs torchvision"
        if n == 0:
            continue
        cap = max(n, 1)
        out, cnt, idx, _ = K.nms(boxes[None].to(backend), scores[None, :, None].to(backend), 0.0, thr, cap, cap, multi_label=True, class_mode=0)
        assert torch.equal(idx[0, : int(cnt[0])].cpu().long(), c["keep"]), "HIP nms vs torchvision.ops.nms"
        ncls = int(classes.max()) + 1
        sc = torch.zeros(n, ncls)
        sc[torch.arange(n), classes] = scores
        out, cnt, idx, _ = K.nms(boxes[None].to(backend), sc[None].to(backend), 0.0, thr, cap, cap, multi_label=True, class_mode=3)
        assert torch.equal(idx[0, : int(cnt[0])].cpu().long() // ncls, c["keep_batched"]), "HIP batched nms vs torchvision.ops.batched_nms"


def test_nms_boundary_ties_overflow_the_candidate_list(backend):
    """More candidates share the k-th key's 22-bit prefix than stage 1's per-image list holds (all scores equal: the whole image sits in the
    boundary bin): stage 2 must fall back to streaming the raw scores and still return the stable-sort answer (lowest candidate index first)."""
    from oracle import nms as onms

    B, L, C = 2, 130, 80  # 10 400 equal-score candidates per image > NMS_LIST_CAP = 8192
    g = np.random.RandomState(3)
    c = g.uniform(50, 590, (B, L, 2))
    wh = g.uniform(10, 60, (B, L, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32))
    scores = torch.full((B, L, C), 0.5)
    scores[1, :, 1::2] = 0.25  # image 1: 5 200 candidates at 0.5 (fits the list), the rest below them
    kw = dict(score_threshold=0.1, nms_threshold=0.6, nms_top_k=64, max_predictions=20, multi_label_per_box=True)
    ref = onms.post_prediction(boxes, scores, class_agnostic_nms=True, **kw)
    out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), 0.1, 0.6, 64, 20, multi_label=True, class_mode=0)
    for b in range(B):
        assert torch.equal(out[b, : int(cnt[b])].cpu(), ref[b]), f"image {b}"
    assert int(ncand[0]) == 64 and int(idx[0, 0]) == 0


@pytest.mark.parametrize("multi_label,class_mode", [(True, 0), (False, 0), (True, 1), (False, 2), (True, 2)])
def test_nms(backend, multi_label, class_mode):
    from oracle import nms as onms

    B, L, C, topk, maxp = _sizes(backend, (4, 8400, 80, 1000, 300), (2, 84, 4, 40, 12))
    boxes, scores = _nms_case(B, L, C, seed=5)
    thr = 0.25 if multi_label else 0.05
    if class_mode == 1:
        topk = min(topk, 1000)  # coordinate-offset trick applies while 4*K <= 4000
    kw = dict(score_threshold=thr, nms_threshold=0.6, nms_top_k=topk, max_predictions=maxp, multi_label_per_box=multi_label)
    if class_mode == 0:
        ref = onms.post_prediction(boxes, scores, class_agnostic_nms=True, **kw)
    elif class_mode == 1:
        ref = onms.post_prediction(boxes, scores, class_agnostic_nms=False, **kw)
    else:
        ref = onms.post_prediction(boxes, scores, class_agnostic_nms=False, force_vanilla=True, **kw)
    out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), thr, 0.6, topk, maxp, multi_label=multi_label, class_mode=class_mode)
    out, cnt = out.cpu(), cnt.cpu()
    for b in range(B):
        n = int(cnt[b])
        assert n == ref[b].shape[0], f"image {b}: kept {n} vs oracle {ref[b].shape[0]}"
        assert torch.equal(out[b, :n], ref[b]), f"image {b}: rows differ"  # bit-exact boxes/scores/classes


@pytest.mark.parametrize("case", ["random", "sample_overestimates", "sample_underestimates", "few_candidates"])
def test_nms_sampled_selection(backend, case):
    """Round 5: stage 1 selects behind a threshold estimated from a 1/32 line sample of the scores (one pass instead of three); stage 2 stays
    exact - it takes the list only if it holds at least min(k, candidates) keys and did not overflow, else it streams the image.  The rows
    must equal the oracle's AND the exact three-pass selection's bit for bit: on a random case; when every large score sits in a SAMPLED
    line (threshold estimated too high, list shorter than k: fallback); when none does (threshold too low: a long list or an overflow);
    and with fewer candidates than the sample rank asks for (threshold 0: everything selected)."""
    from oracle import nms as onms
    from super_gradients_amd._lib import lib

    B, L, C, topk, maxp = _sizes(backend, (3, 4200, 80, 1000, 300), (1, 128, 80, 64, 20))
    boxes, scores = _nms_case(B, L, C, seed=21)
    E = L * C
    flat = scores.reshape(B, E)
    line = torch.arange(E) // 32
    if case == "sample_overestimates":      # large scores only in the sampled lines (0, 32, 64, ...), a sea of medium ones elsewhere
        flat[:] = torch.where((line % 32 == 0)[None], 0.6 + 0.39 * flat, 0.3 * flat)
    elif case == "sample_underestimates":   # the sampled lines hold nothing above the score threshold's neighbourhood
        flat[:] = torch.where((line % 32 == 0)[None], 0.06 + 0.001 * flat, flat)
    elif case == "few_candidates":
        flat[:] = torch.where(flat > 0.97, flat, torch.zeros_like(flat))
    scores = flat.reshape(B, L, C)
    kw = dict(score_threshold=0.05, nms_threshold=0.6, nms_top_k=topk, max_predictions=maxp, multi_label_per_box=True)
    ref = onms.post_prediction(boxes, scores, class_agnostic_nms=True, **kw)
    res = {}
    try:
        for sampled in (1, 0):
            assert lib().sgx_debug_set_nms_selection(sampled) == 0
            out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), 0.05, 0.6, topk, maxp, multi_label=True, class_mode=0)
            res[sampled] = (out.cpu(), cnt.cpu(), idx.cpu(), ncand.cpu())
            if sampled:
                fallbacks = K.nms_fallbacks()
    finally:
        lib().sgx_debug_set_nms_selection(1)
    #  the streaming fallback of stage 2 is exact but slow, and it used to be silent: the call reports it per image
    if case == "random":
        assert fallbacks == 0, "the sampled selection fell back to streaming on a benign input"
    elif case == "sample_overestimates":
        assert fallbacks == B, f"{fallbacks} of {B} images report the fallback this case forces"
    for b in range(B):
        n = int(res[1][1][b])
        assert n == ref[b].shape[0], f"{case} image {b}: kept {n} vs oracle {ref[b].shape[0]}"
        assert torch.equal(res[1][0][b, :n], ref[b]), f"{case} image {b}: rows differ from the oracle"
    for a, c_ in zip(res[1], res[0]):
        assert torch.equal(a, c_), f"{case}: sampled and exact selection disagree"


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 128, 129, 200])
def test_nms_candidate_count_boundaries(backend, n):
    """Candidate counts around the 64-candidate flag words of the suppression matrix (empty list, one candidate, a full word, one past it,
    two words, ...): the split suppression's work items, the walk's register-resident diagonal words and its tail mask all change shape
    there.  Three images with different counts in one call (n, n + 1 clamped, 0), class-agnostic, against the oracle bit for bit."""
    from oracle import nms as onms

    L = max(n + 1, 2)
    g = np.random.RandomState(100 + n)
    c = g.uniform(100, 540, (3, L, 2)) + g.normal(0, 4, (3, L, 2))
    wh = g.uniform(40, 160, (3, L, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32))
    scores = torch.from_numpy(g.uniform(0.2, 0.9, (3, L, 1)).astype(np.float32))
    scores[0, n:] = 0.0       # image 0: exactly n candidates
    scores[2] = 0.0           # image 2: none
    kw = dict(score_threshold=0.1, nms_threshold=0.5, nms_top_k=max(L, 1), max_predictions=max(L, 1), multi_label_per_box=True)
    ref = onms.post_prediction(boxes, scores, class_agnostic_nms=True, **kw)
    out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), 0.1, 0.5, max(L, 1), max(L, 1), multi_label=True, class_mode=0)
    assert int(ncand[0]) == n and int(ncand[1]) == L and int(ncand[2]) == 0
    for b in range(3):
        k = int(cnt[b])
        assert k == ref[b].shape[0], f"n={n} image {b}: kept {k} vs oracle {ref[b].shape[0]}"
        assert torch.equal(out[b, :k].cpu(), ref[b]), f"n={n} image {b}: rows differ"


def test_nms_per_class_case_is_reproducible(backend):
    """A per-class case (detections of a random-init YOLO-NAS on two 64x64 images: 252 candidates, scores within 0.0093..0.0106, boxes far
    larger than the image) on which the chunked suppression walk keeps candidates in all four flag words.  Under the host emulation - one OS
    thread per lane, no wave lock-step - the walk once read the kept-count after lane 0 of the same wave had advanced it; the rows must equal
    the oracle's on every repetition (three per per-class mode)."""
    from oracle import nms as onms

    case = torch.load(os.path.join(os.path.dirname(__file__), "golden", "nms_perclass_case.pt"))
    boxes, scores = case["boxes"], case["scores"]
    for class_mode, agnostic in ((1, False), (2, False)):
        ref = onms.post_prediction(boxes, scores, score_threshold=0.0, nms_threshold=0.6, nms_top_k=200, max_predictions=20, multi_label_per_box=True,
                                   class_agnostic_nms=agnostic)
        for rep in range(3):
            out, cnt, idx, _ = K.nms(boxes.to(backend), scores.to(backend), 0.0, 0.6, 200, 20, multi_label=True, class_mode=class_mode)
            for b in range(2):
                n = int(cnt[b])
                assert n == ref[b].shape[0] and torch.equal(out[b, :n].cpu(), ref[b]), f"mode {class_mode} repetition {rep} image {b}"


def test_nms_large_topk(backend):
    """nms_top_k above 1024 (the 2048 / 4096-candidate instantiations): several candidates per thread in the sort and the scan."""
    from oracle import nms as onms

    B, L, C, topk, maxp = _sizes(backend, (2, 8400, 80, 3000, 1200), (1, 500, 4, 1500, 1100))
    boxes, scores = _nms_case(B, L, C, seed=9, clusters=40)
    kw = dict(score_threshold=0.02, nms_threshold=0.6, nms_top_k=topk, max_predictions=maxp, multi_label_per_box=True)
    modes = ((0, True), (3, False)) if backend.type == "cuda" else ((3, False),)  # host emulation: one mode (1024 OS threads per launch)
    for class_mode, agnostic in modes:
        ref = onms.post_prediction(boxes, scores, class_agnostic_nms=agnostic, **kw)
        out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), 0.02, 0.6, topk, maxp, multi_label=True, class_mode=class_mode)
        assert int(ncand.max()) > 1024, "case must exceed the 1024-candidate kernel"
        for b in range(B):
            n = int(cnt[b])
            assert n == ref[b].shape[0], f"mode {class_mode} image {b}: kept {n} vs oracle {ref[b].shape[0]}"
            assert torch.equal(out[b, :n].cpu(), ref[b]), f"mode {class_mode} image {b}: rows differ"


# --------------------------------------------------------------------------------------------- PP-YOLOE pieces
@pytest.mark.parametrize("act", ["silu", "relu", None])
def test_dual_affine_act(backend, act):
    """RepVGG two-branch BN sum + activation (+ post-activation residual), and the gradient through the activation."""
    n, h, w, c = _sizes(backend, (3, 37, 29, 96), (2, 5, 3, 8))
    g = torch.Generator().manual_seed(11)
    x1, x2, r, dy = (torch.randn(n, c, h, w, generator=g) for _ in range(4))
    s1, t1, s2, t2 = (torch.randn(c, generator=g) for _ in range(4))
    f = {"silu": F.silu, "relu": F.relu, None: lambda v: v}[act]
    v = lambda t: t.view(1, c, 1, 1)  # noqa: E731
    pre = (x1 * v(s1) + v(t1) + x2 * v(s2) + v(t2)).requires_grad_(True)
    ref = f(pre) + r
    (gref,) = torch.autograd.grad(ref, pre, dy)
    d = lambda t: t.to(backend)  # noqa: E731
    out = empty_nhwc(n, h, w, c, backend, ld_pix=c + 8, c_off=4)
    K.dual_affine_act(to_nhwc(x1, backend, ld_pix=c + 4), d(s1), d(t1), to_nhwc(x2, backend), d(s2), d(t2), post_add=to_nhwc(r, backend), act=act, out=out)
    assert_close(to_nchw_cpu(out), ref.detach(), TOL, f"dual_affine_act {act}")
    gg = K.dual_affine_act_bwd(to_nhwc(dy, backend), to_nhwc(x1, backend), d(s1), d(t1), to_nhwc(x2, backend, ld_pix=c + 12, c_off=8), d(s2), d(t2), act=act)
    assert_close(to_nchw_cpu(gg), gref, TOL, f"dual_affine_act_bwd {act}")
    # single branch + post-activation residual (pp_yolo_head.py:205)
    ref1 = f(x1 * v(s1) + v(t1)) + r
    y1 = K.dual_affine_act(to_nhwc(x1, backend), d(s1), d(t1), post_add=to_nhwc(r, backend), act=act)
    assert_close(to_nchw_cpu(y1), ref1, TOL, f"single affine + post add {act}")


@pytest.mark.parametrize("gate", ["hardsigmoid", "sigmoid"])
def test_se_gate(backend, gate):
    """mean over H*W, x * f(pre), and both gradients (EffectiveSEBlock se_blocks.py:39-42, ESEAttn pp_yolo_head.py:90-92)."""
    n, h, w, c = _sizes(backend, (3, 45, 37, 192), (2, 6, 5, 8))   # 1665 pixels: several 512-pixel chunks on the GPU
    g = torch.Generator().manual_seed(12)
    x = torch.randn(n, c, h, w, generator=g).requires_grad_(True)
    pre = (torch.randn(n, c, generator=g) * 3).requires_grad_(True)
    dy = torch.randn(n, c, h, w, generator=g)
    f = {"hardsigmoid": F.hardsigmoid, "sigmoid": torch.sigmoid}[gate]
    y = x * f(pre).view(n, c, 1, 1)
    gx, gpre = torch.autograd.grad(y, (x, pre), dy)
    xd, dyd, pd = to_nhwc(x.detach(), backend, ld_pix=c + 4), to_nhwc(dy, backend), pre.detach().to(backend)
    mean = K.image_colsum(xd, scale=1.0 / (h * w))
    assert_close(mean.cpu(), x.detach().mean((2, 3)), TOL, "image mean")
    yd = K.channel_gate(xd, pd, gate)
    assert_close(to_nchw_cpu(yd), y.detach(), TOL, f"gate fwd {gate}")
    dpre = K.image_colsum(dyd, v=xd, pre=pd, gate=gate)
    assert_close(dpre.cpu(), gpre, 5e-5, f"gate dpre {gate}")
    bias = torch.randn(n, c, generator=g)
    base = torch.randn(n, c, h, w, generator=g)
    acc = to_nhwc(base, backend, ld_pix=c + 8, c_off=4)
    K.channel_gate(dyd, pd, gate, bias=bias.to(backend), bias_scale=0.25, out=acc, accumulate=True)
    assert_close(to_nchw_cpu(acc), base + gx + 0.25 * bias.view(n, c, 1, 1), TOL, f"gate bwd (accumulate + bias) {gate}")
    inplace = to_nhwc(dy, backend)
    K.channel_gate(inplace, pd, gate, out=inplace)
    assert_close(to_nchw_cpu(inplace), gx, TOL, f"gate in place {gate}")


def test_upsample2x(backend):
    n, h, w, c = _sizes(backend, (2, 20, 24, 96), (2, 3, 4, 8))
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, c, h, w, generator=g).requires_grad_(True)
    ref = F.interpolate(x, scale_factor=2, mode="nearest")
    dy = torch.randn(ref.shape, generator=g)
    (gx,) = torch.autograd.grad(ref, x, dy)
    out = empty_nhwc(n, 2 * h, 2 * w, c, backend, ld_pix=c + 8)
    K.upsample2x_fwd(to_nhwc(x.detach(), backend, ld_pix=c + 4, c_off=4), out=out)
    assert torch.equal(to_nchw_cpu(out), ref.detach()), "nearest up-sampling is a copy: bit-exact"
    dx = K.upsample2x_bwd(to_nhwc(dy, backend, ld_pix=c + 4))
    assert_close(to_nchw_cpu(dx), gx, 1e-6, "upsample bwd")
    base = torch.randn(n, c, h, w, generator=g)
    acc = to_nhwc(base, backend)
    K.upsample2x_bwd(to_nhwc(dy, backend), out=acc, accumulate=True)
    assert_close(to_nchw_cpu(acc), base + gx, 1e-6, "upsample bwd accumulate")


def test_standardize_u8_and_device_collate(backend):
    """Device side of the input pipeline: uint8 HWC -> standardized NHWC fp32, bit-identical to the reference's host
    arithmetic (DetectionStandardize: numpy image / 255 -> float32) for EVERY uint8 value; the collate's view is consumed zero-copy."""
    from super_gradients_amd.training.utils.collate_fn import DetectionCollateFN, DeviceDetectionCollateFN

    allv = torch.arange(256, dtype=torch.uint8).reshape(1, 16, 16, 1).repeat(1, 1, 1, 3)
    y = K.standardize_u8(allv.to(backend))
    ref = torch.from_numpy((allv.numpy() / 255.0).astype(np.float32))
    assert torch.equal(y[..., :3].cpu(), ref) and float(y[..., 3].abs().max()) == 0.0
    g = torch.Generator().manual_seed(5)
    n, h, w = _sizes(backend, (4, 96, 80), (2, 6, 5))
    items = [(torch.randint(0, 256, (h, w, 3), dtype=torch.uint8, generator=g).numpy(), np.random.RandomState(i).rand(i + 1, 5).astype(np.float32))
             for i in range(n)]
    mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
    x_dev, t_dev = DeviceDetectionCollateFN(device=backend, mean=mean, std=std)(items)
    host = [((img / 255.0).astype(np.float32), t) for img, t in items]   # DetectionStandardize on the host ...
    x_ref, t_ref = DetectionCollateFN()(host)                            # ... then the reference collate
    x_ref = (x_ref - torch.tensor(mean).view(1, 3, 1, 1)) / torch.tensor(std).view(1, 3, 1, 1)
    assert tuple(x_dev.shape) == tuple(x_ref.shape) == (n, 3, h, w)
    assert_close(x_dev.cpu(), x_ref, 1e-6, "device collate images")
    assert torch.equal(t_dev.cpu(), t_ref) and t_ref.shape == (sum(i + 1 for i in range(n)), 6)
    nhwc = K.input_to_nhwc(x_dev)
    assert nhwc.data_ptr() == x_dev.data_ptr() and tuple(nhwc.shape) == (n, h, w, 4), "the model entrance must reuse the collated buffer"
    plain = K.input_to_nhwc(x_ref.to(backend))   # a plain NCHW batch takes the re-layout kernel
    assert torch.equal(plain.cpu(), nhwc.cpu()) or rel_err(plain.cpu(), nhwc.cpu()) < 1e-6


def test_device_collate_pads_ragged_images(backend):
    """SURVEY 8f-4, device-side pad + collate: images of different sizes -> one padded, standardized batch; against the reference's host chain
    (DetectionPadIfNeeded with center / bottom-right padding coordinates, transforms/utils.py:79-133,155-166, then DetectionStandardize and
    the collate) restated with numpy - bit-identical pixels, shifted boxes."""
    from super_gradients_amd.training.utils.collate_fn import DetectionCollateFN, DeviceDetectionCollateFN

    H, W = _sizes(backend, (96, 128), (9, 12))
    rs = np.random.RandomState(3)
    sizes = [(H, W), (H - 3, W - 5), (H // 2, W // 3), (1, 1)]
    items = [(rs.randint(0, 256, (h, w, 3)).astype(np.uint8), np.concatenate([rs.randint(0, 5, (i + 1, 1)), rs.rand(i + 1, 4) * min(h, w)], 1).astype(np.float32))
             for i, (h, w) in enumerate(sizes)]
    for mode in ("center", "bottom_right"):
        x_dev, t_dev = DeviceDetectionCollateFN(device=backend, pad_to=(H, W), pad_value=114, padding_mode=mode)(items)
        host = []
        for img, t in items:
            h, w = img.shape[:2]
            ph, pw = H - h, W - w
            top, left = (ph // 2, pw // 2) if mode == "center" else (0, 0)
            padded = np.pad(img, ((top, ph - top), (left, pw - left), (0, 0)), mode="constant", constant_values=114)
            t = t.copy()
            t[:, 1] += left
            t[:, 2] += top
            host.append(((padded / 255.0).astype(np.float32), t))
        x_ref, t_ref = DetectionCollateFN()(host)
        assert torch.equal(x_dev.cpu(), x_ref), f"{mode}: padded pixels"
        assert torch.equal(t_dev.cpu(), t_ref.float()), f"{mode}: shifted targets"
    with pytest.raises(ValueError, match="larger than pad_to"):
        DeviceDetectionCollateFN(device=backend, pad_to=(H - 1, W))(items)


def test_device_collate_padded_rescale(backend):
    """the reference's last image transforms of the YOLO-NAS dataset recipe - DetectionPaddedRescale (r = min(H/h, W/w), resize to
    (int(h r), int(w r)), bottom-right pad 114, boxes * r; transforms.py:945-975, transforms/utils.py:202-227), DetectionStandardize and the
    collate - as ONE launch over the ragged uint8 batch, against oracle/image.py (whose resize restates cv2's arithmetic: unpinned)."""
    from oracle import image as O
    from super_gradients_amd.training.utils.collate_fn import DeviceDetectionCollateFN

    H, W = _sizes(backend, (96, 128), (24, 32))
    rs = np.random.RandomState(5)
    sizes = [(H, W), (2 * H, 2 * W), (H + 7, W // 2), (H // 3, W - 3), (3 * H, W)]
    items = [(rs.randint(0, 256, (h, w, 3)).astype(np.uint8), np.concatenate([rs.randint(0, 5, (i + 1, 1)), rs.rand(i + 1, 4) * min(h, w)], 1).astype(np.float32))
             for i, (h, w) in enumerate(sizes)]
    x_dev, t_dev = DeviceDetectionCollateFN(device=backend, rescale_to=(H, W), pad_value=114)(items)
    assert tuple(x_dev.shape) == (len(items), 3, H, W)
    rows = []
    for i, (img, t) in enumerate(items):
        h, w = img.shape[:2]
        r = min(H / h, W / w)
        small = O.resize_linear_u8(img, (int(h * r), int(w * r)))
        ref = O.standardize(O.pad(small, O.bottom_right_padding(small.shape[:2], (H, W)), 114)).transpose(2, 0, 1)
        assert np.array_equal(x_dev[i].cpu().numpy(), ref), f"image {i} ({h}x{w} -> {small.shape[:2]})"
        boxes = O.rescale_boxes(t[:, 1:], (r, r))
        rows.append(np.concatenate([np.full((len(t), 1), i, np.float32), t[:, :1], boxes], 1))
    assert np.array_equal(t_dev.cpu().numpy(), np.concatenate(rows, 0))
    with pytest.raises(ValueError, match="not both"):
        DeviceDetectionCollateFN(device=backend, rescale_to=(H, W), pad_to=(H, W))


def test_wtrans_batch_equals_per_conv_transposes(backend):
    """sgx_conv2d_transpose_jobs + ONE sgx_wtrans_batch launch == the per-convolution sgx_conv2d_transpose_weights launches (bit-exact),
    over 1x1 / 3x3 / stride-2 / 7x7-stride-2 filters (1, 1, 4 and 4 output-parity classes)."""
    import ctypes

    from super_gradients_amd import _lib

    cfgs = [(8, 4, 1, 1, 0), (12, 8, 3, 1, 1), (8, 8, 3, 2, 1), (4, 4, 7, 2, 3), (8, 12, 1, 2, 0)]   # (K, C, R, stride, pad)
    g = torch.Generator().manual_seed(3)
    ws, singles, batched, table = [], [], [], b""
    for k, c, r, s, p in cfgs:
        w = K.to_ohwi(torch.randn(k, c, r, r, generator=g).to(backend))
        a, b = K.conv2d_wt_buffer(w, backend).fill_(-1.0), K.conv2d_wt_buffer(w, backend).fill_(-1.0)
        K.conv2d_transpose_weights(w, a, stride=s, pad=p)
        table += K.conv2d_transpose_jobs(w, b, stride=s, pad=p)
        ws.append(w), singles.append(a), batched.append(b)
    n = len(table) // ctypes.sizeof(_lib.WtransJob)
    assert n == 1 + 1 + 4 + 4 + 1   # a 1x1 stride-2 filter reaches one parity class only
    K.wtrans_batch(torch.frombuffer(bytearray(table), dtype=torch.uint8).to(backend), n)
    for (k, c, r, s, p), a, b in zip(cfgs, singles, batched):
        assert torch.equal(a.cpu(), b.cpu()), f"batched transpose differs for K={k} C={c} R={r} s={s}"


def _adversarial_targets(kind, size, C):
    """Target sets built to hit the assigners' tie / degenerate branches (ppyolo_loss.py:165-230, 301-434, 454-561)."""
    s = float(size)
    if kind == "duplicates":       # the same box twice (same class), and once more with another class: exact metric ties between GTs
        rows = [[0, 1, s / 2, s / 2, s / 3, s / 3], [0, 1, s / 2, s / 2, s / 3, s / 3], [0, 2, s / 2, s / 2, s / 3, s / 3], [1, 0, s / 4, s / 4, s / 5, s / 5]]
    elif kind == "nested":         # boxes nested in each other: one anchor is a candidate of several GTs (max-IoU resolution)
        # (centres kept off the anchor lattice's symmetry axes: see test_atss_distance_tie_policy for exact distance ties)
        cx, cy = s / 2 + 1.3, s / 2 - 0.7
        rows = [[0, 0, cx, cy, s * 0.9, s * 0.9], [0, 1, cx, cy, s * 0.5, s * 0.5], [0, 2, cx, cy, s * 0.2, s * 0.2],
                [1, 3, cx, cy, s * 0.6, s * 0.3], [1, 3, cx, cy, s * 0.3, s * 0.6]]
    elif kind == "tiny_and_outside":  # a box smaller than a stride-8 cell between anchor centres, and boxes sticking out of the image
        rows = [[0, 4, 12.0, 12.0, 3.0, 3.0], [0, 5, 2.0, s / 2, 40.0, 30.0], [1, 6, s - 3.0, s - 3.0, 30.0, 30.0], [1, 7, s / 2, s / 2, 2 * s, 2 * s]]
    elif kind == "many":           # more GTs than anchors of the coarsest level, on a regular lattice (equal distances / IoUs everywhere)
        rows = [[b, (i * 4 + j) % C, (i + 0.5) * s / 4, (j + 0.5) * s / 4, s / 5, s / 5] for b in range(2) for i in range(4) for j in range(3 + b)]
    elif kind == "one_image_empty":
        rows = [[1, 0, s / 2, s / 2, s / 2, s / 3], [1, 0, s / 3, s / 2, s / 4, s / 3]]
    else:
        raise KeyError(kind)
    return torch.tensor(rows, dtype=torch.float32)


@pytest.mark.parametrize("kind", ["duplicates", "nested", "tiny_and_outside", "many", "one_image_empty"])
def test_ppyoloe_assignment_adversarial(backend, kind):
    """Bit-exact labels / boxes and matching scores, sums and gradients on target sets built to hit exact ties and degenerate GTs,
    TAL and ATSS.  Zero logits make every anchor's class scores equal, so candidate ranking is decided by IoU ties and index order."""
    from oracle.ppyolo_loss import PPYoloELossOracle

    B, hw, C = 2, [(12, 12), (6, 6), (3, 3)], 8
    logits, distri, anchors, pts, pts_grid, counts, strides, _ = _head_case(B, hw, C, seed=5)
    logits = torch.zeros_like(logits) if kind in ("duplicates", "many") else logits
    targets = _adversarial_targets(kind, hw[0][0] * 8, C)
    w = (1.0, 2.5, 0.5)
    for static in (False, True):
        orc = PPYoloELossOracle(C, use_varifocal_loss=True, use_static_assigner=static)
        _, a_label, a_box, a_score = orc.assign((logits, distri, anchors, pts, counts, strides), targets)
        sums = torch.stack(orc.sums((logits, distri, anchors, pts, counts, strides), targets)).detach()
        out = K.ppyoloe_loss_fwd(logits.to(backend), distri.to(backend), anchors.to(backend), pts.to(backend), strides.to(backend), targets.to(backend),
                                 counts, static, True, w)
        assert torch.equal(out["label"].cpu().long(), a_label), f"{kind} static={static}: assigned labels differ"
        pos = a_label != C
        if int(pos.sum()):
            assert torch.equal(out["box"].cpu()[pos], a_box[pos]), f"{kind} static={static}: assigned boxes differ"
        assert_close(out["score"].cpu(), a_score, 2e-5, f"{kind} static={static}: assigned scores")
        for i, name in enumerate(["cls", "iou", "dfl", "score"]):
            if float(sums[i].abs()) > 0:
                assert_close(out["sums"].cpu()[i:i + 1], sums[i:i + 1], 2e-5, f"{kind} static={static}: sum {name}")
            else:
                assert float(out["sums"][i]) == 0.0


def test_atss_distance_tie_policy(backend):
    """A GT centred exactly on an anchor-lattice symmetry axis has anchors at EXACTLY equal centre distance; when such a tie straddles
    the 9th place of the per-level top-k, the reference's result is whatever ATen's CPU top-k (std::nth_element on (value, index) pairs
    with a value-only comparator) leaves there - an artefact of the selection algorithm, different again on the reference's CUDA path.
    The kernel's rule is deterministic and documented: among equal distances the LOWER anchor index wins.  This pins that rule (and
    that everything away from the tie agrees with the oracle)."""
    from oracle.ppyolo_loss import PPYoloELossOracle

    B, hw, C = 1, [(12, 12), (6, 6), (3, 3)], 8
    logits, distri, anchors, pts, pts_grid, counts, strides, _ = _head_case(B, hw, C, seed=5)
    s = 96.0
    targets = torch.tensor([[0, 3, s / 2, s / 2, s * 0.6, s * 0.3]])   # centre (48, 48): a lattice corner of every level
    orc = PPYoloELossOracle(C, use_varifocal_loss=True, use_static_assigner=True)
    _, a_label, _, _ = orc.assign((logits, distri, anchors, pts, counts, strides), targets)
    out = K.ppyoloe_loss_fwd(logits.to(backend), distri.to(backend), anchors.to(backend), pts.to(backend), strides.to(backend), targets.to(backend), counts,
                             True, True, (1.0, 2.5, 0.5))
    kl = out["label"].cpu().long()
    # the kernel's own rule, restated: per level the 9 nearest anchor centres by (distance, index); threshold = mean + unbiased std of their
    # IoUs with the GT; positives = candidates with IoU >= threshold whose centre lies inside the GT
    gt = torch.tensor([s / 2 - s * 0.3, s / 2 - s * 0.15, s / 2 + s * 0.3, s / 2 + s * 0.15])
    ctr = (anchors[:, :2] + anchors[:, 2:]) / 2
    dist = ((ctr - torch.tensor([s / 2, s / 2])) ** 2).sum(-1).sqrt()
    cand, off = [], 0
    for n in counts:
        order = sorted(range(n), key=lambda i: (float(dist[off + i]), i))[:9]
        cand += [off + i for i in order]
        off += n
    cand = torch.tensor(cand)
    lt, rb = torch.max(anchors[cand, :2], gt[:2]), torch.min(anchors[cand, 2:], gt[2:])
    inter = (rb - lt).clamp(min=0).prod(-1)
    area = lambda b: (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])  # noqa: E731
    iou = inter / (area(anchors[cand]) + area(gt) - inter + 1e-10)
    thr = iou.mean() + iou.std()
    inside = (ctr[cand] > gt[:2]).all(-1) & (ctr[cand] < gt[2:]).all(-1)
    expect = torch.full((anchors.shape[0],), C, dtype=torch.long)
    expect[cand[(iou >= thr) & inside]] = 3
    assert torch.equal(kl[0], expect), "kernel tie rule: lower anchor index among equal distances"
    away = (kl[0] == a_label[0])
    assert int((~away).sum()) <= 2, "the kernel and ATen's CPU top-k may differ only in which member of a tied pair is kept"


def test_nms_degenerate_boxes(backend):
    """Zero-area boxes (IoU = 0/0 = NaN: never suppresses, never suppressed), inverted boxes (x2 < x1: negative 'area'), identical boxes
    (IoU exactly 1), boxes touching along an edge (IoU exactly 0), IoU exactly AT the threshold (kept: suppression is `>`), equal scores."""
    from oracle import nms as onms

    bx = torch.tensor([
        [10, 10, 50, 50], [10, 10, 50, 50],        # identical pair
        [50, 10, 90, 50],                           # touches the first along x = 50
        [20, 20, 20, 60], [20, 20, 20, 60],        # zero width, twice
        [30, 30, 30, 30],                           # a point
        [80, 80, 40, 40],                           # inverted
        [0, 0, 40, 40], [0, 0, 40, 20],             # IoU exactly 0.5 (20*40 / 40*40)
        [100, 100, 140, 140], [100, 100, 140, 141], [100, 100, 141, 140],
    ], dtype=torch.float32)
    L, C = bx.shape[0], 3
    sc = torch.zeros(1, L, C)
    base = torch.tensor([0.9, 0.9, 0.8, 0.7, 0.7, 0.6, 0.5, 0.95, 0.94, 0.3, 0.3, 0.3])
    sc[0, :, 0] = base
    sc[0, :, 1] = base.flip(0) * 0.5
    for multi, mode in ((True, 0), (False, 0), (True, 1), (True, 2)):
        kw = dict(score_threshold=0.1, nms_threshold=0.5, nms_top_k=64, max_predictions=32, multi_label_per_box=multi)
        ref = onms.post_prediction(bx[None], sc, class_agnostic_nms=(mode == 0), force_vanilla=(mode == 2), **kw)
        out, cnt, idx, ncand = K.nms(bx[None].to(backend), sc.to(backend), 0.1, 0.5, 64, 32, multi_label=multi, class_mode=mode)
        n = int(cnt[0])
        assert n == ref[0].shape[0], f"multi={multi} mode={mode}: kept {n} vs oracle {ref[0].shape[0]}"
        assert torch.equal(out[0, :n].cpu(), ref[0]), f"multi={multi} mode={mode}: rows differ"


@pytest.mark.parametrize("case", [(1, 5, 4, 32, 40, 3, 1, 1), (1, 6, 6, 32, 32, 3, 2, 1), (1, 5, 9, 96, 40, 1, 1, 0)])
@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
def test_conv_deep_slabs(backend, case, math):
    if math == "bf16x3" and case[5] != 3:
        pytest.skip("bf16x3 shares the slab addressing with fp32: one 3x3 case (stride 1 / 2) covers its plane layout")
    """32-deep slabs (igemm_kernel<..., KD = 32>; the default where C % 32 == 0; sgx_debug_set_variant(6): two LDS buffers, (7): the 16-deep loop): every tile shape on problems with ragged
    edges in both tile dimensions, 3x3 / stride-2 (parity-class data gradient) / 1x1.  The reduction runs in the same order as with
    16-deep slabs, so the results must be BIT-identical to the default kernel's, not just close."""
    from super_gradients_amd._lib import lib

    n, h, w, c, k, r, s, p = case
    x, wt, b = _conv_case(case)
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dy)
    xd, wd, dyd = to_nhwc(x.detach(), backend), K.to_ohwi(wt.to(backend)), to_nhwc(dy, backend)
    K.set_conv_math(math)
    try:
        # (every instantiated tile compiles from the same template; tile plumbing itself is test_conv_every_tile_shape's job)
        tiles = [(64, 64), (128, 32), (64, 96)] if r == 3 and s == 1 else [(64, 64)] if s == 2 else [(64, 32), (128, 96)]
        for bm, bn in tiles:
            lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
            res = {}
            for var in (7, 0, 6, 11):  # 7 = the 16-deep loop, 0 = default (32-deep, one LDS buffer), 6 = the pipelined bf16x3 loop, 11 = all slabs up front (fp32, <= 4 slabs)
                lib().sgx_debug_set_variant(var)
                yd, parts = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p, stat_partials=True)
                dx = K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=s, pad=p)
                res[var] = (yd.cpu().clone(), parts[0].cpu().clone(), dx.cpu().clone())
            for var in (0, 6, 11):
                assert_close(to_nchw_cpu(res[var][0]), y.detach(), TOL, f"variant {var} fwd tile {bm}x{bn}")
                assert_close(to_nchw_cpu(res[var][2]), x.grad, TOL, f"variant {var} dgrad tile {bm}x{bn}")
                for a, bb, what in zip(res[7], res[var], ("fwd", "stats", "dgrad")):
                    assert torch.equal(a, bb), f"{what} tile {bm}x{bn}: 32-deep slabs (variant {var}) differ from 16-deep slabs"
    finally:
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        lib().sgx_debug_set_variant(0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


@pytest.mark.parametrize("nblk,C", [(33, 4), (300, 36), (4096, 8), (4200, 4)])
def test_fused_finalize(backend, nblk, C):
    """sgx_bn_set_fused_finalize (experiment switch): the BatchNorm forward / backward finalize and the column sum as ONE cooperative launch
    must give what the pre-reduction + finalize pair gives (same fp32 partial rows, fp64 sums regrouped: equal to ~1e-7), for channel
    counts that do not fill a workgroup, row counts around the lane count, at and above the cooperative limit (4096 rows; 4200: unchanged path)."""
    from super_gradients_amd._lib import lib

    g = torch.Generator().manual_seed(nblk + C)
    parts = (torch.randn(2, nblk, C, generator=g) * 3 + 1).to(backend)
    parts[1] = parts[1].abs() * 50 + 20          # sum of squares partials: keep the variance positive
    M = nblk * 64
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(backend), torch.randn(C, generator=g).to(backend)
    mean, invstd = torch.randn(C, generator=g).to(backend), (torch.rand(C, generator=g) + 0.5).to(backend)
    x = torch.randn(2, 40, nblk // 8 + 3, C, generator=g).to(backend)
    res = {}
    try:
        for fused in (0, 1):
            lib().sgx_bn_set_fused_finalize(fused)
            rm, rv = torch.zeros(C, device=backend), torch.ones(C, device=backend)
            fwd = K.bn_finalize(parts, M, gamma, beta, 1e-3, 0.03, rm, rv)
            coef, dg, db = torch.empty(5, C, device=backend), torch.zeros(C, device=backend), torch.zeros(C, device=backend)
            ws = K.WORKSPACE.get(lib().sgx_reduce_workspace(nblk, C), parts.device)
            K.check(lib().sgx_bn_bwd_finalize(K.ptr(parts), nblk, M, C, K.ptr(gamma), K.ptr(mean), K.ptr(invstd), K.ptr(dg), K.ptr(db), K.ptr(coef), K.ptr(ws),
                                              ws.numel(), K.stream()), "sgx_bn_bwd_finalize")
            cs = torch.zeros(C, device=backend)
            K.colsum(x, cs, accumulate=False)
            res[fused] = [t.cpu().clone() for t in (*fwd, rm, rv, coef, dg, db, cs)]
        assert lib().sgx_bn_get_fused_finalize() == 1
    finally:
        lib().sgx_bn_set_fused_finalize(0)
    for a, b_ in zip(res[0], res[1]):
        assert_close(b_, a, 2e-6, f"fused finalize nblk={nblk} C={C}")


def test_conv_tuning_table(backend):
    """sgx_conv_tuning_load: a problem listed in the table runs with the listed tile / variant (results bit-identical to the heuristic's),
    the forward statistics rows follow the table's M tile, other problems keep the heuristic, bad entries are rejected, [] clears."""
    from super_gradients_amd._lib import lib, load_conv_tuning

    K.set_conv_math("fp32")  # the table's forward / data-gradient entries address igemm_kernel (in "patch" mode this 3x3 problem runs pconv_kernel)
    case = (1, 9, 8, 32, 40, 3, 1, 1)
    n, h, w, c, k, r, s, p = case
    x, wt, b = _conv_case(case)
    xd, wd = to_nhwc(x, backend), K.to_ohwi(wt.to(backend))
    dyd = to_nhwc(torch.randn(n, k, h, w, generator=torch.Generator().manual_seed(2)), backend)
    y0, parts0 = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p, stat_partials=True)
    dx0 = K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=s, pad=p)
    key = dict(N=n, H=h, W=w, C=c, K=k, R=r, stride=s, pad=p)
    try:
        assert load_conv_tuning([dict(kind="fwd", bm=128, bn=32, variant=7, **key), dict(kind="dgrad", bm=64, bn=32, variant=0, **key)]) == 2
        assert lib().sgx_conv_tuning_size() == 2
        y1, parts1 = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p, stat_partials=True)
        dx1 = K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=s, pad=p)
        assert torch.equal(y1.cpu(), y0.cpu()) and torch.equal(dx1.cpu(), dx0.cpu())
        assert parts1.shape[1] == 1 and parts0.shape[1] == 2, "statistics rows: one 128-row tile from the table vs two 64-row tiles (72 pixels)"
        assert_close(parts1.sum(1).cpu(), parts0.sum(1).cpu(), 1e-5, "statistics")
        # the table is consulted per problem: an entry for another problem changes nothing here
        load_conv_tuning([dict(kind="fwd", bm=128, bn=128, variant=7, **dict(key, K=k + 4))])
        assert torch.equal(K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p).cpu(), y0.cpu())
        with pytest.raises(RuntimeError, match="no kernel"):
            load_conv_tuning([dict(kind="fwd", bm=48, bn=32, variant=0, **key)])
        with pytest.raises(RuntimeError, match="no kernel"):
            load_conv_tuning([dict(kind="fwd", bm=64, bn=64, variant=3, **key)])
        # weight gradient: filter tile x (tap, channel) tile x split target; a different split regroups the pixel sum (rounding level)
        g0 = K.to_ohwi(torch.zeros_like(wt).to(backend))
        K.conv2d_bwd_weight(xd, dyd, g0, None, stride=s, pad=p)
        load_conv_tuning([dict(kind="wgrad", bm=32, bn=96, variant=2048, **key)])
        g1 = K.to_ohwi(torch.zeros_like(wt).to(backend))
        K.conv2d_bwd_weight(xd, dyd, g1, None, stride=s, pad=p)
        assert_close(g1.cpu(), g0.cpu(), 1e-5, "weight gradient with a tuned tile / split")
    finally:
        load_conv_tuning([])
        K.set_conv_math(K.DEFAULT_CONV_MATH)
    assert lib().sgx_conv_tuning_size() == 0


@pytest.mark.parametrize("math", ["fp32", "bf16x3"])
def test_conv_every_tile_shape(backend, math):
    """Every instantiated (BM, BN) tile of the implicit-GEMM kernel, forced through the measurement override (sgx_debug_set_tiles), on
    one forward + data-gradient problem with ragged edges in both tile dimensions - the heuristics only ever pick a few of them."""
    from super_gradients_amd._lib import lib

    n, h, w, c, k, r, s, p = (1, 9, 8, 20, 72, 3, 1, 1)   # M = 72 pixels, N = 72 filters: partial tiles everywhere
    x, wt, b = _conv_case((n, h, w, c, k, r, s, p))
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(dy)
    xd, wd = to_nhwc(x.detach(), backend), K.to_ohwi(wt.to(backend))
    K.set_conv_math(math)
    try:
        for bm in (64, 128):
            for bn in (32, 64, 96, 128):
                lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                yd, parts = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s, pad=p, stat_partials=True)
                assert_close(to_nchw_cpu(yd), y.detach(), TOL, f"{math} fwd tile {bm}x{bn}")
                assert_close(parts[0].sum(0).cpu() / (n * h * w), y.detach().mean((0, 2, 3)), 1e-4, f"{math} stats tile {bm}x{bn}")
                dx = K.conv2d_bwd_data(to_nhwc(dy, backend), wd, (n, h, w, c), stride=s, pad=p)
                assert_close(to_nchw_cpu(dx), x.grad, TOL, f"{math} dgrad tile {bm}x{bn}")
    finally:
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


def test_wgrad_group(backend, monkeypatch):
    """sgx_conv2d_bwd_weight_group: several weight gradients of different shapes / tile shapes in one call, pixel splits folded inside the
    launch by the arrival-walked binary tree - an odd split count (nodes without a sibling pass up), several levels, the direct form
    (one split); dw accumulates; the result does not depend on the order in which workgroups arrive (the emulation dispatches them in a
    shuffled order), the tickets are left zero (a second call on the same buffers is correct) and the same call under another arrival
    order is bit-identical (fixed association)."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    # (N, H, W, C, K, R, stride, pad)
    cases = [(2, 80, 80, 64, 64, 3, 1, 1), (2, 80, 80, 96, 96, 1, 1, 0), (4, 40, 40, 32, 48, 3, 2, 1), (1, 20, 20, 128, 32, 1, 1, 0),
             (2, 160, 160, 4, 48, 3, 2, 1)] if gpu else \
            [(1, 42, 42, 4, 8, 1, 1, 0), (1, 9, 7, 8, 36, 3, 1, 1), (2, 12, 12, 8, 8, 3, 2, 1), (1, 20, 26, 4, 40, 1, 1, 0),
             # rows of >= 16 pixels with row AND image wraps inside a split (the incremental lane offsets), stride 1 and 2
             (2, 17, 18, 4, 8, 3, 1, 1), (2, 34, 36, 4, 8, 3, 2, 1)]
    ents, refs = [], []
    for i, shape in enumerate(cases):
        n, h, w, c, k, r, s, p = shape
        x, wt, _ = _conv_case(shape, seed=10 + i)
        wt.requires_grad_(True)
        y = F.conv2d(x, wt, None, stride=s, padding=p)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(20 + i))
        y.backward(dy)
        refs.append(wt.grad)
        dw = K.ohwi_empty(k, c, r, r, backend)
        dw.fill_(0.25)
        # operands as channel slices of wider buffers (explicit pixel strides), like the concat slices of the networks
        ents.append((to_nhwc(x, backend, ld_pix=c + 8, c_off=4), to_nhwc(dy, backend, ld_pix=k + 4, c_off=0), dw, s, p))
    if not gpu:
        monkeypatch.setenv("SGX_EMU_SHUFFLE", "7")
    try:
        # small items: the first emu case (1764 pixels) is cut into 7 splits of 256 pixels -> a three-level tree with sibling-less nodes
        lib().sgx_debug_set_wgrad_group(6, 1, 1)
        K.conv2d_bwd_weight_group(ents)
        first = [e[2].clone() for e in ents]
        for (shape, e, ref) in zip(cases, ents, refs):
            assert_close(e[2].cpu(), ref + 0.25, TOL, f"grouped wgrad {shape}")
        K.conv2d_bwd_weight_group(ents)  # accumulates; tickets were left zero
        for (shape, e, ref) in zip(cases, ents, refs):
            assert_close(e[2].cpu(), 2 * ref + 0.25, TOL, f"grouped wgrad, second call {shape}")
        # same inputs, other arrival order / block order: bit-identical
        if not gpu:
            monkeypatch.setenv("SGX_EMU_SHUFFLE", "1234")
        lib().sgx_debug_set_wgrad_group(6, 1, 0)
        for e in ents:
            e[2].fill_(0.25)
        K.conv2d_bwd_weight_group(ents)
        for a, e in zip(first, ents):
            assert torch.equal(a.cpu(), e[2].cpu()), "grouped wgrad must not depend on the workgroup order"
        # default item size: every small job is a single split (direct accumulate)
        lib().sgx_debug_set_wgrad_group(0, 0, 1)
        for e in ents:
            e[2].zero_()
        K.conv2d_bwd_weight_group(ents)
        for (shape, e, ref) in zip(cases, ents, refs):
            assert_close(e[2].cpu(), ref, TOL, f"grouped wgrad, default split {shape}")
        # the loop's variants: 32-pixel slabs (small tiles), one slab of loads in flight instead of two; small items again
        lib().sgx_debug_set_wgrad_group(6, 1, 1)
        # (the bf16x3 loop has its own test; here it runs on the emulation only - on the chip this case list would reach tile shapes of it
        # that have not met hardware yet)
        for loop, what in ((1, "32-pixel slabs"), (2, "one slab in flight"), (4, "64x64 tile on two waves")) + (() if gpu else ((8, "bf16x3 loop"),)):
            lib().sgx_debug_set_wgrad_loop(loop, 0)
            for e in ents:
                e[2].zero_()
            K.conv2d_bwd_weight_group(ents)
            for (shape, e, ref) in zip(cases, ents, refs):
                assert_close(e[2].cpu(), ref, TOL, f"grouped wgrad, {what} {shape}")
    finally:
        lib().sgx_debug_set_wgrad_group(0, 0, 1)
        lib().sgx_debug_set_wgrad_loop(0, 0)
        if gpu:  # the next use re-binds the library and re-applies the SGX_* environment switches (a gate run sets SGX_WGRAD_MATH suite-wide)
            from super_gradients_amd import _lib as _l

            _l._LIB = None


@pytest.mark.parametrize("mode", ["patch_auto", "patch_bf3"])
def test_conv_math_patch_auto(backend, mode):
    """Conv math modes 4 / 5: the patch kernel on the 3x3 stride-1 problems, the per-problem bf16x3 / fp32 rule on the rest - mode 5
    ("patch_bf3", the default since round 4) extends the rule to the QARepVGG two-branch forward and two-source data gradient (three
    accumulators per block there: leading products, corrections, second output).  Forward (with the statistics rows, whose count follows the
    dispatch), data gradient, and for mode 5 the two-output / two-source launches (stride 2: the forms the patch kernel does not take)
    against ATen on a problem of each kind; the bf16x3 results must be at least as close to fp64 as a small multiple of the fp32 pipe's."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    cases = [(2, 40, 40, 32, 64, 3, 1, 1), (2, 20, 20, 192, 64, 1, 1, 0), (2, 24, 24, 8, 16, 1, 1, 0), (2, 40, 40, 64, 128, 3, 2, 1)] if gpu else \
            [(1, 9, 20, 16, 32, 3, 1, 1),    # 3x3 stride 1 -> patch kernel (variant 9 lifts the 40 x 40 floor for the small map)
             (1, 6, 6, 192, 16, 1, 1, 0),    # depth 192 -> bf16x3 GEMM
             (1, 8, 8, 8, 16, 1, 1, 0),      # shallow -> fp32 pipe
             (1, 10, 10, 32, 16, 3, 2, 1)]   # 3x3 stride 2, depth 288 -> bf16x3 GEMM
    K.set_conv_math(mode)
    lib().sgx_debug_set_variant(0 if gpu else 9)
    try:
        assert K.get_conv_math() == mode
        for i, shape in enumerate(cases):
            n, h, w, c, k, r, s_, p_ = shape
            x, wt, b = _conv_case(shape, seed=60 + i)
            x.requires_grad_(True)
            y = F.conv2d(x, wt, b, stride=s_, padding=p_)
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(70 + i))
            y.backward(dy)
            xd, wd = to_nhwc(x.detach(), backend), K.to_ohwi(wt.to(backend))
            yd, parts = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s_, pad=p_, stat_partials=True)
            assert_close(to_nchw_cpu(yd), y.detach(), TOL, f"{mode} fwd {shape}")
            assert_close(parts[0].sum(0).cpu() / (y.numel() // k), y.detach().mean((0, 2, 3)), 1e-4, f"{mode} stats {shape}")
            dx = K.conv2d_bwd_data(to_nhwc(dy, backend), wd, (n, h, w, c), stride=s_, pad=p_)
            assert_close(to_nchw_cpu(dx), x.grad, TOL, f"{mode} dgrad {shape}")
            if mode == "patch_bf3" and r == 3 and c >= 16 and k >= 16:
                # the QARepVGG forms on the bf16 pipe: y3 / u / five moments, and dx = dgrad3x3(dy) + dgrad1x1(ds)
                g = torch.Generator().manual_seed(80 + i)
                w1 = torch.randn(k, c, 1, 1, generator=g) / c ** 0.5
                w1d = K.to_ohwi(w1.to(backend))
                y3, ud, st5 = K.conv2d_fwd_dual(xd, wd, w1d, b.to(backend), stride=s_)
                y3r, ur = F.conv2d(x.detach(), wt, None, stride=s_, padding=1), F.conv2d(x.detach(), w1, b, stride=s_)
                assert_close(to_nchw_cpu(y3), y3r, TOL, f"{mode} dual y {shape}")
                assert_close(to_nchw_cpu(ud), ur, TOL, f"{mode} dual u {shape}")
                M = y3r.numel() // k
                u0 = ur - b.view(1, -1, 1, 1)
                for q, ref in enumerate((y3r, y3r * y3r, u0, u0 * u0, y3r * u0)):
                    assert_close(st5[q].sum(0).cpu() / M, ref.mean((0, 2, 3)), 1e-4, f"{mode} dual moment {q} {shape}")
                ds = torch.randn(y.shape, generator=g)
                xs = x.detach().clone().requires_grad_(True)
                (F.conv2d(xs, wt, None, stride=s_, padding=1) * dy).sum().backward()
                gx3 = xs.grad.clone()
                xs.grad = None
                (F.conv2d(xs, w1, None, stride=s_) * ds).sum().backward()
                wtb = K.conv2d_wt_buffer(wd, backend)
                K.conv2d_transpose_weights(wd, wtb, stride=s_, pad=1)
                dxd = K.conv2d_bwd_data_dual(to_nhwc(dy, backend), wd, wtb, to_nhwc(ds, backend), w1d.reshape(k, c).t().contiguous(), (n, h, w, c), stride=s_)
                assert_close(to_nchw_cpu(dxd), gx3 + xs.grad, TOL, f"{mode} dual dgrad {shape}")
    finally:
        lib().sgx_debug_set_variant(0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


def test_wgrad_bf16x3_loop(backend):
    """The weight gradient's bf16x3 slab loop (the default arithmetic since round 4 - sgx_conv_set_wgrad_math; sgx_debug_set_wgrad_loop bit 3
    selects it, bit 5 the fp32 loop: three bf16 planes per slab, MFMA operands through the LDS transpose read) on tile shapes of every wave layout, against ATen's fp32 gradient: fp32-level agreement (the six-product scheme drops terms
    <= 2^-24 of a product), i.e. much tighter than the conv tolerance; pixel splits + several images + stride 2 + channel-slice operands."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    shapes = [(2, 40, 40, 32, 48, 3, 1, 1), (2, 40, 40, 16, 40, 3, 2, 1)] if gpu else [(2, 9, 18, 4, 12, 3, 1, 1), (1, 10, 36, 8, 8, 1, 2, 0)]
    try:
        # (on the chip: the four shapes of the loop's first run, r3zj; the other wave layouts have only met the emulation so far)
        tiles = ((64, 64), (96, 128), (128, 64), (32, 128)) + (() if gpu else ((96, 96), (64, 32), (32, 32), (128, 128)))
        for bnk, bj in tiles:
            lib().sgx_debug_set_tiles(0, 0, bnk, bj, 0)
            lib().sgx_debug_set_wgrad_group(6, 1, 1)  # small items: several splits -> the fold tail runs too
            for i, shape in enumerate(shapes):
                n, h, w, c, k, r, s_, p_ = shape
                x, wt, _ = _conv_case(shape, seed=40 + i)
                wt.requires_grad_(True)
                y = F.conv2d(x, wt, None, stride=s_, padding=p_)
                dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(50 + i))
                y.backward(dy)
                ent = (to_nhwc(x, backend, ld_pix=c + 4, c_off=4), to_nhwc(dy, backend, ld_pix=k + 4, c_off=0), K.ohwi_empty(k, c, r, r, backend), s_, p_)
                got = {}
                for loop in (32, 8):  # bit 5: the fp32 loop, bit 3: the bf16x3 loop (the tile override keeps the patch kernel out)
                    lib().sgx_debug_set_wgrad_loop(loop, 0)
                    ent[2].zero_()
                    K.conv2d_bwd_weight_group([ent])
                    got[loop] = ent[2].cpu().clone()
                scale = float(wt.grad.abs().max())
                e_fp32 = float((got[32] - wt.grad).abs().max()) / scale
                e_bf = float((got[8] - wt.grad).abs().max()) / scale
                assert e_bf <= max(2e-6, 4.0 * e_fp32), f"tile {bnk}x{bj} {shape}: bf16x3 {e_bf:.2e}, fp32 loop {e_fp32:.2e}"
                wt.grad = None
    finally:
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        lib().sgx_debug_set_wgrad_group(0, 0, 1)
        lib().sgx_debug_set_wgrad_loop(0, 0)
        if gpu:  # (as in test_wgrad_group)
            from super_gradients_amd import _lib as _l

            _l._LIB = None


def test_dgrad_carries_bn_reduce(backend):
    """sgx_conv2d_bwd_data_wt_req / _dual_req: the data gradient that finalises a layer's output gradient also leaves that layer's
    BatchNorm-backward reduce (sum g, sum g (t - mean) per channel; g = dx * act'(scale t + shift)) - per tile row, the rows
    sgx_bn_bwd_reduce would have produced with its own pass.  Checked: the column totals of the rows against an fp64 evaluation of the
    same sums on the launch's own dx (forms: 1x1 and 3x3 on the slab kernels, 3x3 on the patch kernel, stride 2 with its four parity-class
    launches, accumulate, two channel ranges = two layers behind one concat gradient, the two-source QARepVGG launch), and the whole
    BatchNorm backward fed with those rows against the same backward fed by the reduce sweep."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    # (N, H, W, C, K, R, stride, pad, two_source, accumulate, ranges)
    cases = [(2, 80, 80, 64, 64, 1, 1, 0, False, True, [(0, 64, "relu")]), (2, 48, 56, 64, 32, 3, 1, 1, False, False, [(16, 48, "relu")]),
             (2, 40, 40, 96, 48, 3, 2, 1, False, False, [(0, 96, "silu")]), (2, 20, 20, 192, 96, 1, 1, 0, False, False, [(0, 64, "relu"), (128, 192, None)]),
             (2, 80, 80, 32, 32, 3, 1, 1, True, True, [(0, 32, "relu")]), (2, 24, 24, 32, 32, 3, 1, 1, True, False, [(0, 32, "relu")])] if gpu else \
            [(1, 9, 10, 8, 8, 1, 1, 0, False, True, [(0, 8, "relu")]), (2, 6, 7, 16, 16, 3, 1, 1, False, False, [(4, 12, "relu")]),
             (1, 9, 11, 8, 16, 3, 2, 1, False, False, [(0, 8, "silu")]), (1, 5, 6, 24, 8, 1, 1, 0, False, False, [(0, 8, "relu"), (16, 24, None)]),
             (1, 6, 6, 16, 16, 3, 1, 1, True, True, [(0, 16, "relu")])]
    if not gpu:  # the patch kernel's epilogue too: variant 9 lifts its map-size condition (on the chip the 80 x 80 / 48 x 56 cases reach it)
        cases += [(1, 8, 16, 8, 16, 3, 1, 1, False, True, [(0, 8, "relu")], 9), (1, 9, 17, 16, 16, 3, 1, 1, True, False, [(4, 16, "silu")], 9)]
    for ci, case in enumerate(cases):
        n, h, w, c, k, r, s, p, dual, acc, ranges = case[:11]
        lib().sgx_debug_set_variant(case[11] if len(case) > 11 else 0)
        g = torch.Generator().manual_seed(100 + ci)
        ho, wo = (h + 2 * p - r) // s + 1, (w + 2 * p - r) // s + 1
        dy = to_nhwc(torch.randn(n, k, ho, wo, generator=g), backend)
        wt_l = K.to_ohwi((torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5).to(backend))
        wbuf = K.conv2d_wt_buffer(wt_l, backend)
        K.conv2d_transpose_weights(wt_l, wbuf, stride=s, pad=p)
        dx0 = torch.randn(n, h, w, c + 8, generator=g).to(backend)[..., 4:4 + c]  # a channel slice of a wider buffer, pre-filled (accumulate)
        reqs, layers = [], []
        for (lo, hi, act) in ranges:
            cr = hi - lo
            t = torch.randn(n, h, w, cr, generator=g).to(backend)
            scale, shift, mean = (torch.rand(cr, generator=g) + 0.5).to(backend), (torch.randn(cr, generator=g) * 0.3).to(backend), (torch.randn(cr, generator=g) * 0.2).to(backend)
            reqs.append(K.BnReduceRequest(t, scale, shift, mean, act, lo, hi))
            layers.append((lo, hi, act, t, scale, shift, mean))
        dx_plain, dx_req = dx0.clone(), dx0.clone()
        if dual:
            ds = to_nhwc(torch.randn(n, k, ho, wo, generator=g), backend)
            w1pt = (torch.randn(c, k, generator=g) / k ** 0.5).to(backend)
            K.conv2d_bwd_data_dual(dy, wt_l, wbuf, ds, w1pt, (n, h, w, c), stride=s, out=dx_plain, accumulate=acc)
            K.conv2d_bwd_data_dual(dy, wt_l, wbuf, ds, w1pt, (n, h, w, c), stride=s, out=dx_req, accumulate=acc, reqs=reqs)
        else:
            K.conv2d_bwd_data_wt(dy, wt_l, wbuf, (n, h, w, c), stride=s, pad=p, out=dx_plain, accumulate=acc)
            K.conv2d_bwd_data_wt(dy, wt_l, wbuf, (n, h, w, c), stride=s, pad=p, out=dx_req, accumulate=acc, reqs=reqs)
        assert torch.equal(dx_plain.cpu(), dx_req.cpu()), f"case {ci}: the requests must not change dx"
        for rq, (lo, hi, act, t, scale, shift, mean) in zip(reqs, layers):
            assert rq.parts is not None, f"case {ci}: the launch did not take the request"
            gsl = dx_req[..., lo:hi].cpu().double()
            td = t.cpu().double()
            pre = scale.cpu().double() * td + shift.cpu().double()
            if act == "relu":
                gg = gsl * (pre > 0)
            elif act == "silu":
                sg = torch.sigmoid(pre)
                gg = gsl * (sg * (1 + pre * (1 - sg)))
            else:
                gg = gsl
            s0, s1 = gg.sum((0, 1, 2)), (gg * (td - mean.cpu().double())).sum((0, 1, 2))
            got = rq.parts.cpu().double().sum(1)
            den = (gg.abs().sum((0, 1, 2)) + 1e-30)
            assert float(((got[0] - s0).abs() / den).max()) < 2e-6 and float(((got[1] - s1).abs() / (den * 3)).max()) < 2e-6, f"case {ci} range {lo}:{hi}"
            # the whole BatchNorm backward on these rows == the one that sweeps (dy, t) itself
            gamma, invstd = (torch.rand(hi - lo, generator=g) + 0.5).to(backend), (torch.rand(hi - lo, generator=g) + 0.5).to(backend)
            outs = []
            for parts in (None, rq.parts):
                dg, db = torch.zeros(hi - lo, device=backend), torch.zeros(hi - lo, device=backend)
                dxb = K.bn_bwd(dx_req[..., lo:hi], t, scale, shift, gamma, mean, invstd, dg, db, act=act, parts=parts)
                outs.append((dxb.cpu(), dg.cpu(), db.cpu()))
            for a, b, what in zip(outs[0], outs[1], ("dx", "dgamma", "dbeta")):
                assert_close(b, a, 2e-5, f"case {ci} range {lo}:{hi}: BatchNorm backward {what} from the data gradient's rows")
    lib().sgx_debug_set_variant(0)


def test_wgrad_patch_kernel(backend, monkeypatch):
    """The weight gradient's PATCH kernel (wgrad_patch.hip; sgx_conv_set_wgrad_math mode 2, the default) against ATen's fp32 gradient AND
    against the fp32 slab loop: every kernel form - stride 1 / 2, tile columns 16 / 8 / 4 (by the map's width), one / two / three filter
    blocks per workgroup (by K) - with border tiles on all four sides, ragged maps (tile rows / columns beyond the map), several images,
    several pixel ranges (the fold tree: odd counts, several levels), filter / channel counts that do not fill their 32-wide blocks,
    operands that are channel slices of wider buffers; dw accumulates; a second call on the same tickets is correct; the result does
    not depend on the order in which the workgroups arrive."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    # (N, H, W, C, K, R, stride, pad)
    cases = [(4, 80, 80, 64, 64, 3, 1, 1), (2, 160, 160, 32, 32, 3, 1, 1), (3, 40, 40, 96, 96, 3, 1, 1), (5, 20, 20, 192, 192, 3, 1, 1),
             (2, 160, 160, 48, 96, 3, 2, 1), (2, 80, 80, 96, 192, 3, 2, 1), (3, 40, 40, 192, 384, 3, 2, 1), (2, 23, 37, 40, 72, 3, 1, 1),
             (2, 45, 31, 36, 100, 3, 2, 1)] if gpu else \
            [(2, 12, 16, 8, 8, 3, 1, 1),    # 16 columns, one filter block, 2 x 6 tiles -> two pixel ranges
             (1, 9, 20, 4, 40, 3, 1, 1),    # 20 wide -> 4-column tiles; two filter blocks, the second one partly empty; ragged tile rows
             (3, 7, 8, 12, 72, 3, 1, 1),    # 8-column tiles; three filter blocks; several images inside one range
             (3, 13, 34, 8, 16, 3, 2, 1),   # stride 2: 17 x 7 outputs, parity-split patch, ragged both ways, 15 tiles -> two ranges
             (5, 16, 16, 36, 8, 3, 2, 1),   # stride 2, 8 x 8 outputs, two channel chunks (the second one 4 wide), two ranges
             (4, 40, 8, 4, 8, 3, 2, 1)]     # stride 2, 4 x 20 outputs -> 4-column tiles, tall, two ranges
    ents, refs = [], []
    for i, shape in enumerate(cases):
        n, h, w, c, k, r, s, p = shape
        x, wt, _ = _conv_case(shape, seed=60 + i)
        wt.requires_grad_(True)
        y = F.conv2d(x, wt, None, stride=s, padding=p)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(70 + i))
        y.backward(dy)
        refs.append(wt.grad)
        ents.append((to_nhwc(x, backend, ld_pix=c + 8, c_off=4), to_nhwc(dy, backend, ld_pix=k + 4, c_off=0), K.ohwi_empty(k, c, r, r, backend), s, p))
    if not gpu:
        monkeypatch.setenv("SGX_EMU_SHUFFLE", "11")
    try:
        assert lib().sgx_conv_get_wgrad_math() == 2
        lib().sgx_debug_set_wgrad_patch(1, 0, 1)  # smallest items (8 tiles = 256 pixels), every 3x3 pad-1 job takes the kernel whatever its fill
        got = {}
        for mode, loop in (("patch", 0), ("fp32", 16 + 32)):
            lib().sgx_debug_set_wgrad_loop(loop, 0)
            for e in ents:
                e[2].fill_(0.25)
            K.conv2d_bwd_weight_group(ents)
            got[mode] = [e[2].cpu().clone() for e in ents]
        for shape, a, b, ref in zip(cases, got["patch"], got["fp32"], refs):
            scale = float(ref.abs().max())
            e_patch, e_fp32 = float((a - 0.25 - ref).abs().max()) / scale, float((b - 0.25 - ref).abs().max()) / scale
            assert e_patch <= max(2e-6, 4.0 * e_fp32), f"patch wgrad {shape}: error {e_patch:.2e} (fp32 slab loop {e_fp32:.2e})"
        # the other workgroup shapes (two / three filter blocks = six / nine waves; the default policy takes one block below 192 filters)
        lib().sgx_debug_set_wgrad_loop(0, 0)
        for kb in (2, 3):
            lib().sgx_debug_set_wgrad_patch(1, kb, 1)
            for e in ents:
                e[2].fill_(0.25)
            K.conv2d_bwd_weight_group(ents)
            for shape, e, ref in zip(cases, ents, refs):
                err = float((e[2].cpu() - 0.25 - ref).abs().max()) / float(ref.abs().max())
                assert err <= 1e-5, f"patch wgrad, {kb} filter blocks {shape}: error {err:.2e}"
        lib().sgx_debug_set_wgrad_patch(1, 0, 1)
        # accumulates; tickets were left zero; another arrival order is bit-identical
        lib().sgx_debug_set_wgrad_loop(0, 0)
        if not gpu:
            monkeypatch.setenv("SGX_EMU_SHUFFLE", "4321")
        K.conv2d_bwd_weight_group(ents)
        for shape, e, a, ref in zip(cases, ents, got["patch"], refs):
            assert_close(e[2].cpu(), 2 * ref + 0.25, TOL, f"patch wgrad, second call {shape}")
            e[2].fill_(0.25)
        lib().sgx_debug_set_wgrad_group(0, 0, 0)  # plain workgroup order instead of the XCD-aware one
        K.conv2d_bwd_weight_group(ents)
        for shape, e, a in zip(cases, ents, got["patch"]):
            assert torch.equal(e[2].cpu(), a), f"patch wgrad must not depend on the workgroup order {shape}"
        # default policy (fill threshold, item size): whichever kernel takes a job, the gradient is the same to fp32 accuracy
        lib().sgx_debug_set_wgrad_patch(0, 0, 0)
        lib().sgx_debug_set_wgrad_group(0, 0, 1)
        for e in ents:
            e[2].zero_()
        K.conv2d_bwd_weight_group(ents)
        for shape, e, ref in zip(cases, ents, refs):
            assert_close(e[2].cpu(), ref, TOL, f"wgrad, default policy {shape}")
    finally:
        lib().sgx_debug_set_wgrad_patch(0, 0, 0)
        lib().sgx_debug_set_wgrad_group(0, 0, 1)
        lib().sgx_debug_set_wgrad_loop(0, 0)


# (N, H, W, C, K): 3x3 stride-1 pad-1 problems for the patch kernel - ragged 8 x 16 tiles in both directions, both chunk depths (C % 32),
# every N tile (32 / 64 / 96 filters, ragged 40), several images
PCONV_GPU = [(2, 40, 40, 96, 96), (1, 80, 80, 64, 64), (2, 20, 20, 192, 192), (2, 23, 37, 48, 40), (1, 160, 160, 32, 32), (3, 9, 20, 16, 128)]
PCONV_EMU = [(1, 9, 20, 16, 32), (2, 8, 16, 32, 40), (1, 5, 7, 48, 96)]


@pytest.mark.parametrize("idx", range(max(len(PCONV_GPU), len(PCONV_EMU))))
def test_pconv(backend, idx):
    """Conv math mode "patch" (pconv_kernel: bf16x3 from an LDS-resident input patch): forward with the fused epilogue and the BatchNorm
    statistics rows, the QARepVGG two-branch forward with its five moments, the data gradient (accumulate + addend) and the two-source
    data gradient with the scaled second addend - each against ATen's CPU convolution at the fp32 tolerance, and against the fp32-MFMA
    kernels of the same library."""
    shapes = _sizes(backend, PCONV_GPU, PCONV_EMU)
    if idx >= len(shapes):
        pytest.skip("no such case")
    n, h, w, c, k = shapes[idx]
    shape = (n, h, w, c, k, 3, 1, 1)
    x, wt, b = _conv_case(shape)
    g = torch.Generator().manual_seed(7)
    w1 = torch.randn(k, c, 1, 1, generator=g) / c ** 0.5
    x.requires_grad_(True)
    y = F.conv2d(x, wt, b, padding=1)
    u = F.conv2d(x, w1, b)
    dy = torch.randn(y.shape, generator=g)
    ds = torch.randn(y.shape, generator=g)
    (y * dy).sum().backward(retain_graph=True)
    gx3 = x.grad.clone()
    x.grad = None
    (u * ds).sum().backward()
    gx1 = x.grad.clone()
    xd, wd, w1d, dyd, dsd = to_nhwc(x.detach(), backend), K.to_ohwi(wt.to(backend)), K.to_ohwi(w1.to(backend)), to_nhwc(dy, backend), to_nhwc(ds, backend)
    add = torch.randn(y.shape, generator=g)
    from super_gradients_amd._lib import lib

    K.set_conv_math("patch")
    lib().sgx_debug_set_variant(9)  # (the product sends maps under 40 x 40 to the fp32 kernels; the logic is checked on every size)
    try:
        # forward: bias + addend + relu into a channel slice of a wider buffer, input from a channel slice, statistics of the pre-activation
        xs = to_nhwc(x.detach(), backend, ld_pix=c + 8, c_off=4)
        out = empty_nhwc(n, h, w, k, backend, ld_pix=k + 12, c_off=8)
        y2, parts = K.conv2d_fwd(xs, wd, bias=b.to(backend), addend=to_nhwc(add, backend, ld_pix=k + 12, c_off=8), out=out, act="relu", stride=1, pad=1,
                                 stat_partials=True)
        pre = (y + add).detach()
        assert_close(to_nchw_cpu(y2), F.relu(pre), TOL, "pconv fwd fused")
        M = n * h * w
        assert parts.shape[1] == n * ((h + 7) // 8) * ((w + 15) // 16), "one statistics row per 8 x 16 tile and image"
        assert_close(parts[0].sum(0).cpu() / M, pre.mean((0, 2, 3)), 1e-4, "pconv stat sum")
        assert_close(parts[1].sum(0).cpu() / M, (pre * pre).mean((0, 2, 3)), 1e-4, "pconv stat sumsq")
        yd = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=1, pad=1)
        assert_close(to_nchw_cpu(yd), y.detach(), TOL, "pconv fwd")
        # two-branch forward: y3 = conv3x3(x), u = conv1x1(x) + b, five moments
        y3, ud, st5 = K.conv2d_fwd_dual(xd, wd, w1d, b.to(backend), stride=1)
        y3r = F.conv2d(x.detach(), wt, None, padding=1)
        assert_close(to_nchw_cpu(y3), y3r, TOL, "pconv dual y")
        assert_close(to_nchw_cpu(ud), u.detach(), TOL, "pconv dual u")
        u0 = u.detach() - b.view(1, -1, 1, 1)
        for i, ref in enumerate((y3r, y3r * y3r, u0, u0 * u0, y3r * u0)):
            assert_close(st5[i].sum(0).cpu() / M, ref.mean((0, 2, 3)), 1e-4, f"pconv dual moment {i}")
        # data gradient, accumulate + addend
        dx = K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=1, pad=1)
        assert_close(to_nchw_cpu(dx), gx3, TOL, "pconv dgrad")
        addx = torch.randn(x.shape, generator=g)
        dx2 = to_nhwc(torch.ones_like(x.detach()), backend)
        K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=1, pad=1, addend=to_nhwc(addx, backend), out=dx2, accumulate=True)
        assert_close(to_nchw_cpu(dx2), gx3 + addx + 1.0, TOL, "pconv dgrad acc")
        # two-source data gradient: dx = dgrad3x3(dy) + dgrad1x1(ds) + addend + 0.5 * addend2 (its own strides)
        wtb = K.conv2d_wt_buffer(wd, backend)
        K.conv2d_transpose_weights(wd, wtb, stride=1, pad=1)
        w1t = w1d.reshape(k, c).t().contiguous()
        a2 = torch.randn(x.shape, generator=g)
        dxd = K.conv2d_bwd_data_dual(dyd, wd, wtb, dsd, w1t, (n, h, w, c), stride=1, addend=to_nhwc(addx, backend),
                                     addend2=to_nhwc(a2, backend, ld_pix=c + 4, c_off=0), addend2_scale=0.5)
        assert_close(to_nchw_cpu(dxd), gx3 + gx1 + addx + 0.5 * a2, TOL, "pconv dual dgrad")
    finally:
        lib().sgx_debug_set_variant(0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


def test_bn_backward_input_gradient_sums_to_zero(backend):
    """A training-mode BatchNorm's input gradient sums to zero per channel.  The weight gradient of the convolution in front of it is
    sum_pixels dx * activation, and activations have per-channel means - so a constant offset in dx as small as the fp32 rounding of
    mean(g) (2^-24 |mean g| in EVERY element) is amplified by M * mean(activation) against a sum that only grows like sqrt(M).  Found by
    the flip-free gradient check at 32 x 640^2 (r3b: 60x further from fp64 than ATen's CPU kernel, which forms g - mean g in double).
    The apply sweeps therefore take mean(g) as hi + lo floats: |sum dx| must stay at the random-rounding level, far below the
    single-float offset bound M * 2^-24 * |mean g| * c1 (both the plain BatchNorm backward and the QARepVGG pair's)."""
    n, h, w, c = _sizes(backend, (8, 80, 80, 16), (1, 150, 150, 4))
    M = n * h * w
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=g) * 0.3 + 4.0
    dy = torch.randn(n, c, h, w, generator=g) * 0.01 + torch.tensor([3.1, -2.7, 5.3, 1.9] * (c // 4)).view(1, c, 1, 1)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    xd = to_nhwc(x, backend)
    parts = K.channel_stats_partial(xd)
    rm, rv = torch.zeros(c, device=backend), torch.ones(c, device=backend)
    scale, shift, mean, invstd = K.bn_finalize(parts, M, gamma.to(backend), beta.to(backend), 1e-3, 0.03, rm, rv)
    dgamma, dbeta = torch.zeros(c, device=backend), torch.zeros(c, device=backend)
    dx = K.bn_bwd(to_nhwc(dy, backend), xd, scale, shift, gamma.to(backend), mean, invstd, dgamma, dbeta, act=None)
    s = to_nchw_cpu(dx).double().sum((0, 2, 3)).abs()
    c1 = (gamma.double() * invstd.cpu().double()).abs()
    single_float_offset = M * 2.0 ** -24 * dy.double().mean((0, 2, 3)).abs() * c1
    assert bool((s <= 0.05 * single_float_offset).all()), f"sum dx {s.tolist()} vs single-float offset bound {single_float_offset.tolist()}"


@pytest.mark.parametrize("case", [(1, 12, 14, 16, 32), (2, 9, 11, 32, 16)])
def test_pconv_stride2_data_gradient(backend, case):
    """Conv math mode "patch" on the output-parity classes of a 3x3 stride-2 data gradient (2x2 / 2x1 / 1x2 / 1x1 taps at input offsets
    {0, +1}, outputs written with stride 2; odd sizes give the classes different extents) and its two-source form (the QARepVGG downsample
    blocks: the 1x1 branch reaches parity class (0, 0) only)."""
    n, h, w, c, k = case if backend.type != "cuda" else (case[0] * 4, case[1] * 8 + 1, case[2] * 8, case[3] * 2, case[4] * 2)
    shape = (n, h, w, c, k, 3, 2, 1)
    x, wt, b = _conv_case(shape)
    g = torch.Generator().manual_seed(9)
    w1 = torch.randn(k, c, 1, 1, generator=g) / c ** 0.5
    x.requires_grad_(True)
    y = F.conv2d(x, wt, None, stride=2, padding=1)
    u = F.conv2d(x, w1, None, stride=2)
    dy, ds = torch.randn(y.shape, generator=g), torch.randn(y.shape, generator=g)
    (y * dy).sum().backward(retain_graph=True)
    gx3 = x.grad.clone()
    x.grad = None
    (u * ds).sum().backward()
    gx1 = x.grad.clone()
    wd, w1d, dyd, dsd = K.to_ohwi(wt.to(backend)), K.to_ohwi(w1.to(backend)), to_nhwc(dy, backend), to_nhwc(ds, backend)
    from super_gradients_amd._lib import lib

    K.set_conv_math("patch")
    lib().sgx_debug_set_variant(9)  # (off in the product: the parity classes carry too few taps to pay for the patch - measurement variant)
    try:
        dx = K.conv2d_bwd_data(dyd, wd, (n, h, w, c), stride=2, pad=1)
        assert_close(to_nchw_cpu(dx), gx3, TOL, "pconv stride-2 dgrad")
        wtb = K.conv2d_wt_buffer(wd, backend)
        K.conv2d_transpose_weights(wd, wtb, stride=2, pad=1)
        addx = torch.randn(x.shape, generator=g)
        dxd = K.conv2d_bwd_data_dual(dyd, wd, wtb, dsd, w1d.reshape(k, c).t().contiguous(), (n, h, w, c), stride=2, addend=to_nhwc(addx, backend))
        assert_close(to_nchw_cpu(dxd), gx3 + gx1 + addx, TOL, "pconv stride-2 dual dgrad")
    finally:
        lib().sgx_debug_set_variant(0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


@pytest.mark.gpu
def test_partial_chip_stream(gpu_device):
    """sgx_stream_create_partial: a HIP stream confined to part of the CUs (the weight gradients' side stream).  Kernels launched on it
    give the results of the ordinary stream; the mask it was created with has the requested share of bits, evenly spread; bad shares are
    refused."""
    import ctypes

    from super_gradients_amd._lib import check, lib

    shape = (2, 40, 40, 32, 64, 3, 1, 1)
    x, wt, b = _conv_case(shape, seed=3)
    xd, wd = to_nhwc(x, gpu_device), K.to_ohwi(wt.to(gpu_device))
    ref = K.conv2d_fwd(xd, wd, bias=b.to(gpu_device), stride=1, pad=1)
    torch.cuda.synchronize()
    for pct in (25, 75):
        h = ctypes.c_void_p()
        check(lib().sgx_stream_create_partial(pct, ctypes.byref(h)), "sgx_stream_create_partial")
        assert h.value
        st = torch.cuda.ExternalStream(h.value, device=gpu_device)
        with torch.cuda.stream(st):
            y = K.conv2d_fwd(xd, wd, bias=b.to(gpu_device), stride=1, pad=1)
        st.synchronize()
        assert torch.equal(y, ref)
        del st
        check(lib().sgx_stream_destroy(h), "sgx_stream_destroy")
    h = ctypes.c_void_p()
    assert lib().sgx_stream_create_partial(5, ctypes.byref(h)) == -1 and lib().sgx_stream_create_partial(101, ctypes.byref(h)) == -1


def test_round4_measurement_switches(backend):
    """The measurement switches added in round 4 keep results correct and restore cleanly: the depth from which the per-problem rule runs a
    problem in bf16x3 arithmetic (sgx_debug_set_bf3_min_depth; 0 = the default 192), and the LDS the weight-gradient launches leave to other
    streams (sgx_conv_set_wgrad_lds_reserve: a launch-time dynamic-LDS request, no effect on results; 0..120 KB)."""
    from super_gradients_amd._lib import lib

    shape = _sizes(backend, (2, 24, 24, 64, 32, 1, 1, 0), (1, 6, 6, 64, 16, 1, 1, 0))  # depth 64: fp32 pipe under the default rule
    n, h, w, c, k, r, s_, p_ = shape
    x, wt, b = _conv_case(shape, seed=7)
    ref = F.conv2d(x, wt, b, stride=s_, padding=p_)
    xd, wd = to_nhwc(x, backend), K.to_ohwi(wt.to(backend))
    try:
        y_default = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s_, pad=p_).clone()
        assert lib().sgx_debug_set_bf3_min_depth(64) == 0
        K.clear_desc_cache()
        y_bf3 = K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s_, pad=p_).clone()
        assert_close(to_nchw_cpu(y_default), ref, TOL, "depth-64 problem, default rule (fp32 pipe)")
        assert_close(to_nchw_cpu(y_bf3), ref, TOL, "depth-64 problem in bf16x3 arithmetic")
        assert not torch.equal(y_default, y_bf3), "the depth switch did not change the arithmetic of a depth-64 problem"
        assert lib().sgx_debug_set_bf3_min_depth(-1) == -1
    finally:
        lib().sgx_debug_set_bf3_min_depth(0)
        K.clear_desc_cache()
    assert torch.equal(K.conv2d_fwd(xd, wd, bias=b.to(backend), stride=s_, pad=p_), y_default)
    # LDS reserve: the same weight gradient with and without it
    wshape = _sizes(backend, (2, 24, 24, 32, 32, 3, 1, 1), (1, 8, 8, 16, 16, 3, 1, 1))
    n, h, w, c, k, r, s_, p_ = wshape
    x, wt, _ = _conv_case(wshape, seed=8)
    dy = torch.randn(n, k, h, w, generator=torch.Generator().manual_seed(9))
    xd, dyd = to_nhwc(x, backend), to_nhwc(dy, backend)
    outs = []
    try:
        for kb in (0, 48):
            assert lib().sgx_conv_set_wgrad_lds_reserve(kb) == 0 and lib().sgx_conv_get_wgrad_lds_reserve() == kb
            dw = K.to_ohwi(torch.zeros(k, c, r, r, device=backend))
            K.conv2d_bwd_weight_group([(xd, dyd, dw, s_, p_)])
            outs.append(dw.clone())
        assert lib().sgx_conv_set_wgrad_lds_reserve(121) == -1 and lib().sgx_conv_set_wgrad_lds_reserve(-1) == -1
    finally:
        lib().sgx_conv_set_wgrad_lds_reserve(0)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("case", ["random", "crowded", "per_class", "boundary", "few_kept"])
def test_nms_suppression_forms_agree(backend, case):
    """The two forms of the suppression stage (sgx_debug_set_nms_split: 0 inside the per-image kernel, 1 = default: bit matrix built by a
    chip-wide launch + one-wave walk) give the same rows bit for bit, and the oracle's: sparse boxes (the scan stops at max_predictions after
    a few chunks), crowded boxes (almost everything suppressed: the scan visits every chunk), exact per-class mode, candidate counts around
    the 64-candidate chunks, and max_predictions cutting a chunk in the middle.  (Round 4 built and measured a third form - one workgroup
    per image, 64 candidates at a time against the kept list: the same rows, 102 us against 72 for the pair of launches it replaced,
    profiles/r4_nms_chunked_form.txt; removed.)"""
    from oracle import nms as onms
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    g = np.random.RandomState({"random": 1, "crowded": 2, "per_class": 3, "boundary": 4, "few_kept": 5}[case])
    B, L, C = (4, 2100, 8) if gpu else (2, 150, 3)
    topk, maxp, class_mode, iou = (1000 if gpu else 130), (300 if gpu else 40), 0, 0.6
    spread = 4.0
    if case == "crowded":
        spread, iou = 0.15, 0.3
    elif case == "per_class":
        class_mode = 2
    elif case == "boundary":
        topk = 129 if not gpu else 961
        maxp = topk
    elif case == "few_kept":
        maxp = 37
    c = g.uniform(100, 540, (B, L, 2)) * (1.0 if spread > 1 else 0.0) + (320 if spread < 1 else 0) + g.normal(0, 40 * spread, (B, L, 2))
    wh = g.uniform(40, 160, (B, L, 2))
    boxes = torch.from_numpy(np.concatenate([c - wh / 2, c + wh / 2], -1).astype(np.float32))
    scores = torch.from_numpy(g.uniform(0.0, 1.0, (B, L, C)).astype(np.float32))
    kw = dict(score_threshold=0.05, nms_threshold=iou, nms_top_k=topk, max_predictions=maxp, multi_label_per_box=True)
    ref = onms.post_prediction(boxes, scores, class_agnostic_nms=class_mode == 0, force_vanilla=class_mode == 2, **kw) if class_mode else \
        onms.post_prediction(boxes, scores, class_agnostic_nms=True, **kw)
    outs = []
    try:
        for mode in (0, 1):
            assert lib().sgx_debug_set_nms_split(mode) == 0
            out, cnt, idx, ncand = K.nms(boxes.to(backend), scores.to(backend), 0.05, iou, topk, maxp, multi_label=True, class_mode=class_mode)
            outs.append((out.cpu().clone(), cnt.cpu().clone(), idx.cpu().clone(), ncand.cpu().clone()))
    finally:
        lib().sgx_debug_set_nms_split(1)
    for a, b_, what in zip(outs[0], outs[1], ("rows", "counts", "indices", "candidate counts")):
        assert torch.equal(a, b_), f"{case}: {what} of the split form differ from the per-image kernel's"
    out, cnt = outs[1][0], outs[1][1]
    for b in range(B):
        n = int(cnt[b])
        assert n == ref[b].shape[0], f"{case} image {b}: kept {n} vs oracle {ref[b].shape[0]}"
        assert torch.equal(out[b, :n], ref[b]), f"{case} image {b}: rows differ from the oracle"
    if case == "crowded":
        assert int(cnt.max()) < maxp  # the scan ran through every chunk


@pytest.mark.gpu
def test_wgrad_patch_bias_is_pinned(gpu_device):
    """The patch weight-gradient kernel chains the six bf16x3 products of a step into ONE accumulator per (filter block, tap); the bf16 MFMA's
    accumulate floors what it shifts out, so every element of its result is LOW by ~1-2e-7 of the gradient's rms (the two-accumulator
    slab loop and the fp32 pipe sit at ~1e-9).  The offset is a property of the hardware's accumulate, invisible to the host emulation, so it
    is pinned here, on the chip, against fp64: |mean signed error| <= 2.5e-7 of the gradient's rms per layer for the patch kernel (measured
    0.9-2.1e-7, profiles/r4_error_probe_final_build.txt), <= 3e-8 for the two-accumulator bf16x3 slab loop and the fp32 loop (the probe itself),
    and the rms error of all three below ATen's own fp32 CPU gradient's - so the bias cannot grow unnoticed when the kernel is reworked."""
    from super_gradients_amd._lib import lib

    g = torch.Generator().manual_seed(0)
    worst = {}
    try:
        for (n, h, w, c, k, r, s, p, mean) in [(8, 80, 80, 64, 64, 3, 1, 1, 4.0), (8, 80, 80, 64, 64, 3, 1, 1, 0.0), (8, 80, 80, 64, 128, 3, 2, 1, 4.0),
                                               (8, 40, 40, 192, 192, 3, 1, 1, 4.0)]:
            x = torch.randn(n, c, h, w, generator=g) + mean
            ho, wo = (h + 2 * p - r) // s + 1, (w + 2 * p - r) // s + 1
            dy = torch.randn(n, k, ho, wo, generator=g)
            wref = torch.zeros(k, c, r, r, dtype=torch.float64, requires_grad=True)
            (F.conv2d(x.double(), wref, None, stride=s, padding=p) * dy.double()).sum().backward()
            ref = wref.grad
            w32 = torch.zeros(k, c, r, r, requires_grad=True)
            (F.conv2d(x, w32, None, stride=s, padding=p) * dy).sum().backward()
            sc = float(ref.pow(2).mean().sqrt())
            aten_rms = float((w32.grad.double() - ref).pow(2).mean().sqrt()) / sc
            xd, dyd = to_nhwc(x, gpu_device), to_nhwc(dy, gpu_device)
            for mode, name, bias_bar in ((0, "fp32", 3e-8), (1, "bf16x3", 3e-8), (2, "patch", 2.5e-7)):
                lib().sgx_conv_set_wgrad_math(mode)
                dw = K.to_ohwi(torch.zeros(k, c, r, r, device=gpu_device))
                K.conv2d_bwd_weight_group([(xd, dyd, dw, s, p)])
                e = dw.cpu().double() - ref
                bias, rms = float(e.mean()) / sc, float(e.pow(2).mean().sqrt()) / sc
                worst[name] = max(worst.get(name, 0.0), abs(bias))
                assert abs(bias) <= bias_bar, f"{name} weight gradient {(n, h, w, c, k, r, s, mean)}: signed offset {bias:.2e} of the gradient's rms (bar {bias_bar:.1e})"
                assert rms <= 1.2 * aten_rms, f"{name} weight gradient {(n, h, w, c, k, r, s, mean)}: rms error {rms:.2e} against ATen's {aten_rms:.2e}"
    finally:
        lib().sgx_conv_set_wgrad_math(2)
    print("wgrad signed offsets / rms, worst per mode:", {k_: f"{v:.2e}" for k_, v in worst.items()})


def _planes_for(filters, backend):
    """Plan + produce the pre-split planes of `filters` ((ptr, rows, taps, ch) records) -> (host jobs, keep-alive tensors)."""
    plan, total = K.filter_planes_plan(filters)
    assert plan, "no filter of the case qualifies for planes"
    buf = torch.empty(total, dtype=torch.uint8, device=backend)
    jobs, dev = K.filter_planes_table(plan, buf)
    K.filter_planes_batch(jobs, dev)
    return jobs, (buf, dev)


def _wt_filters(w, wtb, stride, pad):
    import ctypes

    from super_gradients_amd import _lib

    raw = K.conv2d_transpose_jobs(w, wtb, stride=stride, pad=pad)
    n = len(raw) // ctypes.sizeof(_lib.WtransJob)
    return [(r.wt, r.C, r.T, r.K) for r in (_lib.WtransJob * n).from_buffer_copy(raw)]


@pytest.mark.parametrize("case", ["patch32", "patch64", "gemm1x1", "gemm3x3s2", "qarep_s1", "qarep_s2", "gemm1x1-regs", "gemm3x3s2-regs", "qarep_s2-regs",
                                  "gemm1x1-pp", "gemm3x3s2-pp"])
def test_filter_planes_launches_are_bit_identical(backend, case):
    """Pre-split filter planes (sgx_filter_planes_batch; round 5): a bf16x3 launch that copies its filter's planes must produce exactly the
    bits of the launch that splits the fp32 filter while staging - forward, data gradient (through the transposed filters' planes) and the
    QARepVGG two-output / two-source forms, on the patch kernel and on the 32-deep GEMM loop.  The hit counter proves the planes path ran;
    with the scope closed, or after an invalidate, the same calls must not take it."""
    from super_gradients_amd._lib import lib

    gpu = backend.type == "cuda"
    # "-regs": planes mode 2 - the GEMM loop's one-block-per-wave tiles read their filter fragments straight from the planes into registers
    mode = 2 if case.endswith("-regs") else 1
    # "-pp" (round 6): conv variant 14 - the ping-pong GEMM loop (two tiles per 512-thread workgroup, staging and matrix phases half a period
    # apart; both shapes have an ODD tile count: the last workgroup's second wave group runs without a tile of its own)
    pp = case.endswith("-pp")
    if not gpu and case == "gemm1x1-regs":
        pytest.skip("host emulation: the register-fragment form is covered by the 3x3 stride-2 and two-source cases (CPU suite time)")
    case = case.split("-")[0]
    # (N, H, W, C, K, R, stride)
    shape = {"patch32": (2, 40, 40, 32, 32, 3, 1) if gpu else (1, 9, 20, 16, 32, 3, 1),
             "patch64": (2, 40, 48, 64, 128, 3, 1) if gpu else (1, 8, 16, 32, 64, 3, 1),
             "gemm1x1": (2, 20, 20, 384, 192, 1, 1) if gpu else (1, 6, 6, 192, 192, 1, 1),
             "gemm3x3s2": (2, 40, 40, 64, 128, 3, 2) if gpu else (1, 10, 10, 32, 64, 3, 2),
             "qarep_s1": (2, 40, 40, 64, 64, 3, 1) if gpu else (1, 9, 20, 32, 32, 3, 1),
             "qarep_s2": (2, 40, 40, 64, 128, 3, 2) if gpu else (1, 10, 10, 32, 64, 3, 2)}[case]
    n, h, w_, c, k, r, st = shape
    pad = r // 2
    g = torch.Generator().manual_seed(500 + len(case))
    x = to_nhwc(torch.randn(n, c, h, w_, generator=g), backend)
    wd = K.to_ohwi((torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5).to(backend))
    ho, wo = (h + 2 * pad - r) // st + 1, (w_ + 2 * pad - r) // st + 1
    dy = to_nhwc(torch.randn(n, k, ho, wo, generator=g), backend)
    wtb = K.conv2d_wt_buffer(wd, backend)
    K.conv2d_transpose_weights(wd, wtb, stride=st, pad=pad)
    qarep = case.startswith("qarep")
    if qarep:
        w1p = K.to_ohwi((torch.randn(k, c, 1, 1, generator=g) / c ** 0.5).to(backend))
        w1pt = w1p.reshape(k, c).t().contiguous()
        b1 = torch.randn(k, generator=g).to(backend)
        ds = to_nhwc(torch.randn(n, k, ho, wo, generator=g), backend)

    def run():
        if qarep:
            y, u, st5 = K.conv2d_fwd_dual(x, wd, w1p, b1, stride=st)
            dx = K.conv2d_bwd_data_dual(dy, wd, wtb, ds, w1pt, (n, h, w_, c), stride=st)
            return [y, u, st5, dx]
        y, parts = K.conv2d_fwd(x, wd, stride=st, pad=pad, stat_partials=True)
        dx = K.conv2d_bwd_data_wt(dy, wd, wtb, (n, h, w_, c), stride=st, pad=pad)
        return [y, parts, dx]

    K.set_conv_math("patch_bf3")
    lib().sgx_debug_set_variant(14 if pp else 0 if gpu else 9)  # (host emulation: small maps - variant 9 lifts the patch kernel's 40 x 40 floor)
    lib().sgx_debug_set_filter_planes(mode)
    jobs = None
    try:
        K.filter_planes_invalidate(None)
        K.filter_planes_scope(False)
        ref = run()
        filters = [(wd.data_ptr(), k, r * r, c)] + _wt_filters(wd, wtb, st, pad)
        if qarep:
            filters += [(w1p.data_ptr(), k, 1, c), (w1pt.data_ptr(), c, 1, k)]
        jobs, keep = _planes_for(filters, backend)
        # scope closed: entries exist and are valid, yet no launch may use them
        h0 = lib().sgx_debug_filter_planes_hits()
        out = run()
        assert lib().sgx_debug_filter_planes_hits() == h0, "a launch outside a step's scope read filter planes"
        K.filter_planes_scope(True)
        out = run()
        hits = lib().sgx_debug_filter_planes_hits() - h0
        assert hits >= 2, f"{case}: only {hits} launch(es) took the planes path"
        for a, b in zip(out, ref):
            assert torch.equal(a, b), f"{case}: planes launch differs from the splitting launch (max {float((a - b).abs().max()):.3e})"
        # the weights change: without a new batch the stale planes must not serve (the result follows the new weights)
        K.filter_planes_invalidate(jobs)
        wd.mul_(1.5)
        K.conv2d_transpose_weights(wd, wtb, stride=st, pad=pad)
        h1 = lib().sgx_debug_filter_planes_hits()
        out2 = run()
        assert lib().sgx_debug_filter_planes_hits() == h1, "an invalidated entry served a launch"
        K.filter_planes_scope(False)
        ref2 = run()
        for a, b in zip(out2, ref2):
            assert torch.equal(a, b)
        assert not torch.equal(out2[0], ref[0])
    finally:
        K.filter_planes_scope(False)
        K.filter_planes_invalidate(None)
        lib().sgx_debug_set_filter_planes(K.DEFAULT_FILTER_PLANES)
        lib().sgx_debug_set_variant(0)
        K.set_conv_math(K.DEFAULT_CONV_MATH)


@pytest.mark.parametrize("mode", [1, 2])
def test_filter_planes_every_tile_shape(backend, mode):
    """The planes forms of the 32-deep GEMM loop on EVERY instantiated tile (sgx_debug_set_tiles) - the heuristics pick a few of them - on a
    problem with ragged edges in both tile dimensions (20 output pixels, 96 filter rows: partial tiles, filter rows past the last one, a
    wave whose plane pieces run past the tile): forward and data gradient, planes copied into LDS (mode 1) and read straight into the
    fragment registers (mode 2, the tiles of one 32-filter block per wave; the others fall back to mode 1) - bit-identical to the
    splitting launches."""
    from super_gradients_amd._lib import lib

    # (96 channels both ways: 32-deep slabs need multiples of 32 on the reduction axis - the filters of the forward, the transposed filters
    # of the data gradient - and 96 rows leave the 64- and 128-wide tiles ragged)
    n, h, w_, c, k, r, st, pad = (1, 9, 8, 96, 96, 3, 2, 1) if backend.type != "cuda" else (2, 19, 17, 96, 96, 3, 2, 1)
    g = torch.Generator().manual_seed(900 + mode)
    x = to_nhwc(torch.randn(n, c, h, w_, generator=g), backend)
    wd = K.to_ohwi((torch.randn(k, c, r, r, generator=g) / (c * r * r) ** 0.5).to(backend))
    ho, wo = (h + 2 * pad - r) // st + 1, (w_ + 2 * pad - r) // st + 1
    dy = to_nhwc(torch.randn(n, k, ho, wo, generator=g), backend)
    wtb = K.conv2d_wt_buffer(wd, backend)
    K.conv2d_transpose_weights(wd, wtb, stride=st, pad=pad)
    K.set_conv_math("bf16x3")
    lib().sgx_debug_set_filter_planes(mode)
    try:
        K.filter_planes_invalidate(None)
        jobs, keep = _planes_for([(wd.data_ptr(), k, r * r, c)] + _wt_filters(wd, wtb, st, pad), backend)
        for bm in (64, 128):
            for bn in (32, 64, 96, 128):
                lib().sgx_debug_set_tiles(bm, bn, 0, 0, 0)
                K.clear_desc_cache()
                K.filter_planes_scope(False)
                y0 = K.conv2d_fwd(x, wd, stride=st, pad=pad)
                dx0 = K.conv2d_bwd_data_wt(dy, wd, wtb, (n, h, w_, c), stride=st, pad=pad)
                K.filter_planes_scope(True)
                h0 = lib().sgx_debug_filter_planes_hits()
                y1 = K.conv2d_fwd(x, wd, stride=st, pad=pad)
                dx1 = K.conv2d_bwd_data_wt(dy, wd, wtb, (n, h, w_, c), stride=st, pad=pad)
                assert lib().sgx_debug_filter_planes_hits() - h0 >= 2, f"tile {bm}x{bn}: the planes path did not run"
                assert torch.equal(y0, y1), f"mode {mode} forward differs on tile {bm}x{bn}: {float((y0 - y1).abs().max()):.3e}"
                assert torch.equal(dx0, dx1), f"mode {mode} data gradient differs on tile {bm}x{bn}: {float((dx0 - dx1).abs().max()):.3e}"
    finally:
        K.filter_planes_scope(False)
        K.filter_planes_invalidate(None)
        lib().sgx_debug_set_filter_planes(K.DEFAULT_FILTER_PLANES)
        lib().sgx_debug_set_tiles(0, 0, 0, 0, 0)
        K.clear_desc_cache()
        K.set_conv_math(K.DEFAULT_CONV_MATH)


@pytest.mark.gpu
def test_dot_repeats_its_bits_beside_weight_gradients(gpu_device):
    """Round 6 (tools/dot_race_probe.py): while sgx_dot's TwoSum lanes were compiled into packed fp32 instructions, its fp64 partial
    rows moved in 8 - 14 % of the calls made while a weight-gradient kernel of another stream was resident (one ulp of a YOLO-NAS bottleneck's
    d alpha in ~0.5 % of the train steps).  The library is built without such instructions; the dot must leave the same bits - result AND
    first-stage partial rows - alone and beside 3x3 / 1x1 weight gradients."""
    dev = gpu_device
    g = torch.Generator().manual_seed(0)
    n, h, w, c = 4, 40, 40, 64
    x = torch.randn(n, h, w, c, generator=g).to(dev)
    dz = (torch.randn(n, h, w, c, generator=g) * 1e-3).to(dev)
    out = torch.zeros(1, device=dev)
    bx, bdy = torch.randn(8, 80, 80, 96, generator=g).to(dev), torch.randn(8, 80, 80, 96, generator=g).to(dev)
    dw3, dw1 = K.ohwi_empty(96, 96, 3, 3, dev), K.ohwi_empty(96, 96, 1, 1, dev)
    side = torch.cuda.Stream(device=dev)
    npart = K.stats_blocks(n * h * w) * ((c // 4 + 15) // 16)

    def dot():
        K.dot_sum(x, dz, out, accumulate=False)
        return torch.cat([out.view(torch.int32).to(torch.int64), K.WORKSPACE.get(1, dev)[: npart * 8].view(torch.int64)])

    torch.cuda.synchronize()
    ref = dot().clone()
    for pad, dw in ((1, dw3), (0, dw1)):
        vals = []
        for i in range(400):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    dw.zero_()
                    K.conv2d_bwd_weight(bx, bdy, dw, stride=1, pad=pad)
            vals.append(dot())
            if i % 16 == 15:
                torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        v = torch.stack(vals)
        moved = int((v != ref).any(dim=1).sum())
        assert moved == 0, f"sgx_dot beside {'3x3' if pad else '1x1'} weight gradients: {moved} of 400 calls left other bits than the dot alone"
