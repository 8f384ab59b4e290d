"""YOLO-NAS train-step parity: the HIP model (super_gradients_amd) against the CPU oracle (oracle/yolo_nas.py - pinned to the
reference's own modules by tests/test_oracle_vs_reference.py) on identical weights and inputs.
Checks: state_dict key/shape identity, forward outputs (decoded + raw), PPYoloELoss value, EVERY parameter gradient,
BatchNorm running statistics.  Tolerance: the north star's 1e-4 relative (fp32 both sides, different summation order).
"""
import os

import pytest
import torch

from util import assert_close, assert_gradient_arenas_match, rel_err, synthetic_targets


def _build_pair(variant, num_classes, device, seed=0):
    from oracle.yolo_nas import YoloNAS as OracleYoloNAS
    from super_gradients_amd.training import models

    torch.manual_seed(seed)
    ref = OracleYoloNAS(variant, num_classes=num_classes)
    # make BN affine / running stats non-trivial so that every path is exercised
    g = torch.Generator().manual_seed(seed + 1)
    for name, p in ref.named_parameters():
        if name.endswith("bn.weight") or name.endswith("post_bn.weight"):
            p.data.uniform_(0.5, 1.5, generator=g)
        elif name.endswith("bn.bias") or name.endswith("alpha"):
            p.data.add_(torch.randn(p.shape, generator=g) * 0.1)
    net = models.get(f"yolo_nas_{variant}", num_classes=num_classes)
    missing = net.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return ref, net


def test_state_dict_matches_oracle():
    from oracle.yolo_nas import YoloNAS as OracleYoloNAS
    from super_gradients_amd.training import models

    for v in ("s", "m", "l"):
        a = OracleYoloNAS(v, num_classes=80).state_dict()
        b = models.get(f"yolo_nas_{v}", num_classes=80).state_dict()
        assert list(a.keys()) == list(b.keys())
        for k in a:
            assert tuple(a[k].shape) == tuple(b[k].shape), k


def _err(a, b, scale=None):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / (scale if scale is not None else max(float(b.abs().max()), 1e-30))


def _train_step_parity(variant, B, size, device, tol, static=False):
    """Three-way comparison: HIP fp32  vs  oracle CPU fp32 (the reference's arithmetic)  vs  the same oracle in fp64 (truth).

    forward   : activations / decoded predictions / loss items: HIP within `tol` (north star: 1e-4 rel) of the CPU fp32
                path, or - if the CPU fp32 path itself is further than that from the truth - at least as close to the truth
                as 2x the CPU path is.
    backward A: a seeded zero-mean random upstream gradient (well conditioned): EVERY parameter gradient within the
                same bar.  This is the check of the hand-written backward.
    backward B: the real PPYoloELoss gradient at random init.  It is dominated by a per-channel constant (all background
                logits are pushed down by the same amount) which the training-mode BatchNorms annihilate, so what
                remains is round-off amplified ~1e3-1e4x on BOTH fp32 paths (the CPU reference is ~5e-3 from the fp64
                truth in the backbone).  Bar: the relative L2 error of the whole gradient against the fp64 truth is
                no worse than 4x the CPU fp32 path's.
    The same upstream gradient is fed to all three backward passes: the assigner is discontinuous (top-k / arg-max over
    near-tied candidates), so a 1e-6 forward difference may legitimately move a positive to a neighbouring anchor; the
    loss kernels' own gradient parity on identical inputs is in test_kernels.py."""
    import copy

    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd.training.losses import PPYoloELoss

    C = 80
    ref, net = _build_pair(variant, C, device)
    ref64 = copy.deepcopy(ref).double()
    ref.train()
    ref64.train()
    net.train()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, 3, size, size, generator=g)
    targets = synthetic_targets(B, seed=11, kmax=6, size=size, num_classes=C)

    def bar(name, hip, cpu32, truth, scale=None, slack=2.0):
        e_pair, e_hip, e_cpu = _err(hip, cpu32, scale), _err(hip, truth, scale), _err(cpu32, truth, scale)
        assert e_pair <= tol or e_hip <= max(tol, slack * e_cpu), f"{name}: hip-cpu32 {e_pair:.2e}, hip-fp64 {e_hip:.2e}, cpu32-fp64 {e_cpu:.2e}"

    # ---------------------------------------------------------------- forward + loss
    out_ref = ref(x)
    out_ref[1][0].retain_grad()
    out_ref[1][1].retain_grad()
    loss_ref, items_ref = PPYoloELossOracle(C, use_static_assigner=static)(out_ref, targets)
    out64 = ref64(x.double())
    out = net(x.to(device))
    loss, items = PPYoloELoss(num_classes=C, use_static_assigner=static)(out, targets.to(device))
    (bx, sc), (lg, ds, an, pt, cnt, st) = out
    (bx_r, sc_r), (lg_r, ds_r, an_r, pt_r, cnt_r, st_r) = out_ref
    (bx_t, sc_t), (lg_t, ds_t, _, _, _, _) = out64
    assert list(cnt) == list(cnt_r)
    assert torch.equal(an.cpu(), an_r) and torch.equal(pt.cpu(), pt_r) and torch.equal(st.cpu(), st_r)
    bar("cls_logits", lg, lg_r, lg_t)
    bar("reg_distri", ds, ds_r, ds_t)
    bar("pred_bboxes", bx, bx_r, bx_t)
    bar("pred_scores", sc, sc_r, sc_t)
    assert_close(items.cpu(), items_ref, tol, "loss items")
    ref_bufs = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(ref_bufs[name])
        else:
            assert_close(b.cpu(), ref_bufs[name], tol, name)

    ref_params, ref64_params = dict(ref.named_parameters()), dict(ref64.named_parameters())
    live = [n for n, _ in net.named_parameters() if ".rbr_reparam." not in n]

    def run_backward(up_l, up_d, retain):
        for m in (ref, ref64, net):
            m.zero_grad()
        torch.autograd.backward([out_ref[1][0], out_ref[1][1]], [up_l, up_d], retain_graph=retain)
        torch.autograd.backward([out64[1][0], out64[1][1]], [up_l.double(), up_d.double()], retain_graph=retain)

    # ---------------------------------------------------------------- backward B: the loss's own gradient (global L2 bar)
    loss_ref.backward(retain_graph=True)
    up = (out_ref[1][0].grad.clone(), out_ref[1][1].grad.clone())
    run_backward(up[0], up[1], retain=True)
    torch.autograd.backward([lg, ds], [up[0].to(device), up[1].to(device)])
    num_h = num_c = den = 0.0
    for n in live:
        t = ref64_params[n].grad
        num_h += float((dict(net.named_parameters())[n].grad.cpu().double() - t).pow(2).sum())
        num_c += float((ref_params[n].grad.double() - t).pow(2).sum())
        den += float(t.pow(2).sum())
    l2_h, l2_c = (num_h / den) ** 0.5, (num_c / den) ** 0.5
    if os.environ.get("SGX_TEST_DUMP"):
        with open(os.environ["SGX_TEST_DUMP"], "a") as f:
            f.write(f"backward B {variant} B={B} {size}: hip {l2_h:.3e} cpu fp32 {l2_c:.3e} ratio {l2_h / max(l2_c, 1e-30):.2f}\n")
    # measured under the default conv arithmetic of round 4 (bf16x3 per problem, "patch_bf3"; r4o): S 0.51 - 0.57, M 0.74, L 0.20 - the HIP
    # path is CLOSER to fp64 than ATen's CPU fp32 path on every model (single convolutions: 3-4x, profiles/r4n_conv_arithmetic_error_probe.txt),
    # and the bar is the 2x of every other three-way check.  Under the fp32-pipe modes ("fp32", "patch": sequential fp32 MFMA chains) the
    # same aggregate measured S 0.61, M 0.80 - 0.97, L 2.96 - 3.04 (r3 / r4i; this aggregate is the ILL-conditioned one, round-off amplified
    # 1e3 - 1e4x on both fp32 paths - see the docstring): those measurement modes keep the 3.5x of round 3.
    from super_gradients_amd import kernels as K_

    ratio_bar = 3.5 if K_.get_conv_math() in ("fp32", "patch") else 2.0
    assert l2_h <= max(10 * tol, ratio_bar * l2_c), f"loss-gradient L2 error vs fp64: hip {l2_h:.2e}, cpu fp32 {l2_c:.2e}"
    for n in [k for k, _ in net.named_parameters() if ".rbr_reparam." in k]:
        assert ref_params[n].grad is None

    # ---------------------------------------------------------------- backward A: well-conditioned upstream, per parameter
    out = net(x.to(device))  # the HIP blocks free their saved tensors in backward: run the forward again (same batch)
    gg = torch.Generator().manual_seed(21)
    up_l, up_d = torch.randn(lg_r.shape, generator=gg), torch.randn(ds_r.shape, generator=gg)
    run_backward(up_l, up_d, retain=False)
    torch.autograd.backward([out[1][0], out[1][1]], [up_l.to(device), up_d.to(device)])
    # Per-parameter relative L2 error (not max-norm): with ~1e-5 forward round-off a handful of ReLU pre-activations per
    # tensor change sign between ANY two fp32 implementations (measured: 0-3 flips per 2e5 elements, for the CPU fp32 path
    # as well as for HIP, both against fp64); each flip is an O(1) local error in a 3x3 patch of the input gradient, which
    # a max-norm would report as percent-level error although the tensors agree to ~1e-6 in L2.
    nmax = max(float(ref64_params[n].grad.norm()) for n in live)
    net_params = dict(net.named_parameters())
    worst = (0.0, "")
    acc_h = acc_c = acc_d = 0.0
    for n in live:
        t = ref64_params[n].grad
        if t.numel() == 1:
            # the bottleneck `alpha` scalars: d alpha = sum(x * dz) cancels ~1e3x (|sum| ~ 1e2 vs sum|.| ~ 1e5), so the O(1)
            # local errors of a single ReLU flip move it by percents on either fp32 path; their kernel (a dot product) is
            # checked exactly in test_kernels / test_blocks, and they are part of the global L2 bar above.
            continue
        # analytically-zero gradients (a per-channel constant in front of a training-mode BatchNorm) are pure round-off:
        # measure against max(own norm, 1e-3 x the largest gradient norm in the network)
        sc = max(float(t.norm()), 1e-3 * nmax)
        e_hip = float((net_params[n].grad.cpu().double() - t).norm()) / sc
        e_cpu = float((ref_params[n].grad.double() - t).norm()) / sc
        # one flipped ReLU element moves a weight gradient by ~1/sqrt(#pixels) of its norm (1.5e-2 at the 8x8 level of a
        # 256^2 image): the per-parameter bar can only exclude gross errors (a missing term is >= 1e-1); exactness of each
        # block's backward is pinned in test_blocks.py, the aggregate below is held to the CPU path's own accuracy.
        assert e_hip <= max(tol, 3.0 * e_cpu, 5e-2), f"grad {n}: L2 error vs fp64 hip {e_hip:.2e}, cpu fp32 {e_cpu:.2e}"
        worst = max(worst, (e_hip, n))
        acc_h += float((net_params[n].grad.cpu().double() - t).pow(2).sum())
        acc_c += float((ref_params[n].grad.double() - t).pow(2).sum())
        acc_d += float(t.pow(2).sum())
    a_h, a_c = (acc_h / acc_d) ** 0.5, (acc_c / acc_d) ** 0.5
    if os.environ.get("SGX_TEST_DUMP"):
        with open(os.environ["SGX_TEST_DUMP"], "a") as f:
            f.write(f"backward A {variant} B={B} {size} static={static}: hip {a_h:.3e} cpu fp32 {a_c:.3e}\n")
            contrib = sorted(((float((net_params[n].grad.cpu().double() - ref64_params[n].grad).pow(2).sum()) / acc_d,
                               float((ref_params[n].grad.double() - ref64_params[n].grad).pow(2).sum()) / acc_d, n)
                              for n in live if ref64_params[n].grad.numel() > 1), reverse=True)[:8]
            for eh, ec, n in contrib:
                f.write(f"    share of the squared error: hip {eh:.3e} cpu {ec:.3e}  {n}\n")
    # (2e-2: the ReLU-flip noise floor of this aggregate at these sizes - measured 1.4-1.6e-2 for the HIP path AND for the CPU fp32 path
    # against fp64, r2g; a CPU run that happens to flip fewer elements must not fail the HIP path.  The flip-free, strict form of this
    # check is test_yolo_nas_s_backward_exact_without_relu_flips below.)
    assert a_h <= max(10 * tol, 3.0 * a_c, 2e-2), f"random-upstream gradient L2 error vs fp64: hip {a_h:.2e}, cpu fp32 {a_c:.2e}"
    if True:
        print(f"[{variant}] loss-gradient L2 err vs fp64: hip {l2_h:.2e} cpu32 {l2_c:.2e}; random upstream: hip {a_h:.2e} cpu32 {a_c:.2e}; "
              f"worst parameter {worst[0]:.2e} {worst[1]}")
    return float(loss.detach()), float(loss_ref.detach())


@pytest.mark.gpu
def test_yolo_nas_s_train_step_parity(gpu_device):
    l, lr = _train_step_parity("s", 2, 320, gpu_device, 1e-4)
    assert abs(l - lr) <= 2e-4 * abs(lr)


def _spp_smallest_top2_gap(ref, x):
    """Smallest gap between the largest and second-largest value of any max-pool window of the reference forward, relative to the pooled map's
    largest magnitude (the oracle's SPP calls F.max_pool2d: wrapped for the duration of one forward)."""
    import oracle.yolo_nas as oracle_yolo_nas
    import torch.nn.functional as F

    real, gaps = F.max_pool2d, []

    def probe(t, k, stride=None, padding=0, *a, **kw):
        win = F.unfold(F.pad(t, (k // 2,) * 4, value=float("-inf")), k).view(t.shape[0], t.shape[1], k * k, -1)
        top = win.topk(2, dim=2).values
        gaps.append(float((top[:, :, 0] - top[:, :, 1]).min()) / float(t.abs().max()))
        return real(t, k, stride, padding, *a, **kw)

    class _Shim:
        def __getattr__(self, name):
            return probe if name == "max_pool2d" else getattr(F, name)

    saved = oracle_yolo_nas.F
    oracle_yolo_nas.F = _Shim()
    try:
        with torch.no_grad():
            ref.train()(x)
    finally:
        oracle_yolo_nas.F = saved
    assert gaps, "the reference forward made no max-pool call"
    return min(gaps)


def _backward_exact_without_relu_flips(variant, B, size, gpu_device, lazy_fp64=False, threads=None, certify=False):
    """The strict form of the whole-model backward check.  At random init a handful of ReLU pre-activations change sign between any two
    fp32 implementations, and every flip is an O(1) local gradient error - which is why the three-way checks can only bound the aggregate.
    Here every BatchNorm that feeds an activation gets bias +4 (weights in [0.5, 1]): all pre-activations stay positive on both paths, the
    network is smooth, and EVERY parameter gradient must agree with the CPU fp32 oracle element-wise within 1e-4 of the gradient's largest
    element - or, where the CPU fp32 path itself is further than that from the same oracle in fp64 (the cancellation-heavy `alpha` dot
    products, a few deep-stage weights), be no further from the fp64 truth than twice the CPU fp32 path.  A dropped or mis-scaled term
    anywhere in the hand-written backward (>= 1e-2) fails this.  lazy_fp64: run the fp64 oracle only if some parameter needs the
    tie-break.  certify: the full-size form - see below for what changes at 32 x 640^2."""
    import copy

    C = 80
    ref, net = _build_pair(variant, C, gpu_device)
    g = torch.Generator().manual_seed(5)
    for name, p in ref.named_parameters():
        if name.endswith("bn.weight") or name.endswith("post_bn.weight"):
            p.data.uniform_(0.5, 1.0, generator=g)
        elif (name.endswith("bn.bias") and "branch_3x3" not in name) or name.endswith("post_bn.bias"):
            p.data.fill_(4.0)
    if threads:
        torch.set_num_threads(threads)
    ref.train()
    # The other selection in the network is the SPP's max pooling (5 / 9 / 13 windows on the last map): where a window's two largest values
    # are a few ulp apart, two fp32 builds route that gradient to DIFFERENT pixels - the same O(1) local error as a ReLU flip.  r4n met it:
    # YOLO-NAS-L with input seed 8 has a 5x5 window whose top two differ by 3.0e-7 of the map's scale (5 ulp); the bf16x3 convolution modes
    # pick the other pixel, and context_module.cv1.weight is then 2.7e-2 off, everything upstream of it 5e-4 - 1e-2, everything downstream
    # 1e-5 (profiles/r4n_*).  So this premise is certified too: the input seed is the first from 8 on whose windows all keep their top two
    # more than 2e-6 of the map's scale apart on the reference forward (S and M: seed 8, smallest gap 4.1e-6 / 4.2e-6; L: seed 9, 1.1e-5).
    for in_seed in range(8, 24):
        x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(in_seed))
        if certify or _spp_smallest_top2_gap(copy.deepcopy(ref), x) >= 2e-6:
            break
    else:
        raise AssertionError("no input seed keeps the SPP's max-pool selections apart")
    if certify:
        # "+4" is not flip-free at 32 x 640^2: BatchNorm outputs are heavy-tailed and with ~1e8 elements per layer the minima reach -5 (r3f:
        # backbone.stage1 ... post_bn -4.98; the HIP and CPU paths then clip DIFFERENT elements and a whole sub-network's gradients differ by
        # 1e-3 - a property of the probe, not of either implementation).  So the premise is CERTIFIED: every activation-feeding BatchNorm whose
        # smallest output on this very batch is below +0.5 gets its bias raised by the shortfall, until no pre-activation is below +0.5.
        feeds = {n: m for n, m in ref.named_modules() if isinstance(m, torch.nn.BatchNorm2d) and not n.endswith("branch_3x3.bn")}
        mins = {}
        hooks = [m.register_forward_hook(lambda mod, inp, out, n=n: mins.__setitem__(n, float(out.min()))) for n, m in feeds.items()]
        state = {n: (m.running_mean.clone(), m.running_var.clone(), m.num_batches_tracked.clone()) for n, m in ref.named_modules()
                 if isinstance(m, torch.nn.BatchNorm2d)}
        for _ in range(10):
            with torch.no_grad():
                ref(x)
            low = {n: v for n, v in mins.items() if v < 0.5}
            if not low:
                break
            for n, v in low.items():
                feeds[n].bias.data += 1.5 * (0.5 - v)
        for h in hooks:
            h.remove()
        assert not low, f"could not certify the flip-free premise: {sorted(low.items(), key=lambda kv: kv[1])[:4]}"
        for n, m in ref.named_modules():  # the probing forwards must not count as training steps
            if n in state:
                m.running_mean.copy_(state[n][0]); m.running_var.copy_(state[n][1]); m.num_batches_tracked.copy_(state[n][2])
    net.load_state_dict(ref.state_dict(), strict=True)
    net.train()
    out_ref = ref(x)
    out = net(x.to(gpu_device))
    (lg, ds), (lg_r, ds_r) = out[1][:2], out_ref[1][:2]
    assert_close(lg.detach().cpu(), lg_r.detach(), 1e-4, "cls_logits")
    assert_close(ds.detach().cpu(), ds_r.detach(), 1e-4, "reg_distri")
    gg = torch.Generator().manual_seed(21)
    up_l, up_d = torch.randn(lg_r.shape, generator=gg), torch.randn(ds_r.shape, generator=gg)
    torch.autograd.backward([lg_r, ds_r], [up_l, up_d])
    torch.autograd.backward([lg, ds], [up_l.to(gpu_device), up_d.to(gpu_device)])
    ref_params = dict(ref.named_parameters())
    ref64_params = {}

    def fp64_truth():
        if not ref64_params:
            ref64 = copy.deepcopy(ref).double().train()
            ref64.zero_grad()
            out64 = ref64(x.double())
            torch.autograd.backward([out64[1][0], out64[1][1]], [up_l.double(), up_d.double()])
            ref64_params.update(dict(ref64.named_parameters()))
        return ref64_params

    if not lazy_fp64:
        fp64_truth()
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    # the bottlenecks' scalar `alpha`: d alpha = <x, dz> with x's per-channel mean at +4 here, so a 1e-7 per-channel offset of dz (the BN
    # backward's mean subtraction) is amplified ~1e3x: 1e-3 for those scalars (the dot-product kernel itself is checked in test_kernels)
    bar = lambda n: 1e-3 if ref_params[n].numel() == 1 else 1e-4  # noqa: E731
    # How much further from fp64 than the CPU fp32 path a gradient may be.  2x at the small sizes.  At 32 x 640^2 the certified premise
    # costs conditioning: ~1e8 elements per layer put BatchNorm minima at -8 .. -18 sigma, the biases that keep those positive reach 12,
    # and with activation mean / deviation ratios of 20-40 BOTH fp32 paths sit 1e-3 .. 1e-2 from fp64 (r3g: CPU up to 1.7e-2); measured
    # there, the HIP path is 2-3.5x the CPU path's distance on tensors (sequential fp32 MFMA chains against ATen's blocked GEMM sums) -
    # bar 5x - and the bottlenecks' scalar alpha (a dot product of a mean-12 activation with its gradient) is held to 2e-2.
    factor, scalar_floor = (5.0, 2e-2) if certify else (2.0, 0.0)
    errs, bad, tie = [], [], 0
    for n, p in net.named_parameters():
        if ".rbr_reparam." in n:
            continue
        rg = ref_params[n].grad
        if n.endswith("branch_3x3.bn.bias") or n.endswith("branch_1x1.bias"):
            # analytically zero (a per-channel constant in front of a training-mode BatchNorm): pure round-off of a sum over all pixels
            # on either path - both must stay at that level
            assert float(p.grad.abs().max()) <= max(10.0 * float(rg.abs().max()), 1e-3 * gmax), f"grad {n}: analytically zero, got {float(p.grad.abs().max()):.2e}"
            continue
        sc = max(float(rg.abs().max()), 1e-3 * gmax)
        e = float((p.grad.cpu().double() - rg.double()).abs().max()) / sc
        errs.append((e, n))
        if e > bar(n):
            tie += 1
            t = fp64_truth()[n].grad
            e_hip, e_cpu = float((p.grad.cpu().double() - t).abs().max()) / sc, float((rg.double() - t).abs().max()) / sc
            if e_hip > max(bar(n), factor * e_cpu, scalar_floor if ref_params[n].numel() == 1 else 0.0):
                bad.append(f"{n}: hip-cpu32 {e:.2e}, hip-fp64 {e_hip:.2e}, cpu32-fp64 {e_cpu:.2e}")
    errs.sort(reverse=True)
    dump = os.environ.get("SGX_TEST_DUMP")
    if dump:  # diagnosis aid: the whole list, worst first (hip vs cpu32; the fp64 columns for the parameters that needed the tie-break)
        with open(dump, "a") as f:
            f.write(f"# {variant} {B}x{size} conv_math={os.environ.get('SGX_CONV_MATH')} tuning={os.environ.get('SGX_CONV_TUNING')} "
                    f"group={os.environ.get('SGX_WGRAD_GROUP_GFLOP')}\n")
            for e, n in errs:
                f.write(f"{e:.3e} {n}\n")
            for b_ in bad:
                f.write(f"BAD {b_}\n")
    assert not bad, (f"{len(bad)} parameter gradients off by more than 1e-4 of their largest element (and further from fp64 than {factor}x the CPU fp32 "
                     f"path): {bad[:8]}")
    assert certify or tie <= 8, f"{tie} parameters needed the fp64 tie-break (more than 8)"
    print(f"[exact {variant} {B}x{size}] worst parameter gradient error {errs[0][0]:.2e} ({errs[0][1]}); fp64 tie-breaks: {tie}")


@pytest.mark.gpu
def test_yolo_nas_s_backward_exact_without_relu_flips(gpu_device):
    _backward_exact_without_relu_flips("s", 2, 256, gpu_device)


@pytest.mark.gpu
def test_yolo_nas_m_backward_exact_without_relu_flips(gpu_device):
    """YOLO-NAS-M (96 / 192-wide stages): the <128, 96, ...> and <128, 32, ...> tile families of the conv kernels and the 96x128 weight-gradient
    tiles get the element-wise backward check too."""
    _backward_exact_without_relu_flips("m", 1, 256, gpu_device)


@pytest.mark.gpu
def test_yolo_nas_l_backward_exact_without_relu_flips(gpu_device):
    """YOLO-NAS-L: the widest stages and the deepest CSP layers get the element-wise backward check too (round 4: the three-way aggregate of
    test_yolo_nas_l_train_step_parity can only bound L's gradient error from above - here every parameter is held to 1e-4)."""
    _backward_exact_without_relu_flips("l", 1, 256, gpu_device)


@pytest.mark.gpu
def test_yolo_nas_s_headline_config_backward_exact(gpu_device):
    """BASELINE.json configs[2] at FULL size (YOLO-NAS-S, 32 x 640^2): the conv problems, pixel splits and tuning-table entries the
    benchmark runs - element-wise gradient check of every parameter against the CPU fp32 oracle (flip-free form, see the helper, with the
    premise certified on this batch: no activation-feeding BatchNorm output below +0.5); the fp64 oracle runs only if a parameter needs
    the tie-break."""
    _backward_exact_without_relu_flips("s", 32, 640, gpu_device, lazy_fp64=True, threads=min(64, torch.get_num_threads() * 4), certify=True)


@pytest.mark.gpu
def test_yolo_nas_s_train_step_parity_atss(gpu_device):
    """(at 320 x 320 like the task-aligned variant above: a 2 x 256^2 batch ends in 8 x 8 maps - 128 values per BatchNorm channel - where ONE
    ReLU pre-activation that changes sign between two fp32 builds moves a tensor's gradient by 10-20 %: r4d measured exactly that between
    two builds whose forward outputs differ by 4e-5 and which both pass the element-wise flip-free checks below; the assigner, which is
    all this variant changes, does not enter the random-upstream backward at all)"""
    _train_step_parity("s", 2, 320, gpu_device, 1e-4, static=True)


@pytest.mark.gpu
def test_yolo_nas_m_train_step_parity(gpu_device):
    _train_step_parity("m", 1, 256, gpu_device, 1e-4)


@pytest.mark.gpu
def test_yolo_nas_l_train_step_parity(gpu_device):
    _train_step_parity("l", 1, 256, gpu_device, 1e-4)


@pytest.mark.gpu
def test_yolo_nas_eval_and_nms(gpu_device):
    """eval-mode forward (running statistics) + PPYoloEPostPredictionCallback against the oracle's post-processing."""
    from oracle import nms as onms

    ref, net = _build_pair("s", 80, gpu_device)
    ref.eval()
    net.eval()
    x = torch.rand(2, 3, 320, 320, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        (bx_r, sc_r), _ = ref(x)
    (bx, sc), raw = net(x.to(gpu_device))
    assert_close(bx.cpu(), bx_r, 1e-4, "eval boxes")
    assert_close(sc.cpu(), sc_r, 1e-4, "eval scores")
    cb = net.get_post_prediction_callback(conf=0.01, iou=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=True)
    res = cb(((bx, sc), raw))
    # NMS decisions are discontinuous in the inputs: feed the oracle the HIP model's own decoded predictions
    ref_res = onms.post_prediction(bx.cpu(), sc.cpu(), score_threshold=0.01, nms_threshold=0.7, nms_top_k=1000, max_predictions=300,
                                   multi_label_per_box=True, class_agnostic_nms=True)
    for a, b in zip(res, ref_res):
        assert torch.equal(a.cpu(), b)


@pytest.mark.gpu
def test_yolo_nas_deployment_form(gpu_device):
    """prep_model_for_conversion(full_fusion=True) (customizable_detector.py:106-118, qarepvgg_block.py:253-321): every QARepVGGBlock
    collapses to one 3x3 conv + bias + ReLU; eval outputs are unchanged within fp32 round-off, against the branch form and the oracle."""
    ref, net = _build_pair("s", 80, gpu_device)
    g = torch.Generator().manual_seed(4)
    for m in ref.modules():  # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.8, 1.2, generator=g)
    net.load_state_dict(ref.state_dict(), strict=True)
    ref.eval()
    net.materialize(gpu_device).eval()
    x = torch.rand(2, 3, 320, 320, generator=g)
    with torch.no_grad():
        (bx_r, sc_r), (lg_r, ds_r, *_r) = ref(x)
        (bx0, sc0), (lg0, ds0, *_r0) = net(x.to(gpu_device))
        net.prep_model_for_conversion(input_size=(320, 320), full_fusion=True)
        (bx1, sc1), (lg1, ds1, *_r1) = net(x.to(gpu_device))
    from super_gradients_amd.modules.qarepvgg_block import QARepVGGBlock

    assert all(m.fully_fused for m in net.modules() if isinstance(m, QARepVGGBlock))
    for a, b, r, name in ((lg1, lg0, lg_r, "logits"), (ds1, ds0, ds_r, "distri"), (bx1, bx0, bx_r, "boxes"), (sc1, sc0, sc_r, "scores")):
        assert_close(a.cpu(), b.cpu(), 1e-4, f"deployment form vs branch form: {name}")
        assert_close(a.cpu(), r, 2e-4, f"deployment form vs oracle: {name}")


@pytest.mark.gpu
def test_trainer_yolo_nas_recipe_shape(gpu_device, tmp_path):
    """The YOLO-NAS recipe's optimisation settings in miniature through Trainer.train(): AdamW lr 2e-4 wd 1e-5 with zero weight decay on
    bias/BN, linear batch warm-up + cosine, EMA (threshold decay), PPYoloELoss(TAL) - loss trajectory against the same loop on the oracle."""
    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.utils.callbacks import CosineLRScheduler
    import numpy as np

    ref, net = _build_pair("s", 80, gpu_device)
    n, bs = 3, 4
    loader = []
    for i in range(n):
        g = torch.Generator().manual_seed(20 + i)
        loader.append((torch.rand(bs, 3, 160, 160, generator=g), synthetic_targets(bs, seed=30 + i, kmax=4, size=160)))
    tp = dict(max_epochs=1, lr_mode="CosineLRScheduler", initial_lr=2e-4, loss=PPYoloELoss(num_classes=80, use_static_assigner=False), optimizer="AdamW",
              optimizer_params=dict(weight_decay=1e-5), zero_weight_decay_on_bias_and_bn=True, warmup_mode="LinearBatchLRWarmup", lr_warmup_steps=2,
              warmup_initial_lr=1e-6, cosine_final_lr_ratio=0.1, ema=True, ema_params=dict(decay=0.9997, decay_type="threshold"), silent_mode=True,
              save_model=False)
    from super_gradients_amd.training.metrics import DetectionMetrics_050

    cb = net.get_post_prediction_callback(conf=0.01, iou=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=False)
    tp.update(valid_metrics_list=[DetectionMetrics_050(num_cls=80, post_prediction_callback=cb, normalize_targets=True)], metric_to_watch="mAP@0.50")
    res = Trainer("yolo_nas_mini", ckpt_root_dir=str(tmp_path)).train(net, tp, loader, valid_loader=loader[:2])
    assert 0.0 <= res[0]["valid"]["mAP@0.50"] <= 1.0 and "Recall@0.50" in res[0]["valid"] and "PPYoloELoss/loss" in res[0]["valid"]
    decay = [p for k, p in ref.named_parameters() if p.dim() > 1]
    no_decay = [p for k, p in ref.named_parameters() if p.dim() <= 1]
    o = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay}], lr=2e-4, weight_decay=1e-5)
    crit = PPYoloELossOracle(80, use_static_assigner=False)
    ref.train()
    tot = torch.zeros(4)
    for b, (x, t) in enumerate(loader):
        if b < 2:
            for pg in o.param_groups:
                pg["lr"] = float(np.linspace(1e-6, 2e-4, 2)[b])
        loss, items = crit(ref(x), t)
        loss.backward()
        o.step()
        o.zero_grad()
        if b >= 2:
            for pg in o.param_groups:
                pg["lr"] = float(CosineLRScheduler.compute_learning_rate(max(0, b - 2), n - 2, 2e-4, 0.1))
        tot += items * bs
    tot /= n * bs
    got = res[0]["train"]
    for i, name in enumerate(["loss_cls", "loss_iou", "loss_dfl", "loss"]):
        # 2e-3 on a three-step AdamW trajectory: AdamW divides every gradient by its own running magnitude, so parameters whose gradient is
        # analytically zero (branch_3x3.bn.bias, branch_1x1.bias: exact zeros here, +-1e-9 round-off in the oracle) take noise-driven steps
        # in the oracle and none here; the per-step parity of loss and gradients is held to 1e-4 above
        assert abs(got["PPYoloELoss/" + name] - float(tot[i])) <= 2e-3 * abs(float(tot[i])), (name, got, float(tot[i]))


@pytest.mark.gpu
def test_yolo_nas_s_headline_config_parity(gpu_device):
    """BASELINE.json configs[2] at FULL size: YOLO-NAS-S, 32 synthetic 640x640 images (L = 8400 anchors), PPYoloELoss(TAL) + NMS,
    HIP path vs the CPU oracle (reference tests/unit_tests/ppyoloe_unit_test.py:42-81 checks loss items to places=4).
      forward  : raw head outputs / decoded predictions within 1e-4 (relative, max-norm) of the CPU fp32 path;
      loss     : the four loss items within 1e-4 of the oracle's on the oracle's own forward (end to end), AND - the assigner being
                 discontinuous - on IDENTICAL predictions (the HIP model's own raw outputs fed to both): assigned labels bit-exact,
                 assigned boxes / scores and loss items within 1e-4;
      NMS      : recipe settings (score 0.01, top-k 1000, IoU 0.7, max 300, multi-label) + class_agnostic_nms=True on the eval
                 forward of the same batch: rows bit-exact against the oracle's post-processing of the same predictions."""
    from oracle import nms as onms
    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.losses import PPYoloELoss

    tol, C, B, size = 1e-4, 80, 32, 640
    ref, net = _build_pair("s", C, gpu_device)
    ref.train()
    net.train()
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(42))
    targets = synthetic_targets(B, seed=42, kmax=20, size=size, num_classes=C)
    torch.set_num_threads(min(64, torch.get_num_threads() * 4))
    with torch.no_grad():
        out_ref = ref(x)
        orc = PPYoloELossOracle(C, use_static_assigner=False)
        loss_ref, items_ref = orc(out_ref, targets)
    out = net(x.to(gpu_device))
    crit = PPYoloELoss(num_classes=C, use_static_assigner=False)
    loss, items = crit(out, targets.to(gpu_device))
    (bx, sc), (lg, ds, an, pt, cnt, st) = out
    (bx_r, sc_r), (lg_r, ds_r, an_r, pt_r, cnt_r, st_r) = out_ref
    assert lg.shape[1] == 8400 and list(cnt) == list(cnt_r)
    assert_close(lg.detach().cpu(), lg_r, tol, "cls_logits @ bs32/640")
    assert_close(ds.detach().cpu(), ds_r, tol, "reg_distri @ bs32/640")
    assert_close(bx.detach().cpu(), bx_r, tol, "pred_bboxes @ bs32/640")
    assert_close(sc.detach().cpu(), sc_r, tol, "pred_scores @ bs32/640")
    assert_close(items.cpu(), items_ref, tol, "loss items, end to end")
    # identical predictions into both assigners + losses
    preds_cpu = (lg.detach().cpu(), ds.detach().cpu(), an_r, pt_r, cnt_r, st_r)
    _, a_label, a_box, a_score = orc.assign(preds_cpu, targets)
    _, items_same = orc((None, preds_cpu), targets)
    w = (1.0, 2.5, 0.5)
    o = K.ppyoloe_loss_fwd(lg.detach(), ds.detach(), an, pt, st, targets.to(gpu_device), [int(c) for c in cnt], False, True, w)
    assert torch.equal(o["label"].cpu().long(), a_label), "assigned labels differ at bs32/640"
    pos = a_label != C
    assert int(pos.sum()) > 32
    assert_close(o["box"].cpu()[pos], a_box[pos], 1e-6, "assigned boxes")
    assert_close(o["score"].cpu(), a_score, tol, "assigned scores")
    it, _ = K.ppyoloe_loss_finalize(o["sums"], w, 1.0)
    assert_close(it.cpu(), items_same, tol, "loss items on identical predictions")
    # NMS at the recipe settings on the eval forward of the same batch
    net.eval()
    with torch.no_grad():
        (ebx, esc), raw = net(x.to(gpu_device))
    kw = dict(score_threshold=0.01, nms_threshold=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=True)
    cb = net.get_post_prediction_callback(conf=0.01, iou=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=True)
    res = cb(((ebx, esc), raw))
    ref_res = onms.post_prediction(ebx.cpu(), esc.cpu(), **kw)
    assert len(res) == B
    nrows = 0
    for a, b in zip(res, ref_res):
        assert torch.equal(a.cpu(), b), "NMS rows differ at bs32/640"
        nrows += int(b.shape[0])
    assert nrows > 0


@pytest.mark.gpu
def test_yolo_nas_s_step_is_bit_identical_with_filter_planes(gpu_device):
    """Pre-split filter planes (round 5) over a whole YOLO-NAS-S train step: with the planes served (default) and on a network without them
    the step must produce the same bits - loss and the whole gradient arena - because a launch that copies planes stages exactly the
    pieces the splitting launch computes.  (The premise that one build repeats itself bit for bit is checked first: without it the
    comparison would say nothing.)  The hit counter shows that the planes step really took the planes path in most of its launches."""
    from super_gradients_amd._lib import lib
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from util import synthetic_targets

    def build(planes):
        torch.manual_seed(3)
        net = models.get("yolo_nas_s", num_classes=80).materialize(gpu_device).train()
        if not planes:  # (not through SGX_FILTER_PLANES: the library reads that variable once, when it is loaded, as its process-wide mode)
            net.drop_filter_planes()
            net._fp_jobs = net._fp_dev = net._fp_buf = None
        return net

    x = torch.rand(4, 3, 320, 320, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    t = synthetic_targets(4, seed=2, kmax=6, size=320).to(gpu_device)
    crit = PPYoloELoss(num_classes=80, use_static_assigner=False)

    def step(net):
        net.zero_grad()
        loss, _ = crit(net(x), t)
        loss.backward()
        net.join_side()
        torch.cuda.synchronize()
        return loss.detach().cpu().clone(), net.g_arena.buf.cpu().clone()

    plain, fast = build(False), build(True)
    assert plain._fp_jobs is None and fast._fp_jobs is not None
    fast.load_state_dict(plain.state_dict())
    l0, g0 = step(plain)
    l1, g1 = step(plain)
    assert torch.equal(l0, l1), "one build does not repeat its own step bit for bit"
    assert_gradient_arenas_match(plain, g0, g1, "one build, the same step twice")
    h0 = lib().sgx_debug_filter_planes_hits()
    l2, g2 = step(fast)
    hits = lib().sgx_debug_filter_planes_hits() - h0
    assert hits >= 100, f"only {hits} launches of the step read filter planes"
    assert torch.equal(l0, l2), f"loss differs with filter planes: {float(l0)} vs {float(l2)}"
    assert_gradient_arenas_match(fast, g0, g2, "filter planes against splitting while staging")


@pytest.mark.gpu
def test_yolo_nas_s_step_is_bit_identical_with_branch_stream(gpu_device):
    """Branch stream (round 6; engine.fork_branch): YoloNASCSPLayer's conv2 chain, the up stages' skip branches, the coarse head levels and the
    batch re-layout run on a second in-order stream beside the main chain, forward and backward.  Same kernels on the same operands: the step
    must produce the same bits as the single-chain step - loss and the whole gradient arena - and keep producing them (a missing join or a
    buffer recycled across streams would show up as a step that does not repeat).  Weight gradients one launch per layer in both networks:
    the grouped launches size their splits by what happens to be queued, and the branch forks flush the queue at other points."""
    from super_gradients_amd.training import models
    from super_gradients_amd.training.losses import PPYoloELoss
    from util import synthetic_targets

    def build(branch):
        torch.manual_seed(3)
        net = models.get("yolo_nas_s", num_classes=80).materialize(gpu_device).train()
        net.wg_group_flops = 0.0
        if branch:
            if net.branch_stream is None:  # (switched off through the environment: build the streams the default would have)
                net.branch_stream = torch.cuda.Stream(device=gpu_device)
                net.branch_lanes = [net.branch_stream, torch.cuda.Stream(device=gpu_device)]
            net.branch_mode, net.branch_sites, net.branch_max_tiles = 3, 63, 1 << 30
        else:
            net.branch_mode = 0
        return net

    x = torch.rand(4, 3, 320, 320, generator=torch.Generator().manual_seed(1)).to(gpu_device)
    t = synthetic_targets(4, seed=2, kmax=6, size=320).to(gpu_device)
    crit = PPYoloELoss(num_classes=80, use_static_assigner=False)

    def step(net):
        net.zero_grad()
        loss, _ = crit(net(x), t)
        loss.backward()
        net.join_side()
        torch.cuda.synchronize()
        return loss.detach().cpu().clone(), net.g_arena.buf.cpu().clone()

    plain, forked = build(False), build(True)
    forked.load_state_dict(plain.state_dict())
    l0, g0 = step(plain)
    for rep in range(4):
        l1, g1 = step(forked)
        assert torch.equal(l0, l1), f"repeat {rep}: loss differs with the branch stream: {float(l0)} vs {float(l1)}"
        assert_gradient_arenas_match(forked, g0, g1, f"repeat {rep}: branch stream against the single chain")  # (bits; the d alpha scalars to 4 ulp: see there)
    assert sum(bool(getattr(m, "_branched", False)) for m in forked.modules()) >= 8, "the forked network did not fork"


@pytest.mark.gpu
@pytest.mark.parametrize("variant,size,batch,anchors", [("m", 640, 32, 8400), ("l", 1280, 8, 33600)], ids=["m_640_bs32", "l_1280_bs8"])
def test_yolo_nas_other_baseline_configs_parity_at_full_size(gpu_device, variant, size, batch, anchors):
    """BASELINE.json configs[3] / [4] AT THEIR OWN SIZE (round 5 held M and L to the oracle at B = 1, 256 x 256 only): YOLO-NAS-M, 32 x
    640 x 640, and YOLO-NAS-L, 8 x 1280 x 1280 (33 600 anchors), training-mode forward + PPYoloELoss(TAL) against the CPU oracle:
    raw head outputs and decoded predictions within 1e-4 (relative, max-norm), the four loss items within 1e-4, and - on IDENTICAL
    predictions - assigned labels bit-exact."""
    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd import kernels as K
    from super_gradients_amd.training.losses import PPYoloELoss

    tol, C = 1e-4, 80
    ref, net = _build_pair(variant, C, gpu_device)
    ref.train()
    net.train()
    x = torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(42))
    targets = synthetic_targets(batch, seed=42, kmax=20, size=size, num_classes=C)
    torch.set_num_threads(min(64, torch.get_num_threads() * 4))
    with torch.no_grad():
        out_ref = ref(x)
        orc = PPYoloELossOracle(C, use_static_assigner=False)
        _, items_ref = orc(out_ref, targets)
        out = net(x.to(gpu_device))
        _, items = PPYoloELoss(num_classes=C, use_static_assigner=False)(out, targets.to(gpu_device))
    (bx, sc), (lg, ds, an, pt, cnt, st) = out
    (bx_r, sc_r), (lg_r, ds_r, an_r, pt_r, cnt_r, st_r) = out_ref
    assert lg.shape[1] == anchors and list(cnt) == list(cnt_r)
    tag = f"@ {variant} bs{batch}/{size}"
    assert_close(lg.detach().cpu(), lg_r, tol, "cls_logits " + tag)
    assert_close(ds.detach().cpu(), ds_r, tol, "reg_distri " + tag)
    assert_close(bx.detach().cpu(), bx_r, tol, "pred_bboxes " + tag)
    assert_close(sc.detach().cpu(), sc_r, tol, "pred_scores " + tag)
    assert_close(items.cpu(), items_ref, tol, "loss items, end to end " + tag)
    preds_cpu = (lg.detach().cpu(), ds.detach().cpu(), an_r, pt_r, cnt_r, st_r)
    _, a_label, _, _ = orc.assign(preds_cpu, targets)
    o = K.ppyoloe_loss_fwd(lg.detach(), ds.detach(), an, pt, st, targets.to(gpu_device), [int(c) for c in cnt], False, True, (1.0, 2.5, 0.5))
    assert torch.equal(o["label"].cpu().long(), a_label), "assigned labels differ " + tag
    assert int((a_label != C).sum()) > batch
