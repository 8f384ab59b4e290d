"""Block-level parity (forward, input gradient, every parameter gradient, BN running stats) of the HIP blocks against the
oracle's CPU modules (oracle/yolo_nas.py, which restate the reference's blocks with identical state_dict keys)."""
import pytest
import torch
from torch import nn

from util import assert_close, to_nchw_cpu, to_nhwc


class _Net(nn.Module):
    """Minimal SgxNetwork around one block, so that arenas / materialisation are exercised exactly as in a model."""


def _wrap(block, device):
    from super_gradients_amd.modules.engine import SgxNetwork

    class One(SgxNetwork):
        def __init__(self):
            super().__init__()
            self.b = block

    net = One()
    net.materialize(device)
    return net


def _randomize(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for name, p in mod.named_parameters():
        if name.endswith("bn.weight") or name.endswith("post_bn.weight"):
            p.data.uniform_(0.5, 1.5, generator=g)
        elif p.dim() <= 1:
            p.data.add_(torch.randn(p.shape, generator=g) * 0.2)
    for m in mod.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.eps, m.momentum = 1e-3, 0.03


def _check(ref, blk, x, device, tol=2e-5, key_prefix="b.", prefetch=False, need_dx=True):
    from super_gradients_amd.modules.layers import BatchNorm

    _randomize(ref, 1)
    for m in blk.modules():
        if isinstance(m, BatchNorm):
            m.eps, m.momentum = 1e-3, 0.03
    net = _wrap(blk, device)
    missing = blk.load_state_dict(ref.state_dict(), strict=True)
    assert not missing.missing_keys
    ref.train()
    net.train()
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(dy)
    net.zero_grad()
    if prefetch:  # what NetFunction.forward does at the start of every training step: per-step weight preparation of all blocks
        net.prefetch_dgrad_weights()
    yd = blk.fwd(to_nhwc(x, device))
    assert_close(to_nchw_cpu(yd), y, tol, "forward")
    dx = blk.bwd(to_nhwc(dy, device)) if need_dx else blk.bwd(to_nhwc(dy, device), need_dx=False)
    net.join_side()  # weight gradients run on the network's side stream
    if need_dx:
        assert_close(to_nchw_cpu(dx), xr.grad, 5 * tol, "input gradient")
    rp = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for name, p in blk.named_parameters():
        if "rbr_reparam" in name:
            continue
        rg = rp[name].grad
        e = float((p.grad.cpu().double() - rg.double()).abs().max()) / max(float(rg.abs().max()), 1e-2 * gmax)
        assert e <= 5 * tol, f"grad {name}: {e:.3e}"
    rb = dict(ref.named_buffers())
    for name, b in blk.named_buffers():
        if not name.endswith("num_batches_tracked"):
            assert_close(b.cpu(), rb[name], tol, name)


def _shape(backend, gpu, emu):
    return gpu if backend.type == "cuda" else emu


@pytest.mark.parametrize("k,s", [(1, 1), (3, 1), (3, 2)])
def test_conv_block(backend, k, s):
    from oracle.yolo_nas import ConvBnAct
    from super_gradients_amd.modules import Conv

    n, c, h, w, co = _shape(backend, (2, 96, 10, 10, 64), (1, 8, 6, 6, 4))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(ConvBnAct(c, co, k, s), Conv(c, co, k, s, "relu"), x, backend)


@pytest.mark.parametrize("stride,cout", [(1, None), (2, 2)])
def test_qarepvgg_block(backend, stride, cout):
    from oracle.yolo_nas import QARep
    from super_gradients_amd.modules import QARepVGGBlock

    n, c, h, w = _shape(backend, (2, 64, 10, 10), (1, 8, 6, 6))
    co = c if cout is None else c * cout
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(QARep(c, co, stride, residual=stride == 1), QARepVGGBlock(c, co, stride=stride, use_residual_connection=stride == 1), x, backend)


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("stride,cout,act", [(1, None, "relu"), (2, 2, "relu"), (1, None, "silu"), (2, 1, "relu")])
def test_qarepvgg_block_two_branch_launch(backend, stride, cout, act, wide):
    """The training form the network runs (after the per-step weight preparation): both convolution branches in one launch with the
    identity folded into the 1x1 filter, both BatchNorms finalised from the five conv-epilogue moments, one forward sweep, two backward
    sweeps, one data-gradient launch over two K-axis sources - against the oracle's QARepVGG block: forward, input gradient, every
    parameter gradient (d beta of branch_3x3.bn is analytically zero), running statistics."""
    from oracle.yolo_nas import QARep
    from super_gradients_amd.modules import QARepVGGBlock

    # channel counts that are / are not multiples of 32 take the 32-deep / 16-deep slab kernels; 96 output channels the 128x32 tile
    if wide:
        n, c, h, w = _shape(backend, (2, 96, 12, 10), (1, 32, 5, 4))
    else:
        n, c, h, w = _shape(backend, (2, 48, 23, 17), (2, 16, 6, 5))
    co = c if cout is None else c * cout
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    kw = {} if act == "relu" else dict(activation_type=nn.SiLU)
    ref = QARep(c, co, stride, residual=stride == 1)
    if act == "silu":  # the oracle block hard-wires YOLO-NAS's ReLU: same arithmetic with the other activation the kernels offer
        import types

        def fwd(self, x):
            s = self.branch_3x3(x) + self.alpha * self.branch_1x1(x)
            return nn.functional.silu(self.post_bn(s + x if self.residual else s))

        ref.forward = types.MethodType(fwd, ref)
    blk = QARepVGGBlock(c, co, stride=stride, use_residual_connection=stride == 1, **kw)
    _check(ref, blk, x, backend, prefetch=True)
    assert blk._w1p is not None and blk._ctx is None


def test_qarepvgg_block_two_branch_launch_rgb_stem(backend):
    """Round 6: the RGB stem (3 input channels, padded to 4; stride 2, no residual) takes the two-output launch too - on the flattened
    (tap, channel) K axis - instead of the general five-sweep sequence at the largest map of the network: forward, every parameter
    gradient, running statistics against the oracle's block (the block produces no input gradient in the network, and none is asked here)."""
    from oracle.yolo_nas import QARep
    from super_gradients_amd.modules import QARepVGGBlock

    n, c, h, w = _shape(backend, (2, 4, 46, 38), (1, 4, 12, 10))  # (4 channels: the padded RGB batch the network's stem reads)
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    blk = QARepVGGBlock(c, 48, stride=2, use_residual_connection=False)
    _check(QARep(c, 48, 2, residual=False), blk, x, backend, prefetch=True, need_dx=False)
    assert blk._w1p is not None and blk._ctx is None


@pytest.mark.parametrize("stride", [1, 2])
def test_qarepvgg_block_learnable_alpha(backend, stride):
    """QARepVGGBlock(use_alpha=True) (qarepvgg_block.py:130-136, 198-203): learnable multiplier of the 1x1 branch - forward, every
    gradient incl. d alpha = <ds, conv1x1(x) + b>, running statistics; then the fused deployment form with that alpha folded in."""
    from oracle.yolo_nas import QARep
    from super_gradients_amd.modules import QARepVGGBlock

    n, c, h, w = _shape(backend, (2, 64, 10, 10), (2, 8, 6, 5))
    co = c * stride
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    ref, blk = QARep(c, co, stride, residual=stride == 1, use_alpha=True), QARepVGGBlock(c, co, stride=stride, use_residual_connection=stride == 1, use_alpha=True)
    ref.alpha.data.fill_(0.7)
    _check(ref, blk, x, backend)
    assert blk.alpha.grad is not None and float(blk.alpha.grad.abs().sum()) > 0
    # deployment form: alpha enters the fused kernel and bias (qarepvgg_block.py:206-230)
    ref.eval()
    blk.eval()
    with torch.no_grad():
        want = ref(x)
        blk.full_fusion()
        got = blk.fwd(to_nhwc(x, backend))
    assert_close(to_nchw_cpu(got), want, 1e-4, "fused forward with alpha")


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("stride", [1, 2])
def test_oracle_qarepvgg_alpha_live(stride):
    """oracle.QARep(use_alpha=True) against the reference's own QARepVGGBlock source executed through the import shim."""
    from oracle import ref_shim
    from oracle.yolo_nas import QARep

    ref_shim.install()
    from super_gradients.modules.qarepvgg_block import QARepVGGBlock as RefBlock

    c, co = 8, 8 * stride
    torch.manual_seed(3)
    r = RefBlock(c, co, stride=stride, use_alpha=True, use_residual_connection=stride == 1).train()
    o = QARep(c, co, stride, residual=stride == 1, use_alpha=True).train()
    r.alpha.data.fill_(0.6)
    missing = o.load_state_dict(r.state_dict(), strict=False)
    assert not missing.missing_keys
    x = torch.randn(2, c, 6, 5, generator=torch.Generator().manual_seed(1))
    xr, xo = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr, yo = r(xr), o(xo)
    assert torch.equal(yr, yo)
    g = torch.randn(yr.shape, generator=torch.Generator().manual_seed(2))
    yr.backward(g)
    yo.backward(g)
    assert torch.equal(xr.grad, xo.grad) and torch.equal(r.alpha.grad, o.alpha.grad)


@pytest.mark.parametrize("k,s", [(1, 1), (3, 2)])
def test_conv_block_folded_eval_form(backend, k, s):
    """prep_model_for_conversion on a Conv block: eval forward = ONE conv launch with the BatchNorm folded into filter and bias and the
    activation in the epilogue; equal to the unfolded eval sequence and to the oracle within fp32 round-off; the parameters are untouched
    and a training forward drops the folded copy (it would be stale after the step)."""
    from oracle.yolo_nas import ConvBnAct
    from super_gradients_amd.modules import Conv

    n, cin, cout, h, w = _shape(backend, (2, 32, 64, 20, 20), (1, 8, 8, 6, 6))
    ref = ConvBnAct(cin, cout, k, s)
    _randomize(ref, 3)
    blk = Conv(cin, cout, k, s, "relu")
    blk.bn.eps, blk.bn.momentum = 1e-3, 0.03  # the YOLO-NAS arch values the oracle blocks are built with
    net = _wrap(blk, backend)
    blk.load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(n, cin, h, w, generator=torch.Generator().manual_seed(1))
    ref.eval()
    net.eval()
    with torch.no_grad():
        y_ref = ref(x)
        y0 = to_nchw_cpu(blk.fwd(to_nhwc(x, backend)))
        before = {kk: v.clone() for kk, v in blk.state_dict().items()}
        net.prep_model_for_conversion(input_size=(h, w))
        assert blk._folded is not None
        y1 = to_nchw_cpu(blk.fwd(to_nhwc(x, backend)))
    assert_close(y1, y0, 1e-5, "folded vs unfolded eval form")
    assert_close(y1, y_ref, 2e-5, "folded eval form vs oracle")
    assert all(torch.equal(v, blk.state_dict()[kk]) for kk, v in before.items())
    net.train()
    assert blk._folded is None
    net.eval()
    net.prep_model_for_conversion(input_size=(h, w))
    net.train()
    blk._folded = ("stale",)  # e.g. prepared while in training mode
    blk.fwd(to_nhwc(x, backend))
    assert blk._folded is None
    # new weights arriving in eval mode (load_state_dict, an EMA swap) must drop the folded copy too - not serve the old filter
    net.eval()
    net.prep_model_for_conversion(input_size=(h, w))
    assert blk._folded is not None
    sd = {kk: (v * 1.5 if kk.endswith("conv.weight") else v) for kk, v in blk.state_dict().items()}
    net.load_state_dict({"b." + kk: v for kk, v in sd.items()})
    assert blk._folded is None
    with torch.no_grad():
        y3 = to_nchw_cpu(blk.fwd(to_nhwc(x, backend)))
    ref.load_state_dict(sd)
    with torch.no_grad():
        assert_close(y3, ref(x), 2e-5, "eval forward after load_state_dict on a prepared model")


@pytest.mark.parametrize("two_branch", [False, True])
@pytest.mark.parametrize("concat", [False, True])
def test_csp_layer(backend, concat, two_branch):
    """two_branch: the blocks run their one-launch-per-pair form, the bottleneck shortcut `alpha * x + cv2(cv1(x))` (yolo_stages.py:61-63)
    rides in cv2's last sweep and its gradient in cv1's data-gradient launch; concat: that launch accumulates onto the concat slice."""
    from oracle.yolo_nas import CSP, _qa
    from super_gradients_amd.modules import QARepVGGBlock
    from super_gradients_amd.training.models.detection_models.yolo_nas.yolo_stages import YoloNASCSPLayer

    n, c, h, w, hid, nb = _shape(backend, (2, 96, 10, 10, 32, 2), (1, 32, 5, 4, 16, 2) if two_branch else (1, 8, 4, 4, 4, 2))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    blk = YoloNASCSPLayer(c, c, nb, QARepVGGBlock, "relu", True, hidden_channels=hid, concat_intermediates=concat)
    _check(CSP(c, c, nb, hid, concat, _qa), blk, x, backend, prefetch=two_branch)
    assert not two_branch or all(b.cv1._w1p is not None for b in blk.bottlenecks)


@pytest.mark.parametrize("kind", ["csp_qarep", "csp_plain", "head"])
def test_bn_reduce_rides_in_data_gradients(backend, kind):
    """Round 4: a plain conv -> BatchNorm -> activation layer whose output gradient is finalised by a convolution's data gradient gets its
    BatchNorm-backward reduce from that launch's epilogue (kernels.BnReduceRequest) instead of a sweep over (dy, saved conv output).
    The same block, same weights, same batch with the hand-over on and off: identical forward, gradients equal to fp32 round-off (the
    partial rows regroup with the launch's tiles; the fp64 finalize is the same) - and the requests are really taken."""
    from super_gradients_amd import kernels as KK
    from super_gradients_amd.modules import Conv, QARepVGGBlock
    from super_gradients_amd.training.models.detection_models.yolo_nas.dfl_heads import YoloNASDFLHead
    from super_gradients_amd.training.models.detection_models.yolo_nas.yolo_stages import YoloNASCSPLayer
    from functools import partial

    gpu = backend.type == "cuda"
    n, c, h, w, hid = (2, 64, 24, 24, 32) if gpu else (1, 32, 5, 6, 16)
    results = {}
    for fuse in (True, False):
        torch.manual_seed(3)
        if kind == "csp_qarep":
            blk = YoloNASCSPLayer(c, c, 2, QARepVGGBlock, "relu", True, hidden_channels=hid, concat_intermediates=False)
        elif kind == "csp_plain":
            blk = YoloNASCSPLayer(c, c, 2, partial(Conv, kernel=3, stride=1), "relu", True, hidden_channels=hid, concat_intermediates=True)
        else:
            blk = YoloNASDFLHead(c, hid, 1.0, 0, 8, 8, 16)
        net = _wrap(blk, backend)
        net.fuse_bn_reduce = fuse
        net.train()
        net.zero_grad()
        net.prefetch_dgrad_weights()
        KK.BN_REQ_STATS["taken"] = KK.BN_REQ_STATS["declined"] = 0
        g = torch.Generator().manual_seed(9)
        x = to_nhwc(torch.randn(n, c, h, w, generator=g) + 0.3, backend)
        if kind == "head":
            reg, cls = torch.empty(n, h, w, 68, device=backend), torch.empty(n, h, w, 8, device=backend)
            blk.fwd(x, out=(reg, cls))
            out = torch.cat([reg.flatten(), cls.flatten()]).cpu()
            dx = blk.bwd(torch.randn(n, h, w, 68, generator=g).to(backend), torch.randn(n, h, w, 8, generator=g).to(backend))
        else:
            y = blk.fwd(x)
            out = y.cpu().clone()
            dx = blk.bwd(torch.randn(tuple(y.shape), generator=g).to(backend))
        net.join_side()
        results[fuse] = (out, dx.cpu().clone(), {k: p.grad.cpu().clone() for k, p in blk.named_parameters() if p.grad is not None}, dict(KK.BN_REQ_STATS))
    on, off = results[True], results[False]
    # conv2 + conv1 (+ the first conv of each plain bottleneck); the two 3x3 blocks + the stem of a head
    expect = {"csp_qarep": 2, "csp_plain": 4, "head": 3}[kind]
    assert on[3]["taken"] == expect and on[3]["declined"] == 0 and off[3]["taken"] == 0, f"requests taken: {on[3]} / {off[3]}"
    assert torch.equal(on[0], off[0])
    assert_close(on[1], off[1], 2e-5, "input gradient")
    gmax = max(float(v.abs().max()) for v in off[2].values())
    for k, v in off[2].items():
        e = float((on[2][k].double() - v.double()).abs().max()) / max(float(v.abs().max()), 1e-2 * gmax)
        assert e <= 5e-5, f"grad {k}: {e:.3e}"


def test_spp(backend):
    from oracle.yolo_nas import SPP as OSPP
    from super_gradients_amd.modules.detection_modules import SPP

    n, c, h, w = _shape(backend, (2, 64, 10, 10), (1, 8, 6, 6))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(OSPP(c, c), SPP(c, c, (5, 9, 13), "relu"), x, backend)


@pytest.mark.parametrize("kind,stride", [("basic", 1), ("basic", 2), ("bottleneck", 1), ("bottleneck", 2)])
def test_resnet_block(backend, kind, stride):
    """BasicResNetBlock / Bottleneck (classification_models/resnet.py:26-84): identity and conv shortcuts, ReLU after the add."""
    from oracle.resnet import BasicBlock, BottleneckBlock
    from super_gradients_amd.training.models.classification_models.resnet import BasicResNetBlock, Bottleneck

    n, cin, planes, hw = _shape(backend, (2, 64, 64, 28), (1, 8, 8, 6))
    if kind == "basic":
        cin_eff = cin if stride == 1 else cin // 2
        ref, blk = BasicBlock(cin_eff, planes, stride, 1), BasicResNetBlock(cin_eff, planes, stride, 1)
    else:
        cin_eff = planes * 4 if stride == 1 else cin
        ref, blk = BottleneckBlock(cin_eff, planes, stride, 4), Bottleneck(cin_eff, planes, stride, 4)
    x = torch.randn(n, cin_eff, hw, hw, generator=torch.Generator().manual_seed(0))
    _check(ref, blk, x, backend)


@pytest.mark.parametrize("full", [False, True])
@pytest.mark.parametrize("stride,cout", [(1, None), (2, 2)])
def test_qarepvgg_fusion(backend, full, stride, cout):
    """Re-parameterisation (reference qarepvgg_block.py:206-321; test pattern of tests/unit_tests/repvgg_unit_test.py:43-113): the fused
    3x3 kernel / bias equal the reference's algebra, and the deployment-form eval forward equals the branch-form eval forward."""
    from oracle import ref_shim
    from super_gradients_amd.modules.layers import BatchNorm
    from super_gradients_amd.modules.qarepvgg_block import QARepVGGBlock

    n, c, hw = _shape(backend, (2, 64, 20), (1, 8, 6))
    co = c if cout is None else c * cout
    blk = QARepVGGBlock(c, co, stride=stride)
    g = torch.Generator().manual_seed(3)
    for m in blk.modules():
        if isinstance(m, BatchNorm):
            m.eps = 1e-3
            m.weight.data.uniform_(0.5, 1.5, generator=g)
            m.bias.data.normal_(0, 0.2, generator=g)
            m.running_mean.normal_(0, 0.3, generator=g)
            m.running_var.uniform_(0.5, 1.5, generator=g)
    net = _wrap(blk, backend)
    net.eval()
    x = torch.randn(n, c, hw, hw, generator=g)
    y_branches = to_nchw_cpu(blk.fwd(to_nhwc(x, backend)))
    blk.prep_model_for_conversion(full_fusion=full)
    y_fused = to_nchw_cpu(blk.fwd(to_nhwc(x, backend)))
    assert_close(y_fused, y_branches, 2e-5, "fused eval forward vs branch eval forward")
    if ref_shim.available():
        ref_shim.install()
        from super_gradients.modules.qarepvgg_block import QARepVGGBlock as Ref

        ref = Ref(c, co, stride=stride, use_residual_connection=(stride == 1 and co == c))
        for m in ref.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eps = 1e-3
        sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
        branch_sd = {k: v for k, v in sd.items() if not k.startswith("rbr_reparam")}
        ref.load_state_dict(branch_sd, strict=False)
        ref.eval()
        (ref.full_fusion if full else ref.partial_fusion)()
        assert_close(blk.rbr_reparam.weight.detach().cpu(), ref.rbr_reparam.weight.detach(), 1e-6, "fused kernel vs reference")
        assert_close(blk.rbr_reparam.bias.detach().cpu(), ref.rbr_reparam.bias.detach(), 1e-6, "fused bias vs reference")
        assert_close(y_fused, ref(x).detach(), 2e-5, "fused forward vs reference fused forward")


# --------------------------------------------------------------------------------------------- PP-YOLOE blocks
def test_repvgg_block(backend):
    from oracle.pp_yolo_e import RepVGGBlock as O
    from super_gradients_amd.modules.repvgg_block import RepVGGBlock

    n, c, h, w = _shape(backend, (2, 64, 10, 10), (1, 8, 6, 6))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(O(c, c, nn.SiLU), RepVGGBlock(c, c, activation_type="silu", use_residual_connection=False), x, backend)


def test_repvgg_block_learnable_alpha(backend):
    """RepVGGBlock(use_alpha=True) (PP-YOLOE+; reference modules/repvgg_block.py:31,77-87,94-104): y = act(bn3(conv3(x)) + alpha * bn1(conv1(x)))
    with alpha a learnable [1] parameter - forward, input gradient, every parameter gradient INCLUDING alpha's and the 1x1 branch's
    BatchNorm parameters (which see alpha * g), running statistics; then the deployment form, where alpha enters the fused kernel and bias."""
    from oracle.pp_yolo_e import RepVGGBlock as O
    from super_gradients_amd.modules.repvgg_block import RepVGGBlock

    n, c, h, w = _shape(backend, (2, 64, 10, 10), (1, 8, 6, 6))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    ref, blk = O(c, c, nn.SiLU, use_alpha=True), RepVGGBlock(c, c, activation_type="silu", use_residual_connection=False, use_alpha=True)
    with torch.no_grad():
        ref.alpha.fill_(0.7)
    _check(ref, blk, x, backend)
    assert abs(float(blk.alpha.detach().cpu()) - float(ref.alpha.detach())) < 1e-6 and abs(float(ref.alpha.detach()) - 1.0) > 0.05 and blk.alpha.grad is not None
    ref.eval()
    blk.eval()
    blk.fuse_block_residual_branches()
    k, b = ref.fused()
    assert_close(blk.rbr_reparam.weight.detach().cpu(), k.detach(), 1e-6, "fused kernel")
    assert_close(blk.rbr_reparam.bias.detach().cpu(), b.detach(), 1e-6, "fused bias")
    with torch.no_grad():
        assert_close(to_nchw_cpu(blk.fwd(to_nhwc(x, backend))), ref(x), 2e-5, "fused forward")


@pytest.mark.skipif(not __import__("oracle.ref_shim", fromlist=["x"]).available(), reason="/root/reference not present (GPU box)")
def test_oracle_repvgg_alpha_live():
    """oracle.pp_yolo_e.RepVGGBlock(use_alpha=True) against the reference's own RepVGGBlock source executed through the import shim."""
    from oracle import ref_shim
    from oracle.pp_yolo_e import RepVGGBlock as O

    ref_shim.install()
    from super_gradients.modules.repvgg_block import RepVGGBlock as RefBlock

    c = 8
    r = RefBlock(c, c, activation_type=nn.SiLU, use_residual_connection=False, use_alpha=True).train()
    o = O(c, c, nn.SiLU, use_alpha=True).train()
    _randomize(r, 3)
    _randomize(o, 3)  # (the same BatchNorm eps / momentum on both sides; the parameters are then copied from the reference block)
    o.load_state_dict(r.state_dict(), strict=True)
    x = torch.randn(2, c, 6, 6, generator=torch.Generator().manual_seed(1))
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    ya, yb = r(xa), o(xb)
    assert torch.equal(ya, yb)
    ya.sum().backward()
    yb.sum().backward()
    assert torch.equal(xa.grad, xb.grad) and torch.equal(r.alpha.grad, o.alpha.grad)


def test_effective_se_block(backend):
    from oracle.pp_yolo_e import EffectiveSEBlock as O
    from super_gradients_amd.modules.se_blocks import EffectiveSEBlock

    n, c, h, w = _shape(backend, (3, 96, 24, 23), (2, 8, 5, 4))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) * 2
    ref = O(c)
    with torch.no_grad():
        ref.project.weight.mul_(4.0)  # pre-activations on both sides of the hardsigmoid knees
    _check(ref, EffectiveSEBlock(c), x, backend)


@pytest.mark.parametrize("residual", [True, False])
def test_csp_resnet_basic_block(backend, residual):
    from oracle.pp_yolo_e import BasicBlock as O
    from super_gradients_amd.training.models.detection_models.csp_resnet import CSPResNetBasicBlock

    n, c, h, w = _shape(backend, (2, 48, 12, 12), (1, 8, 5, 5))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(O(c, c, nn.SiLU, residual=residual), CSPResNetBasicBlock(c, c, "silu", use_residual_connection=residual), x, backend)


def test_csp_res_stage(backend):
    from oracle.pp_yolo_e import CSPResStage as O
    from super_gradients_amd.training.models.detection_models.csp_resnet import CSPResStage

    n, c, co, h, w, nb = _shape(backend, (2, 64, 128, 16, 16, 2), (2, 8, 8, 6, 6, 1))
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(O(c, co, nb, nn.SiLU), CSPResStage(c, co, nb, stride=2, activation_type="silu"), x, backend)


@pytest.mark.parametrize("nblk,spp", [(1, True), (3, True), (2, False)])
def test_ppyoloe_csp_stage(backend, nblk, spp):
    from oracle.pp_yolo_e import CSPStage as O
    from super_gradients_amd.training.models.detection_models.pp_yolo_e.pan import CSPStage

    n, c, co, h, w = _shape(backend, (2, 96, 64, 10, 10), (1, 8, 8, 5, 5))
    if backend.type == "cpu" and nblk == 3:
        pytest.skip("host emulation: the 1-block and 2-block forms cover the wiring")
    x = torch.randn(n, c, h, w, generator=torch.Generator().manual_seed(0)) + 0.5
    _check(O(c, co, nblk, nn.SiLU, spp), CSPStage(c, co, nblk, "silu", spp), x, backend)


def test_batchnorm_single_value_per_channel_raises(backend):
    """Training-mode BatchNorm over ONE value per channel: the reference raises (F.batch_norm: 'Expected more than 1 value per channel
    when training'); so does the HIP path, instead of writing inf/NaN into the running variance.  Eval mode is fine."""
    from super_gradients_amd.modules import Conv

    blk = Conv(8, 8, 1, 1, "relu")
    net = _wrap(blk, backend)
    x = to_nhwc(torch.randn(1, 8, 1, 1), backend)
    net.train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        blk.fwd(x)
    ref = nn.Sequential(nn.Conv2d(8, 8, 1, bias=False), nn.BatchNorm2d(8)).train()
    with pytest.raises(ValueError, match="Expected more than 1 value per channel"):
        ref(torch.randn(1, 8, 1, 1))
    net.eval()
    assert torch.isfinite(blk.fwd(x).cpu()).all()


@pytest.mark.parametrize("ksize,nout", [(1, 6), (3, 5), (1, 17)])
def test_prediction_conv_any_class_count(backend, ksize, nout):
    """The class-prediction convs (dfl_heads.py:57-66 cls_pred, pp_yolo_head.py:139 pred_cls) with a class count that is NOT a multiple
    of 4 (the reference's own test builds 17 classes, tests/unit_tests/yolo_nas_tests.py:13-18): forward, input / weight / bias gradient."""
    from super_gradients_amd.training.models.detection_models.yolo_nas.dfl_heads import _PredConv

    n, c, h, w = 2, 8, 5, 4
    g = torch.Generator().manual_seed(0)
    x = torch.randn(n, c, h, w, generator=g, requires_grad=True)
    ref = nn.Conv2d(c, nout, ksize, padding=ksize // 2)
    blk = _PredConv(c, nout, ksize, 1, ksize // 2, bias=True)
    net = _wrap(blk, backend)
    blk.load_state_dict(ref.state_dict(), strict=True)
    net.train()
    y = ref(x)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    net.zero_grad()
    out = torch.full((n, h, w, nout + 3), float("nan"), device=backend)[..., :nout]   # a slice of a wider prediction buffer? no: rows of [B,L,C]
    out = torch.empty(n, h, w, nout, device=backend)
    yd = blk.fwd(to_nhwc(x.detach(), backend), out=out)
    assert_close(to_nchw_cpu(yd), y.detach(), 2e-5, "forward")
    dx = blk.bwd(to_nhwc(dy, backend))
    net.join_side()
    assert_close(to_nchw_cpu(dx), x.grad, 1e-4, "input gradient")
    assert_close(blk.weight.grad.cpu(), ref.weight.grad, 1e-4, "weight gradient")
    assert_close(blk.bias.grad.cpu(), ref.bias.grad, 1e-4, "bias gradient")


def test_stem_with_custom_in_channels(backend):
    """tests/unit_tests/yolo_nas_tests.py:13-18 builds YOLO-NAS with in_channels=2: the stem QARepVGG block on a 2-channel image
    (channels are zero-padded to 4 at the NCHW -> NHWC entrance; weights carry the same physical padding)."""
    from oracle.yolo_nas import QARep
    from super_gradients_amd import kernels as K
    from super_gradients_amd.modules import QARepVGGBlock
    from super_gradients_amd.training import models

    net = models.get("yolo_nas_s", arch_params=dict(in_channels=2), num_classes=17)
    sd = net.state_dict()
    assert tuple(sd["backbone.stem.conv.branch_3x3.conv.weight"].shape) == (48, 2, 3, 3) and tuple(sd["heads.head1.cls_pred.weight"].shape)[0] == 17
    x = torch.randn(2, 2, 8, 8, generator=torch.Generator().manual_seed(0)) + 0.5
    ref, blk = QARep(2, 8, 2, residual=False), QARepVGGBlock(2, 8, stride=2, use_residual_connection=False)
    _randomize(ref, 1)
    wrapped = _wrap(blk, backend)
    blk.load_state_dict(ref.state_dict(), strict=True)
    from super_gradients_amd.modules.layers import BatchNorm

    for m in blk.modules():
        if isinstance(m, BatchNorm):
            m.eps, m.momentum = 1e-3, 0.03
    ref.train(), wrapped.train()
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
    y.backward(dy)
    wrapped.zero_grad()
    xh = K.input_to_nhwc(x.to(backend))
    assert tuple(xh.shape) == (2, 8, 8, 4)
    yd = blk.fwd(xh)
    assert_close(to_nchw_cpu(yd), y, 2e-5, "forward")
    blk.bwd(to_nhwc(dy, backend), need_dx=False)
    wrapped.join_side()
    rp = dict(ref.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in ref.parameters() if p.grad is not None)
    for name, p in blk.named_parameters():
        if "rbr_reparam" in name:
            continue
        rg = rp[name].grad   # (gradients that are analytically zero in front of a training-mode BatchNorm are judged against the largest one)
        e = float((p.grad.cpu().double() - rg.double()).abs().max()) / max(float(rg.abs().max()), 1e-2 * gmax)
        assert e <= 1e-4, f"grad {name}: {e:.3e}"


def test_experiment_switches_compose(backend, monkeypatch):
    """All experiment switches at once - 32-deep conv slabs (variant 5 / 6), a tuning-table entry, the fused BatchNorm-backward sweep and the
    one-launch finalize - on a QARepVGG chain whose channel counts make every switch engage (C % 32 == 0): forward, input gradient and every
    parameter gradient stay within rounding of the default path (the conv switches are bit-exact, the finalize regroups fp64 sums)."""
    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib, load_conv_tuning
    from super_gradients_amd.modules import QARepVGGBlock
    from super_gradients_amd.modules.engine import SgxNetwork


    class Chain(SgxNetwork):
        def __init__(self):
            super().__init__()
            self.b1 = QARepVGGBlock(32, 32, stride=1)
            self.b2 = QARepVGGBlock(32, 64, stride=2)

    n, h, w = 2, 6, 5
    x = to_nhwc(torch.randn(n, 32, h, w, generator=torch.Generator().manual_seed(0)) + 0.5, backend)
    dy = to_nhwc(torch.randn(n, 64, 3, 3, generator=torch.Generator().manual_seed(1)), backend)

    def run():
        torch.manual_seed(11)
        net = Chain()
        net.materialize(backend).train()
        y = net.b2.fwd(net.b1.fwd(x))
        dx = net.b1.bwd(net.b2.bwd(dy))
        net.join_side()
        return [y.cpu().clone(), dx.cpu().clone(), net.g_arena.buf.cpu().clone()]

    base = run()
    key = dict(N=n, H=h, W=w, C=32, K=32, R=3, stride=1, pad=1)
    try:
        for variant in (7, 6):
            lib().sgx_debug_set_variant(variant)
            lib().sgx_bn_set_fused_finalize(1)
            load_conv_tuning([dict(kind="fwd", bm=128, bn=32, variant=0, **key), dict(kind="dgrad", bm=64, bn=32, variant=0, **key),
                              dict(kind="wgrad", bm=32, bn=96, variant=2048, **key)])
            got = run()
            for a, b, what in zip(base, got, ("forward", "input gradient", "parameter gradients")):
                assert_close(b, a, 2e-5, f"variant {variant}: {what}")
    finally:
        lib().sgx_debug_set_variant(0)
        lib().sgx_bn_set_fused_finalize(0)
        load_conv_tuning([])


def test_deepcopy_of_a_materialised_network_trains(backend):
    """copy.deepcopy(net) after materialisation - nn.Parameter.__deepcopy__ clones .data and drops .grad, so the copy's parameters
    must be re-attached to the COPIED arenas: parameters / gradients are views of the copy's arenas (not of the original's), one training
    step of the copy produces the gradients the original produces, writes them into the copy's gradient arena only, and load_state_dict
    on the copy reaches the weights its kernels read."""
    import copy

    from super_gradients_amd.modules import QARepVGGBlock

    blk = QARepVGGBlock(16, 16, stride=1, use_residual_connection=True)
    net = _wrap(blk, backend)
    net.train()
    x = torch.randn(2, 16, 6, 5, generator=torch.Generator().manual_seed(0)) + 0.5
    dy = torch.randn(2, 16, 6, 5, generator=torch.Generator().manual_seed(1))

    def step(n):
        n.zero_grad()
        n.prefetch_dgrad_weights()
        y = n.b.fwd(to_nhwc(x, backend))
        n.b.bwd(to_nhwc(dy, backend))
        n.join_side()
        return to_nchw_cpu(y)

    y0 = step(net)
    g0 = net.g_arena.buf.clone()
    cp = copy.deepcopy(net)

    def inside(t, arena):
        return arena.buf.data_ptr() <= t.data_ptr() < arena.buf.data_ptr() + arena.buf.numel() * 4

    assert cp.p_arena.buf.data_ptr() != net.p_arena.buf.data_ptr()
    for name, p in cp.named_parameters():
        if "rbr_reparam" in name:
            continue
        assert inside(p.data, cp.p_arena) and p.grad is not None and inside(p.grad, cp.g_arena), name
    for b in cp.buffers():
        assert not inside(b, net.b_arena) and (b.dtype != torch.float32 or inside(b, cp.b_arena))
    net.zero_grad()
    y1 = step(cp)
    assert torch.equal(y1, y0) and torch.equal(cp.g_arena.buf, g0)
    assert float(net.g_arena.buf.abs().max()) == 0.0, "the copy's step must not touch the original's gradient arena"
    # load_state_dict on the copy reaches the arena its convolutions read
    sd = {k: (v + 0.25 if v.dtype == torch.float32 and "rbr_reparam" not in k else v) for k, v in net.state_dict().items()}
    cp.load_state_dict(sd)
    y2 = step(cp)
    assert not torch.equal(y2, y0)
    net.load_state_dict(sd)
    assert torch.equal(step(net), y2)


def test_filter_planes_serve_a_training_step_and_nothing_else(backend):
    """Pre-split filter planes at network level (engine.py, round 5): inside NetFunction.forward / backward the bf16x3 launches of a QARepVGG
    chain read the planes the step's prefetch made - same bits as a network without planes (forward, every parameter
    gradient) - while an eval-mode forward after an in-place weight change, outside any step, must follow the NEW weights (no stale planes),
    and a second step after the change must too."""
    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import lib
    from super_gradients_amd.modules import QARepVGGBlock
    from super_gradients_amd.modules.engine import SgxNetwork

    class Chain(SgxNetwork):
        def __init__(self):
            super().__init__()
            self.b1 = QARepVGGBlock(32, 32, stride=1)
            self.b2 = QARepVGGBlock(32, 64, stride=2)

        def _fwd(self, x):
            return (self.b2.fwd(self.b1.fwd(K.input_to_nhwc(x))),)

        def _bwd(self, dy):
            self.b1.bwd(self.b2.bwd(dy.contiguous()), need_dx=False)

    gpu = backend.type == "cuda"
    n, h, w = (4, 40, 40) if gpu else (1, 8, 16)
    x = (torch.randn(n, 32, h, w, generator=torch.Generator().manual_seed(0)) + 0.5).to(backend)
    lib().sgx_debug_set_variant(0 if gpu else 9)  # (host emulation: small maps - variant 9 lets the patch kernel take them)

    def step(net):
        net.zero_grad()
        y = net(x)
        dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(backend)
        y.backward(dy)
        net.join_side()
        return [y.detach().cpu().clone(), net.g_arena.buf.cpu().clone()]

    def build(planes):
        torch.manual_seed(11)
        net = Chain()
        net.materialize(backend).train()
        if not planes:  # (not through SGX_FILTER_PLANES: the library reads that variable once, when it is loaded, as its process-wide mode)
            net.drop_filter_planes()
            net._fp_jobs = net._fp_dev = net._fp_buf = None
        return net

    try:
        plain, fast = build(False), build(True)
        assert plain._fp_jobs is None and fast._fp_jobs is not None
        h0 = lib().sgx_debug_filter_planes_hits()
        ref = step(plain)
        assert lib().sgx_debug_filter_planes_hits() == h0
        got = step(fast)
        assert lib().sgx_debug_filter_planes_hits() - h0 >= 3, "the step's launches did not read the planes"  # (b1 forward, b2 forward, b2 data gradient)
        for a, b, what in zip(ref, got, ("forward", "parameter gradients")):
            assert torch.equal(a, b), f"planes step differs in {what}: {float((a - b).abs().max()):.3e}"
        # weights change in place outside a step: the next launches - eval forward, then a whole step - follow the new weights
        for net in (plain, fast):
            with torch.no_grad():
                net.p_arena.buf.mul_(1.25)
            net.weights_changed()
        h1 = lib().sgx_debug_filter_planes_hits()
        plain.eval(), fast.eval()
        with torch.no_grad():
            assert torch.equal(plain(x), fast(x))
        assert lib().sgx_debug_filter_planes_hits() == h1, "a launch outside a training step read filter planes"
        if gpu:  # (a second pair of steps costs the host emulation another minute; the kernel-level test covers re-validation there)
            plain.train(), fast.train()
            for a, b in zip(step(plain), step(fast)):
                assert torch.equal(a, b)
            assert lib().sgx_debug_filter_planes_hits() > h1
    finally:
        lib().sgx_debug_set_variant(0)
        K.filter_planes_scope(False)
        K.filter_planes_invalidate(None)


def test_branch_stream_policy_and_host_behaviour():
    """engine.SgxNetwork.branches / fork_branch (round 6): which call sites fork is a pure function of the mode bits, the site bits and the size
    limit; without HIP streams (host emulation, SGX_SIDE_STREAM=0) nothing forks - fork_branch runs the chain in place and its join is a no-op."""
    from types import SimpleNamespace

    from super_gradients_amd.modules.engine import SgxNetwork, _nothing

    net = SimpleNamespace(branch_stream=object(), branch_mode=3, branch_sites=1 | 4, branch_max_tiles=100)
    assert SgxNetwork.branches(net, 1, 64 * 10, 64 * 10) and SgxNetwork.branches(net, 4, 64 * 10, 64 * 10, backward=True)
    assert not SgxNetwork.branches(net, 2, 64, 64), "site bit off"
    assert not SgxNetwork.branches(net, 1, 64 * 11, 64 * 10), "110 tiles > the limit of 100"
    net.branch_mode = 1
    assert SgxNetwork.branches(net, 1, 64, 64) and not SgxNetwork.branches(net, 1, 64, 64, backward=True), "forward bit only"
    net.branch_stream = None
    assert not SgxNetwork.branches(net, 1, 64, 64)
    ran = []
    out, joined = SgxNetwork.fork_branch(net, lambda: ran.append(1) or "result", backward=True)
    assert out == "result" and ran == [1] and joined is _nothing and joined() is None


def test_data_parallel_run_gives_a_branch_lane_to_the_collectives(monkeypatch):
    """engine.SgxNetwork.data_parallel_streams (r6ah): with collectives on, one branch lane and no d alpha site - unless set explicitly."""
    from types import SimpleNamespace

    from super_gradients_amd.modules.engine import SgxNetwork

    monkeypatch.delenv("SGX_BRANCH_LANES", raising=False)
    monkeypatch.delenv("SGX_BRANCH_SITES", raising=False)
    a, b = object(), object()
    net = SimpleNamespace(branch_stream=a, branch_lanes=[a, b], branch_sites=63)
    SgxNetwork.data_parallel_streams(net)
    assert net.branch_lanes == [a] and net.branch_sites == 31
    monkeypatch.setenv("SGX_BRANCH_LANES", "2")
    monkeypatch.setenv("SGX_BRANCH_SITES", "63")
    net = SimpleNamespace(branch_stream=a, branch_lanes=[a, b], branch_sites=63)
    SgxNetwork.data_parallel_streams(net)
    assert net.branch_lanes == [a, b] and net.branch_sites == 63
    net = SimpleNamespace(branch_stream=None, branch_lanes=[], branch_sites=63)
    SgxNetwork.data_parallel_streams(net)
    assert net.branch_lanes == []
