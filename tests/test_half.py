"""Half-precision inference path (csrc/half.hip; SURVEY 8f-3: `predict(fp16=True)`, reference training/pipelines/pipelines.py:76,223).

Kernel level (both back ends: the product library on the GPU, the host emulation of the same sources here): every bf16 kernel against ATen on
the SAME bf16-rounded operands in float64 - what remains is the fp32 accumulation order and the one rounding of the stored result, so the
bars are an fp32-sum bound for fp32 outputs and one bf16 ulp for bf16 outputs.  Model level: the fused YOLO-NAS copy on the half path
against (a) the CPU oracle under torch.autocast(bfloat16) - the reference's own arithmetic for this leg - and (b) the fp32 product path.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

BF = torch.bfloat16


def _bf(t):
    return t.to(BF)


def _conv_ref(x_nhwc, w, bias, stride, pad, act, post=None, post_scale=1.0):
    """float64 reference on the bf16-rounded operands; x_nhwc / post bf16 NHWC, w fp32 [K,C,R,S] (rounded here like the product's cache)."""
    x = x_nhwc.double().permute(0, 3, 1, 2)
    wq = _bf(w).double()
    if wq.shape[1] < x.shape[1]:
        wq = F.pad(wq, (0, 0, 0, 0, 0, x.shape[1] - wq.shape[1]))
    y = F.conv2d(x, wq, None if bias is None else bias.double(), stride=stride, padding=pad)
    if act == "relu":
        y = F.relu(y)
    elif act == "silu":
        y = F.silu(y)
    y = y.permute(0, 2, 3, 1)
    if post is not None:
        y = y + post_scale * post.double()
    return y


def _check(out, ref, what):
    ref = ref.to(out.device)
    if out.dtype == BF:
        # one rounding to bf16 (2^-9 relative at most) on top of the fp32 accumulation error
        err = (out.double() - ref).abs()
        bound = ref.abs() * 2.0 ** -8 + 1e-3 * ref.abs().max().clamp_min(1e-6) * 2.0 ** -8 + 1e-30
        bad = err > bound
        assert not bool(bad.any()), f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond one bf16 ulp (worst {float((err / bound).max()):.2f}x)"
    else:
        scale = float(ref.abs().max().clamp_min(1e-6))
        err = float((out.double() - ref).abs().max())
        assert err <= 2e-5 * scale, f"{what}: fp32 output differs by {err / scale:.2e} of the output's scale"


CASES = [
    # N, H, W, C, K, R, stride, pad, act, extras
    dict(N=2, H=9, W=11, C=8, K=48, R=3, s=2, p=1, act="relu"),                    # the stem's shape: 8-channel flat K axis, stride 2
    dict(N=1, H=12, W=10, C=8, K=32, R=3, s=1, p=1, act=None),                     # flat, 12 taps slots for 9 taps
    dict(N=2, H=10, W=10, C=32, K=64, R=3, s=1, p=1, act="relu"),
    dict(N=2, H=10, W=10, C=48, K=96, R=3, s=1, p=1, act="relu"),                  # C % 32 != 0: half-empty last chunk; N = 3 x 32
    dict(N=1, H=16, W=16, C=64, K=128, R=1, s=1, p=0, act="silu"),                 # 64-deep slabs, 128-wide tile
    dict(N=1, H=16, W=16, C=128, K=64, R=3, s=2, p=1, act="relu"),
    dict(N=3, H=7, W=5, C=96, K=80, R=1, s=1, p=0, act=None, f32=True),            # prediction conv: fp32 output, 80 classes
    dict(N=2, H=6, W=6, C=32, K=17, R=1, s=1, p=0, act=None, f32=True),            # class count not a multiple of 4: scalar epilogue
    dict(N=2, H=8, W=8, C=64, K=64, R=3, s=1, p=1, act="relu", post=0.75),         # bottleneck shortcut in the epilogue
    dict(N=2, H=8, W=8, C=32, K=32, R=3, s=1, p=1, act="relu", post=1.0, post_dev=True, slice_out=True, slice_in=True),
    dict(N=1, H=40, W=40, C=32, K=32, R=3, s=1, p=1, act="relu"),                  # more than one pixel tile per image row run
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c['N']}x{c['H']}x{c['W']}x{c['C']}-k{c['K']}r{c['R']}s{c['s']}")
def test_hconv_against_float64(backend, case):
    from super_gradients_amd import kernels as K

    g = torch.Generator().manual_seed(3)
    N, H, W, C, Kf, R, s, p = (case[k] for k in ("N", "H", "W", "C", "K", "R", "s", "p"))
    if case.get("slice_in"):  # the input is a channel slice of a wider concat buffer
        wide = _bf(torch.randn(N, H, W, 2 * C, generator=g)).to(backend)
        x = wide[..., C:]
    else:
        x = _bf(torch.randn(N, H, W, C, generator=g)).to(backend)
    cw = C if C > 8 else 4  # the stem's filter has the fp32 path's 4 (3 + 1 padding) channels; the activation is padded to 8
    w = (torch.randn(Kf, cw, R, R, generator=g) / (cw * R * R) ** 0.5)
    w_k = K.to_ohwi(w.to(backend))
    bias = torch.randn(Kf, generator=g).to(backend)
    ho, wo = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    post = post_scale = None
    ps = 1.0
    if "post" in case:
        post = _bf(torch.randn(N, ho, wo, Kf, generator=g)).to(backend)
        ps = case["post"]
        post_scale = torch.tensor([ps], device=backend) if case.get("post_dev") else ps
    out = None
    if case.get("f32"):
        out = torch.empty(N, ho, wo, Kf, device=backend, dtype=torch.float32)
    elif case.get("slice_out"):
        out = torch.full((N, ho, wo, Kf + 16), 7.0, device=backend, dtype=BF)[..., 8:8 + Kf]
    y = K.conv2d_fwd(x, w_k, bias=bias, out=out, act=case["act"], stride=s, pad=p, post_add=post, post_scale=post_scale)
    assert y.dtype == (torch.float32 if case.get("f32") else BF) and tuple(y.shape) == (N, ho, wo, Kf)
    ref = _conv_ref(x.cpu(), w, bias.cpu(), s, p, case["act"], None if post is None else post.cpu(), ps)
    _check(y.cpu(), ref, "hconv")
    if case.get("slice_out"):  # nothing outside the slice was touched
        base = out._base if out._base is not None else out
        full = torch.as_strided(base, (N, ho, wo, Kf + 16), (ho * wo * (Kf + 16), wo * (Kf + 16), Kf + 16, 1))
        assert bool((full[..., :8].float() == 7.0).all()) and bool((full[..., 8 + Kf:].float() == 7.0).all())


def test_hconv_tile_and_depth_overrides_agree(backend):
    """Every (tile, slab depth) instantiation computes the same function (the heuristic's choice is a speed matter only)."""
    from super_gradients_amd import kernels as K
    from super_gradients_amd._lib import check, lib

    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(2, 12, 12, 64, generator=g)).to(backend)
    w = torch.randn(128, 64, 3, 3, generator=g) / 24.0
    w_k = K.to_ohwi(w.to(backend))
    ref = _conv_ref(x.cpu(), w, None, 1, 1, "relu")
    try:
        for bm, bn in ((64, 32), (64, 64), (128, 32), (128, 64), (64, 128), (128, 128)):
            for kd in (32, 64):
                check(lib().sgx_hconv_debug_set_tile(bm, bn, kd), "set_tile")
                _check(K.conv2d_fwd(x, w_k, act="relu", stride=1, pad=1).cpu(), ref, f"hconv tile {bm}x{bn} depth {kd}")
    finally:
        check(lib().sgx_hconv_debug_set_tile(0, 0, 0), "set_tile")


def test_half_helpers(backend):
    """cast (== torch's round-to-nearest-even), copy into a slice, max pooling (exact), transposed convolution 2x2."""
    from super_gradients_amd import kernels as K

    g = torch.Generator().manual_seed(9)
    x32 = torch.randn(2, 5, 7, 4, generator=g)
    x32[0, 0, 0, 0] = 1.00390625  # a tie between two bf16 neighbours: rounds to even
    h = K.cast_bf16(x32.to(backend), cpad=8).cpu()
    assert h.dtype == BF and tuple(h.shape) == (2, 5, 7, 8)
    assert torch.equal(h[..., :4], x32.to(BF)) and bool((h[..., 4:].float() == 0).all())

    src = _bf(torch.randn(2, 4, 4, 16, generator=g)).to(backend)
    cat = torch.zeros(2, 4, 4, 40, device=backend, dtype=BF)
    K.axpy(src, out=cat[..., 24:])
    assert torch.equal(cat[..., 24:].cpu(), src.cpu()) and bool((cat[..., :24].float() == 0).all())

    xm = _bf(torch.randn(2, 9, 10, 16, generator=g)).to(backend)
    for k in (5, 9, 13):
        y, am = K.maxpool_fwd(xm, k, 1, k // 2, want_argmax=False)
        assert am is None
        ref = F.max_pool2d(xm.cpu().float().permute(0, 3, 1, 2), k, 1, k // 2).permute(0, 2, 3, 1)
        assert torch.equal(y.cpu().float(), ref)

    xt = _bf(torch.randn(2, 5, 6, 32, generator=g)).to(backend)
    wt32 = torch.randn(32, 16, 2, 2, generator=g) / 6.0
    wt = K.convT_empty(32, 16, backend)
    wt.copy_(wt32)
    bias = torch.randn(16, generator=g).to(backend)
    out = torch.zeros(2, 10, 12, 24, device=backend, dtype=BF)
    yt = K.convT2x2_fwd(xt, wt, bias, out=out[..., :16])
    ref = F.conv_transpose2d(xt.cpu().double().permute(0, 3, 1, 2), _bf(wt32).double(), bias.cpu().double(), stride=2).permute(0, 2, 3, 1)
    _check(yt.cpu(), ref, "hconvT2x2")
    assert bool((out[..., 16:].float() == 0).all())


def test_half_rejects_what_it_cannot_run(backend):
    from super_gradients_amd import _lib
    from super_gradients_amd import kernels as K

    x = torch.zeros(1, 4, 4, 12, device=backend, dtype=BF)  # 12 channels: not a multiple of 8
    w = K.to_ohwi(torch.zeros(8, 12, 1, 1, device=backend))
    with pytest.raises(_lib.SgxError, match="multiple of 8"):
        K.conv2d_fwd(x, w)
    x = torch.zeros(1, 4, 4, 16, device=backend, dtype=BF)
    w = K.to_ohwi(torch.zeros(8, 16, 1, 1, device=backend))
    with pytest.raises(_lib.SgxError, match="inference form only"):
        K.conv2d_fwd(x, w, stat_partials=True)
    with pytest.raises(_lib.SgxError):
        K.conv2d_fwd(torch.zeros(1, 4, 4, 16, device=backend), w, post_add=torch.zeros(1, 4, 4, 8, device=backend))


def _detector(device):
    from test_predict import _small_detector

    return _small_detector(device)


def test_spp_pool_chain_is_exact(backend):
    """SPP in eval mode runs 5 / 9 / 13 as three chained 5-wide pools: bit-identical to the three direct pools (fp32 and bf16)."""
    from super_gradients_amd import kernels as K

    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 7, 9, 16, generator=g).to(backend)
    for t in (x, _bf(x)):
        p5, _ = K.maxpool_fwd(t, 5, 1, 2, want_argmax=False)
        p9, _ = K.maxpool_fwd(t, 9, 1, 4, want_argmax=False)
        p13, _ = K.maxpool_fwd(t, 13, 1, 6, want_argmax=False)
        c9, _ = K.maxpool_fwd(p5, 5, 1, 2, want_argmax=False)
        c13, _ = K.maxpool_fwd(c9, 5, 1, 2, want_argmax=False)
        assert torch.equal(c9, p9) and torch.equal(c13, p13)


def _iou(a, b):
    x1, y1 = np.maximum(a[:, None, 0], b[None, :, 0]), np.maximum(a[:, None, 1], b[None, :, 1])
    x2, y2 = np.minimum(a[:, None, 2], b[None, :, 2]), np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    area = lambda t: (t[:, 2] - t[:, 0]) * (t[:, 3] - t[:, 1])  # noqa: E731
    return inter / (area(a)[:, None] + area(b)[None, :] - inter + 1e-12)


def test_half_forward_against_autocast_oracle_and_fp32_path(backend):
    """The fused copy on the half path: raw scores / boxes against (a) the oracle network under torch.autocast(cpu, bfloat16) - the
    reference's arithmetic for predict(fp16=True) - and (b) this build's fp32 path, whose distance from the half path bounds what bf16
    costs: scores 2e-2 absolute, boxes 2 % of the image side.  The half path must be no further from the fp32 truth than autocast is
    (x 2: different rounding points - autocast keeps BatchNorm unfolded)."""
    import copy

    from oracle.yolo_nas import YoloNAS as OracleYoloNAS

    net = _detector(backend)
    size = 64
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(4))
    fused32 = copy.deepcopy(net).eval()
    fused32.prep_model_for_conversion(input_size=(size, size), full_fusion=True)
    fused16 = copy.deepcopy(fused32).eval().half_inference(True)
    with torch.no_grad():
        (b32, s32), (l32, d32, *_) = fused32(x.to(backend))
        (b16, s16), (l16, d16, *_) = fused16(x.to(backend))
    assert l16.dtype == torch.float32 and s16.dtype == torch.float32  # the heads' outputs stay fp32
    b32, s32, b16, s16 = (t.cpu() for t in (b32, s32, b16, s16))
    assert float((s16 - s32).abs().max()) < 2e-2, "scores: half path vs fp32 path"
    assert float((b16 - b32).abs().max()) < 0.02 * size, "boxes: half path vs fp32 path"
    assert float((s16 - s32).abs().max()) > 0.0  # (it IS a different arithmetic: the switch is not a no-op)

    if backend.type != "cuda":
        return  # the oracle network exists for the full S / M / L architectures only: the emulation runs the shrunk wiring (part (b) above)
    ref = OracleYoloNAS("s", num_classes=3)
    ref.load_state_dict({k: v.cpu() for k, v in net.state_dict().items()}, strict=True)
    ref.eval()
    with torch.no_grad():
        (br, sr), _ = ref(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            (ba, sa), _ = ref(x)
    sa, ba = sa.float(), ba.float()
    assert float((s16 - sa).abs().max()) < 2e-2, "scores: half path vs the oracle under autocast(bfloat16)"
    e_half, e_auto = float((s16 - sr).abs().max()), float((sa - sr).abs().max())
    assert e_half <= 2.0 * e_auto + 1e-3, f"half path is {e_half:.2e} from the fp32 oracle, autocast {e_auto:.2e}"


def test_predict_fp16_runs_the_half_path(backend):
    """predict(fp16=True) (the reference's default) takes the fused copy onto the bf16 kernels; fp16=False keeps fp32; detections agree:
    every confident fp32 detection has a half-path partner of the same class with IoU >= 0.9."""
    net = _detector(backend)
    proc = [{"DetectionCenterPadding": {"output_shape": (64, 64), "pad_value": 114}}, {"StandardizeImage": {"max_value": 255.0}},
            {"ImagePermute": {"permutation": (2, 0, 1)}}]
    net.set_dataset_processing_params(class_names=["a", "b", "c"], image_processor=proc, iou=0.6, conf=0.0)
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, (64, 50, 3), dtype=np.uint8), rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)]
    r16 = net.predict(images, max_predictions=20, nms_top_k=100)
    p16 = net._get_pipeline(max_predictions=20, nms_top_k=100)
    assert p16.half and p16.model._half_inference and not net._half_inference
    r32 = net.predict(images, max_predictions=20, nms_top_k=100, fp16=False)
    p32 = net._get_pipeline(max_predictions=20, nms_top_k=100, fp16=False)
    assert not p32.half and not p32.model._half_inference
    for a, b in zip(r32, r16):
        pa, pb = a.prediction, b.prediction
        assert len(pa) > 0 and len(pb) > 0
        top = np.argsort(-pa.confidence)[:5]
        iou = _iou(pa.bboxes_xyxy[top], pb.bboxes_xyxy)
        same = pa.labels[top][:, None] == pb.labels[None, :]
        assert bool(((iou * same).max(axis=1) >= 0.9).all()), "a confident fp32 detection has no half-path partner"


def test_half_needs_the_folded_form(backend):
    net = _detector(backend).eval()
    net.half_inference(True)
    with pytest.raises(RuntimeError, match="deployment form"):
        with torch.no_grad():
            net(torch.rand(1, 3, 64, 64).to(backend))
    net.half_inference(False)
    from super_gradients_amd.training import models

    with pytest.raises(NotImplementedError, match="half-precision"):
        models.get("resnet18", num_classes=10).half_inference(True)


def test_half_filters_follow_raw_pointer_weight_updates(backend):
    """the bf16 copies of LIVE filters (prediction convs, transposed convs) were validated by tensor identity and
    `_version` only - the optimizer / EMA kernels write the parameter arena through raw pointers and bump neither, so a model that ran
    half-precision inference and was then trained further served the old bf16 filters for those layers.  Here: half forward, a raw-pointer
    update of the whole parameter arena (the EMA kernel, as the optimizers write), half forward again - the second result must equal a fresh
    copy's (nothing cached) and differ from the first."""
    import copy

    from super_gradients_amd import kernels as K

    net = _detector(backend).eval()
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(9)).to(backend)

    def half_forward(m):
        m.eval()
        m.prep_model_for_conversion(input_size=(64, 64), full_fusion=True)
        m.half_inference(True)
        with torch.no_grad():
            (b, s), _ = m(x)
        m.half_inference(False)
        return b.clone(), s.clone()

    b1, s1 = half_forward(net)
    arena = net.p_arena.buf
    K.ema_update(arena, torch.randn(arena.numel(), generator=torch.Generator().manual_seed(10)).to(backend) * 0.05 + arena, 0.5)
    net.train()
    net.eval()
    b2, s2 = half_forward(net)
    fresh = copy.deepcopy(net)
    b3, s3 = half_forward(fresh)
    assert torch.equal(s2, s3) and torch.equal(b2, b3), "stale bf16 filters served after a raw-pointer weight update"
    assert not torch.equal(s1, s2)


@pytest.mark.gpu
def test_half_yolo_nas_s_80_classes_640_against_autocast_oracle(gpu_device):
    """Review item 7 (round 5 checked the half path on a 3-class 64 x 64 detector only): the REAL YOLO-NAS-S - 80 classes, 640 x 640 - fused
    onto the bf16 kernels against the oracle network under torch.autocast(cpu, bfloat16) (the reference's arithmetic for
    predict(fp16=True), pipelines.py:76,222-247) and against the fp32 oracle.  Bars: raw scores 2e-2 absolute everywhere; for the
    autocast oracle's top-k (anchor, class) pairs of each image the half path's box at that anchor has IoU >= 0.9 with the oracle's and its
    score is within 2e-2; the half path is no further from the fp32 oracle than 2 x autocast is."""
    import copy

    from oracle.yolo_nas import YoloNAS as OracleYoloNAS
    from super_gradients_amd.training import models

    torch.manual_seed(5)
    ref = OracleYoloNAS("s", num_classes=80)
    g = torch.Generator().manual_seed(11)
    for m in ref.modules():  # non-trivial running statistics and a spread of class logits (random-init heads sit at the 0.01 prior)
        if hasattr(m, "running_var"):
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.8, 1.2, generator=g)
    sd = ref.state_dict()
    for k, v in sd.items():
        if "cls_pred" in k and k.endswith("bias"):
            v.add_(torch.randn(v.shape, generator=g) * 1.5 + 2.0)
    ref.load_state_dict(sd)
    ref.eval()
    net = models.get("yolo_nas_s", num_classes=80)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.materialize(gpu_device).eval()
    size = 640
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(6))
    fused = copy.deepcopy(net).eval()
    fused.prep_model_for_conversion(input_size=(size, size), full_fusion=True)
    fused.half_inference(True)
    with torch.no_grad():
        (b16, s16), _ = fused(x.to(gpu_device))
        (br, sr), _ = ref(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            (ba, sa), _ = ref(x)
    b16, s16, sa, ba = b16.cpu(), s16.cpu(), sa.float(), ba.float()
    assert s16.shape == (2, 8400, 80)
    assert float((s16 - sa).abs().max()) < 2e-2, f"scores: half path vs autocast oracle {float((s16 - sa).abs().max()):.3e}"
    e_half, e_auto = float((s16 - sr).abs().max()), float((sa - sr).abs().max())
    assert e_half <= 2.0 * e_auto + 1e-3, f"half path is {e_half:.2e} from the fp32 oracle, autocast {e_auto:.2e}"
    assert float(sa.max()) - float(sa.min()) > 0.2  # (the scores are a spread, not the constant prior)
    for i in range(2):
        top = torch.topk(sa[i].flatten(), 50).indices
        anchors, classes = top // 80, top % 80
        assert float((s16[i, anchors, classes] - sa[i, anchors, classes]).abs().max()) < 2e-2
        iou = np.diag(_iou(ba[i, anchors].numpy(), b16[i, anchors].numpy()))
        assert float(iou.min()) >= 0.9, f"image {i}: top-50 boxes of the autocast oracle vs the half path: min IoU {float(iou.min()):.3f}"
        # and the half path ranks the same pairs at the top: at least 40 of its own top 50 are among the oracle's top 100
        mine = set(torch.topk(s16[i].flatten(), 50).indices.tolist())
        assert len(mine & set(torch.topk(sa[i].flatten(), 100).indices.tolist())) >= 40
    print(f"YOLO-NAS-S 80 classes @640: half-vs-autocast scores {float((s16 - sa).abs().max()):.2e}, half-vs-fp32 {e_half:.2e}, autocast-vs-fp32 {e_auto:.2e}")


# ---- PP-YOLOE on the half-precision path (round 6) ------------------------------------------------------------------------------------
def test_half_squeeze_excitation_and_upsample_kernels(backend):
    """The three bf16 ops PP-YOLOE's deployment form needs beside its convolutions (csrc/half.hip: sgx_himage_colsum, sgx_hchannel_gate,
    sgx_hupsample2x_fwd) against float64 on the bf16-rounded operands: the per-image channel means are fp32 sums (2e-5 relative), the gate
    rounds once to bf16 (1 ulp), the up-sampling is a copy (exact) - also into a channel slice of a wider tensor."""
    from super_gradients_amd import kernels as K

    g = torch.Generator().manual_seed(21)
    x32 = torch.randn(2, 9, 7, 24, generator=g)
    x = _bf(x32).to(backend)
    xr = x.float().cpu().double()
    mean = K.image_colsum(x, scale=1.0 / 63)
    assert mean.dtype == torch.float32 and tuple(mean.shape) == (2, 24)
    ref = xr.sum(dim=(1, 2)) / 63
    assert float((mean.cpu().double() - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1.0)
    pre = torch.randn(2, 24, generator=g).to(backend) * 3
    for gate, fn in (("hardsigmoid", lambda p: torch.clamp(p / 6 + 0.5, 0, 1)), ("sigmoid", torch.sigmoid)):
        wide = torch.zeros(2, 9, 7, 40, device=backend, dtype=BF)
        y = K.channel_gate(x, pre, gate, out=wide[..., 8:32])
        _check(y.cpu(), xr * fn(pre.cpu().double()).view(2, 1, 1, 24), f"hchannel_gate[{gate}]")
        assert bool((wide[..., :8].float() == 0).all()) and bool((wide[..., 32:].float() == 0).all())
    up = torch.zeros(2, 18, 14, 32, device=backend, dtype=BF)
    K.upsample2x_fwd(x, out=up[..., :24])
    exp = x.cpu().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)
    assert torch.equal(up[..., :24].cpu(), exp) and bool((up[..., 24:].float() == 0).all())
    with pytest.raises(Exception):
        K.image_colsum(x, v=x)


def _small_ppyoloe(device, num_classes=3):
    from super_gradients_amd.training import models

    arch = None if device.type == "cuda" else {"depth_mult": 0.2}  # (host emulation: the S widths - the bf16 path wants channel counts in multiples of 8 - at the least depth)
    net = models.get("ppyoloe_s", num_classes=num_classes, arch_params=arch)
    g = torch.Generator().manual_seed(12)
    for m in net.modules():
        if hasattr(m, "running_var"):
            m.running_mean.normal_(0, 0.1, generator=g)
            m.running_var.uniform_(0.8, 1.2, generator=g)
    for n_, p in net.named_parameters():  # (the prediction convs start at zero weights: give the scores something to vary with)
        if "pred_" in n_ and n_.endswith("weight"):
            p.data.normal_(0, 0.05, generator=g)
        elif "pred_cls" in n_ and n_.endswith("bias"):  # a spread of class priors: confident, distinct detections instead of a flat 0.01
            p.data.add_(torch.randn(p.shape, generator=g) * 1.5 + 2.0)
    net.materialize(device)
    return net


def test_half_ppyoloe_forward_against_fp32_path(backend):
    """PP-YOLOE's fused copy on the half path (round 6: `PPYoloE.supports_half_inference()`): RepVGG blocks as one bf16 3x3 convolution with
    the block's `x + y` in its epilogue, squeeze-excitation means in fp32 / gates in bf16, nearest up-sampling and concats in bf16, the
    RGB stem's channel-padded filter folded for this path - raw scores / boxes against this build's fp32 path: scores 2e-2
    absolute, boxes 2 % of the image side; the switch is not a no-op."""
    import copy

    net = _small_ppyoloe(backend)
    size = 64
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(4))
    fused32 = copy.deepcopy(net).eval()
    fused32.prep_model_for_conversion(input_size=(size, size), full_fusion=True)
    fused16 = copy.deepcopy(fused32).eval().half_inference(True)
    with torch.no_grad():
        (b32, s32), (l32, d32, *_) = fused32(x.to(backend))
        (b16, s16), (l16, d16, *_) = fused16(x.to(backend))
    assert l16.dtype == torch.float32 and s16.dtype == torch.float32
    b32, s32, b16, s16 = (t.cpu() for t in (b32, s32, b16, s16))
    assert float((s16 - s32).abs().max()) < 2e-2, f"scores: half path vs fp32 path {float((s16 - s32).abs().max()):.3e}"
    assert float((b16 - b32).abs().max()) < 0.02 * size, "boxes: half path vs fp32 path"
    assert float((s16 - s32).abs().max()) > 0.0


def test_predict_fp16_runs_the_half_path_for_ppyoloe(backend):
    """predict(fp16=True) on PP-YOLOE takes the bf16 kernels too (rounds 2 - 5: a warning and the fp32 path)."""
    net = _small_ppyoloe(backend)
    proc = [{"DetectionRescale": {"output_shape": (64, 64)}}, {"StandardizeImage": {"max_value": 255.0}}, {"ImagePermute": {"permutation": (2, 0, 1)}}]
    net.set_dataset_processing_params(class_names=["a", "b", "c"], image_processor=proc, iou=0.6, conf=0.0)
    rng = np.random.default_rng(3)
    images = [rng.integers(0, 256, (64, 50, 3), dtype=np.uint8), rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)]
    r16 = net.predict(images, max_predictions=20, nms_top_k=100)
    p16 = net._get_pipeline(max_predictions=20, nms_top_k=100)
    assert p16.half and p16.model._half_inference and not net._half_inference
    r32 = net.predict(images, max_predictions=20, nms_top_k=100, fp16=False)
    for a, b in zip(r32, r16):
        pa, pb = a.prediction, b.prediction
        assert len(pa) > 0 and len(pb) > 0
        top = np.argsort(-pa.confidence)[:5]
        iou = _iou(pa.bboxes_xyxy[top], pb.bboxes_xyxy)
        same = pa.labels[top][:, None] == pb.labels[None, :]
        assert bool(((iou * same).max(axis=1) >= 0.85).all()), "a confident fp32 detection has no half-path partner"


@pytest.mark.gpu
def test_half_ppyoloe_s_80_classes_640_against_autocast_oracle(gpu_device):
    """The real PP-YOLOE-S (80 classes, 640 x 640) fused onto the bf16 kernels against the oracle network under torch.autocast(cpu, bfloat16)
    (the reference's arithmetic for predict(fp16=True)) and the fp32 oracle: scores 2e-2 absolute everywhere; the autocast oracle's top-50
    (anchor, class) pairs per image: half-path box IoU >= 0.9 and scores 2e-2; the half path no further from the fp32 oracle than 2 x autocast."""
    import copy

    from oracle import golden_util as G
    from oracle.pp_yolo_e import PPYoloE as Oracle
    from super_gradients_amd.training import models

    torch.manual_seed(5)
    ref = Oracle("s", num_classes=80)
    G.deterministic_fill(ref, seed=6)
    ref.eval()
    net = models.get("ppyoloe_s", num_classes=80)
    net.load_state_dict(ref.state_dict(), strict=True)
    net.materialize(gpu_device).eval()
    size = 640
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(6))
    fused = copy.deepcopy(net).eval()
    fused.prep_model_for_conversion(input_size=(size, size), full_fusion=True)
    fused.half_inference(True)
    with torch.no_grad():
        (b16, s16), _ = fused(x.to(gpu_device))
        (br, sr), _ = ref(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            (ba, sa), _ = ref(x)
    b16, s16, sa, ba = b16.cpu(), s16.cpu(), sa.float(), ba.float()
    assert s16.shape == (2, 8400, 80)
    assert float((s16 - sa).abs().max()) < 2e-2, f"scores: half path vs autocast oracle {float((s16 - sa).abs().max()):.3e}"
    e_half, e_auto = float((s16 - sr).abs().max()), float((sa - sr).abs().max())
    assert e_half <= 2.0 * e_auto + 1e-3, f"half path is {e_half:.2e} from the fp32 oracle, autocast {e_auto:.2e}"
    for i in range(2):
        top = torch.topk(sa[i].flatten(), 50).indices
        anchors, classes = top // 80, top % 80
        assert float((s16[i, anchors, classes] - sa[i, anchors, classes]).abs().max()) < 2e-2
        iou = np.diag(_iou(ba[i, anchors].numpy(), b16[i, anchors].numpy()))
        assert float(iou.min()) >= 0.9, f"image {i}: min IoU {float(iou.min()):.3f}"
    print(f"PP-YOLOE-S 80 classes @640: half-vs-autocast scores {float((s16 - sa).abs().max()):.2e}, half-vs-fp32 {e_half:.2e}, autocast-vs-fp32 {e_auto:.2e}")
