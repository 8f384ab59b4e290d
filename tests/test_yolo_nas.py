"""YOLO-NAS train-step parity: the HIP model (super_gradients_amd) against the CPU oracle (oracle/yolo_nas.py - pinned to the
reference's own modules by tests/test_oracle_vs_reference.py) on identical weights and inputs.
Checks: state_dict key/shape identity, forward outputs (decoded + raw), PPYoloELoss value, EVERY parameter gradient,
BatchNorm running statistics.  Tolerance: the north star's 1e-4 relative (fp32 both sides, different summation order).
"""
import pytest
import torch

from util import assert_close, rel_err, synthetic_targets


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


def _train_step_parity(variant, B, size, device, tol, static=False):
    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd.training.losses import PPYoloELoss

    C = 80
    ref, net = _build_pair(variant, C, device)
    ref.train()
    net.train()
    g = torch.Generator().manual_seed(7)
    x = torch.rand(B, 3, size, size, generator=g)
    targets = synthetic_targets(B, seed=11, kmax=6, size=size, num_classes=C)

    out_ref = ref(x)
    loss_ref, items_ref = PPYoloELossOracle(C, use_static_assigner=static)(out_ref, targets)
    loss_ref.backward()

    out = net(x.to(device))
    crit = PPYoloELoss(num_classes=C, use_static_assigner=static)
    loss, items = crit(out, targets.to(device))
    loss.backward()

    (bx, sc), (lg, ds, an, pt, cnt, st) = out
    (bx_r, sc_r), (lg_r, ds_r, an_r, pt_r, cnt_r, st_r) = out_ref
    assert list(cnt) == list(cnt_r)
    assert torch.equal(an.cpu(), an_r) and torch.equal(pt.cpu(), pt_r) and torch.equal(st.cpu(), st_r)
    assert_close(lg.cpu(), lg_r, tol, "cls_logits")
    assert_close(ds.cpu(), ds_r, tol, "reg_distri")
    assert_close(bx.cpu(), bx_r, tol, "pred_bboxes")
    assert_close(sc.cpu(), sc_r, tol, "pred_scores")
    assert_close(items.cpu(), items_ref, tol, "loss items")
    ref_params = dict(ref.named_parameters())
    worst = ("", 0.0)
    for name, p in net.named_parameters():
        if ".rbr_reparam." in name:
            assert ref_params[name].grad is None
            continue
        e = rel_err(p.grad, ref_params[name].grad)
        if e > worst[1]:
            worst = (name, e)
    assert worst[1] <= 10 * tol, f"parameter gradient {worst[0]}: rel err {worst[1]:.3e}"
    ref_bufs = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(ref_bufs[name])
        else:
            assert_close(b.cpu(), ref_bufs[name], tol, name)
    return float(loss), float(loss_ref)


@pytest.mark.gpu
def test_yolo_nas_s_train_step_parity(gpu_device):
    l, lr = _train_step_parity("s", 2, 320, gpu_device, 1e-4)
    assert abs(l - lr) <= 1e-4 * abs(lr)


@pytest.mark.gpu
def test_yolo_nas_s_train_step_parity_atss(gpu_device):
    _train_step_parity("s", 2, 256, gpu_device, 1e-4, static=True)


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
