"""PP-YOLOE train-step parity: the HIP model against the CPU oracle (oracle/pp_yolo_e.py - pinned bit-exactly to the
reference's own modules by tests/test_oracle_vs_reference.py and tests/golden/ppyoloe_s.pt) on identical weights and inputs.
Checks: state_dict key/shape identity, training forward (raw 6-tuple), PPYoloELoss value, EVERY parameter gradient against the fp64
truth, BatchNorm running statistics, eval forward (decoded + raw), NMS, and the fused deployment form.
Tolerance: the north star's 1e-4 relative (fp32 both sides, different summation order).
"""
import copy

import pytest
import torch

from util import assert_close, synthetic_targets


def _build_pair(variant, num_classes, device, seed=0):
    from oracle import golden_util as G
    from oracle.pp_yolo_e import PPYoloE as Oracle
    from super_gradients_amd.training import models

    torch.manual_seed(seed)
    ref = Oracle(variant, num_classes=num_classes)
    G.deterministic_fill(ref, seed=seed + 1)   # non-trivial BN affine / running statistics, non-zero prediction convs
    net = models.get(f"ppyoloe_{variant}", num_classes=num_classes)
    res = net.load_state_dict(ref.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    net.materialize(device)
    return ref, net


def test_state_dict_matches_oracle():
    from oracle.pp_yolo_e import PPYoloE as Oracle
    from super_gradients_amd.training import models

    for v in ("s", "m", "l", "x"):
        a = Oracle(v, num_classes=80).state_dict()
        b = models.get(f"ppyoloe_{v}", num_classes=80).state_dict()
        assert list(a.keys()) == list(b.keys()), v
        for k in a:
            assert tuple(a[k].shape) == tuple(b[k].shape), k


def test_initialisation_matches_reference_rules():
    """pp_yolo_head.py:150-165: zero prediction weights, class bias -log(99), regression bias 1; ESEAttn fc ~ N(0, 0.001)."""
    import math

    from super_gradients_amd.training import models

    net = models.get("ppyoloe_s", num_classes=7)
    for i in range(3):
        assert float(net.head.pred_cls[i].weight.abs().max()) == 0.0 and float(net.head.pred_reg[i].weight.abs().max()) == 0.0
        assert torch.allclose(net.head.pred_cls[i].bias, torch.full((7,), -math.log(99.0)))
        assert torch.allclose(net.head.pred_reg[i].bias, torch.ones(68))
        assert float(net.head.stem_cls[i].fc.weight.std()) < 2e-3
    assert net.num_classes == 7 and net.get_input_shape_steps() == (32, 32)


def _err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def _train_step_parity(variant, B, size, device, tol, static=False):
    """Three-way: HIP fp32 vs oracle CPU fp32 (the reference's arithmetic) vs the same oracle in fp64 (truth); see
    tests/test_yolo_nas.py::_train_step_parity for the rationale of each bar."""
    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd.training.losses import PPYoloELoss

    C = 80
    ref, net = _build_pair(variant, C, device)
    ref64 = copy.deepcopy(ref).double()
    ref.train(), ref64.train(), net.train()
    x = torch.rand(B, 3, size, size, generator=torch.Generator().manual_seed(7))
    targets = synthetic_targets(B, seed=11, kmax=6, size=size, num_classes=C)

    def bar(name, hip, cpu32, truth, slack=2.0):
        e_pair, e_hip, e_cpu = _err(hip, cpu32), _err(hip, truth), _err(cpu32, truth)
        assert e_pair <= tol or e_hip <= max(tol, slack * e_cpu), f"{name}: hip-cpu32 {e_pair:.2e}, hip-fp64 {e_hip:.2e}, cpu32-fp64 {e_cpu:.2e}"

    out_ref = ref(x)
    out_ref[0].retain_grad(), out_ref[1].retain_grad()
    loss_ref, items_ref = PPYoloELossOracle(C, use_static_assigner=static)(out_ref, targets)
    out64 = ref64(x.double())
    out = net(x.to(device))
    assert isinstance(out, tuple) and len(out) == 6, "training forward returns the reference's raw 6-tuple (pp_yolo_head.py:199-216)"
    loss, items = PPYoloELoss(num_classes=C, use_static_assigner=static)(out, targets.to(device))
    lg, ds, an, pt, cnt, st = out
    lg_r, ds_r, an_r, pt_r, cnt_r, st_r = out_ref
    assert list(cnt) == list(cnt_r)
    assert torch.equal(an.cpu(), an_r) and torch.equal(pt.cpu(), pt_r) and torch.equal(st.cpu(), st_r)
    bar("cls_logits", lg, lg_r, out64[0])
    bar("reg_distri", ds, ds_r, out64[1])
    assert_close(items.cpu(), items_ref, 2 * tol, "loss items")
    ref_bufs = dict(ref.named_buffers())
    for name, b in net.named_buffers():
        if name.endswith("num_batches_tracked"):
            assert int(b) == int(ref_bufs[name])
        else:
            assert_close(b.cpu(), ref_bufs[name], tol, name)
    ref_params, ref64_params, net_params = dict(ref.named_parameters()), dict(ref64.named_parameters()), dict(net.named_parameters())
    assert set(net_params) == set(ref_params)

    def l2(up_l, up_d, lg_, ds_):
        for m in (ref, ref64, net):
            m.zero_grad()
        torch.autograd.backward([out_ref[0], out_ref[1]], [up_l, up_d], retain_graph=True)
        torch.autograd.backward([out64[0], out64[1]], [up_l.double(), up_d.double()], retain_graph=True)
        torch.autograd.backward([lg_, ds_], [up_l.to(device), up_d.to(device)])
        nmax = max(float(p.grad.norm()) for p in ref64_params.values())
        acc_h = acc_c = acc_d = 0.0
        worst = (0.0, "")
        for n, p in net_params.items():
            t = ref64_params[n].grad
            sc = max(float(t.norm()), 1e-3 * nmax)
            e_hip = float((p.grad.cpu().double() - t).norm()) / sc
            e_cpu = float((ref_params[n].grad.double() - t).norm()) / sc
            worst = max(worst, (e_hip, n, e_cpu))
            acc_h += float((p.grad.cpu().double() - t).pow(2).sum())
            acc_c += float((ref_params[n].grad.double() - t).pow(2).sum())
            acc_d += float(t.pow(2).sum())
        return (acc_h / acc_d) ** 0.5, (acc_c / acc_d) ** 0.5, worst

    # backward B: the loss's own gradient (ill-conditioned through the training-mode BatchNorms: global L2 bar against the truth)
    loss_ref.backward(retain_graph=True)
    up = (out_ref[0].grad.clone(), out_ref[1].grad.clone())
    l2_h, l2_c, _ = l2(up[0], up[1], lg, ds)
    assert l2_h <= max(10 * tol, 3.0 * l2_c), f"loss-gradient L2 error vs fp64: hip {l2_h:.2e}, cpu fp32 {l2_c:.2e}"
    # backward A: a seeded zero-mean random upstream gradient (well conditioned): EVERY parameter gradient
    out = net(x.to(device))   # the HIP blocks free their saved tensors in backward: run the forward again (same batch)
    gg = torch.Generator().manual_seed(21)
    a_h, a_c, worst = l2(torch.randn(lg_r.shape, generator=gg), torch.randn(ds_r.shape, generator=gg), out[0], out[1])
    assert worst[0] <= max(10 * tol, 3.0 * worst[2]), f"grad {worst[1]}: L2 error vs fp64 hip {worst[0]:.2e}, cpu fp32 {worst[2]:.2e}"
    assert a_h <= max(10 * tol, 3.0 * a_c), f"random-upstream gradient L2 error vs fp64: hip {a_h:.2e}, cpu fp32 {a_c:.2e}"
    print(f"[ppyoloe_{variant}] loss-gradient L2 err vs fp64: hip {l2_h:.2e} cpu32 {l2_c:.2e}; random upstream: hip {a_h:.2e} cpu32 {a_c:.2e}; "
          f"worst parameter {worst[0]:.2e} ({worst[1]}, cpu32 {worst[2]:.2e})")
    return float(loss.detach()), float(loss_ref.detach())


# Whole-model tests run on the GPU only: under the host emulation (one OS thread per HIP thread) one PP-YOLOE-S train step takes
# ~30 minutes (it passes); the CPU suite covers the same kernels and every block's forward/backward in test_kernels / test_blocks.
@pytest.mark.gpu
def test_ppyoloe_s_train_step_parity(gpu_device):
    l, lr = _train_step_parity("s", 2, 320, gpu_device, 1e-4)
    assert abs(l - lr) <= 2e-4 * abs(lr)


@pytest.mark.gpu
def test_ppyoloe_s_train_step_parity_atss(gpu_device):
    _train_step_parity("s", 2, 256, gpu_device, 1e-4, static=True)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["m", "l"])
def test_ppyoloe_ml_train_step_parity(gpu_device, variant):
    _train_step_parity(variant, 1, 256, gpu_device, 1e-4)


@pytest.mark.gpu
def test_ppyoloe_eval_nms_and_deployment_form(gpu_device):
    """eval forward (running statistics, decoded boxes) + PPYoloEPostPredictionCallback + prep_model_for_conversion
    (pp_yolo_e.py:358-377: every RepVGGBlock -> one 3x3 conv + bias): outputs unchanged within fp32 round-off."""
    from oracle import nms as onms
    from super_gradients_amd.modules.repvgg_block import RepVGGBlock

    size, backend = 320, gpu_device
    ref, net = _build_pair("s", 80, backend)
    ref.eval()
    net.eval()
    x = torch.rand(2, 3, size, size, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        (bx_r, sc_r), (lg_r, ds_r, *_r) = ref(x)
        (bx, sc), raw = net(x.to(backend))
    assert len(raw) == 6
    for a, r, name in ((raw[0], lg_r, "logits"), (raw[1], ds_r, "distri"), (bx, bx_r, "boxes"), (sc, sc_r, "scores")):
        assert_close(a.cpu(), r, 1e-4, f"eval {name}")
    cb = net.get_post_prediction_callback(conf=0.01, iou=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=True)
    res = cb(((bx, sc), raw))
    ref_res = onms.post_prediction(bx.cpu(), sc.cpu(), score_threshold=0.01, nms_threshold=0.7, nms_top_k=1000, max_predictions=300,
                                   multi_label_per_box=True, class_agnostic_nms=True)
    for a, b in zip(res, ref_res):
        assert torch.equal(a.cpu(), b)
    with torch.no_grad():
        net.prep_model_for_conversion(input_size=(size, size))
        (bx1, sc1), raw1 = net(x.to(backend))
    blocks = [m for m in net.modules() if isinstance(m, RepVGGBlock)]
    assert blocks and all(not m.build_residual_branches and hasattr(m, "rbr_reparam") for m in blocks)
    for a, b, r, name in ((raw1[0], raw[0], lg_r, "logits"), (raw1[1], raw[1], ds_r, "distri"), (bx1, bx, bx_r, "boxes"), (sc1, sc, sc_r, "scores")):
        assert_close(a.cpu(), b.cpu(), 1e-4, f"deployment form vs branch form: {name}")
        assert_close(a.cpu(), r, 2e-4, f"deployment form vs oracle: {name}")
    with pytest.raises(RuntimeError):
        net.train()
        net(x.to(backend))


@pytest.mark.gpu
def test_trainer_ppyoloe_recipe_shape(gpu_device, tmp_path):
    """The PP-YOLOE recipe's optimisation settings in miniature through Trainer.train() (AdamW, zero weight decay on bias/BN, EMA,
    PPYoloELoss with the static ATSS assigner for the first epochs as coco2017_ppyoloe_train_params does, DetectionMetrics on the
    validation pass), with the DEVICE-SIDE input pipeline (uint8 HWC images -> DeviceDetectionCollateFN): the loss trajectory against the
    same loop on the oracle fed by the reference's host-side standardisation + collate."""
    import numpy as np

    from oracle.ppyolo_loss import PPYoloELossOracle
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.metrics import DetectionMetrics_050
    from super_gradients_amd.training.utils.collate_fn import DetectionCollateFN, DeviceDetectionCollateFN

    ref, net = _build_pair("s", 80, gpu_device)
    n, bs, size = 3, 4, 160
    dev_collate, host_collate = DeviceDetectionCollateFN(device=gpu_device, max_value=255.0), DetectionCollateFN()
    dev_loader, host_loader = [], []
    for i in range(n):
        g = torch.Generator().manual_seed(20 + i)
        t = synthetic_targets(bs, seed=30 + i, kmax=4, size=size)
        items = [(torch.randint(0, 256, (size, size, 3), dtype=torch.uint8, generator=g).numpy(), t[t[:, 0] == b][:, 1:].numpy()) for b in range(bs)]
        dev_loader.append(dev_collate(items))
        host_loader.append(host_collate([((img / 255.0).astype(np.float32), lab) for img, lab in items]))
    cb = net.get_post_prediction_callback(conf=0.01, iou=0.7, nms_top_k=1000, max_predictions=300, multi_label_per_box=True, class_agnostic_nms=False)
    tp = dict(max_epochs=1, lr_mode="StepLRScheduler", lr_updates=[5], lr_decay_factor=0.1, initial_lr=2e-4,
              loss=PPYoloELoss(num_classes=80, use_static_assigner=True), optimizer="AdamW",
              optimizer_params=dict(weight_decay=1e-5), zero_weight_decay_on_bias_and_bn=True, lr_warmup_epochs=0, ema=True,
              ema_params=dict(decay=0.9997, decay_type="threshold"), silent_mode=True, save_model=False,
              valid_metrics_list=[DetectionMetrics_050(num_cls=80, post_prediction_callback=cb, normalize_targets=True)], metric_to_watch="mAP@0.50")
    res = Trainer("ppyoloe_mini", ckpt_root_dir=str(tmp_path)).train(net, tp, dev_loader, valid_loader=dev_loader[:2])
    assert 0.0 <= res[0]["valid"]["mAP@0.50"] <= 1.0 and "PPYoloELoss/loss" in res[0]["valid"]
    decay = [p for k, p in ref.named_parameters() if p.dim() > 1]
    no_decay = [p for k, p in ref.named_parameters() if p.dim() <= 1]
    o = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0}, {"params": decay}], lr=2e-4, weight_decay=1e-5)
    crit = PPYoloELossOracle(80, use_static_assigner=True)
    ref.train()
    tot = torch.zeros(4)
    for x, t in host_loader:
        loss, items = crit(ref(x), t)
        loss.backward()
        o.step()
        o.zero_grad()
        tot += items * bs
    tot /= n * bs
    got = res[0]["train"]
    for i, name in enumerate(["loss_cls", "loss_iou", "loss_dfl", "loss"]):
        assert abs(got["PPYoloELoss/" + name] - float(tot[i])) <= 1e-3 * abs(float(tot[i])), (name, got, float(tot[i]))


def test_ppyoloe_gradient_buckets_tile_the_arena(backend):
    """Data-parallel exchange of PP-YOLOE (bench.py --workload ppyoloe --gpus N): the bucket prefixes the backward announces
    (head -> neck -> backbone stages -> stem) must tile the gradient arena exactly; GradientAllReducer refuses anything else."""
    from super_gradients_amd.training import models
    from super_gradients_amd.training.utils.distributed_training_utils import GradientAllReducer

    for v in ("s", "m"):
        net = models.get(f"ppyoloe_{v}", num_classes=80).materialize(backend)
        red = GradientAllReducer(net, net.gradient_buckets())
        assert set(red.ranges) == set(net.gradient_buckets())
        spans = sorted(red.ranges.values())
        assert spans[0][0] == 0 and spans[-1][1] == net.g_arena.size and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        # every live parameter's gradient view lies inside the bucket its name announces
        for s in net.slots:
            pref = next(p for p in red.ranges if s.name.startswith(p))
            a, b = red.ranges[pref]
            assert a <= s.start and s.start + s.numel <= b, s.name


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["ppyoloe_s", "ppyoloe_x"])
def test_reference_unit_test_from_name_and_from_cls(gpu_device, name):
    """The reference's own unit test transplanted (tests/unit_tests/ppyoloe_unit_test.py:11-39): build by name and by class with
    arch_params={}, eval forward of a NON-square 640x480 batch; here additionally held to the oracle's outputs."""
    from oracle import golden_util as G
    from oracle.pp_yolo_e import PPYoloE as Oracle
    from super_gradients_amd.training import models
    from super_gradients_amd.training.models import PPYoloE_S, PPYoloE_X

    x = torch.randn(1, 3, 640, 480, generator=torch.Generator().manual_seed(0))
    by_name = models.get(name, num_classes=80).eval()
    by_cls = {"ppyoloe_s": PPYoloE_S, "ppyoloe_x": PPYoloE_X}[name](arch_params={}).eval()
    ref = Oracle(name[-1], num_classes=80).eval()
    G.deterministic_fill(ref, seed=2)
    with torch.no_grad():
        (bx_r, sc_r), raw_r = ref(x)
    for net in (by_name, by_cls):
        assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
        net.load_state_dict(ref.state_dict(), strict=True)
        net.materialize(gpu_device)
        with torch.no_grad():
            out = net(x.to(gpu_device))
        assert out is not None
        (bx, sc), raw = out
        assert list(raw[4]) == list(raw_r[4]) == [20 * 15, 40 * 30, 80 * 60]
        assert_close(raw[0].cpu(), raw_r[0], 1e-4, f"{name} logits (640x480)")
        assert_close(bx.cpu(), bx_r, 1e-4, f"{name} boxes (640x480)")
