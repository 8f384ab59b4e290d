"""Trainer plumbing: required parameters, LR warm-up + schedules, zero-weight-decay grouping, EMA,
the per-batch order of operations, checkpoints - on a tiny conv network, against the same loop written with torch.optim on the
CPU oracle modules.  `backend` = host emulation of the kernels (CPU) and the MI355X (gpu)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn


def _tiny_models(device):
    from super_gradients_amd.modules.conv_bn_act_block import Conv
    from super_gradients_amd.modules.engine import SgxNetwork
    from super_gradients_amd.modules.layers import LinearLayer
    from super_gradients_amd import kernels as K

    class Tiny(SgxNetwork):
        def __init__(self):
            super().__init__()
            self.c1 = Conv(4, 8, 3, 1, "relu")
            self.c2 = Conv(8, 8, 3, 2, "relu")
            self.linear = LinearLayer(8, 6)  # 6 outputs: exercises the padded (multiple-of-4) linear path

        def _fwd(self, x):
            a = self.c2.fwd(self.c1.fwd(K.nchw_to_nhwc(x.float())))
            self._shape = tuple(a.shape)
            return (self.linear.fwd(K.avgpool_fwd(a)).contiguous(),)

        def _bwd(self, d):
            d = K.avgpool_bwd(self.linear.bwd(d.contiguous()).contiguous(), self._shape)
            self.c1.bwd(self.c2.bwd(d), need_dx=False)

        def gradient_buckets(self):
            return ["c1.", "c2.", "linear."]

    class Ref(nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = nn.Sequential()
            self.c1.add_module("conv", nn.Conv2d(4, 8, 3, 1, 1, bias=False))
            self.c1.add_module("bn", nn.BatchNorm2d(8))
            self.c2 = nn.Sequential()
            self.c2.add_module("conv", nn.Conv2d(8, 8, 3, 2, 1, bias=False))
            self.c2.add_module("bn", nn.BatchNorm2d(8))
            self.linear = nn.Linear(8, 6)

        def forward(self, x):
            x = F.relu(self.c1.bn(self.c1.conv(x)))
            x = F.relu(self.c2.bn(self.c2.conv(x)))
            return self.linear(x.mean((2, 3)))

    torch.manual_seed(0)
    ref = Ref()
    net = Tiny()
    net.load_state_dict(ref.state_dict(), strict=True)
    return ref, net


def _loader(n_batches, bs, seed):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(bs, 4, 8, 8, generator=g), torch.randint(0, 4, (bs,), generator=g)) for _ in range(n_batches)]


def _reference_lr(step_in_epoch, epoch, n, max_epochs, initial_lr, warmup_steps, warmup_initial_lr, final_ratio):
    """LinearBatchLRWarmup (callbacks.py:374-392) + CosineLRScheduler (:489-514) as the reference applies them: the warm-up value
    is set at batch start; the cosine value is set after the optimizer step and holds for the NEXT batch."""
    g = step_in_epoch + epoch * n
    if g < warmup_steps:
        return float(np.linspace(warmup_initial_lr, initial_lr, warmup_steps)[g])
    return None


@pytest.mark.parametrize("opt", ["SGD", "AdamW"])
def test_trainer_matches_torch_loop(backend, tmp_path, opt):
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.losses import CrossEntropyLoss
    from super_gradients_amd.training.utils.callbacks import CosineLRScheduler

    ref, net = _tiny_models(backend)
    n, epochs, bs = (3, 2, 4) if backend.type == "cuda" else (2, 2, 2)  # host emulation: keep the CPU suite short
    loader = _loader(n, bs, 1)
    initial_lr, warm_steps, warm_lr, ratio = 0.05, 2, 1e-3, 0.1
    oparams = dict(momentum=0.9, weight_decay=1e-2) if opt == "SGD" else dict(weight_decay=1e-2, betas=(0.9, 0.99))
    tp = dict(max_epochs=epochs, lr_mode="CosineLRScheduler", initial_lr=initial_lr, loss=CrossEntropyLoss(), optimizer=opt, optimizer_params=oparams,
              zero_weight_decay_on_bias_and_bn=True, warmup_mode="LinearBatchLRWarmup", lr_warmup_steps=warm_steps, warmup_initial_lr=warm_lr,
              cosine_final_lr_ratio=ratio, ema=True, ema_params=dict(decay=0.9, decay_type="threshold"), silent_mode=True, seed=7,
              valid_metrics_list=["Accuracy"], metric_to_watch="Accuracy")
    trainer = Trainer("tiny", ckpt_root_dir=str(tmp_path))
    res = trainer.train(net, tp, loader, valid_loader=loader)

    # the same loop on the torch modules (what the reference Trainer executes: sg_trainer.py:461-647)
    decay = [p for nme, p in ref.named_parameters() if p.dim() > 1]
    no_decay = [p for nme, p in ref.named_parameters() if p.dim() <= 1]
    groups = [{"params": no_decay, "weight_decay": 0.0}, {"params": decay}]
    o = torch.optim.SGD(groups, lr=initial_lr, **oparams) if opt == "SGD" else torch.optim.AdamW(groups, lr=initial_lr, **oparams)
    ema = {k: v.clone() for k, v in ref.state_dict().items() if v.dtype.is_floating_point}
    ref.train()
    losses = []
    for epoch in range(epochs):
        tot = 0.0
        for b, (x, y) in enumerate(loader):
            g = b + epoch * n
            if g < warm_steps:
                for pg in o.param_groups:
                    pg["lr"] = float(np.linspace(warm_lr, initial_lr, warm_steps)[g])
            loss = F.cross_entropy(ref(x), y)
            loss.backward()
            o.step()
            o.zero_grad()
            step = g + 1
            d = min(0.9, (1 + step) / (10 + step))
            for k, v in ref.state_dict().items():
                if v.dtype.is_floating_point:
                    ema[k].mul_(d).add_(v.detach(), alpha=1 - d)
            if g >= warm_steps:
                it, max_it = max(0, g - warm_steps), n * epochs - warm_steps
                lr = float(CosineLRScheduler.compute_learning_rate(it, max_it, initial_lr, ratio))
                for pg in o.param_groups:
                    pg["lr"] = lr
            tot += float(loss) * bs
        losses.append(tot / (n * bs))
    for r, l in zip(res, losses):
        assert abs(r["train"]["CrossEntropyLoss"] - l) <= 2e-4 * abs(l), (r, l)
    assert abs(res[-1]["lr"] - o.param_groups[0]["lr"]) < 1e-12
    sd = net.state_dict()
    for k, v in ref.state_dict().items():
        if v.dtype.is_floating_point:
            e = float((sd[k].cpu() - v).abs().max()) / max(float(v.abs().max()), 1e-3)
            assert e <= 5e-4, f"{k}: {e:.2e}"
    ema_sd = trainer.ema_model.state_dict()
    for k, v in ema.items():
        e = float((ema_sd[k].cpu() - v).abs().max()) / max(float(v.abs().max()), 1e-3)
        assert e <= 5e-4, f"ema {k}: {e:.2e}"
    # checkpoints with the reference's layout, resumable
    ck = torch.load(os.path.join(str(tmp_path), "tiny", "ckpt_latest.pth"), weights_only=False)
    assert set(["net", "ema_net", "optimizer_state_dict", "epoch"]).issubset(ck.keys()) and ck["epoch"] == epochs - 1
    assert os.path.exists(os.path.join(str(tmp_path), "tiny", "ckpt_best.pth"))
    assert "Accuracy" in res[-1]["valid"] and 0.0 <= res[-1]["valid"]["Accuracy"] <= 1.0


def test_trainer_argument_contract(tmp_path):
    from super_gradients_amd.training import Trainer

    with pytest.raises(KeyError):
        Trainer("x", device="cuda")
    with pytest.raises(KeyError):
        Trainer("x", multi_gpu="DDP")
    t = Trainer("x", ckpt_root_dir=str(tmp_path))
    with pytest.raises(ValueError):
        t.train(nn.Linear(2, 2), {"max_epochs": 1}, train_loader=None)
    with pytest.raises(TypeError):
        t.train(nn.Linear(2, 2), {"max_epochs": 1, "lr_mode": "StepLRScheduler", "initial_lr": 0.1, "loss": "CrossEntropyLoss"}, train_loader=[1])


def test_lr_schedules_match_reference():
    """Closed forms of the schedules against the reference's own callback classes (live, when /root/reference is present) and
    against the values the YOLO-NAS recipe implies (coco2017_yolo_nas_train_params.yaml: warm-up 1e-6 -> 2e-4 over 1000 steps, cosine to 0.1)."""
    from oracle import ref_shim
    from super_gradients_amd.training.utils import callbacks as C
    from super_gradients_amd.training.utils.utils import HpmStruct

    tp = HpmStruct(max_epochs=5, lr_warmup_epochs=0, lr_warmup_steps=7, lr_cooldown_epochs=0, batch_accumulate=1, warmup_initial_lr=1e-6)
    n = 10

    class Opt:
        param_groups = [{"name": "default", "lr": 0.0}]

    ours = [C.LinearBatchLRWarmup(warmup_initial_lr=1e-6, initial_lr=2e-4, train_loader_len=n, lr_warmup_steps=7, training_params=tp, net=None),
            C.CosineLRScheduler(max_epochs=5, cosine_final_lr_ratio=0.1, initial_lr=2e-4, update_param_groups=False, train_loader_len=n, net=None, training_params=tp)]
    refs = None
    if ref_shim.available():
        ref_shim.install()
        import super_gradients.training.utils.callbacks.callbacks as R

        refs = [R.LinearBatchLRWarmup(warmup_initial_lr=1e-6, initial_lr=2e-4, train_loader_len=n, lr_warmup_steps=7, training_params=tp, net=None),
                R.CosineLRScheduler(max_epochs=5, cosine_final_lr_ratio=0.1, initial_lr=2e-4, update_param_groups=False, train_loader_len=n, net=None,
                                    training_params=tp)]
    trace = []
    for side in ([ours] + ([refs] if refs else [])):
        o = Opt()
        o.param_groups = [{"name": "default", "lr": 2e-4}]
        vals = []
        for epoch in range(5):
            for b in range(n):
                ctx = C.PhaseContext(epoch=epoch, batch_idx=b, optimizer=o)
                side[0].on_train_batch_start(ctx)
                vals.append(o.param_groups[0]["lr"])
                side[1].on_train_batch_gradient_step_end(ctx) if hasattr(side[1], "on_train_batch_gradient_step_end") else side[1](ctx)
        trace.append(vals)
    v = trace[0]
    assert abs(v[0] - 1e-6) < 1e-15 and abs(v[6] - 2e-4) < 1e-15  # linspace end points
    it, max_it = 49 - 1 - 7, 50 - 7  # lr in force during the last batch was set after batch 48
    expect = 0.5 * 2e-4 * (1 + math.cos(it / (max_it + 1) * math.pi)) * 0.9 + 2e-4 * 0.1
    assert abs(v[-1] - expect) < 1e-15
    if refs:
        assert np.allclose(trace[0], trace[1], rtol=0, atol=1e-18)


@pytest.mark.gpu
def test_trainer_resnet18_cifar_recipe_shape(gpu_device, tmp_path):
    """BASELINE.json configs[0] in miniature: ResNet-18 CIFAR, bs 32, SGD lr 0.1 momentum 0.9 wd 1e-4, CE, step LR
    (recipes/training_hyperparams/cifar10_resnet_train_params.yaml) - a few batches through Trainer.train() against the same loop
    on the CPU oracle with torch.optim.SGD."""
    from oracle.resnet import build
    from super_gradients_amd.training import Trainer, models

    torch.manual_seed(1)
    ref = build("resnet18_cifar", 10).train()
    net = models.get("resnet18_cifar", num_classes=10)
    net.load_state_dict(ref.state_dict(), strict=True)
    g = torch.Generator().manual_seed(2)
    loader = [(torch.randn(32, 3, 32, 32, generator=g), torch.randint(0, 10, (32,), generator=g)) for _ in range(4)]
    tp = dict(max_epochs=1, lr_mode="StepLRScheduler", lr_updates=[100, 150, 200], lr_decay_factor=0.1, initial_lr=0.01, loss="CrossEntropyLoss", optimizer="SGD",
              optimizer_params=dict(momentum=0.9, weight_decay=1e-4), silent_mode=True, valid_metrics_list=["Accuracy"], metric_to_watch="Accuracy")
    trainer = Trainer("cifar_resnet18", ckpt_root_dir=str(tmp_path))
    res = trainer.train(net, tp, loader, valid_loader=loader[:1])
    o = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)  # recipe lr is 0.1; 0.01 keeps 4 steps on random data out of the chaotic regime
    tot = 0.0
    for x, y in loader:
        loss = F.cross_entropy(ref(x), y)
        loss.backward()
        o.step()
        o.zero_grad()
        tot += float(loss.detach()) * 32
    assert abs(res[0]["train"]["CrossEntropyLoss"] - tot / 128) <= 2e-3 * tot / 128, (res[0], tot / 128)
    sd = net.state_dict()
    worst = max(float((sd[k].cpu() - v).norm() / v.norm().clamp_min(1e-6)) for k, v in ref.state_dict().items() if v.dtype.is_floating_point and v.dim() > 1)
    assert worst <= 2e-2, f"weights after 4 SGD steps differ by {worst:.2e} (relative L2, worst tensor)"


def test_batched_weight_transposes_do_not_change_the_step(backend, monkeypatch):
    """SGX_WT_BATCH=1 (all data-gradient weight transposes as one launch per step): outputs and every gradient are bit-identical to
    the per-convolution path, over two optimizer steps (the transposed copies must follow the updated weights)."""
    from super_gradients_amd.training.utils.optimizers import ArenaSGD

    x = torch.randn(4, 4, 8, 8, generator=torch.Generator().manual_seed(1))
    up = torch.randn(4, 6, generator=torch.Generator().manual_seed(2))
    results = []
    for mode in ("0", "1"):
        monkeypatch.setenv("SGX_WT_BATCH", mode)
        _, net = _tiny_models(backend)
        net.materialize(backend).train()
        assert net.wt_batch == (mode == "1") and (net._wt_njobs == 1 + 4 if mode == "1" else net._wt_jobs is None)
        opt = ArenaSGD(net, lr=0.1, momentum=0.9)
        outs = []
        for _ in range(2):
            y = net(x.to(backend))
            y.backward(up.to(backend))
            net.join_side()
            outs.append((y.detach().cpu().clone(), net.g_arena.buf.cpu().clone()))
            opt.step()
            opt.zero_grad()
        results.append(outs)
    for (y0, g0), (y1, g1) in zip(*results):
        assert torch.equal(y0, y1) and torch.equal(g0, g1)


def _yolo_nas_recipe_containers(num_classes=3):
    """training_hyperparams/coco2017_yolo_nas_train_params.yaml over default_train_params.yaml as yaml.safe_load leaves them (the reference's
    files when /root/reference is present; otherwise the same entries written out), `${arch_params.num_classes}` - a hydra interpolation -
    substituted.  Exponent-form numbers arrive as strings from a YAML 1.1 loader, `_target_` entries as plain mappings."""
    ref = "/root/reference/src/super_gradients/recipes/training_hyperparams/"
    if os.path.isdir(ref):
        import yaml

        tp = yaml.safe_load(open(ref + "default_train_params.yaml"))
        tp.update(yaml.safe_load(open(ref + "coco2017_yolo_nas_train_params.yaml")))
        tp.pop("defaults", None)
    else:
        tp = dict(max_epochs=300, warmup_mode="LinearBatchLRWarmup", warmup_initial_lr="1e-6", lr_warmup_steps=1000, lr_warmup_epochs=0, initial_lr="2e-4",
                  lr_mode="CosineLRScheduler", cosine_final_lr_ratio=0.1, zero_weight_decay_on_bias_and_bn=True, batch_accumulate=1,
                  save_ckpt_epoch_list=[100, 200, 250], loss="PPYoloELoss", criterion_params=dict(use_static_assigner=False, num_classes="${arch_params.num_classes}"),
                  optimizer="AdamW", optimizer_params=dict(weight_decay=0.00001), ema=True, ema_params=dict(decay=0.9997, decay_type="threshold"),
                  mixed_precision=False, sync_bn=True, lr_updates={"_target_": "super_gradients.training.utils.utils.empty_list"},
                  valid_metrics_list=[{"DetectionMetrics": dict(
                      score_thres=0.1, top_k_predictions=300, num_cls="${arch_params.num_classes}", normalize_targets=True,
                      post_prediction_callback={"_target_": "super_gradients.training.models.detection_models.pp_yolo_e.PPYoloEPostPredictionCallback",
                                                "score_threshold": 0.01, "nms_top_k": 1000, "max_predictions": 300, "nms_threshold": 0.7})}],
                  pre_prediction_callback=None, metric_to_watch="mAP@0.50:0.95", greater_metric_to_watch_is_better=True, _convert_="all")

    def sub(o):
        if isinstance(o, dict):
            return {k: sub(v) for k, v in o.items()}
        if isinstance(o, list):
            return [sub(v) for v in o]
        return num_classes if o == "${arch_params.num_classes}" else o

    return sub(tp)


def test_reference_recipe_containers_resolve(tmp_path):
    """The reference's YOLO-NAS recipe as a YAML loader hands it over is accepted as it is: exponent-form strings become numbers, the
    `_target_` entries (post-prediction callback of the metric, the empty lr_updates list) become objects of the classes registered here, the
    metric list goes through the factory."""
    from super_gradients_amd.common.factories import MetricsFactory
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.metrics import DetectionMetrics
    from super_gradients_amd.training.models.detection_models.pp_yolo_e.post_prediction_callback import PPYoloEPostPredictionCallback

    tp = Trainer("recipe", ckpt_root_dir=str(tmp_path))._params(_yolo_nas_recipe_containers())
    assert tp.initial_lr == 2e-4 and tp.warmup_initial_lr == 1e-6 and tp.optimizer_params["weight_decay"] == 1e-5
    assert tp.lr_updates == [] and tp.sync_bn is True and tp.ema_params["decay_type"] == "threshold"
    (m,) = [MetricsFactory().get(c) for c in tp.valid_metrics_list]
    assert isinstance(m, DetectionMetrics) and isinstance(m.post_prediction_callback, PPYoloEPostPredictionCallback)
    assert m.post_prediction_callback.nms_top_k == 1000 and m.score_threshold == 0.1 and m.top_k_predictions == 300
    assert tp.metric_to_watch in m.component_names


@pytest.mark.gpu
def test_reference_recipe_dict_trains(gpu_device, tmp_path):
    """... and one epoch of it runs: YOLO-NAS-S, the recipe's own loss / optimizer / EMA / warm-up + cosine / DetectionMetrics entries."""
    from super_gradients_amd.training import Trainer, models
    from util import synthetic_targets

    tp = _yolo_nas_recipe_containers(num_classes=3)
    tp.update(max_epochs=1, lr_warmup_steps=2, sync_bn=False)
    net = models.get("yolo_nas_s", num_classes=3)
    g = torch.Generator().manual_seed(0)
    loader = [(torch.rand(2, 3, 128, 128, generator=g), synthetic_targets(2, seed=i, size=128, num_classes=3)) for i in range(3)]
    res = Trainer("recipe_run", ckpt_root_dir=str(tmp_path)).train(net, training_params=tp, train_loader=loader, valid_loader=loader[:1])
    row = res[-1] if isinstance(res, list) else res
    assert np.isfinite(row["train"]["PPYoloELoss/loss"]) and "mAP@0.50:0.95" in row["valid"]
    assert os.path.exists(os.path.join(str(tmp_path), "recipe_run", "ckpt_latest.pth"))


class _CollectLR:
    """The reference tests' CollectLRCallback / TestLRCallback (tests/unit_tests/lr_warmup_test.py:12-21, callbacks.py:768-780) in one."""

    def __new__(cls):
        from super_gradients_amd.training.utils.callbacks import Callback

        class Collect(Callback):
            def __init__(self):
                self.per_step, self.per_epoch, self.after_validation = [], [], []

            def on_train_batch_end(self, context):
                self.per_step.append(context.optimizer.param_groups[0]["lr"])

            def on_train_loader_end(self, context):
                self.per_epoch.append(context.optimizer.param_groups[0]["lr"])

            def on_validation_loader_end(self, context):
                self.after_validation.append(context.optimizer.param_groups[0]["lr"])

        return Collect()


@pytest.fixture
def no_op_kernels():
    """Host logic only: every entry point bound to a function that returns at once (tools/host_overhead.py's generated library), tensors on the
    CPU.  Results of 'kernels' are garbage; the tests that use this read none."""
    import ctypes
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import host_overhead
    from super_gradients_amd import _lib
    from super_gradients_amd import kernels as K

    _lib._LIB = _lib.bind(ctypes.CDLL(host_overhead.build_null_library()))
    _lib._TEST_HOST_MODE = True
    K.clear_caches()
    try:
        yield torch.device("cpu")
    finally:
        _lib._LIB = None
        _lib._TEST_HOST_MODE = False
        K.clear_caches()


@pytest.mark.parametrize("case", ["warmup_step", "warmup_cosine", "warmup_cosine_cooldown", "warmup_from_above", "batch_warmup_cosine"])
def test_reference_lr_unit_test_vectors(no_op_kernels, tmp_path, case):
    """The reference's own expected learning-rate sequences (tests/unit_tests/lr_warmup_test.py:54-218, lr_cooldown_test.py:11-52), produced here
    by Trainer.train() on a small network: LinearEpochLRWarmup into StepLR / cosine (with and without cool-down epochs), a warm-up that starts
    above the initial rate, and LinearBatchLRWarmup into a per-step cosine.  The numbers are the reference's."""
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.utils.callbacks import CosineLRScheduler

    _, net = _tiny_models(no_op_kernels)
    n_train = 2 if case != "batch_warmup_cosine" else 6  # the reference's loaders have two batches per epoch (5 samples, batch 4): the cosine is per step
    loader = _loader(n_train, 4, 7)
    cb = _CollectLR()
    common = dict(initial_lr=1, loss="CrossEntropyLoss", optimizer="SGD", optimizer_params={"weight_decay": 1e-4, "momentum": 0.9}, ema=False,
                  phase_callbacks=[cb], silent_mode=True, save_model=False)
    if case == "warmup_step":
        tp = dict(common, max_epochs=5, lr_updates=[10], lr_decay_factor=0.1, lr_mode="StepLRScheduler", lr_warmup_epochs=3, warmup_mode="LinearEpochLRWarmup")
        expected, got = [0.25, 0.5, 0.75, 1.0, 1.0], "after_validation"
    elif case == "warmup_cosine":
        tp = dict(common, max_epochs=5, cosine_final_lr_ratio=0.2, lr_mode="CosineLRScheduler", lr_warmup_epochs=3, warmup_mode="LinearEpochLRWarmup")
        expected, got = [0.25, 0.5, 0.75, 0.9236067977499791, 0.4763932022500211], "after_validation"
    elif case == "warmup_cosine_cooldown":
        tp = dict(common, max_epochs=7, cosine_final_lr_ratio=0.2, lr_mode="CosineLRScheduler", lr_cooldown_epochs=2, lr_warmup_epochs=3)
        expected, got = [0.25, 0.5, 0.75, 0.9236067977499791, 0.4763932022500211, 0.4763932022500211, 0.4763932022500211], "after_validation"
    elif case == "warmup_from_above":
        tp = dict(common, max_epochs=5, lr_updates=[10], lr_decay_factor=0.1, lr_mode="StepLRScheduler", lr_warmup_epochs=3, warmup_initial_lr=4.0,
                  warmup_mode="LinearEpochLRWarmup")
        expected, got = [4.0, 3.0, 2.0, 1.0, 1.0], "per_epoch"
    else:
        steps, epochs = 4, 3
        tp = dict(common, max_epochs=epochs, lr_mode="CosineLRScheduler", cosine_final_lr_ratio=0.2, warmup_initial_lr=0.05, warmup_mode="LinearBatchLRWarmup",
                  lr_warmup_steps=steps)
        total = epochs * n_train - steps
        expected = np.linspace(0.05, 1, steps).tolist() + list(CosineLRScheduler.compute_learning_rate(step=np.arange(0, total), total_steps=total,
                                                                                                       initial_lr=1, final_lr_ratio=0.2))
        got = "per_step"
    Trainer("lr_vectors", ckpt_root_dir=str(tmp_path)).train(net, tp, loader, valid_loader=loader[:1])
    np.testing.assert_allclose(np.array(getattr(cb, got)), np.array(expected), rtol=1e-6 if got != "per_step" else 1e-4)


def test_reference_loss_logging_names(no_op_kernels, tmp_path):
    """tests/unit_tests/loss_loggings_test.py:23-105 transplanted: a scalar loss logs under the criterion's class name, a tuple loss without
    names under "<Class>/loss_i", with `component_names` under "<Class>/<name>" (sg_trainer.py:2407-2420); a bare component name given as
    metric_to_watch is prefixed the same way."""
    from super_gradients_amd.training import Trainer

    class Unnamed(torch.nn.CrossEntropyLoss):
        def forward(self, input, target):
            loss = super().forward(input, target)
            return loss, torch.cat((loss.unsqueeze(0), loss.unsqueeze(0))).detach()

    class Named(Unnamed):
        component_names = ["loss_A", "loss_B"]

    def run(loss, **kw):
        _, net = _tiny_models(no_op_kernels)
        tr = Trainer("loss_names", ckpt_root_dir=str(tmp_path))
        tp = dict(max_epochs=1, lr_updates=[1], lr_decay_factor=0.1, lr_mode="StepLRScheduler", initial_lr=0.1, loss=loss, optimizer="SGD",
                  optimizer_params={"weight_decay": 1e-4, "momentum": 0.9}, silent_mode=True, save_model=False, **kw)
        res = tr.train(net, tp, _loader(1, 4, 3), valid_loader=_loader(1, 4, 4))
        return tr, res

    tr, _ = run(torch.nn.CrossEntropyLoss())
    assert tr.loss_logging_items_names == ["CrossEntropyLoss"]
    tr, _ = run(Unnamed())
    assert tr.loss_logging_items_names == ["Unnamed/loss_0", "Unnamed/loss_1"]
    tr, res = run(Named(), metric_to_watch="loss_B", greater_metric_to_watch_is_better=False)
    assert tr.loss_logging_items_names == ["Named/loss_A", "Named/loss_B"] and tr.training_params.metric_to_watch == "Named/loss_B"
    assert set(res[0]["train"]) == {"Named/loss_A", "Named/loss_B"} and "Named/loss_B" in res[0]["valid"]


def test_reference_optimizer_params_defaults(no_op_kernels):
    """tests/unit_tests/optimizer_params_override_test.py:9-70 transplanted: SGD's recipe defaults (weight decay 1e-4, momentum 0.9:
    training/params.py:88) sit under whatever `optimizer_params` the recipe gives, and the merged dictionary is written back to the training
    params; AdamW has no recipe defaults (torch's own apply)."""
    from super_gradients_amd.training.utils.optimizers import build_optimizer
    from super_gradients_amd.training.utils.utils import HpmStruct

    _, net = _tiny_models(no_op_kernels)
    net.materialize(no_op_kernels)
    tp = HpmStruct(optimizer="SGD", optimizer_params={"momentum": 0.8}, zero_weight_decay_on_bias_and_bn=True)
    opt = build_optimizer(net, 0.1, tp)
    assert tp.optimizer_params == {"weight_decay": 1e-4, "momentum": 0.8}
    assert opt.defaults["momentum"] == 0.8 and opt.defaults["weight_decay"] == 1e-4
    tp = HpmStruct(optimizer="SGD", optimizer_params={}, zero_weight_decay_on_bias_and_bn=False)
    opt = build_optimizer(net, 0.1, tp)
    assert opt.defaults["momentum"] == 0.9 and opt.defaults["weight_decay"] == 1e-4 and tp.optimizer_params == {"weight_decay": 1e-4, "momentum": 0.9}
    tp = HpmStruct(optimizer="AdamW", optimizer_params={}, zero_weight_decay_on_bias_and_bn=False)
    assert build_optimizer(net, 0.1, tp).defaults["weight_decay"] == 1e-2
