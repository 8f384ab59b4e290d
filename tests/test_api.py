"""The Python drop-in boundary: factory / registry / models.get() contracts and error behaviour, the C-ABI symbol table,
the "no CPU fallback" rule.  Host logic only - runs everywhere."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    """Every function include/sgx_hip.h declares is exported by the built library and has a ctypes prototype (and vice versa)."""
    from super_gradients_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "sgx_hip.h")).read()
    declared = set(re.findall(r"\b(sgx_[a-zA-Z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), f"header vs binding: {sorted(declared ^ set(_lib.PROTOTYPES))}"
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsgx_hip.so not built")
    cdll = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(cdll, name), f"{name} declared in the header but not exported"
    assert cdll.sgx_version() >= 1


def test_default_arithmetic_constants_match_the_library():
    """kernels.DEFAULT_CONV_MATH / DEFAULT_WGRAD_MATH are what the tests restore after switching arithmetic: they must be the modes the
    library STARTS in (round 4: a stale constant left every whole-model test of a full-suite run in the previous round's mode after the first
    kernel test had "restored" it).  Read from a fresh process: no test of this session can have switched anything there."""
    import subprocess
    import sys

    from super_gradients_amd import _lib
    from super_gradients_amd import kernels as K

    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsgx_hip.so not built")
    code = ("import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); l.sgx_conv_get_math.restype = ctypes.c_int32; l.sgx_conv_get_wgrad_math.restype = ctypes.c_int32; "
            "print(l.sgx_conv_get_math(), l.sgx_conv_get_wgrad_math())")
    env = {k: v for k, v in os.environ.items() if not k.startswith("SGX_")}
    out = subprocess.run([sys.executable, "-c", code, _lib.LIB_PATH], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    conv, wgrad = (int(v) for v in out.stdout.split())
    assert K.CONV_MATH[K.DEFAULT_CONV_MATH] == conv, f"library starts in conv math {conv}, kernels.DEFAULT_CONV_MATH = {K.DEFAULT_CONV_MATH!r}"
    assert _lib.WGRAD_MATH[K.DEFAULT_WGRAD_MATH] == wgrad, f"library starts in weight-gradient math {wgrad}, kernels.DEFAULT_WGRAD_MATH = {K.DEFAULT_WGRAD_MATH!r}"


def test_no_cpu_fallback():
    from super_gradients_amd import _lib
    from super_gradients_amd import kernels as K
    from super_gradients_amd.training import models

    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    _lib._TEST_HOST_MODE = False
    with pytest.raises(_lib.SgxError):
        _lib.ptr(torch.zeros(8))  # a CPU tensor reaching a kernel wrapper is rejected before any library call
    net = models.get("yolo_nas_s", num_classes=3)
    with pytest.raises(Exception):
        net(torch.zeros(1, 3, 64, 64))  # no device to materialize on / CPU batch rejected


def test_models_get_contract():
    """model_factory.py:88-92,137-138,179-183: unknown names, missing num_classes, _sg_model_name; state_dict layout of the reference."""
    from super_gradients_amd.common.factories import UnknownTypeException
    from super_gradients_amd.training import models

    with pytest.raises(UnknownTypeException):
        models.get("yolo_nas_xxl", num_classes=80)
    with pytest.raises(ValueError):
        models.get("yolo_nas_s")
    with pytest.raises(ValueError):
        models.get(123, num_classes=80)
    net = models.get("yolo_nas_s", num_classes=17)
    assert models.get_model_name(net) == "yolo_nas_s" and net.num_classes == 17
    sd = net.state_dict()
    assert len(sd) == 921 and sum(1 for _ in net.parameters()) == 534
    assert sd["heads.head1.cls_pred.weight"].shape[0] == 17 and "backbone.stem.conv.rbr_reparam.weight" in sd
    assert net.get_input_shape_steps() == (32, 32)
    for name in ("resnet18", "resnet50", "resnet18_cifar", "yolo_nas_m", "yolo_nas_l"):
        assert models.get(name, num_classes=10) is not None


def test_registry_and_factory_semantics():
    """registry.py:36-41 (re-registering a different class raises), base_factory.py:37-72 (str / single-entry mapping / passthrough, fuzzy names)."""
    from super_gradients_amd.common.factories import BaseFactory, DetectionModulesFactory, UnknownTypeException
    from super_gradients_amd.common.registry import Registry

    reg = Registry()

    @reg.register("Thing", deprecated_name="thing_old")
    class Thing:
        def __init__(self, a=1):
            self.a = a

    reg.register("Thing")(Thing)  # same class again: fine
    with pytest.raises(Exception):
        reg.register("Thing")(type("Other", (), {}))
    f = BaseFactory(reg)
    assert f.get("Thing").a == 1 and f.get({"Thing": {"a": 5}}).a == 5 and f.get("thing").a == 1  # fuzzy name match
    with pytest.warns(DeprecationWarning):
        f.get("thing_old")
    obj = object()
    assert f.get(obj) is obj
    with pytest.raises(UnknownTypeException):
        f.get("Nope")
    with pytest.raises(RuntimeError):
        f.get({"Thing": {}, "Other": {}})
    assert DetectionModulesFactory.insert_module_param({"X": {"a": 1}}, "in_channels", 3) == {"X": {"a": 1, "in_channels": 3}}
    assert DetectionModulesFactory.insert_module_param("X", "in_channels", 3) == {"X": {"in_channels": 3}}


def test_loss_and_callback_protocols():
    """PPYoloELoss component_names / constructor (ppyolo_loss.py:643-653,990-992); keyword-only callback constructor (post_prediction_callback.py:13-40)."""
    from super_gradients_amd.training.losses import PPYoloELoss
    from super_gradients_amd.training.models.detection_models.pp_yolo_e.post_prediction_callback import PPYoloEPostPredictionCallback

    crit = PPYoloELoss(num_classes=80)
    assert crit.component_names == ["loss_cls", "loss_iou", "loss_dfl", "loss"] and crit.use_static_assigner is True
    with pytest.warns(DeprecationWarning):
        PPYoloELoss(num_classes=80, reg_max=16)
    with pytest.raises(TypeError):
        PPYoloEPostPredictionCallback(0.1, 0.6, 1000, 300)  # positional arguments are rejected like in the reference
    cb = PPYoloEPostPredictionCallback(score_threshold=0.1, nms_threshold=0.6, nms_top_k=1000, max_predictions=300)
    assert cb.multi_label_per_box is True and cb.class_agnostic_nms is False


def test_trainer_resume(backend, tmp_path):
    """Checkpoint round trip: train 1 epoch, resume for a second one == training 2 epochs in one go (weights, optimizer state, EMA, LR)."""
    from super_gradients_amd.training import Trainer
    from super_gradients_amd.training.losses import CrossEntropyLoss
    from test_trainer import _loader, _tiny_models

    loader = _loader(2, 2, 1)

    def params(epochs, **kw):
        return dict(max_epochs=epochs, lr_mode="StepLRScheduler", lr_updates=[1], lr_decay_factor=0.5, initial_lr=0.05, loss=CrossEntropyLoss(), optimizer="SGD",
                    optimizer_params=dict(momentum=0.9, weight_decay=1e-3), ema=True, ema_params=dict(decay=0.9, decay_type="constant"), silent_mode=True, **kw)

    _, net_a = _tiny_models(backend)
    Trainer("a", ckpt_root_dir=str(tmp_path)).train(net_a, params(2), loader)
    _, net_c = _tiny_models(backend)
    # the schedule must see max_epochs=2 from the start for both runs: the interrupted run trains with max_epochs=2 and is stopped after
    # epoch 0 by a callback
    from super_gradients_amd.training.utils.callbacks import Callback

    class StopAfterFirst(Callback):
        def on_train_loader_end(self, context):
            context.stop_training = True

    _, net_b = _tiny_models(backend)
    Trainer("b2", ckpt_root_dir=str(tmp_path)).train(net_b, params(2, phase_callbacks=[StopAfterFirst()]), loader)
    tp = params(2, resume_path=os.path.join(str(tmp_path), "b2", "ckpt_latest.pth"))
    tr = Trainer("c", ckpt_root_dir=str(tmp_path))
    tr.train(net_c, tp, loader)
    for (k, va), vc in zip(net_a.state_dict().items(), net_c.state_dict().values()):
        if va.dtype.is_floating_point:
            assert torch.allclose(va.cpu(), vc.cpu(), rtol=1e-5, atol=1e-6), k


def test_replace_head_transfer_learning(tmp_path):
    """tests/unit_tests/replace_head_test.py:26-40 and model_factory.py:227-251: build with the checkpoint's class count, load it, then
    replace_head(new_num_classes): YOLO-NAS swaps only the class-prediction convs (statistics-matched random weights), PP-YOLOE re-creates
    pred_cls with zero weights and the prior bias; everything else keeps the loaded values; models.get(checkpoint_num_classes=...) does both."""
    import math

    from super_gradients_amd.training import models

    src = models.get("yolo_nas_s", num_classes=80)
    ckpt = str(tmp_path / "ckpt.pth")
    torch.save({"net": src.state_dict()}, ckpt)
    net = models.get("yolo_nas_s", num_classes=100, checkpoint_path=ckpt, checkpoint_num_classes=80, strict_load=True)
    assert net.num_classes == 100
    sd, ref = net.state_dict(), src.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k in sd:
        if ".cls_pred." in k:
            assert sd[k].shape[0] == 100
        else:
            assert torch.equal(sd[k], ref[k]), k
    w_old, w_new = ref["heads.head1.cls_pred.weight"], sd["heads.head1.cls_pred.weight"]
    assert abs(float(w_new.std()) - float(w_old.std())) < 0.2 * float(w_old.std())   # drawn with the old layer's statistics
    with pytest.raises(ValueError):
        net.replace_head()
    pp = models.get("ppyoloe_s", num_classes=80)
    pp.replace_head(new_num_classes=100)
    assert pp.num_classes == 100
    for i in range(3):
        assert tuple(pp.head.pred_cls[i].weight.shape)[:1] == (100,) and float(pp.head.pred_cls[i].weight.abs().max()) == 0.0
        assert torch.allclose(pp.head.pred_cls[i].bias.detach(), torch.full((100,), -math.log(99.0)))
    rn = models.get("resnet18", num_classes=1000)
    rn.replace_head(new_num_classes=10)
    assert rn.linear.weight.shape[0] == 10


def test_models_get_adaptive_load_and_input_channels(tmp_path):
    """models.get: checkpoint loading follows the reference's adaptive_load_state_dict (checkpoint_utils.py:79-106) - every mode except
    "off" is strict first; "no_key_matching" falls back to pairing tensors by position and shape and raises when that fails (a checkpoint
    that fits nothing can no longer be 'loaded' silently); num_input_channels rebuilds the stem (model_factory.py:253-254)."""
    import torch

    from super_gradients_amd.training import models

    net = models.get("yolo_nas_s", num_classes=80)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    torch.save({"net": sd}, tmp_path / "ok.pth")
    a = models.get("yolo_nas_s", num_classes=80, checkpoint_path=str(tmp_path / "ok.pth"))
    assert all(torch.equal(v, sd[k]) for k, v in a.state_dict().items())
    renamed = {("model." + k): v for k, v in sd.items()}      # other layer names, same order and shapes
    torch.save(renamed, tmp_path / "renamed.pth")
    b = models.get("yolo_nas_s", num_classes=80, checkpoint_path=str(tmp_path / "renamed.pth"))
    assert all(torch.equal(v, sd[k]) for k, v in b.state_dict().items())
    with pytest.raises(RuntimeError):
        models.get("yolo_nas_s", num_classes=80, checkpoint_path=str(tmp_path / "renamed.pth"), strict_load="on")
    torch.save({"w": torch.zeros(3)}, tmp_path / "junk.pth")
    with pytest.raises(RuntimeError):
        models.get("yolo_nas_s", num_classes=80, checkpoint_path=str(tmp_path / "junk.pth"))
    c = models.get("yolo_nas_s", num_classes=80, checkpoint_path=str(tmp_path / "ok.pth"), num_input_channels=5)
    csd = c.state_dict()
    assert c.get_input_channels() == 5 and tuple(csd["backbone.stem.conv.branch_3x3.conv.weight"].shape) == (48, 5, 3, 3)
    assert torch.equal(csd["backbone.stage1.downsample.branch_3x3.conv.weight"], sd["backbone.stage1.downsample.branch_3x3.conv.weight"])
    r = models.get("resnet18", num_classes=10, num_input_channels=5)
    assert r.get_input_channels() == 5 and tuple(r.state_dict()["conv1.weight"].shape)[1] == 5


def test_registry_mirrors_follow_reassignment(backend):
    """After materialisation sub-modules / parameters / buffers are mirrored into the instance dictionaries (plain attribute reads on the
    hot path); re-assigning a name must drop its mirror so that nn.Module's registries stay the single source of truth."""
    from test_trainer import _tiny_models

    _, net = _tiny_models(backend)
    net.materialize(backend)
    bn = net.c1.bn
    assert bn.__dict__["running_mean"] is bn._buffers["running_mean"] and net.__dict__["c1"] is net._modules["c1"]
    new = torch.ones(8, device=backend)
    bn.running_mean = new
    assert bn.running_mean is new and bn._buffers["running_mean"] is new and "running_mean" not in bn.__dict__
    assert dict(bn.named_buffers())["running_mean"] is new


def test_reference_strict_load_unit_test_scenarios(tmp_path):
    """tests/unit_tests/strictload_enum_test.py:104-158 transplanted (a second random-init model stands in for the downloaded ImageNet weights):
    StrictLoad.ON loads an exact checkpoint and raises RuntimeError on missing or renamed keys; OFF tolerates the missing classifier;
    NO_KEY_MATCHING loads a checkpoint whose keys were renamed to their positions."""
    from super_gradients_amd.training import StrictLoad, models

    def same(a, b, skip=()):
        return all(torch.equal(v, b.state_dict()[k]) for k, v in a.state_dict().items() if not k.startswith(skip))

    torch.manual_seed(0)
    pre = models.get("resnet18", num_classes=1000)
    torch.manual_seed(1)
    assert not same(models.get("resnet18", num_classes=1000), pre)
    p = str(tmp_path / "on.pth")
    torch.save(pre.state_dict(), p)
    assert same(models.get("resnet18", num_classes=1000, checkpoint_path=p, strict_load=StrictLoad.ON), pre)
    p = str(tmp_path / "off.pth")
    torch.save({k: v for k, v in pre.state_dict().items() if not k.startswith("linear.")}, p)
    with pytest.raises(RuntimeError):
        models.get("resnet18", num_classes=1000, checkpoint_path=p, strict_load=StrictLoad.ON)
    assert same(models.get("resnet18", num_classes=1000, checkpoint_path=p, strict_load=StrictLoad.OFF), pre, skip=("linear.",))
    p = str(tmp_path / "renamed.pth")
    torch.save({str(i): v for i, (k, v) in enumerate(pre.state_dict().items())}, p)
    with pytest.raises(RuntimeError):
        models.get("resnet18", num_classes=1000, checkpoint_path=p, strict_load=StrictLoad.ON)
    assert same(models.get("resnet18", num_classes=1000, checkpoint_path=p, strict_load=StrictLoad.NO_KEY_MATCHING), pre)


class _ForeignProcessor:  # stands in for a class of another package pickled into a checkpoint (the reference's Processing objects)
    constructed = 0

    def __init__(self):
        type(self).constructed += 1

    def __setstate__(self, state):
        type(self).constructed += 1


def test_checkpoint_with_pickled_objects_loads_weights_without_constructing_them(tmp_path):
    """the reference's Trainer pickles its image processor into "processing_params" (sg_trainer.py:710-712); weights_only=True
    refuses such a file.  read_checkpoint reads it with every foreign global turned into an inert placeholder: the weights arrive, nothing of
    the foreign class runs, models.get warns that the processing parameters are to be set by hand."""
    import warnings

    import torch

    from super_gradients_amd.training import models
    from super_gradients_amd.training.utils.checkpoint_utils import OpaqueObject, contains_opaque, read_checkpoint

    net = models.get("resnet18_cifar", num_classes=10)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    path = str(tmp_path / "ref_style.pth")
    obj = _ForeignProcessor()
    torch.save({"net": sd, "epoch": 3, "processing_params": {"class_names": ["a"], "image_processor": obj}}, path)
    _ForeignProcessor.constructed = 0
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ckpt = read_checkpoint(path)
    assert _ForeignProcessor.constructed == 0, "the pickled object's class must not be constructed or have its state set"
    assert any("NOT constructed" in str(x.message) for x in w)
    assert isinstance(ckpt["processing_params"]["image_processor"], OpaqueObject) and contains_opaque(ckpt)
    assert ckpt["epoch"] == 3 and all(torch.equal(ckpt["net"][k], sd[k]) for k in sd)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        net2 = models.get("resnet18_cifar", num_classes=10, checkpoint_path=path)
    assert all(torch.equal(a, b) for a, b in zip(net2.state_dict().values(), sd.values()))
    # a clean file takes the weights_only path silently
    torch.save({"net": sd}, path)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        read_checkpoint(path)
    assert not w


def test_inert_unpickler_admits_exact_names_only(tmp_path):
    """the second reading of a checkpoint admitted any ("torch", name) ending in "Storage", and protocol 4 resolves dotted names by
    attribute traversal.  Exact storage / dtype names only, never a dotted name; and a scalar field that arrives as a placeholder (the
    reference's Trainer may pickle numpy scalars for epoch / metrics) is refused by name, not as a TypeError inside int()."""
    import io
    import pickle

    import numpy as np
    import pytest as _pytest
    import torch

    from super_gradients_amd.training.utils.checkpoint_utils import OpaqueObject, _InertUnpickler, plain_number, read_checkpoint

    def resolve(module, name):
        return _InertUnpickler(io.BytesIO(b"")).find_class(module, name)

    assert resolve("torch", "FloatStorage") is torch.FloatStorage and resolve("torch", "float32") is torch.float32
    for module, name in (("torch", "nn.Module.load_state_dict"), ("torch", "serialization.load.EvilStorage"), ("torch", "MadeUpStorage"),
                         ("os", "system"), ("torch", "hub.load")):
        cls = resolve(module, name)
        assert isinstance(cls, type) and issubclass(cls, OpaqueObject) and cls.pickled_name == f"{module}.{name}"
    # a numpy scalar where a number is expected: weights_only refuses the file, the inert reading makes it a placeholder, the field is refused by name
    path = str(tmp_path / "np_scalar.pth")
    torch.save({"net": {}, "epoch": np.int64(7), "acc": 0.5}, path, pickle_protocol=pickle.DEFAULT_PROTOCOL)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ckpt = read_checkpoint(path)
    if isinstance(ckpt["epoch"], OpaqueObject) or not isinstance(ckpt["epoch"], (int, np.integer)):
        with _pytest.raises(ValueError, match="epoch"):
            plain_number(ckpt, "epoch", int)
    else:  # (a torch that admits numpy scalars under weights_only=True)
        assert plain_number(ckpt, "epoch", int) == 7
    assert plain_number(ckpt, "acc", float) == 0.5 and plain_number(ckpt, "missing", int, -1) == -1
    assert plain_number({"steps": torch.tensor(4)}, "steps", int) == 4
