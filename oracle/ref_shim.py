"""TEST INFRASTRUCTURE ONLY - never imported by the product package.

Import shim that lets the *reference's own source files* for the train-step hot path
(QARepVGGBlock, YOLO-NAS stages/neck/heads, ResNet, PPYoloELoss, PPYoloEPostPredictionCallback)
execute unmodified on CPU PyTorch in this container, although `import super_gradients` itself
fails here (omegaconf / hydra / torchvision / torchmetrics / cv2 ... are not installed and there
is no network).  Mechanism:

  1. a meta-path finder that manufactures stub modules for the missing third-party packages;
  2. a meta-path finder that turns every *directory* under /root/reference/src/super_gradients
     into a lazy package whose __init__.py is parsed (ast) instead of executed, so only the
     hot-path .py files are really imported - and those run byte-for-byte as the reference wrote
     them.

It is used ONLY (a) by oracle/make_golden.py to generate tests/golden/* fixtures and (b) by the
`not gpu` tests that pin oracle/* (our CPU restatement) against the real reference when
/root/reference is present.  /root/reference does not exist on the GPU box; nothing in the
`-m gpu` tests, smoke() or bench.py touches this file.
"""
import ast
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import tempfile
import types

REF_ROOT = os.environ.get("SG_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")

_MISSING = {
    "omegaconf", "hydra", "torchvision", "torchmetrics", "cv2", "onnx", "onnxruntime", "albumentations",
    "data_gradients", "pycocotools", "tensorboard", "treelib", "termcolor", "stringcase", "rapidfuzz",
    "json_tricks", "deprecated", "boto3", "botocore", "jsonschema", "imagesize", "onnxsim",
    "onnx_graphsurgeon", "pytorch_quantization", "deci_platform_client", "deci_lab_client", "wandb",
    "clearml", "dagshub", "mlflow", "coverage", "pip_tools", "piptools", "matplotlib", "PIL", "scipy_stub_never",
}
_FULL = {"torch.utils.tensorboard"}


def available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "super_gradients"))


class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        sub = _Meta(name, (_StubBase,), {})
        setattr(cls, name, sub)
        return sub

    def __call__(cls, *a, **k):
        if cls.__dict__.get("_is_stub_leaf", False) and len(a) == 1 and callable(a[0]) and not k:
            return a[0]  # stub used as a decorator
        return super().__call__(*a, **k)


class _StubBase(metaclass=_Meta):
    def __init__(self, *a, **k):
        pass

    def __init_subclass__(cls, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Meta(name, (_StubBase,), {"_is_stub_leaf": True})

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return self

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        cls = _Meta(name, (_StubBase,), {"__module__": self.__name__, "__class_getitem__": classmethod(lambda c, i: c), "_is_stub_leaf": True})
        setattr(self, name, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in _MISSING or fullname in _FULL or any(fullname.startswith(f + ".") for f in _FULL):
            # only stub what is really absent
            if top in _MISSING:
                try:
                    for finder in sys.meta_path:
                        if finder is self or isinstance(finder, _LazyFinder):
                            continue
                        spec = finder.find_spec(top, None) if hasattr(finder, "find_spec") else None
                        if spec is not None:
                            return None
                except Exception:
                    pass
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _LazyPackage(types.ModuleType):
    def _build_map(self):
        m = {}
        init = os.path.join(self.__path__[0], "__init__.py")
        try:
            tree = ast.parse(open(init).read())
        except Exception:
            tree = ast.Module(body=[], type_ignores=[])
        for node in tree.body:
            if isinstance(node, ast.ImportFrom) and node.module:
                mod = node.module if node.level == 0 else self.__name__ + "." + node.module
                for a in node.names:
                    m[a.asname or a.name] = (mod, a.name)
            elif isinstance(node, ast.ImportFrom) and node.level == 1 and node.module is None:
                for a in node.names:
                    m[a.asname or a.name] = (self.__name__ + "." + a.name, None)
        self.__dict__["_lazy_map"] = m

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        if "_lazy_map" not in self.__dict__:
            self._build_map()
        if name in self._lazy_map:
            mod, attr = self._lazy_map[name]
            module = importlib.import_module(mod)
            val = module if attr is None else getattr(module, attr)
            setattr(self, name, val)
            return val
        if not self.__dict__.get("_lazy_execd"):
            self.__dict__["_lazy_execd"] = True
            init = os.path.join(self.__path__[0], "__init__.py")
            if os.path.exists(init):
                exec(compile(open(init).read(), init, "exec"), self.__dict__)
                if name in self.__dict__:
                    return self.__dict__[name]
        try:
            return importlib.import_module(self.__name__ + "." + name)
        except ImportError:
            raise AttributeError(f"{self.__name__}.{name}")


class _LazyFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname != "super_gradients" and not fullname.startswith("super_gradients."):
            return None
        d = os.path.join(REF_SRC, *fullname.split("."))
        if os.path.isdir(d):
            spec = importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            spec.submodule_search_locations = [d]
            return spec
        return None

    def create_module(self, spec):
        m = _LazyPackage(spec.name)
        m.__path__ = list(spec.submodule_search_locations)
        return m

    def exec_module(self, module):
        pass


_installed = False


def install():
    """Install both finders (idempotent).  Raises if the reference tree is absent."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError(f"reference tree not found under {REF_SRC}; the shim only works in the build container")
    os.environ.setdefault("SUPER_GRADIENTS_LOG_DIR", tempfile.mkdtemp(prefix="sg_logs_"))
    sys.meta_path.insert(0, _StubFinder())
    sys.meta_path.insert(0, _LazyFinder())
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    _installed = True


def _register_detection_modules():
    import super_gradients.modules.detection_modules  # noqa: F401  NStageBackbone
    import super_gradients.training.models.detection_models.csp_darknet53  # noqa: F401  SPP
    import super_gradients.training.models.detection_models.yolo_nas.yolo_stages  # noqa: F401
    import super_gradients.training.models.detection_models.yolo_nas.dfl_heads  # noqa: F401
    import super_gradients.training.models.detection_models.yolo_nas.panneck  # noqa: F401


def load_arch_yaml(variant: str) -> dict:
    import yaml

    p = os.path.join(REF_SRC, "super_gradients", "recipes", "arch_params", f"yolo_nas_{variant}_arch_params.yaml")
    cfg = yaml.safe_load(open(p))
    cfg.pop("_convert_", None)
    return cfg


def build_reference_yolo_nas(variant: str = "s", num_classes: int = 80, in_channels: int = 3):
    """Builds the reference's own YoloNAS (customizable_detector.py:30, yolo_nas_variants.py:75) bypassing hydra."""
    install()
    _register_detection_modules()
    from super_gradients.training.models.detection_models.yolo_nas.yolo_nas_variants import YoloNAS

    cfg = load_arch_yaml(variant)
    return YoloNAS(
        backbone=cfg["backbone"],
        neck=cfg["neck"],
        heads=cfg["heads"],
        num_classes=num_classes,
        bn_eps=float(cfg["bn_eps"]),
        bn_momentum=float(cfg["bn_momentum"]),
        inplace_act=cfg["inplace_act"],
        in_channels=in_channels,
    )


def build_reference_ppyoloe(variant: str = "s", num_classes: int = 80):
    """The reference's own PPYoloE (pp_yolo_e/pp_yolo_e.py:95-111) from its arch YAMLs (recipes/arch_params/ppyoloe_*.yaml), bypassing hydra."""
    import yaml

    install()
    from super_gradients.training.models.detection_models.pp_yolo_e.pp_yolo_e import PPYoloE

    d = os.path.join(REF_SRC, "super_gradients", "recipes", "arch_params")
    with open(os.path.join(d, "ppyoloe_arch_params.yaml")) as f:
        cfg = yaml.safe_load(f)
    with open(os.path.join(d, f"ppyoloe_{variant}_arch_params.yaml")) as f:
        var = yaml.safe_load(f)
    cfg["depth_mult"], cfg["width_mult"] = var["depth_mult"], var["width_mult"]
    cfg["num_classes"] = num_classes
    cfg["backbone"]["pretrained_weights"] = None   # no network here
    return PPYoloE(cfg)


def reference_ppyolo_loss(**kw):
    install()
    from super_gradients.training.losses.ppyolo_loss import PPYoloELoss

    return PPYoloELoss(**kw)


def reference_loss_module():
    install()
    import super_gradients.training.losses.ppyolo_loss as m

    return m


def reference_resnet(name: str, num_classes: int):
    install()
    import super_gradients.training.models.classification_models.resnet as r
    from super_gradients.training.utils.utils import HpmStruct

    return getattr(r, name)(arch_params=HpmStruct(num_classes=num_classes))


def reference_post_prediction_callback(nms_fn, batched_nms_fn, **kw):
    """The reference's PPYoloEPostPredictionCallback with torchvision.ops.boxes.{nms,batched_nms}
    (absent here) bound to the given restatements (oracle/nms.py)."""
    install()
    import torchvision  # the stub

    torchvision.ops.boxes.nms = staticmethod(nms_fn)
    torchvision.ops.boxes.batched_nms = staticmethod(batched_nms_fn)
    from super_gradients.training.models.detection_models.pp_yolo_e.post_prediction_callback import PPYoloEPostPredictionCallback

    return PPYoloEPostPredictionCallback(**kw)
