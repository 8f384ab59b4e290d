"""Name -> object factories (reference: common/factories/base_factory.py:37-72, detection_modules_factory.py:15-28,
activations_type_factory.py).  A configuration is a type name, a single-entry `{TypeName: {kwargs}}` mapping, or an
already-built object (returned unchanged)."""
import re
from typing import Mapping, Union

from .registry import ALL_DETECTION_MODULES, CALLBACKS, LOSSES, warn_if_deprecated


class UnknownTypeException(Exception):
    """Same name and meaning as common/exceptions/factory_exceptions.py:7."""

    def __init__(self, unknown_type: str, choices: list, message: str = None):
        self.message = message or f"Unknown object type: {unknown_type} in configuration. valid types are: \n{choices}"
        super().__init__(self.message)


def _fuzzy(s: str) -> str:
    return re.sub(r"[^a-z0-9]", "", s.lower())


class BaseFactory:
    def __init__(self, type_dict: Mapping[str, type]):
        self.type_dict = type_dict

    def _lookup(self, name: str):
        warn_if_deprecated(name, self.type_dict)
        if name in self.type_dict:
            return self.type_dict[name]
        fz = {_fuzzy(k): v for k, v in self.type_dict.items() if isinstance(k, str)}
        if _fuzzy(name) in fz:
            return fz[_fuzzy(name)]
        raise UnknownTypeException(name, [k for k in self.type_dict.keys() if not k.startswith("_")])

    def get(self, conf: Union[str, Mapping, object]):
        if isinstance(conf, str):
            return self._lookup(conf)()
        if isinstance(conf, Mapping):
            if len(conf) != 1:
                raise RuntimeError("Malformed object definition in configuration. Expecting either a string of object type or a single entry "
                                   f"dictionary {{type_name(str): {{parameters...}}}}. received: {conf}")
            (name, params), = conf.items()
            return self._lookup(name)(**(params or {}))
        return conf


class DetectionModulesFactory(BaseFactory):
    def __init__(self):
        super().__init__(ALL_DETECTION_MODULES)

    @staticmethod
    def insert_module_param(conf, name: str, value):
        """Adds/overrides a constructor argument inside a `{Type: {kwargs}}` config (detection_modules_factory.py:15-28)."""
        if isinstance(conf, str):
            return {conf: {name: value}}
        if isinstance(conf, Mapping):
            (k, v), = conf.items()
            v = dict(v or {})
            v[name] = value
            return {k: v}
        return conf


class ProcessingFactory(BaseFactory):
    """`{TypeName: {kwargs}}` (or a list of them, composed) -> Processing (common/factories/processing_factory.py:8-19)."""

    def __init__(self):
        from .registry import PROCESSINGS

        super().__init__(PROCESSINGS)

    def get(self, conf):
        from ..training.processing import processing as _p  # noqa: F401  (fills the registry)

        if isinstance(conf, (list, tuple)):
            return _p.ComposeProcessing([self.get(c) for c in conf])
        if isinstance(conf, Mapping) and len(conf) == 1 and "ComposeProcessing" in conf:
            kw = dict(conf["ComposeProcessing"])
            kw["processings"] = [self.get(c) for c in kw["processings"]]
            return _p.ComposeProcessing(**kw)
        return super().get(conf)


class LossesFactory(BaseFactory):
    def __init__(self):
        super().__init__(LOSSES)


class CallbacksFactory(BaseFactory):
    def __init__(self):
        super().__init__(CALLBACKS)


class MetricsFactory(BaseFactory):
    def __init__(self):
        from .registry import METRICS

        super().__init__(METRICS)


def _target_table():
    """Short class name -> class for hydra-style `_target_: dotted.path.ClassName` entries of the reference's recipes: the recipe names the
    reference's module path, the class of the same name registered here is what gets built."""
    from . import registry as R
    from ..training.models.detection_models.pp_yolo_e.post_prediction_callback import PPYoloEPostPredictionCallback

    table = {"PPYoloEPostPredictionCallback": PPYoloEPostPredictionCallback, "empty_list": list}
    for reg in (R.METRICS, R.LOSSES, R.CALLBACKS, R.ALL_DETECTION_MODULES, R.PROCESSINGS, R.LR_SCHEDULERS_CLS_DICT, R.LR_WARMUP_CLS_DICT):
        for k, v in reg.items():
            if isinstance(k, str) and not k.startswith("_") and isinstance(v, type):
                table.setdefault(v.__name__, v)
    return table


_EXP_FLOAT = re.compile(r"^[+-]?\d+(\.\d*)?[eE][+-]?\d+$")


def resolve_recipe_values(conf):
    """Recursively turn the plain containers of a recipe (yaml.safe_load of the reference's training_hyperparams files, or an
    already-resolved hydra config) into objects: `{_target_: a.b.Class, **kwargs}` -> Class(**kwargs) by short class name, and exponent-form
    numbers that YAML 1.1 loaders leave as strings (`2e-4`, `1e-6` - hydra/omegaconf parse them as floats) -> float.  Anything else is
    returned unchanged."""
    if isinstance(conf, Mapping):
        if "_target_" in conf:
            name = str(conf["_target_"]).rsplit(".", 1)[-1]
            table = _target_table()
            if name not in table:
                raise UnknownTypeException(name, sorted(table))
            kwargs = {k: resolve_recipe_values(v) for k, v in conf.items() if k not in ("_target_", "_convert_", "_recursive_", "_partial_")}
            return table[name](**kwargs)
        return {k: resolve_recipe_values(v) for k, v in conf.items()}
    if isinstance(conf, (list, tuple)):
        return type(conf)(resolve_recipe_values(v) for v in conf)
    if isinstance(conf, str) and _EXP_FLOAT.match(conf):
        return float(conf)
    return conf
