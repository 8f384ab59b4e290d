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
