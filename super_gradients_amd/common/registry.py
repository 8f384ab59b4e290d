"""Plugin registries of the MI355X build - the drop-in boundary the reference's recipes address by name.

Mirrors the public surface of common/registry/registry.py:14-196 for the hot path: decorator factories
`register_model / register_detection_module / register_loss / register_callback / register_optimizer /
register_lr_scheduler / register_metric / register_dataloader / register_processing(name=None, deprecated_name=None)` and the dictionaries they fill.
These registries are owned by this package (the reference raises when a different class is re-registered under an
existing name, registry.py:36-41, so sharing its dictionaries is not an option).
"""
import warnings
from typing import Callable, Dict, Optional

_DEPRECATED = "_deprecated_objects"


class Registry(dict):
    def register(self, name: Optional[str] = None, deprecated_name: Optional[str] = None) -> Callable:
        def deco(obj):
            for key in filter(None, (name or obj.__name__, deprecated_name)):
                prev = self.get(key)
                if prev is not None and prev is not obj:
                    raise Exception(f"`{key}` is already registered and points to `{prev.__module__}.{prev.__name__}`")
                self[key] = obj
            if deprecated_name:
                self.setdefault(_DEPRECATED, {})[deprecated_name] = name or obj.__name__
            return obj

        return deco


def warn_if_deprecated(name: str, registry: dict):
    new = registry.get(_DEPRECATED, {}).get(name)
    if new is not None:
        warnings.warn(f"Object name `{name}` is now deprecated. Please replace it with `{new}`.", DeprecationWarning)


ARCHITECTURES = Registry()
ALL_DETECTION_MODULES = Registry()
LOSSES = Registry()
CALLBACKS = Registry()
OPTIMIZERS = Registry()
LR_SCHEDULERS_CLS_DICT = Registry()
LR_WARMUP_CLS_DICT = Registry()
METRICS = Registry()
ALL_DATALOADERS = Registry()
PROCESSINGS = Registry()

register_model = ARCHITECTURES.register
register_detection_module = ALL_DETECTION_MODULES.register
register_loss = LOSSES.register
register_callback = CALLBACKS.register
register_optimizer = OPTIMIZERS.register
register_lr_scheduler = LR_SCHEDULERS_CLS_DICT.register
register_lr_warmup = LR_WARMUP_CLS_DICT.register
register_metric = METRICS.register
register_dataloader = ALL_DATALOADERS.register
register_processing = PROCESSINGS.register
