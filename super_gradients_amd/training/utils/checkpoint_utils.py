"""Reading checkpoint files without running what they carry.

`read_checkpoint(path)`: torch.load(weights_only=True) first - tensors, numbers, strings and plain containers.  The reference's Trainer
pickles live objects into its checkpoints (sg_trainer.py:710-712: the image processor inside "processing_params"), which that mode
refuses.  Such files are read a second time with an unpickler that builds tensors as torch does and turns EVERY other global it meets into
an inert placeholder (`OpaqueObject`): no foreign constructor, `__setstate__` or `__reduce__` target is ever called, the weights load,
and the caller sees which entries were objects (model_factory warns that the processing parameters have to be set by hand).
"""
import pickle
import warnings

import torch


class OpaqueObject:
    """Stands in for an object of a class this package does not construct from a file; keeps what the pickle stream said about it."""

    def __init__(self, *args, **kwargs):
        self.pickled_args, self.pickled_kwargs, self.pickled_state = args, kwargs, None

    def __setstate__(self, state):
        self.pickled_state = state

    def __repr__(self):
        return f"<OpaqueObject {getattr(type(self), 'pickled_name', '?')}>"


# what torch.save needs to rebuild tensors and plain containers (the same set weights_only=True admits for a state dict)
_ALLOWED = {
    ("collections", "OrderedDict"), ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"),
    ("torch", "Size"), ("torch", "device"), ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
}
# torch's typed storage classes and dtypes by NAME (protocol 4 resolves a dotted name by attribute traversal - "nn.Module.load_state_dict"
# under "torch" would be reachable through any suffix rule - so: exact names only, and a '.' in a name is never admitted)
_TORCH_STORAGES = {n for n in ("UntypedStorage", "TypedStorage", "DoubleStorage", "FloatStorage", "HalfStorage", "BFloat16Storage", "LongStorage",
                               "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage", "ComplexFloatStorage", "ComplexDoubleStorage")
                   if hasattr(torch, n)}
_TORCH_DTYPES = {n for n in dir(torch) if isinstance(getattr(torch, n), torch.dtype)}


class _InertUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if "." not in name and ((module, name) in _ALLOWED or (module == "torch" and (name in _TORCH_STORAGES or name in _TORCH_DTYPES))):
            return super().find_class(module, name)
        return type(name.rsplit(".", 1)[-1], (OpaqueObject,), {"pickled_name": f"{module}.{name}"})


class _InertPickle:
    """The `pickle_module` protocol torch.load expects."""

    __name__ = "inert_pickle"
    Unpickler = _InertUnpickler
    load = staticmethod(lambda f, **kw: _InertUnpickler(f, **kw).load())


def contains_opaque(obj, _depth=0) -> bool:
    if isinstance(obj, OpaqueObject):
        return True
    if _depth > 6:
        return False
    if isinstance(obj, dict):
        return any(contains_opaque(v, _depth + 1) for v in obj.values())
    if isinstance(obj, (list, tuple)):
        return any(contains_opaque(v, _depth + 1) for v in obj)
    return False


def plain_number(ckpt, key, kind=float, default=None):
    """ckpt[key] as a Python number.  A field the reading above turned into a placeholder (a numpy scalar pickled by the reference's Trainer,
    say) is refused HERE, by name - not as a TypeError somewhere inside int() later."""
    if key not in ckpt or ckpt[key] is None:
        return default
    v = ckpt[key]
    if contains_opaque(v):
        raise ValueError(f"checkpoint field {key!r} is a pickled object ({v!r}), not a number: this package does not construct objects from "
                         "checkpoint files - re-save the field as a plain int / float")
    if isinstance(v, torch.Tensor):
        v = v.item()
    return kind(v)


def read_checkpoint(path):
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except pickle.UnpicklingError as first:
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_InertPickle)
        except Exception:  # noqa: BLE001  (whatever the second reading trips over, the first refusal is the error to report)
            raise first
        warnings.warn(f"{path}: the file pickles objects (e.g. the reference Trainer's image processor); they were NOT constructed - tensors and "
                      "plain values are loaded, object entries are opaque placeholders")
        return ckpt
