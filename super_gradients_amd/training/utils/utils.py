"""HpmStruct / get_param (reference: training/utils/utils.py:49-81, 212-235): the parameter containers recipes pass around."""
import copy
from collections.abc import Mapping


def recursive_override(base: dict, extension: dict):
    for k, v in extension.items():
        if k in base and isinstance(v, Mapping) and isinstance(base[k], Mapping):
            base[k] = dict(base[k])
            recursive_override(base[k], v)
        else:
            base[k] = extension[k]


class HpmStruct:
    def __init__(self, **entries):
        self.__dict__.update(entries)
        self.schema = None

    def set_schema(self, schema: dict):
        self.schema = schema

    def override(self, **entries):
        recursive_override(self.__dict__, entries)

    def to_dict(self, include_schema=True) -> dict:
        out = self.__dict__.copy()
        if not include_schema:
            out.pop("schema")
        return out


def get_param(params, name, default_val=None):
    """Value of `name` from a dict / HpmStruct, `default_val` if absent (dict defaults are merged, utils.py:212-235)."""
    if isinstance(params, Mapping):
        if name in params:
            v = params[name]
            if isinstance(v, Mapping) and isinstance(default_val, Mapping) and default_val:
                merged = copy.deepcopy(dict(default_val))
                recursive_override(merged, v)
                return merged
            return v
        return default_val
    if hasattr(params, name):
        return getattr(params, name)
    return default_val
