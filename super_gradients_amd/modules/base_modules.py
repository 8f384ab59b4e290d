"""Detection-module protocol (reference: modules/base_modules.py:9-27): constructor takes `in_channels`, the
module reports `out_channels` so that the next module in the YAML chain can be sized."""
from typing import List, Union

from .engine import SgxBlock


class BaseDetectionModule(SgxBlock):
    def __init__(self, in_channels: Union[List[int], int], **kwargs):
        super().__init__()
        self.in_channels = in_channels

    @property
    def out_channels(self) -> Union[List[int], int]:
        raise NotImplementedError()

    def on_materialize(self):
        pass


def width_multiplier(original, factor, divisor: int = None):
    """modules/utils.py:63-74"""
    import math

    if divisor is None:
        return int(original * factor)
    return math.ceil(int(original * factor) / divisor) * divisor
