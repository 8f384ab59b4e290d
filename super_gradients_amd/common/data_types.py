"""Enumerations of the reference's public API that the hot path's call signatures accept (reference: common/data_types/enum/*.py)."""
from enum import Enum


class StrictLoad(Enum):
    """`strict_load=` of models.get() / checkpoint loading (common/data_types/enum/strict_load.py:4-23): torch's own on / off, or one of the two
    adaptive strategies (training/models/model_factory.py::adaptive_load_state_dict here)."""

    OFF = False
    ON = True
    NO_KEY_MATCHING = "no_key_matching"
    KEY_MATCHING = "key_matching"
