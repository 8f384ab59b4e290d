"""MI355X-native train-step of super-gradients' detection/classification hot path.

    from super_gradients_amd.training import models
    net = models.get("yolo_nas_s", num_classes=80)          # runs on libsgx_hip.so (gfx950), no CPU fallback
"""
from . import modules  # noqa: F401
from .training import losses, models  # noqa: F401

__version__ = "0.1.0"
