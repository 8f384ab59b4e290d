from . import dfl_heads, panneck, yolo_stages  # noqa: F401  (registers the detection modules)
from .yolo_nas_variants import YoloNAS, YoloNAS_L, YoloNAS_M, YoloNAS_S  # noqa: F401
