from .detection_collate_fn import DetectionCollateFN, DeviceDetectionCollateFN  # noqa: F401
