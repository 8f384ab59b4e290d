from .detection_metrics import (DetectionMetrics, DetectionMetrics_050, DetectionMetrics_050_095, DetectionMetrics_075, IouThreshold,  # noqa: F401
                                compute_detection_metrics)
