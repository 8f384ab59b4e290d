from .processing import (  # noqa: F401
    AutoPadding, ComposeProcessing, ComposeProcessingMetadata, DetectionAutoPadding, DetectionBottomRightPadding, DetectionCenterPadding,
    DetectionLongestMaxSizeRescale, DetectionPadToSizeMetadata, DetectionRescale, ImagePermute, NormalizeImage, PaddingCoordinates, Processing,
    RescaleMetadata, ReverseImageChannels, StandardizeImage, default_ppyoloe_coco_processing_params, default_yolo_nas_coco_processing_params,
)
