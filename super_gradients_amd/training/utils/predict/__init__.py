"""Prediction containers of predict() (reference: training/utils/predict/predictions.py:16-66, prediction_results.py:24-52,99-110,
386-400): numpy boxes / confidences / labels per image, and the image + prediction + class-name wrappers predict() returns.  Drawing /
show / save are visualisation built on cv2 in the reference and are outside the hot path: they raise NotImplementedError here."""
from abc import ABC
from dataclasses import dataclass
from typing import Iterator, List, Tuple

import numpy as np


@dataclass
class Prediction(ABC):
    pass


@dataclass
class DetectionPrediction(Prediction):
    """Boxes in xyxy pixels.  `bbox_format` other than "xyxy" is converted on construction ("xywh", "cxcywh")."""

    bboxes_xyxy: np.ndarray
    confidence: np.ndarray
    labels: np.ndarray

    def __init__(self, bboxes: np.ndarray, bbox_format: str, confidence: np.ndarray, labels: np.ndarray, image_shape: Tuple[int, int]):
        if not (bboxes.shape[0] == confidence.shape[0] == labels.shape[0]):
            raise ValueError(f"The number of bounding boxes ({bboxes.shape[0]}) does not match the number of confidence scores "
                             f"({confidence.shape[0]}) and labels ({labels.shape[0]}).")
        fmt = str(bbox_format).lower()
        if fmt == "xyxy":
            xyxy = bboxes.copy()
        elif fmt == "xywh":
            xyxy = np.concatenate([bboxes[:, :2], bboxes[:, :2] + bboxes[:, 2:4]], axis=1)
        elif fmt == "cxcywh":
            x1y1 = bboxes[:, :2] - 0.5 * bboxes[:, 2:4]  # data_formats/bbox_formats/cxcywh.py:36-56: x2 = x1 + w
            xyxy = np.concatenate([x1y1, x1y1 + bboxes[:, 2:4]], axis=1)
        else:
            raise NotImplementedError(f"DetectionPrediction: bbox_format {bbox_format!r} (xyxy, xywh, cxcywh are covered)")
        self.bboxes_xyxy, self.confidence, self.labels, self.image_shape = xyxy, confidence, labels, image_shape

    def __len__(self):
        return len(self.bboxes_xyxy)


def _no_visualisation(*_a, **_k):
    raise NotImplementedError("drawing / showing / saving predictions is cv2 visualisation, outside the MI355X hot path; use .prediction")


@dataclass
class ImagePrediction(ABC):
    image: np.ndarray
    prediction: Prediction
    class_names: List[str]

    draw = show = save = _no_visualisation


@dataclass
class ImageDetectionPrediction(ImagePrediction):
    image: np.ndarray
    prediction: DetectionPrediction
    class_names: List[str]


@dataclass
class ImagesPredictions(ABC):
    _images_prediction_lst: List[ImagePrediction]

    def __len__(self) -> int:
        return len(self._images_prediction_lst)

    def __getitem__(self, index: int) -> ImagePrediction:
        return self._images_prediction_lst[index]

    def __iter__(self) -> Iterator[ImagePrediction]:
        return iter(self._images_prediction_lst)

    show = save = _no_visualisation


@dataclass
class ImagesDetectionPrediction(ImagesPredictions):
    _images_prediction_lst: List[ImageDetectionPrediction]
