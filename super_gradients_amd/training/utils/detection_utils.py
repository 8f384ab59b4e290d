"""Post-prediction callback protocol (reference: training/utils/detection_utils.py:213-228)."""
from abc import ABC, abstractmethod

from torch import nn


class DetectionPostPredictionCallback(ABC, nn.Module):
    def __init__(self) -> None:
        super().__init__()

    @abstractmethod
    def forward(self, x, device: str = None):
        """
        :param x:       the output of your model
        :param device:  (deprecated in the reference) the device to move all output tensors into
        :return:        a list with length batch_size, each item a detections tensor [Ni, 6] = x1, y1, x2, y2, confidence, class
        """
        raise NotImplementedError
