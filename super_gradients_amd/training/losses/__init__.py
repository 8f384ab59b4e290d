from .label_smoothing_cross_entropy_loss import CrossEntropyLoss  # noqa: F401
from .ppyolo_loss import PPYoloELoss  # noqa: F401
