from .model_factory import get, get_model_name, instantiate_model  # noqa: F401
from .detection_models.yolo_nas import YoloNAS, YoloNAS_L, YoloNAS_M, YoloNAS_S  # noqa: F401
from .detection_models.customizable_detector import CustomizableDetector  # noqa: F401
from .classification_models import ResNet, ResNet18, ResNet18Cifar, ResNet34, ResNet50, CifarResNet  # noqa: F401
from .detection_models.pp_yolo_e.pp_yolo_e import PPYoloE, PPYoloE_L, PPYoloE_M, PPYoloE_S, PPYoloE_X  # noqa: F401
