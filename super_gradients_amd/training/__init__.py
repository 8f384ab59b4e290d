from . import losses, metrics, models  # noqa: F401
from .sg_trainer import Trainer  # noqa: F401
