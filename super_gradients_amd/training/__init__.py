from . import losses, models  # noqa: F401
from .sg_trainer import Trainer  # noqa: F401
