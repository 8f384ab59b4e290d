from . import losses, metrics, models  # noqa: F401
from .sg_trainer import Trainer  # noqa: F401
from ..common.data_types import StrictLoad  # noqa: F401,E402
