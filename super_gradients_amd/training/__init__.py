from . import losses, models  # noqa: F401
