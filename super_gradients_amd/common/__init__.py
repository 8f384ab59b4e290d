
from .data_types import StrictLoad  # noqa: F401,E402
