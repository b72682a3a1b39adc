"""Import-path alias: the reference keeps `Identity` under `pretorched.models` (`pretorched/models/__init__.py:79`,
`pretorched/models/utils.py:81`).  The model factories themselves live on the package, as upstream (`pretorched.__dict__`)."""
from . import utils  # noqa: F401
from .utils import Identity  # noqa: F401
