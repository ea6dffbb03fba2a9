from .simplepose import SimplePose
from . import criterion  # noqa: F401  (registers MSELoss)

__all__ = ["SimplePose"]
