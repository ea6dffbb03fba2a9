from .fastpose import FastPose
from .hrnet import PoseHighResolutionNet
from .simplepose import SimplePose
from . import criterion  # noqa: F401  (registers MSELoss)

__all__ = ["FastPose", "SimplePose", "PoseHighResolutionNet"]
