"""MI355X-native drop-in for the hot-path subset of the reference's ``alphapose`` package.

Only what ``scripts/Run_active_learning.py`` needs for pose inference,
uncertainty scoring and fine-tuning is provided; the arithmetic
runs in libvatl_hip.so (hand-written gfx950 kernels) through ``vatl_hip``.
"""
from .version import __version__, short_version

__all__ = ["__version__", "short_version"]
