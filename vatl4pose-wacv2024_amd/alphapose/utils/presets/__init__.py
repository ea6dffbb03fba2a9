"""Input presets (reference: alphapose/utils/presets/__init__.py) — only the ``simple`` preset feeds the pose path."""
from .simple_transform import SimpleTransform

__all__ = ["SimpleTransform"]
