"""Dataset registry entries shipped with the MI355X build.

``Posetrack21`` / ``JRDB2022`` (the two video datasets of the active-learning loop) read the reference's COCO-format
annotation json and decode frames on the host, then crop on the device (coco_video.py).  ``Mscoco`` / ``Mpii`` (image-only pre-training
datasets) share the same loader and yield ``CustomDataset``'s 5-tuple.  ``SyntheticVideo`` produces
the same 11-tuple item contract (posetrack21.py:181,205) from a seed, for tests, smoke and benchmarks; ``FrameVideo``
produces it from decoded uint8 frames + annotations with the crops made on the device (SimpleTransform on MI355X).
"""
from .coco_video import JRDB2022, Mpii, Mscoco, Posetrack21
from .frame_video import FrameVideo
from .synthetic import SyntheticVideo

__all__ = ["FrameVideo", "JRDB2022", "Mpii", "Mscoco", "Posetrack21", "SyntheticVideo"]
