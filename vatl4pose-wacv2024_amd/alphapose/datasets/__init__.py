"""Dataset registry entries shipped with the MI355X build.

The reference's COCO-json datasets (Posetrack21, JRDB2022, Mscoco, Mpii) are host-side I/O on files
that are not part of the hot path; they register themselves in
``alphapose.models.builder.DATASET`` when installed beside this package.  ``SyntheticVideo`` produces
the same 11-tuple item contract (posetrack21.py:181,205) from a seed, for tests, smoke and benchmarks; ``FrameVideo``
produces it from decoded uint8 frames + annotations with the crops made on the device (SimpleTransform on MI355X).
"""
from .frame_video import FrameVideo
from .synthetic import SyntheticVideo

__all__ = ["FrameVideo", "SyntheticVideo"]
