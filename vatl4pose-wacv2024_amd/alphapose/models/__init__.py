"""Pose networks of the MI355X path.  Importing this package fills the ``SPPE`` and ``LOSS`` registries of
``alphapose.models.builder`` (the reference's ``build_sppe`` / ``build_loss`` look their classes up there)."""
import importlib as _importlib

# module -> class it registers; the HIP plans behind the forwards live in hip_engine / hip_train
_REGISTERED = {"simplepose": "SimplePose", "fastpose": "FastPose", "hrnet": "PoseHighResolutionNet", "criterion": "L1JointRegression"}
for _mod, _cls in _REGISTERED.items():
    globals()[_cls] = getattr(_importlib.import_module(f"{__name__}.{_mod}"), _cls)
del _mod, _cls
