"""Hot-path subset of the reference's ``active_learning`` package on MI355X.

``AnnoyTransformer`` / ``ActiveLearning`` are resolved lazily so that a missing
optional dependency cannot kill the import.
"""
from .version import __version__

__all__ = ["ActiveLearning", "AnnoyTransformer", "__version__"]


def __getattr__(name):
    if name == "ActiveLearning":
        from .ActiveLearning import ActiveLearning
        return ActiveLearning
    if name == "AnnoyTransformer":
        raise ImportError("AnnoyTransformer (annoy-based kNN) is outside the hot path and not provided")
    raise AttributeError(name)
