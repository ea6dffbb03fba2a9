"""Hot-path subset of the reference's ``active_learning`` package on MI355X.

``ActiveLearning`` is imported eagerly like the reference does (active_learning/__init__.py:1);
``AnnoyTransformer`` (annoy-based kNN, never used by the loop: SURVEY.md §9 item 12) resolves lazily so a
missing optional dependency cannot kill the import.
"""
from .ActiveLearning import ActiveLearning
from .version import __version__

__all__ = ["ActiveLearning", "AnnoyTransformer", "__version__"]


def __getattr__(name):
    if name == "AnnoyTransformer":
        raise ImportError("AnnoyTransformer (annoy-based kNN) is outside the hot path and not provided")
    raise AttributeError(name)
