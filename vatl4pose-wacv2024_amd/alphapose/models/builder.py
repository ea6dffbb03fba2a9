"""The three registries of the plugin surface and the constructors the reference's drivers call on them
(alphapose/models/builder.py:4-42): ``build_sppe(cfg.MODEL, preset_cfg=cfg.DATA_PRESET)``, ``build_loss(cfg.LOSS)``,
``build_dataset(cfg.DATASET.X, preset_cfg=..., train=..., get_prenext=...)`` and ``retrieve_dataset``.

A config node names its class in ``TYPE``; every other key becomes a constructor keyword, joined by the caller's
extras (``PRESET`` = the data preset).  A list of nodes yields an ``nn.Sequential`` of the built objects.
"""
import importlib

from torch import nn

from alphapose.utils import Registry, build_from_cfg, retrieve_from_cfg

SPPE, LOSS, DATASET = Registry("sppe"), Registry("loss"), Registry("dataset")


def build(cfg, registry, default_args=None):
    if not isinstance(cfg, list):
        return build_from_cfg(cfg, registry, default_args)
    return nn.Sequential(*(build_from_cfg(node, registry, default_args) for node in cfg))


def _with_preset(preset_cfg, extras):
    merged = dict(extras)
    merged["PRESET"] = preset_cfg
    return merged


def _ensure_dataset_registered(type_name):
    """Dataset classes register themselves when ``alphapose.datasets`` is imported; the COCO-json datasets of the
    reference are host-side I/O outside the MI355X path and may be installed beside this package."""
    try:
        importlib.import_module("alphapose.datasets")
    except ImportError:
        pass
    if DATASET.get(type_name) is None:
        raise KeyError(f"{type_name} is not in the {DATASET.name} registry")


def build_loss(cfg):
    return build(cfg, LOSS)


def build_sppe(cfg, preset_cfg, **kwargs):
    return build(cfg, SPPE, default_args=_with_preset(preset_cfg, kwargs))


def build_dataset(cfg, preset_cfg, **kwargs):
    _ensure_dataset_registered(cfg["TYPE"])
    return build(cfg, DATASET, default_args=_with_preset(preset_cfg, kwargs))


def retrieve_dataset(cfg):
    _ensure_dataset_registered(cfg["TYPE"])
    return retrieve_from_cfg(cfg, DATASET)
