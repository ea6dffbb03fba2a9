"""Registries and build helpers (reference: alphapose/models/builder.py:4-42)."""
import importlib

from torch import nn

from alphapose.utils import Registry, build_from_cfg, retrieve_from_cfg

SPPE = Registry("sppe")
LOSS = Registry("loss")
DATASET = Registry("dataset")


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_sppe(cfg, preset_cfg, **kwargs):
    return build(cfg, SPPE, default_args={"PRESET": preset_cfg, **kwargs})


def build_loss(cfg):
    return build(cfg, LOSS)


def _import_dataset(type_name):
    # datasets are host-side I/O outside the hot path; they
    # register themselves on import when a datasets package is installed beside us.
    try:
        importlib.import_module("alphapose.datasets")
    except ImportError:
        pass
    if DATASET.get(type_name) is None:
        raise KeyError(f"{type_name} is not in the {DATASET.name} registry")


def build_dataset(cfg, preset_cfg, **kwargs):
    _import_dataset(cfg["TYPE"])
    return build(cfg, DATASET, default_args={"PRESET": preset_cfg, **kwargs})


def retrieve_dataset(cfg):
    _import_dataset(cfg["TYPE"])
    return retrieve_from_cfg(cfg, DATASET)
