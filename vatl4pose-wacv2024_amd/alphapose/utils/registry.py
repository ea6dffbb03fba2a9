"""Name -> class registries behind ``builder.SPPE / LOSS / DATASET``.

Same surface and error behaviour as the reference's alphapose/utils/registry.py
(:4-43 Registry, :46-71 build_from_cfg, :74-98 retrieve_from_cfg): classes are
registered under ``cls.__name__``; a config dict's ``TYPE`` selects the class and
the remaining (upper-case) keys become constructor kwargs.
"""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f"{type(self).__name__}(name={self._name}, items={list(self._module_dict)})"

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f"module must be a class, but got {type(cls)}")
        if cls.__name__ in self._module_dict:
            raise KeyError(f"{cls.__name__} is already registered in {self._name}")
        self._module_dict[cls.__name__] = cls
        return cls


def _resolve(cfg, registry):
    assert isinstance(cfg, dict) and "TYPE" in cfg
    kwargs = dict(cfg)
    kind = kwargs.pop("TYPE")
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f"{kind} is not in the {registry.name} registry")
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f"type must be a str or valid type, but got {type(kind)}")
    return cls, kwargs


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(default_args, dict) or default_args is None
    cls, kwargs = _resolve(cfg, registry)
    for k, v in (default_args or {}).items():
        kwargs.setdefault(k, v)
    return cls(**kwargs)


def retrieve_from_cfg(cfg, registry):
    return _resolve(cfg, registry)[0]
