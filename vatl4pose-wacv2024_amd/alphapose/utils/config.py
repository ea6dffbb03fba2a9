"""yaml -> attribute dict (reference: alphapose/utils/config.py:5-8 uses easydict)."""
import yaml

try:                                    # the reference's dependency, when present
    from easydict import EasyDict as edict
except ImportError:                     # same access pattern, no dependency
    class edict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in {**(d or {}), **kw}.items():
                self[k] = v

        def __setitem__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, edict):
                v = edict(v)
            elif isinstance(v, (list, tuple)):
                v = type(v)(edict(e) if isinstance(e, dict) and not isinstance(e, edict) else e for e in v)
            super().__setitem__(k, v)

        __setattr__ = __setitem__

        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e


def update_config(config_file):
    with open(config_file) as f:
        return edict(yaml.load(f, Loader=yaml.FullLoader))
