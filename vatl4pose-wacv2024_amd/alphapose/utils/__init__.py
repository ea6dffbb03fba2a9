"""Registry helpers under the names the reference's packages import from ``alphapose.utils``
(the builder and the dataset modules do ``from alphapose.utils import Registry, build_from_cfg``)."""
from . import registry as _registry

Registry = _registry.Registry
build_from_cfg = _registry.build_from_cfg
retrieve_from_cfg = _registry.retrieve_from_cfg
