"""Drop-in for the ``nvblox_torch`` Python API surface the reference uses.

``import nvblox_mindmap_amd.nvblox_torch as nvblox_torch`` -- or call
``nvblox_mindmap_amd.install_as_nvblox_torch()`` so that the reference's own
``from nvblox_torch.mapper import Mapper`` statements resolve to this package.
"""
from . import constants, indexing, layer, mapper, mapper_params, projective_integrator_types, timer  # noqa: F401
