"""nvblox_torch.projective_integrator_types (import site: mindmap/mapping/helpers/nvblox_mapping_helpers.py:20,74)."""
from enum import Enum


class ProjectiveIntegratorType(Enum):
    TSDF = 0
    OCCUPANCY = 1  # accepted for API parity; only TSDF is implemented


class WeightingFunctionType(Enum):
    kConstantWeight = 0
    kInverseSquareWeight = 1
