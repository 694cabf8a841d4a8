"""Per-task mapping configuration (mirror of mindmap/mapping/nvblox_mapper_constants.py:20-169).

The numeric values are the reference's task constants; ``NvbloxMappingCfg`` takes the task name directly
instead of a ``Tap`` argument object (typed-argument-parser is not part of the hot path).
"""
from dataclasses import dataclass, field
from enum import Enum
from typing import Dict, List, Optional, Tuple

import torch

DEPTH_SCALE_FACTOR = 1000.0
CAMERA_NAME_TO_ID = {"table_rgb": 0, "wrist_rgb": 1}


class MAPPER_TO_ID(int, Enum):  # noqa: N801
    STATIC = 0
    DYNAMIC = 1


COMMON_NVBLOX_MAPPER_CFG = {
    "projective_integrator_max_integration_distance_m": 5.0,
    "voxel_size_m": 0.01,
    "unobserved_value": 0.0,
    "required_tensor_shape_dict": {"x": 128, "y": 128, "z": 64},
    "upscaled_feature_image_size": (512, 512),
    "feature_mask_border_percent": 5,
    "static_mask_erosion_iterations": 17,
    "dynamic_mask_erosion_iterations": 3,
    "projective_appearance_integrator_measurement_weight": 1.0,
}

TASK_TO_NVBLOX_MAPPER_CFG = {
    "MUG_IN_DRAWER": {
        "tsdf_decay_factor": 0.999,
        "aabb_min_m": torch.tensor([-0.2, -0.8, -0.2]),
        "aabb_max_m": torch.tensor([0.9, 0.8, 1.0]),
        "min_integration_distance_m": 0.37,
        "use_dynamic_mask": True,
        "dynamic_class_labels": ["robot_arm"],
        "valid_depth_mask_erosion_iterations": 10,
    },
    "CUBE_STACKING": {
        "tsdf_decay_factor": 0.98,
        "aabb_min_m": torch.tensor([-0.25, -0.65, -0.07]),
        "aabb_max_m": torch.tensor([1.0, 0.62, 0.56]),
        "min_integration_distance_m": 0.10,
        "use_dynamic_mask": True,
        "dynamic_class_labels": ["robot_arm"],
        "valid_depth_mask_erosion_iterations": 20,
    },
    "DRILL_IN_BOX": {
        "tsdf_decay_factor": 0.98,
        "aabb_min_m": torch.tensor([-0.37, -0.75, -0.13]),
        "aabb_max_m": torch.tensor([0.95, 0.75, 0.65]),
        "min_integration_distance_m": 0.30,
        "use_dynamic_mask": True,
        "dynamic_class_labels": ["robot"],
        "valid_depth_mask_erosion_iterations": 20,
    },
    "STICK_IN_BIN": {
        "tsdf_decay_factor": 0.98,
        "aabb_min_m": torch.tensor([3.7, 1.5, 0.44]),
        "aabb_max_m": torch.tensor([5.5, 3.2, 1.25]),
        "min_integration_distance_m": 0.30,
        "use_dynamic_mask": True,
        "dynamic_class_labels": ["robot"],
        "valid_depth_mask_erosion_iterations": 20,
    },
}


def _task_name(task) -> str:
    name = getattr(task, "name", task)
    return str(name).upper()


def get_workspace_bounds(task) -> torch.Tensor:
    """Workspace bounds as a 2x3 tensor given a task (name or enum with .name)."""
    cfg = TASK_TO_NVBLOX_MAPPER_CFG[_task_name(task)]
    return torch.stack([cfg["aabb_min_m"], cfg["aabb_max_m"]])


@dataclass
class NvbloxMappingCfg:
    """Mapping parameters of a task (fields and meaning as in the reference dataclass, :91-169)."""

    task: str = "DRILL_IN_BOX"
    voxel_size_m_override: Optional[float] = None
    measurement_weight_override: Optional[float] = None

    projective_integrator_max_integration_distance_m: float = None
    tsdf_decay_factor: float = None
    voxel_size_m: float = None
    aabb_min_m: torch.Tensor = None
    aabb_max_m: torch.Tensor = None
    unobserved_value: float = None
    min_integration_distance_m: float = None
    use_dynamic_mask: bool = None
    dynamic_class_labels: List[str] = None
    required_tensor_shape_dict: Dict[str, int] = None
    upscaled_feature_image_size: Tuple[int, int] = None
    feature_mask_border_percent: int = None
    static_mask_erosion_iterations: int = None
    dynamic_mask_erosion_iterations: int = None
    valid_depth_mask_erosion_iterations: int = None
    projective_appearance_integrator_measurement_weight: float = None
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        name = _task_name(self.task)
        assert name in TASK_TO_NVBLOX_MAPPER_CFG, f"{name} is not a recognized task."
        self.task = name
        for k, v in COMMON_NVBLOX_MAPPER_CFG.items():
            setattr(self, k, v)
        for k, v in TASK_TO_NVBLOX_MAPPER_CFG[name].items():
            setattr(self, k, v)
        if self.voxel_size_m_override is not None:
            self.voxel_size_m = self.voxel_size_m_override
        if self.measurement_weight_override is not None:
            self.projective_appearance_integrator_measurement_weight = self.measurement_weight_override

    @property
    def aabb_min_host(self):
        """``aabb_min_m`` as three host floats (float32 values), for native calls that take the box by value."""
        return [float(x) for x in self.aabb_min_m.to(torch.float32).tolist()]

    @property
    def aabb_max_host(self):
        return [float(x) for x in self.aabb_max_m.to(torch.float32).tolist()]
