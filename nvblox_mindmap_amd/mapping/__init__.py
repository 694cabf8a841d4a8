from .nvblox_mapper_constants import (  # noqa: F401
    COMMON_NVBLOX_MAPPER_CFG, MAPPER_TO_ID, TASK_TO_NVBLOX_MAPPER_CFG, NvbloxMappingCfg, get_workspace_bounds)
