"""Shared helpers of the parity tests: run the same synthetic stream through the HIP Mapper and the
CPU oracle and compare.  (The oracle is only ever the checker here.)"""
import numpy as np

from nvblox_mindmap_amd import synthetic as S

REF_PARAMS = dict(  # what get_nvblox_mapper sets for DRILL_IN_BOX (nvblox_mapping_helpers.py:40-70)
    voxel_size=0.01,
    max_integration_distance_m=5.0,
    raycast_subsampling=1,
    workspace_bounds_type=2,
    ws_min=S.DRILL_IN_BOX_AABB_MIN,
    ws_max=S.DRILL_IN_BOX_AABB_MAX,
    tsdf_decay_factor=0.98,
    appearance_measurement_weight=1.0,
)


def make_oracle(O, channels, **over):
    kw = dict(REF_PARAMS)
    kw.update(over)
    kw.pop("num_preallocated_blocks", None)  # the oracle's block store grows on demand: pool sizing is a device-side matter
    return O.OracleMapper(O.default_params(feature_channels=channels, **kw))


def make_mapper(channels, **over):
    """HIP Mapper configured like the oracle (same parameter names as oracle.OrcParams)."""
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import (
        BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams)

    kw = dict(REF_PARAMS)
    kw.update(over)
    pi = ProjectiveIntegratorParams()
    pi.projective_integrator_max_integration_distance_m = kw["max_integration_distance_m"]
    pi.projective_appearance_integrator_measurement_weight = kw["appearance_measurement_weight"]
    if "truncation_distance_vox" in kw:
        pi.projective_integrator_truncation_distance_vox = kw["truncation_distance_vox"]
    if "weighting_mode" in kw:
        pi.projective_integrator_weighting_mode = ["kConstantWeight", "kInverseSquareWeight"][kw["weighting_mode"]]
    if "max_weight" in kw:
        pi.projective_integrator_max_weight = kw["max_weight"]
    if "st_subsampling" in kw:
        pi.projective_appearance_integrator_sphere_tracing_ray_subsampling_factor = kw["st_subsampling"]
    de = TsdfDecayIntegratorParams()
    de.tsdf_decay_factor = kw["tsdf_decay_factor"]
    if "decayed_weight_threshold" in kw:
        de.tsdf_decayed_weight_threshold = kw["decayed_weight_threshold"]
    vc = ViewCalculatorParams()
    vc.raycast_subsampling_factor = kw["raycast_subsampling"]
    vc.workspace_bounds_type = ["kUnbounded", "kHeightBounds", "kBoundingBox"][kw["workspace_bounds_type"]]
    vc.workspace_bounds_min_corner_x_m = float(kw["ws_min"][0])
    vc.workspace_bounds_min_corner_y_m = float(kw["ws_min"][1])
    vc.workspace_bounds_min_height_m = float(kw["ws_min"][2])
    vc.workspace_bounds_max_corner_x_m = float(kw["ws_max"][0])
    vc.workspace_bounds_max_corner_y_m = float(kw["ws_max"][1])
    vc.workspace_bounds_max_height_m = float(kw["ws_max"][2])
    mp = MapperParams()
    mp.set_projective_integrator_params(pi)
    mp.set_tsdf_decay_integrator_params(de)
    mp.set_view_calculator_params(vc)
    if "num_preallocated_blocks" in kw:
        pool = BlockMemoryPoolParams()
        pool.num_preallocated_blocks = int(kw["num_preallocated_blocks"])
        mp.set_block_memory_pool_params(pool)
    return Mapper(voxel_sizes_m=kw["voxel_size"], mapper_parameters=mp, feature_channels=channels)


def small_cfg(scale=4):
    """640x480 stream scaled down by `scale` (same field of view)."""
    return S.StreamConfig(width=640 // scale, height=480 // scale, fx=525.0 / scale, fy=525.0 / scale,
                          cx=(320.0 / scale) - 0.5, cy=(240.0 / scale) - 0.5)


def sort_rows(idx):
    """Order that sorts an [n,3] int array lexicographically."""
    idx = np.asarray(idx)
    return np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))


def frame_masks(depth, index):
    """A deterministic integration mask with a hole (exercises the mask paths)."""
    m = np.ones(depth.shape, dtype=np.uint8)
    h, w = depth.shape
    m[h // 3: h // 3 + h // 8, (w // 4 + 3 * index) % (w - w // 8): (w // 4 + 3 * index) % (w - w // 8) + w // 8] = 0
    return m
