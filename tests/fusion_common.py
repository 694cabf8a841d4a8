"""Shared helpers of the parity tests: run the same synthetic stream through the HIP Mapper and the
CPU oracle and compare.  (The oracle is only ever the checker here.)"""
import os

import numpy as np

from nvblox_mindmap_amd import synthetic as S

# MMF_FMA_CONTRACTION=1: every mapper of the process (nvblox_torch/mapper_params.py) and every oracle built here carry the spec
# switch fma_contraction (mmf_params / orc_params) -- the parity, fuzz and soak suites then check the contracted arithmetic
FMA = os.environ.get("MMF_FMA_CONTRACTION", "0") == "1"
# MMF_SPEC_FLIPS=name[,name...]: the same for the spec switches block_index_by_division / view_truncation_band_marking /
# bilinear_four_weight_sum (mapper_params.SPEC_FLIP_DEFAULTS): every mapper of the process and every oracle built here carry them
SPEC_FLIPS = tuple(x.strip() for x in os.environ.get("MMF_SPEC_FLIPS", "").split(",") if x.strip())
# frames of such mappers take the un-merged / stand-alone launches and are not pipelined: tests that assert the default ROUTE skip
NOT_DEFAULT_ROUTE = FMA or bool(SPEC_FLIPS)
ROUTE_SKIP_REASON = "asserts a route (merged launch / deferred rows) that mappers under MMF_FMA_CONTRACTION / MMF_SPEC_FLIPS do not take"

REF_PARAMS = dict(  # what get_nvblox_mapper sets for DRILL_IN_BOX (nvblox_mapping_helpers.py:40-70)
    voxel_size=0.01,
    max_integration_distance_m=5.0,
    raycast_subsampling=1,
    workspace_bounds_type=2,
    ws_min=S.DRILL_IN_BOX_AABB_MIN,
    ws_max=S.DRILL_IN_BOX_AABB_MAX,
    tsdf_decay_factor=0.98,
    appearance_measurement_weight=1.0,
    **({"fma_contraction": 1} if FMA else {}),
    **{name: 1 for name in SPEC_FLIPS},
)


def make_oracle(O, channels, **over):
    kw = dict(REF_PARAMS)
    kw.update(over)
    kw.pop("num_preallocated_blocks", None)  # the oracle's block store grows on demand: pool sizing is a device-side matter
    return O.OracleMapper(O.default_params(feature_channels=channels, **kw))


def make_mapper(channels, **over):
    """HIP Mapper configured like the oracle: same parameter names as oracle.OrcParams, every one of them honoured
    (an unknown name raises -- a silently ignored override would make a spec flip a no-op)."""
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import (
        BlockMemoryPoolParams, MapperParams, MeshIntegratorParams, MmfWeightingMode, ProjectiveIntegratorParams, TsdfDecayIntegratorParams,
        ViewCalculatorParams)

    kw = dict(REF_PARAMS)
    kw.update(over)
    pi, de, vc, me, pool = (ProjectiveIntegratorParams(), TsdfDecayIntegratorParams(), ViewCalculatorParams(), MeshIntegratorParams(),
                            BlockMemoryPoolParams())
    routes = {
        "max_integration_distance_m": (pi, "projective_integrator_max_integration_distance_m", float),
        "appearance_measurement_weight": (pi, "projective_appearance_integrator_measurement_weight", float),
        "appearance_max_weight": (pi, "projective_appearance_integrator_max_weight", float),
        "truncation_distance_vox": (pi, "projective_integrator_truncation_distance_vox", float),
        "weighting_mode": (pi, "projective_integrator_weighting_mode", MmfWeightingMode),  # (the oracle's numbering, said so explicitly)
        "max_weight": (pi, "projective_integrator_max_weight", float),
        "lin_interp_max_diff_vox": (pi, "projective_tsdf_integrator_linear_interpolation_max_allowable_difference_vox", float),
        "st_subsampling": (pi, "projective_appearance_integrator_sphere_tracing_ray_subsampling_factor", int),
        "st_max_steps": (pi, "projective_appearance_integrator_sphere_tracing_max_steps", int),
        "st_max_ray_length_m": (pi, "projective_appearance_integrator_sphere_tracing_max_ray_length_m", float),
        "st_surface_eps_vox": (pi, "projective_appearance_integrator_sphere_tracing_surface_epsilon_vox", float),
        "tsdf_decay_factor": (de, "tsdf_decay_factor", float),
        "decayed_weight_threshold": (de, "tsdf_decayed_weight_threshold", float),
        "deallocate_decayed_blocks": (de, "decay_integrator_deallocate_decayed_blocks", bool),
        "decay_appearance_layers": (de, "decay_appearance_layers", bool),
        "raycast_subsampling": (vc, "raycast_subsampling_factor", int),
        "raycast_to_truncation": (vc, "raycast_to_truncation_distance", bool),
        "raycast_walk_from_camera": (vc, "raycast_walk_from_camera", bool),
        "appearance_blend_division": (pi, "projective_appearance_integrator_blend_division", bool),
        "fma_contraction": (pi, "projective_integrator_fma_contraction", bool),
        "bilinear_four_weight_sum": (pi, "projective_integrator_bilinear_four_weight_sum", bool),
        "block_index_by_division": (vc, "block_index_by_division", bool),
        "view_truncation_band_marking": (vc, "view_truncation_band_marking", bool),
        "mesh_min_weight": (me, "mesh_integrator_min_weight", float),
        "num_preallocated_blocks": (pool, "num_preallocated_blocks", int),
    }
    for k, v in kw.items():
        if k in ("voxel_size", "ws_min", "ws_max", "workspace_bounds_type"):
            continue
        if k not in routes:
            raise KeyError(f"make_mapper: no route for parameter '{k}'")
        bag, name, cast = routes[k]
        setattr(bag, name, cast(v))
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
    mp.set_mesh_integrator_params(me)
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
