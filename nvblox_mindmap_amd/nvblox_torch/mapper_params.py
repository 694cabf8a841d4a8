"""nvblox_torch.mapper_params: the parameter bags the reference fills in ``get_nvblox_mapper``
(mindmap/mapping/helpers/nvblox_mapping_helpers.py:13-19,40-70).

Plain attribute bags with the nvblox field names and nvblox defaults (recalled from upstream; see
DESIGN.md section 3 for the list and their confidence).  Unknown attribute names raise, like the
pybind classes upstream do.
"""
import os

from .. import _lib

# default of the spec switch fma_contraction for every mapper of the process (benchmark legs / whole test suites under the switch)
FMA_CONTRACTION_DEFAULT = os.environ.get("MMF_FMA_CONTRACTION", "0") == "1"
# the same for the three spec switches of round 6 (pin tooling: whole parity / fuzz suites under a flip):
# MMF_SPEC_FLIPS=block_index_by_division,view_truncation_band_marking,bilinear_four_weight_sum (any subset)
_SPEC_FLIP_NAMES = ("block_index_by_division", "view_truncation_band_marking", "bilinear_four_weight_sum")
SPEC_FLIP_DEFAULTS = frozenset(x.strip() for x in os.environ.get("MMF_SPEC_FLIPS", "").split(",") if x.strip())
if SPEC_FLIP_DEFAULTS - set(_SPEC_FLIP_NAMES):
    raise ValueError(f"MMF_SPEC_FLIPS: unknown spec switch(es) {sorted(SPEC_FLIP_DEFAULTS - set(_SPEC_FLIP_NAMES))}; known: {_SPEC_FLIP_NAMES}")


class _Bag:
    _fields = {}

    def __init__(self):
        for k, v in self._fields.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, k, v):
        if k not in self._fields:
            raise AttributeError(f"{type(self).__name__} has no parameter '{k}'")
        object.__setattr__(self, k, v)

    def __repr__(self):
        return f"{type(self).__name__}({', '.join(f'{k}={getattr(self, k)!r}' for k in self._fields)})"


class ProjectiveIntegratorParams(_Bag):
    _fields = {
        "projective_integrator_max_integration_distance_m": 7.0,
        "lidar_projective_integrator_max_integration_distance_m": 10.0,  # accepted, unused (no lidar path)
        "projective_integrator_truncation_distance_vox": 4.0,
        "projective_integrator_weighting_mode": "kInverseSquareWeight",
        "projective_integrator_max_weight": 5.0,
        "projective_tsdf_integrator_invalid_depth_decay_factor": -1.0,  # accepted, unused
        "projective_tsdf_integrator_linear_interpolation_max_allowable_difference_vox": 2.0,
        "projective_appearance_integrator_measurement_weight": 1.0,
        "projective_appearance_integrator_max_weight": 5.0,
        "projective_appearance_integrator_sphere_tracing_ray_subsampling_factor": 4,
        # the sphere tracer's remaining constants (upstream: members of its SphereTracer, not exposed as parameters)
        "projective_appearance_integrator_sphere_tracing_max_steps": 100,
        "projective_appearance_integrator_sphere_tracing_max_ray_length_m": 15.0,
        "projective_appearance_integrator_sphere_tracing_surface_epsilon_vox": 0.1,
        # spec switch (not an upstream field): (A W + a w) / (W + w) per channel (True) or one reciprocal per voxel (False)
        "projective_appearance_integrator_blend_division": False,
        # spec switch (not an upstream field): contract a*b + c into fused multiply-adds at the projection, the bilinear samples and
        # the two blends' numerators, as nvcc's default -fmad=true would (include/mmfusion.h: mmf_params.fma_contraction)
        # (MMF_FMA_CONTRACTION=1 in the environment makes it the default of every mapper of the process: benchmarks / whole test suites)
        "projective_integrator_fma_contraction": FMA_CONTRACTION_DEFAULT,
        # spec switch (not an upstream field): bilinear samples (depth, synthetic depth, colour / feature taps) as four weighted taps in
        # upstream's order of terms instead of nested lerps (include/mmfusion.h: mmf_params.bilinear_four_weight_sum)
        "projective_integrator_bilinear_four_weight_sum": "bilinear_four_weight_sum" in SPEC_FLIP_DEFAULTS,
    }


class TsdfDecayIntegratorParams(_Bag):
    _fields = {
        "tsdf_decay_factor": 0.95,
        "tsdf_decayed_weight_threshold": 1e-3,
        "decay_integrator_deallocate_decayed_blocks": True,
        # spec switch (not an upstream field): whether Mapper.decay() also fades the colour / feature weights
        "decay_appearance_layers": False,
    }


class ViewCalculatorParams(_Bag):
    _fields = {
        "raycast_subsampling_factor": 4,
        "workspace_bounds_type": "kUnbounded",  # kUnbounded | kHeightBounds | kBoundingBox
        "workspace_bounds_min_corner_x_m": 0.0,
        "workspace_bounds_min_corner_y_m": 0.0,
        "workspace_bounds_min_height_m": 0.0,
        "workspace_bounds_max_corner_x_m": 0.0,
        "workspace_bounds_max_corner_y_m": 0.0,
        "workspace_bounds_max_height_m": 0.0,
        # spec switch (not an upstream field): blocks in view are marked along each ray up to depth + truncation (True) or depth
        "raycast_to_truncation_distance": True,
        # spec switch (not an upstream field): a ray's block walk starts at the camera (True) or where it enters the workspace bounds
        "raycast_walk_from_camera": False,
        # spec switch (not an upstream field): the block (and voxel) of a point by floor(p / size) (True) or floor(p * (1 / size))
        "block_index_by_division": "block_index_by_division" in SPEC_FLIP_DEFAULTS,
        # spec switch (not an upstream field): a pixel also marks the blocks within `truncation` of its surface point
        "view_truncation_band_marking": "view_truncation_band_marking" in SPEC_FLIP_DEFAULTS,
    }


class BlockMemoryPoolParams(_Bag):
    _fields = {
        "expansion_factor": 1.5,
        "num_preallocated_blocks": 0,
    }


class MeshIntegratorParams(_Bag):
    _fields = {
        "mesh_integrator_min_weight": 1e-4,
        "mesh_integrator_weld_vertices": True,  # vertices are always welded per block
    }


_WS_TYPES = {"kUnbounded": 0, "kHeightBounds": 1, "kBoundingBox": 2}
# upstream's WeightingFunctionType members (recalled); the int is mmf_params.weighting_mode
_WEIGHT_MODES = {"kConstantWeight": 0, "kInverseSquareWeight": 1, "kConstantDropoffWeight": 2, "kInverseSquareDropoffWeight": 3,
                 "kInverseSquareTsdfDistancePenalty": 4, "kLinearWithMax": 5}


class MmfWeightingMode(int):
    """An index in THIS library's numbering of the weighting functions (``mmf_params.weighting_mode``, the order of
    ``_WEIGHT_MODES``) -- the oracle's numbering, used by the tests and the pin kit.  A BARE int is refused by ``to_c``: upstream's
    ``WeightingFunctionType`` may number its members differently (its source is not in the reference tree), and a caller passing
    upstream's integer would silently get another weighting function.  Names and enum members (``.name``) are unambiguous."""

    @property
    def name(self) -> str:
        for k, v in _WEIGHT_MODES.items():
            if v == int(self):
                return k
        raise ValueError(f"no weighting mode with index {int(self)}")


class MapperParams:
    def __init__(self):
        self._projective = ProjectiveIntegratorParams()
        self._decay = TsdfDecayIntegratorParams()
        self._view = ViewCalculatorParams()
        self._pool = BlockMemoryPoolParams()
        self._mesh = MeshIntegratorParams()

    def set_projective_integrator_params(self, p: ProjectiveIntegratorParams) -> None:
        self._projective = p

    def set_tsdf_decay_integrator_params(self, p: TsdfDecayIntegratorParams) -> None:
        self._decay = p

    def set_view_calculator_params(self, p: ViewCalculatorParams) -> None:
        self._view = p

    def set_block_memory_pool_params(self, p: BlockMemoryPoolParams) -> None:
        self._pool = p

    def set_mesh_integrator_params(self, p: MeshIntegratorParams) -> None:
        self._mesh = p

    def get_projective_integrator_params(self) -> ProjectiveIntegratorParams:
        return self._projective

    def get_tsdf_decay_integrator_params(self) -> TsdfDecayIntegratorParams:
        return self._decay

    def get_view_calculator_params(self) -> ViewCalculatorParams:
        return self._view

    def get_block_memory_pool_params(self) -> BlockMemoryPoolParams:
        return self._pool

    def get_mesh_integrator_params(self) -> MeshIntegratorParams:
        return self._mesh

    def to_c(self, voxel_size_m: float, feature_channels: int) -> "_lib.MmfParams":
        """Flatten into the C ABI's ``mmf_params``."""
        p = _lib.default_params()
        pi, de, vc, po, me = self._projective, self._decay, self._view, self._pool, self._mesh
        p.voxel_size_m = float(voxel_size_m)
        p.max_integration_distance_m = float(pi.projective_integrator_max_integration_distance_m)
        p.truncation_distance_vox = float(pi.projective_integrator_truncation_distance_vox)
        p.max_weight = float(pi.projective_integrator_max_weight)
        wm = pi.projective_integrator_weighting_mode
        wm = getattr(wm, "name", wm)
        if isinstance(wm, int):
            raise ValueError(f"projective_integrator_weighting_mode = {wm}: a bare integer is ambiguous (this library's numbering is not "
                             f"known to be upstream's); pass the member's name ({', '.join(_WEIGHT_MODES)}), an enum member, or "
                             "mapper_params.MmfWeightingMode(index) for this library's own numbering")
        if wm not in _WEIGHT_MODES:
            raise ValueError(f"unsupported projective_integrator_weighting_mode: {wm}")
        p.weighting_mode = _WEIGHT_MODES[wm]
        p.lin_interp_max_diff_vox = float(pi.projective_tsdf_integrator_linear_interpolation_max_allowable_difference_vox)
        p.appearance_measurement_weight = float(pi.projective_appearance_integrator_measurement_weight)
        p.appearance_max_weight = float(pi.projective_appearance_integrator_max_weight)
        p.st_subsampling = int(pi.projective_appearance_integrator_sphere_tracing_ray_subsampling_factor)
        p.st_max_steps = int(pi.projective_appearance_integrator_sphere_tracing_max_steps)
        p.st_max_ray_length_m = float(pi.projective_appearance_integrator_sphere_tracing_max_ray_length_m)
        p.st_surface_eps_vox = float(pi.projective_appearance_integrator_sphere_tracing_surface_epsilon_vox)
        p.raycast_subsampling = int(vc.raycast_subsampling_factor)
        ws = getattr(vc.workspace_bounds_type, "name", vc.workspace_bounds_type)
        if ws not in _WS_TYPES:
            raise ValueError(f"unsupported workspace_bounds_type: {ws}")
        p.workspace_bounds_type = _WS_TYPES[ws]
        p.ws_min[0] = float(vc.workspace_bounds_min_corner_x_m)
        p.ws_min[1] = float(vc.workspace_bounds_min_corner_y_m)
        p.ws_min[2] = float(vc.workspace_bounds_min_height_m)
        p.ws_max[0] = float(vc.workspace_bounds_max_corner_x_m)
        p.ws_max[1] = float(vc.workspace_bounds_max_corner_y_m)
        p.ws_max[2] = float(vc.workspace_bounds_max_height_m)
        p.tsdf_decay_factor = float(de.tsdf_decay_factor)
        p.decayed_weight_threshold = float(de.tsdf_decayed_weight_threshold)
        p.deallocate_decayed_blocks = 1 if de.decay_integrator_deallocate_decayed_blocks else 0
        p.mesh_min_weight = float(me.mesh_integrator_min_weight)
        p.feature_channels = int(feature_channels)
        p.num_preallocated_blocks = int(po.num_preallocated_blocks)
        p.expansion_factor = float(po.expansion_factor)
        p.raycast_to_truncation = 1 if vc.raycast_to_truncation_distance else 0
        p.decay_appearance_layers = 1 if de.decay_appearance_layers else 0
        p.raycast_walk_from_camera = 1 if vc.raycast_walk_from_camera else 0
        p.appearance_blend_division = 1 if pi.projective_appearance_integrator_blend_division else 0
        p.fma_contraction = 1 if pi.projective_integrator_fma_contraction else 0
        p.bilinear_four_weight_sum = 1 if pi.projective_integrator_bilinear_four_weight_sum else 0
        p.block_index_by_division = 1 if vc.block_index_by_division else 0
        p.view_truncation_band_marking = 1 if vc.view_truncation_band_marking else 0
        return p
