"""CPU tests of the drop-in boundary: the C-ABI library loads here (no GPU needed for that), exports every
symbol include/mmfusion.h declares, and fails loudly -- no fallback -- when asked to compute without a GPU."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "mmfusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from nvblox_mindmap_amd import _lib

    _lib.build()
    L = C.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"{n} is declared in include/mmfusion.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_params_layout_and_defaults():
    from nvblox_mindmap_amd import _lib

    L = _lib.lib()
    assert L.mmf_params_size() == C.sizeof(_lib.MmfParams)
    assert L.mmf_abi_version() == 1
    p = _lib.default_params()
    assert p.truncation_distance_vox == 4.0 and p.max_weight == 5.0 and p.feature_channels == 768
    assert L.mmf_kernel_name(6).decode().startswith("k_feature_integrate") and L.mmf_kernel_name(9).decode() == "k_feature_flat"


def test_mapper_params_flatten_like_the_reference_config():
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import (
        MapperParams, ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams)

    cfg = NvbloxMappingCfg("drill_in_box")
    pi = ProjectiveIntegratorParams()
    pi.projective_integrator_max_integration_distance_m = cfg.projective_integrator_max_integration_distance_m
    de = TsdfDecayIntegratorParams()
    de.tsdf_decay_factor = cfg.tsdf_decay_factor
    vc = ViewCalculatorParams()
    vc.raycast_subsampling_factor = 1
    vc.workspace_bounds_type = "kBoundingBox"
    vc.workspace_bounds_min_corner_x_m = float(cfg.aabb_min_m[0])
    vc.workspace_bounds_max_height_m = float(cfg.aabb_max_m[2])
    mp = MapperParams()
    mp.set_projective_integrator_params(pi)
    mp.set_tsdf_decay_integrator_params(de)
    mp.set_view_calculator_params(vc)
    c = mp.to_c(cfg.voxel_size_m, 64)
    assert abs(c.voxel_size_m - 0.01) < 1e-9 and c.max_integration_distance_m == 5.0 and c.raycast_subsampling == 1
    assert c.workspace_bounds_type == 2 and abs(c.ws_min[0] + 0.37) < 1e-6 and abs(c.ws_max[2] - 0.65) < 1e-6
    assert abs(c.tsdf_decay_factor - 0.98) < 1e-7 and c.feature_channels == 64
    with pytest.raises(AttributeError):
        vc.no_such_parameter = 1


def test_no_cpu_fallback():
    """Without a GPU the product path must refuse to run (and never route through the oracle)."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nvblox_mindmap_amd.image_processing import erode_mask, get_camera_pointcloud
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper

    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        Mapper(0.01)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        erode_mask(torch.ones((4, 4), dtype=torch.bool), iterations=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_camera_pointcloud(torch.eye(3), torch.ones((4, 4)), torch.zeros(3), torch.tensor([1.0, 0, 0, 0]))
    # the product package must not import the oracle
    import sys

    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers  # noqa: F401
    import nvblox_mindmap_amd.mapping.helpers.nvblox_output_helpers  # noqa: F401
    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules if "nvblox_mindmap_amd" in str(getattr(sys.modules[m], "__file__", "")))
    src_dir = os.path.join(ROOT, "nvblox_mindmap_amd")
    banned = ("import oracle", "from oracle", "libmmf_oracle", "orc_create", "orc_add_", "oracle.oracle", "oracle/_ref")
    for dirpath, _, files in os.walk(src_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in text, f"{f} references the oracle ({b})"


def test_nvblox_torch_surface():
    """Every nvblox_torch symbol the reference imports exists."""
    import nvblox_mindmap_amd as P

    P.install_as_nvblox_torch()
    from nvblox_torch.constants import constants  # noqa
    from nvblox_torch.mapper import Mapper, QueryType  # noqa
    from nvblox_torch.mapper_params import (  # noqa
        BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams)
    from nvblox_torch.projective_integrator_types import ProjectiveIntegratorType  # noqa
    from nvblox_torch.timer import Timer, get_last_time, get_mean_time, print_timers, timer_status_string  # noqa

    assert constants.feature_array_num_elements() % 8 == 0
    for m in ("add_depth_frame", "add_color_frame", "add_feature_frame", "decay", "clear", "update_feature_mesh",
              "get_feature_mesh", "num_mappers", "tsdf_layer_view", "feature_layer_view", "query_layer"):
        assert hasattr(Mapper, m)
    with Timer("a/b"):
        pass
    t = Timer("a/b")
    t.stop()
    assert get_mean_time("a/b") >= 0 and get_last_time("a/b") >= 0 and "a/b" in timer_status_string()

    import torch
    from nvblox_torch.indexing import get_voxel_center_grids
    from nvblox_torch.layer import FeatureLayer, Layer, convert_layer_to_dense_tensor  # noqa
    from nvblox_torch.visualization import get_voxel_mesh

    grids = get_voxel_center_grids(torch.tensor([[0, 0, 0], [-1, 2, 3]], dtype=torch.int32), 0.01)
    assert grids.shape == (2, 8, 8, 8, 3)
    assert torch.allclose(grids[1, 0, 0, 0], torch.tensor([-0.08 + 0.005, 0.16 + 0.005, 0.24 + 0.005]))
    cubes = get_voxel_mesh(grids[0].reshape(-1, 3)[:5], 0.01, colors=torch.full((5, 3), 255, dtype=torch.uint8))
    assert cubes.vertices.shape == (40, 3) and cubes.triangles.shape == (60, 3) and cubes.vertex_colors.shape == (40, 3)
    # every cube is closed and outward-facing: signed volume of its 12 triangles = voxel volume
    v = cubes.vertices[cubes.triangles.long()].double()
    vol = (v[:, 0] * torch.linalg.cross(v[:, 1], v[:, 2])).sum(-1).reshape(5, 12).sum(-1) / 6.0
    assert torch.allclose(vol, torch.full((5,), 1e-6, dtype=torch.float64), rtol=1e-4)


def test_weighting_mode_takes_names_not_bare_integers():
    """Round-3 advisor finding: an integer is ambiguous (upstream's enum order is not known here); names, enum members and the
    explicit MmfWeightingMode index are not."""
    import enum

    import pytest

    from nvblox_mindmap_amd.nvblox_torch.mapper_params import MapperParams, MmfWeightingMode, ProjectiveIntegratorParams

    def mode_of(value):
        pi = ProjectiveIntegratorParams()
        pi.projective_integrator_weighting_mode = value
        mp = MapperParams()
        mp.set_projective_integrator_params(pi)
        return mp.to_c(0.01, 16).weighting_mode

    class Upstream(enum.Enum):  # whatever VALUES upstream gives its members, the NAME decides
        kConstantDropoffWeight = 1
        kInverseSquareWeight = 2

    assert mode_of("kInverseSquareWeight") == 1 and mode_of(Upstream.kInverseSquareWeight) == 1
    assert mode_of(Upstream.kConstantDropoffWeight) == 2 and mode_of(MmfWeightingMode(5)) == 5
    with pytest.raises(ValueError, match="bare integer"):
        mode_of(2)
    with pytest.raises(ValueError):
        mode_of("kNoSuchWeight")
