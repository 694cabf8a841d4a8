"""Map -> model-input tensors (A11 + A12): the fused native path behind ``get_vertices_and_features``
(mmf_model_inputs_prepare / mmf_model_inputs_gather) against a numpy restatement, written here, of the reference's chain
mindmap/mapping/helpers/nvblox_output_helpers.py:49-91 applied to the FULL mesh of ``get_feature_mesh`` (itself bit-exact
against the oracle, tests/test_gpu_fusion_parity.py), followed by the reference's ``sample_to_n_vertices`` semantics
(data_loading/vertex_sampling.py:29-108; its RNG draws are pinned by tests/golden/vertex_sampling.npz).  Everything bit-exact."""
import numpy as np
import pytest
import torch

from fusion_common import make_mapper, small_cfg
from nvblox_mindmap_amd import synthetic as S
from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod
from nvblox_mindmap_amd.mapping.helpers.nvblox_output_helpers import get_vertices_and_features
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg

pytestmark = pytest.mark.gpu


def build_map(channels, scale=4, frames=(0, 9, 18), zero_half=False, **over):
    """A few frames of the synthetic stream through the stand-alone calls.  ``zero_half``: the left half of every feature
    image is zero, so that observed voxels (weight > 0) with an all-zero row exist."""
    cfg = small_cfg(scale)
    gpu = make_mapper(channels, **over)
    for i in frames:
        f = S.frame(cfg, i, channels)
        feat = f["features"].copy()
        if zero_half:
            feat[:, : cfg.width // 2, :] = 0
        T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
        gpu.add_depth_frame(torch.from_numpy(f["depth"]).cuda(), T, K, None, 0)
        gpu.add_feature_frame(torch.from_numpy(feat).cuda(), T, K, None, 0)
    return gpu


def reference_rows(gpu, mcfg, remove_zero_features, num_excess_features):
    """nvblox_output_helpers.py:49-74 in numpy on the full mesh: (vertices [n,3] f32, features [n,used] f16)."""
    mesh = gpu.get_feature_mesh(MAPPER_TO_ID.STATIC)
    v, f = mesh.vertices().cpu().numpy(), mesh.vertex_features().cpu().numpy()
    inside = np.all((v > mcfg.aabb_min_m.numpy()) & (v < mcfg.aabb_max_m.numpy()), axis=1)  # strict (:57-60)
    v, f = v[inside], f[inside]
    if num_excess_features > 0:
        f = f[:, :-num_excess_features]  # (:63-66)
    if remove_zero_features:
        nz = ~np.all(f == 0, axis=1)  # (:68-74); -0.0 == 0
        v, f = v[nz], f[nz]
    return v, f


def reference_sample(v, f, want, method, seed):
    """sample_to_n_vertices (vertex_sampling.py:29-108) in numpy + torch's CPU generator for the draws."""
    n = v.shape[0]
    if method == VertexSamplingMethod.NONE or n == want:
        return v, f, np.ones(n, dtype=bool)
    if n > want:
        torch.manual_seed(seed)
        if method == VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT:
            sel = torch.randperm(n)[:want].numpy()
        elif method == VertexSamplingMethod.RANDOM_WITH_REPLACEMENT:
            sel = torch.randint(0, n, (want,)).numpy()
        else:
            sel = np.argsort(-v[:, 2], kind="stable")[:want]
        return v[sel], f[sel], np.ones(want, dtype=bool)
    pad = want - n
    valid = np.ones(want, dtype=bool)
    valid[n:] = False
    return np.concatenate([v, np.zeros((pad, 3), v.dtype)]), np.concatenate([f, np.zeros((pad, f.shape[1]), f.dtype)]), valid


@pytest.mark.parametrize("channels,excess", [(16, 0), (16, 3), (64, 8), (64, 0), (768, 384)])
@pytest.mark.parametrize("remove_zero", [True, False])
def test_model_inputs_match_the_reference_chain(channels, excess, remove_zero):
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    gpu = build_map(channels, scale=8 if channels == 768 else 4, zero_half=True)
    rv, rf = reference_rows(gpu, mcfg, remove_zero, excess)
    n = rv.shape[0]
    assert n > 500
    full_v, full_f = reference_rows(gpu, mcfg, False, excess)
    if remove_zero:
        assert full_v.shape[0] > n, "the case must hold all-zero rows for the filter to remove"
    # un-sampled (:76-80)
    v, f, valid = get_vertices_and_features(gpu, MAPPER_TO_ID.STATIC, mcfg, remove_zero, excess, sample_vertices=False)
    assert f.dtype == torch.float16 and valid.shape == (1, n) and bool(valid.all())
    assert np.array_equal(v.cpu().numpy(), rv) and np.array_equal(f.cpu().numpy().view(np.uint16), rf.view(np.uint16))
    for method in VertexSamplingMethod:
        for want in (300, n, n + 77):
            for dtype in (None, torch.float32):
                torch.manual_seed(11)
                v, f, valid = get_vertices_and_features(gpu, MAPPER_TO_ID.STATIC, mcfg, remove_zero, excess, sample_vertices=True,
                                                        number_of_vertices_to_sample=want, vertex_sampling_method=method,
                                                        features_dtype=dtype)
                state = torch.get_rng_state()
                ev, ef, evalid = reference_sample(rv, rf, want, method, 11)
                assert torch.equal(torch.get_rng_state(), state), "the helper must leave the generator where the reference does"
                assert v.shape == (1,) + ev.shape and f.shape == (1,) + ef.shape and valid.shape == (1, evalid.shape[0])
                assert f.dtype == (torch.float16 if dtype is None else dtype)
                assert np.array_equal(v[0].cpu().numpy(), ev), (method, want)
                assert np.array_equal(valid[0].cpu().numpy(), evalid)
                got = f[0].cpu().numpy()
                assert np.array_equal(got, ef if dtype is None else ef.astype(np.float32)), (method, want, dtype)


def test_model_inputs_follow_the_map():
    """prepare / gather are tied to the map state: a gather after the map changed is refused, a new prepare sees the change
    (incl. a pending lazy decay and a cleared map)."""
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    gpu = build_map(16, frames=(0,))
    cfg = small_cfg(4)
    n0 = gpu.model_inputs_prepare(0, mcfg.aabb_min_host, mcfg.aabb_max_host, 16, True)
    v0, f0, m0 = gpu.model_inputs_gather(0, None, n0, n0, torch.float16)
    f = S.frame(cfg, 12, 16)
    T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
    gpu.add_depth_frame(torch.from_numpy(f["depth"]).cuda(), T, K, None, 0)
    with pytest.raises(RuntimeError, match="changed since"):
        gpu.model_inputs_gather(0, None, 1, 1, torch.float16)
    gpu.add_feature_frame(torch.from_numpy(f["features"]).cuda(), T, K, None, 0)
    n1 = gpu.model_inputs_prepare(0, mcfg.aabb_min_host, mcfg.aabb_max_host, 16, True)
    assert n1 > n0
    rv, rf = reference_rows(gpu, mcfg, True, 0)
    v1, f1, _ = gpu.model_inputs_gather(0, None, n1, n1, torch.float16)
    assert np.array_equal(v1.cpu().numpy(), rv) and np.array_equal(f1.cpu().numpy().view(np.uint16), rf.view(np.uint16))
    gpu.decay()  # lazy: pending until something reads the map
    with pytest.raises(RuntimeError, match="changed since"):
        gpu.model_inputs_gather(0, None, 1, 1, torch.float16)
    assert gpu.model_inputs_prepare(0, mcfg.aabb_min_host, mcfg.aabb_max_host, 16, True) > 0
    gpu.clear()
    with pytest.raises(AssertionError, match="No vertices"):
        get_vertices_and_features(gpu, 0, mcfg, True, 0, sample_vertices=True, number_of_vertices_to_sample=10,
                                  vertex_sampling_method=VertexSamplingMethod.NONE)
    with pytest.raises(RuntimeError):
        gpu.model_inputs_prepare(0, mcfg.aabb_min_host, mcfg.aabb_max_host, 17, True)  # more channels than the map stores


def test_model_inputs_list_growth_and_unbounded_map():
    """More kept vertices than the internal list's first capacity (65 536: the pass is repeated once, transparently) and more
    live blocks than the gather kernel scans in LDS (8 192: prefix sums from k_mesh_scan), on an unbounded map."""
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    mcfg.aabb_min_m, mcfg.aabb_max_m = torch.tensor([-10.0, -10.0, -10.0]), torch.tensor([10.0, 10.0, 10.0])
    gpu = build_map(8, scale=2, frames=(0, 9, 18), workspace_bounds_type=0, max_integration_distance_m=3.0)
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() > 8192
    rv, rf = reference_rows(gpu, mcfg, False, 0)
    assert rv.shape[0] > 65536
    v, f, valid = get_vertices_and_features(gpu, 0, mcfg, False, 0, sample_vertices=False)
    assert np.array_equal(v.cpu().numpy(), rv) and np.array_equal(f.cpu().numpy().view(np.uint16), rf.view(np.uint16))
    torch.manual_seed(3)
    v, f, valid = get_vertices_and_features(gpu, 0, mcfg, True, 0, sample_vertices=True, number_of_vertices_to_sample=2048,
                                            vertex_sampling_method=VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
    rv, rf = reference_rows(gpu, mcfg, True, 0)
    ev, ef, _ = reference_sample(rv, rf, 2048, VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, 3)
    assert np.array_equal(v[0].cpu().numpy(), ev) and np.array_equal(f[0].cpu().numpy().view(np.uint16), ef.view(np.uint16))
