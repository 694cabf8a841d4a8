"""GPU parity: HIP integrator (through the C ABI) vs the CPU oracle on the same seeded stream.

Bar (BASELINE.json north_star): bit-exact voxel-block indices / occupancy; TSDF and feature values within
1e-5 abs.  Because both sides use the same float32 operation order without FMA contraction the values are
in fact expected to be bit-identical; the tests assert the 1e-5 bar and report exactness.
"""
import os

import numpy as np
import pytest
import torch

from nvblox_mindmap_amd import synthetic as S

import fusion_common
from fusion_common import REF_PARAMS, frame_masks, make_mapper, make_oracle, small_cfg, sort_rows

pytestmark = pytest.mark.gpu
TOL = 1e-5


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t if dtype is None else t.to(dtype)


def run_both(oracle_mod, cfg, channels, frames, use_mask=True, color=True, decay=True, **over):
    orc = make_oracle(oracle_mod, channels, **over)
    gpu = make_mapper(channels, **over)
    for k, i in enumerate(frames):
        f = S.frame(cfg, i, channels)
        mask = frame_masks(f["depth"], k) if use_mask else None
        if decay:
            orc.decay()
            gpu.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], mask)
        gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]),
                            None if mask is None else dev(mask), 0)
        if color:
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], mask)
            gpu.add_color_frame(dev(f["rgb"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]),
                                mask_frame=None if mask is None else dev(mask), mapper_id=0)
        if channels:
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], mask)
            gpu.add_feature_frame(dev(f["features"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]),
                                  None if mask is None else dev(mask), 0)
    return orc, gpu


def compare_tsdf(orc, gpu):
    blocks, idx = gpu.tsdf_layer_view(0).get_all_blocks()
    blocks, idx = blocks.cpu().numpy(), idx.cpu().numpy()
    oidx, oblocks = orc.block_indices(0), orc.all_tsdf()
    assert idx.shape == oidx.shape, f"TSDF block count {idx.shape[0]} vs oracle {oidx.shape[0]}"
    # the allocation order itself is part of the spec: compare unsorted first, then as sets
    assert np.array_equal(idx[sort_rows(idx)], oidx[sort_rows(oidx)]), "TSDF block index SET differs"
    assert np.array_equal(idx, oidx), "TSDF block allocation ORDER differs"
    if idx.shape[0] == 0:
        return 0.0, True
    diff = np.abs(blocks - oblocks)
    assert diff.max() <= TOL, f"TSDF max abs diff {diff.max()}"
    return float(diff.max()), bool(np.array_equal(blocks.view(np.uint32), oblocks.view(np.uint32)))


def compare_features(orc, gpu):
    f, w, idx = gpu.feature_layer_view(0).get_all_blocks_split()
    f, w, idx = f.cpu().numpy(), w.cpu().numpy(), idx.cpu().numpy()
    of, ow = orc.all_features()
    oidx = orc.block_indices(2)
    assert idx.shape == oidx.shape, f"feature block count {idx.shape[0]} vs oracle {oidx.shape[0]}"
    assert np.array_equal(idx, oidx), "feature block indices / order differ"
    assert np.array_equal(w, ow), f"feature weights differ (max {np.abs(w - ow).max()})"
    d = np.abs(f.astype(np.float32) - of.astype(np.float32))
    # f16 storage: 1e-5 abs is below one f16 ulp for |x| > 0.01, so demand bit equality of the halves
    assert np.array_equal(f.view(np.uint16), of.view(np.uint16)), f"feature values differ, max abs {d.max()}"
    return float(d.max()) if d.size else 0.0


def test_tsdf_only_small(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 7, 14], use_mask=False, color=False, decay=False)
    v = gpu.last_view_blocks(0).cpu().numpy()
    assert np.array_equal(v, orc.last_view_blocks()), "blocks-in-view of the last frame differ"
    mx, exact = compare_tsdf(orc, gpu)
    print(f"tsdf max diff {mx} bit-exact {exact} blocks {orc.num_blocks(0)}")
    assert exact


def test_tsdf_mask_decay_small(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 5, 10, 15, 20], use_mask=True, color=False, decay=True)
    mx, exact = compare_tsdf(orc, gpu)
    assert exact


def test_decay_deallocates_and_reuses_slots(oracle_mod):
    """Strong decay so blocks die, get deallocated (order preserved) and their slots are reused."""
    cfg = small_cfg(4)
    over = dict(tsdf_decay_factor=0.05, decayed_weight_threshold=1e-3)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 50, 100, 150, 0, 50], use_mask=False, color=False, decay=True, **over)
    n0 = orc.num_blocks(0)
    assert n0 > 0
    compare_tsdf(orc, gpu)
    counts = [n0]
    for _ in range(4):
        orc.decay()
        gpu.decay()
        assert gpu.tsdf_layer_view(0).num_allocated_blocks() == orc.num_blocks(0)
        compare_tsdf(orc, gpu)
        counts.append(orc.num_blocks(0))
    assert counts[-1] < n0, counts
    # re-observe: freed slots are reused, the hash was rebuilt
    f = S.frame(cfg, 25, 0)
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
    gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    compare_tsdf(orc, gpu)


def test_sphere_trace_small(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 3], use_mask=False, color=False, decay=False)
    f = S.frame(cfg, 1, 0)
    so = orc.render_synthetic_depth(cfg.height, cfg.width, f["T_W_C"], f["K"])
    sg = gpu.render_synthetic_depth(cfg.height, cfg.width, f["T_W_C"], f["K"]).cpu().numpy()
    assert so.shape == sg.shape
    assert (so > 0).mean() > 0.2
    assert np.array_equal(so.view(np.uint32), sg.view(np.uint32)), f"max diff {np.abs(so - sg).max()}"


def test_color_small(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 4, 8], use_mask=True, color=True, decay=True)
    compare_tsdf(orc, gpu)
    rgb, w, idx = gpu.color_layer_view(0).get_all_blocks_split()
    orgb, ow = orc.all_colors()
    assert np.array_equal(idx.cpu().numpy(), orc.block_indices(1))
    assert np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(rgb.cpu().numpy(), orgb)
    assert (ow > 0).sum() > 1000


@pytest.mark.parametrize("channels", [16, 64])
def test_feature_fusion_small(oracle_mod, channels):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, channels, [0, 4, 8, 12], use_mask=True, color=False, decay=True)
    compare_tsdf(orc, gpu)
    compare_features(orc, gpu)
    _, ow = orc.all_features()
    assert (ow > 0).sum() > 1000


def test_feature_mesh_small(oracle_mod):
    cfg = small_cfg(4)
    channels = 16
    orc, gpu = run_both(oracle_mod, cfg, channels, [0, 6, 12], use_mask=True, color=False, decay=True)
    ov, of = orc.feature_mesh()
    gpu.update_feature_mesh(0)
    mesh = gpu.get_feature_mesh(0)
    gv, gf = mesh.vertices().cpu().numpy(), mesh.vertex_features().cpu().numpy()
    assert gv.shape == ov.shape, f"{gv.shape} vs {ov.shape}"
    assert ov.shape[0] > 500
    assert np.abs(gv - ov).max() <= TOL
    assert np.array_equal(gv.view(np.uint32), ov.view(np.uint32))
    assert np.array_equal(gf.view(np.uint16), of.view(np.uint16))
    assert (np.abs(of.astype(np.float32)).sum(1) > 0).mean() > 0.3


def test_query_layer(oracle_mod):
    cfg = small_cfg(4)
    channels = 16
    orc, gpu = run_both(oracle_mod, cfg, channels, [0, 6], use_mask=False, color=False, decay=False)
    ov, _ = orc.feature_mesh()
    pts = np.concatenate([ov[::7], ov[::11] + np.float32(0.013), np.array([[9.0, 9.0, 9.0]], dtype=np.float32)])
    from nvblox_mindmap_amd.nvblox_torch.mapper import QueryType

    qf = gpu.query_layer(QueryType.FEATURE, dev(pts), 0).cpu().numpy()
    qt = gpu.query_layer(QueryType.TSDF, dev(pts), 0).cpu().numpy()
    assert np.array_equal(qf, orc.query_features(pts))
    assert np.array_equal(qt, orc.query_tsdf(pts))


def test_clear(oracle_mod):
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 16, [0, 6], use_mask=False, color=True, decay=False)
    gpu.clear()
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() == 0
    assert gpu.feature_layer_view(0).num_allocated_blocks() == 0
    assert gpu.update_feature_mesh(0) == 0
    orc.clear()
    # the map must be fully usable after clear
    f = S.frame(cfg, 3, 16)
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
    gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    compare_tsdf(orc, gpu)


def test_unbounded_workspace(oracle_mod):
    """kUnbounded workspace: larger voxels, the view grid comes from the frustum only."""
    cfg = small_cfg(4)
    over = dict(workspace_bounds_type=0, voxel_size=0.04, max_integration_distance_m=3.0)
    orc, gpu = run_both(oracle_mod, cfg, 0, [0, 9], use_mask=False, color=False, decay=False, **over)
    compare_tsdf(orc, gpu)


def test_full_resolution_frame(oracle_mod):
    """One BASELINE-sized frame (640x480 depth + 64-channel features) against the oracle."""
    cfg = S.StreamConfig()
    orc, gpu = run_both(oracle_mod, cfg, 64, [0, 10], use_mask=False, color=False, decay=True)
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)


def test_errors_are_loud():
    gpu = make_mapper(16)
    with pytest.raises(ValueError):
        gpu.add_feature_frame(torch.zeros((8, 8, 24), dtype=torch.float16, device="cuda"), torch.eye(4), torch.eye(3), None, 0)
    with pytest.raises(ValueError):
        gpu.add_depth_frame(torch.zeros((8, 8), dtype=torch.float32), torch.eye(4), torch.eye(3), None, 0)
    with pytest.raises(ValueError):
        gpu.add_depth_frame(torch.zeros((8, 8), dtype=torch.float32, device="cuda"), torch.eye(4), torch.eye(3), None, 5)


@pytest.mark.parametrize("scale,channels", [(4, 16), (1, 64)])
def test_integrate_frame_fused_matches_oracle(oracle_mod, scale, channels):
    """The reference's integrate_frame as one native call (masks + depth + colour + feature, five fused launches, lazy
    decay folded in) against the oracle driven with the numpy-oracle masks, over several frames with decay."""
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper, integrate_frame
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg
    from oracle import image_ops as IO

    base = small_cfg(scale)
    cfg = S.StreamConfig(width=base.width, height=base.height, fx=base.fx, fy=base.fy, cx=base.cx, cy=base.cy, hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    k_in, k_depth = (3, 4) if scale > 1 else (mcfg.static_mask_erosion_iterations, mcfg.valid_depth_mask_erosion_iterations)
    gpu = get_nvblox_mapper(mcfg, feature_channels=channels)
    orc = make_oracle(oracle_mod, channels)
    frames = [0, 5, 10] if scale > 1 else [0, 7]
    for k, i in enumerate(frames):
        f = S.frame(cfg, i, channels)
        dyn = np.zeros(f["depth"].shape, dtype=bool)
        dyn[10 + k: 20 + k, 30:50] = True
        static = ~dyn
        odm, ofm = IO.frame_masks(static, f["depth"], mcfg.min_integration_distance_m, k_in, k_depth,
                                  mcfg.feature_mask_border_percent, cfg.height, cfg.width)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
        gpu.decay()
        K = torch.from_numpy(f["K"])
        images = integrate_frame(mapper=gpu, nvblox_mapping_config=mcfg, depth_frame=dev(f["depth"]), feature_frame=dev(f["features"]),
                                 intrinsics=K, camera_pose=torch.from_numpy(f["T_W_C"]), rgb=dev(f["rgb"]), input_mask=dev(static),
                                 input_mask_erosion_iterations=k_in, valid_depth_mask_erosion_iterations=k_depth,
                                 mapper_id=MAPPER_TO_ID.STATIC)
        assert torch.equal(K, torch.from_numpy(f["K"])), "the caller's intrinsics must not be modified"
        assert np.array_equal(images["depth_mask"].cpu().numpy(), odm)
        assert np.array_equal(images["feature_mask"].cpu().numpy().astype(bool), ofm)
        assert set(images.keys()) == {"depth_frame", "depth_mask", "rgb_frame", "rgb_mask", "feature_frame", "feature_mask", "input_mask"}
        assert images["rgb_frame"].shape == (3, cfg.height, cfg.width) and float(images["rgb_frame"].max()) <= 1.0
    compare_tsdf(orc, gpu)
    compare_features(orc, gpu)
    rgb, w, idx = gpu.color_layer_view(0).get_all_blocks_split()
    orgb, ow = orc.all_colors()
    assert np.array_equal(idx.cpu().numpy(), orc.block_indices(1))
    assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(rgb.cpu().numpy(), orgb)
    assert (ow > 0).sum() > 500
    _, fw = orc.all_features()
    assert (fw > 0).sum() > 500
    # and the model-input extraction on top of it
    from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod
    from nvblox_mindmap_amd.mapping.helpers.nvblox_output_helpers import get_vertices_and_features

    v, feats, valid = get_vertices_and_features(gpu, MAPPER_TO_ID.STATIC, mcfg, remove_zero_features=True, num_excess_features=0,
                                                sample_vertices=True, number_of_vertices_to_sample=2048,
                                                vertex_sampling_method=VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
    assert v.shape == (1, 2048, 3) and feats.shape == (1, 2048, channels) and valid.shape == (1, 2048)
    ov, of = orc.feature_mesh()
    inside = np.all((ov > mcfg.aabb_min_m.numpy()) & (ov < mcfg.aabb_max_m.numpy()), axis=1)
    nonzero = np.any(of != 0, axis=1)
    n_expected = int((inside & nonzero).sum())
    assert int(valid.sum()) == min(n_expected, 2048)
    # the single-gather path of the helper (more vertices than requested) returns what the step-by-step form returns
    from nvblox_mindmap_amd.data_loading.vertex_sampling import sample_to_n_vertices

    for method in (VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, VertexSamplingMethod.RANDOM_WITH_REPLACEMENT, VertexSamplingMethod.LOWEST):
        torch.manual_seed(5)
        v1, f1, m1 = get_vertices_and_features(gpu, MAPPER_TO_ID.STATIC, mcfg, remove_zero_features=True, num_excess_features=1,
                                               sample_vertices=True, number_of_vertices_to_sample=300, vertex_sampling_method=method)
        torch.manual_seed(5)
        va, fa, _ = get_vertices_and_features(gpu, MAPPER_TO_ID.STATIC, mcfg, remove_zero_features=True, num_excess_features=1,
                                              sample_vertices=False)
        assert va.shape[0] > 300
        # the un-sampled rows are the reference's chain of boolean-mask copies (nvblox_output_helpers.py:57-74), restated in numpy
        mesh = gpu.get_feature_mesh(MAPPER_TO_ID.STATIC)
        mv, mf = mesh.vertices().cpu().numpy(), mesh.vertex_features().cpu().numpy()
        box = np.all((mv > mcfg.aabb_min_m.numpy()) & (mv < mcfg.aabb_max_m.numpy()), axis=1)
        mv, mf = mv[box], mf[box][:, :-1]
        nz = ~np.all(mf == 0, axis=1)
        assert np.array_equal(va.cpu().numpy(), mv[nz]) and np.array_equal(fa.cpu().numpy(), mf[nz])
        v2, f2, m2 = sample_to_n_vertices(va, fa, 300, method)
        assert torch.equal(v1[0], v2) and torch.equal(f1[0], f2) and torch.equal(m1[0], m2)


def test_reference_channel_count_768(oracle_mod):
    """The reference build's feature width (C_pad = 768, docker/install_nvblox.sh:24-25): 96 pieces per voxel row."""
    cfg = small_cfg(8)
    orc, gpu = run_both(oracle_mod, cfg, 768, [0, 9], use_mask=True, color=False, decay=True)
    compare_tsdf(orc, gpu)
    compare_features(orc, gpu)
    ov, of = orc.feature_mesh()
    mesh = gpu.get_feature_mesh(0)
    assert np.array_equal(mesh.vertices().cpu().numpy().view(np.uint32), ov.view(np.uint32))
    assert np.array_equal(mesh.vertex_features().cpu().numpy().view(np.uint16), of.view(np.uint16))


def test_two_mappers_are_independent(oracle_mod):
    """STATIC / DYNAMIC mappers of one Mapper object (nvblox_mapper_constants.py:27-29) do not interact."""
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg

    cfg = small_cfg(4)
    gpu = get_nvblox_mapper(NvbloxMappingCfg("DRILL_IN_BOX"), feature_channels=16)
    assert gpu.num_mappers() == 2
    o0, o1 = make_oracle(oracle_mod, 16), make_oracle(oracle_mod, 16)
    for i, (orc, mid) in enumerate([(o0, 0), (o1, 1), (o0, 0)]):
        f = S.frame(cfg, 7 * i, 16)
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
        gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, mid)
    for orc, mid in ((o0, 0), (o1, 1)):
        blocks, idx = gpu.tsdf_layer_view(mid).get_all_blocks()
        assert np.array_equal(idx.cpu().numpy(), orc.block_indices(0))
        assert np.array_equal(blocks.cpu().numpy().view(np.uint32), orc.all_tsdf().view(np.uint32))
    gpu.clear(1)
    assert gpu.tsdf_layer_view(1).num_allocated_blocks() == 0
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() == o0.num_blocks(0)


def _lowres_map(index, cin, lh=16, lw=16):
    return np.random.default_rng(1000 + index).standard_normal((lh, lw, cin)).astype(np.float32)


@pytest.mark.parametrize("scale,cin,channels,lh,lw", [(4, 16, 16, 16, 16), (4, 24, 32, 5, 7), (2, 64, 64, 16, 16)])
def test_add_feature_frame_lowres_is_upsample_plus_add(oracle_mod, scale, cin, channels, lh, lw):
    """Fused up-sample + integrate == materialised path, bit for bit: against the HIP
    upsample_features + add_feature_frame pair and against the oracle fed with the same up-sampled f16 image."""
    from nvblox_mindmap_amd.image_processing import upsample_features

    cfg = small_cfg(scale)
    fused, plain, orc = make_mapper(channels), make_mapper(channels), make_oracle(oracle_mod, channels)
    for k, i in enumerate([0, 6, 12]):
        f = S.frame(cfg, i, 0)
        mask = frame_masks(f["depth"], k)
        low = _lowres_map(i, cin, lh, lw)
        T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
        up = upsample_features(dev(np.ascontiguousarray(low.transpose(2, 0, 1))), (cfg.height, cfg.width), channels)
        for m in (fused, plain):
            m.decay()
            m.add_depth_frame(dev(f["depth"]), T, K, dev(mask), 0)
        plain.add_feature_frame(up, T, K, dev(mask), 0)
        fused.add_feature_frame_lowres(dev(low), (cfg.height, cfg.width), T, K, dev(mask), 0)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], mask)
        orc.add_feature_frame(up.cpu().numpy(), f["T_W_C"], f["K"], mask)
    fa, wa, ia = fused.feature_layer_view(0).get_all_blocks_split()
    fb, wb, ib = plain.feature_layer_view(0).get_all_blocks_split()
    assert torch.equal(ia, ib) and torch.equal(wa, wb)
    assert torch.equal(fa.view(torch.int16), fb.view(torch.int16))
    assert int((wa > 0).sum()) > 1000
    if cin < channels:
        assert float(fa[..., cin:].abs().max()) == 0.0
    compare_tsdf(orc, fused)
    compare_features(orc, fused)


def test_integrate_frame_lowres_matches_integrate_frame():
    """One-call frame integration with the low-res source == the same call with the up-sampled image (full 640x480)."""
    from nvblox_mindmap_amd.image_processing import upsample_features
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    a, b = get_nvblox_mapper(mcfg, feature_channels=64), get_nvblox_mapper(mcfg, feature_channels=64)
    for i in [0, 9]:
        f = S.frame(cfg, i, 0)
        low = _lowres_map(i, 64)
        up = upsample_features(dev(np.ascontiguousarray(low.transpose(2, 0, 1))), (cfg.height, cfg.width), 64)
        static = np.ones(f["depth"].shape, dtype=bool)
        args = (dev(static), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), mcfg.min_integration_distance_m, 3, 4,
                mcfg.feature_mask_border_percent, 0)
        a.decay()
        b.decay()
        dm_a, fm_a = a.integrate_frame(dev(f["depth"]), dev(f["rgb"]), up, *args)
        dm_b, fm_b = b.integrate_frame_lowres(dev(f["depth"]), dev(f["rgb"]), dev(low), *args)
        assert torch.equal(dm_a, dm_b) and torch.equal(fm_a, fm_b)
    fa, wa, ia = a.feature_layer_view(0).get_all_blocks_split()
    fb, wb, ib = b.feature_layer_view(0).get_all_blocks_split()
    assert torch.equal(ia, ib) and torch.equal(wa, wb) and torch.equal(fa.view(torch.int16), fb.view(torch.int16))
    assert int((wa > 0).sum()) > 10000
    ca, cwa, _ = a.color_layer_view(0).get_all_blocks_split()
    cb, cwb, _ = b.color_layer_view(0).get_all_blocks_split()
    assert torch.equal(ca, cb) and torch.equal(cwa, cwb)
    with pytest.raises(RuntimeError):  # Cin not a multiple of 8: the fused path refuses, loudly
        b.add_feature_frame_lowres(dev(_lowres_map(0, 12)), (cfg.height, cfg.width), torch.from_numpy(f["T_W_C"]),
                                   torch.from_numpy(f["K"]), None, 0)


def test_inverted_input_mask_equals_materialised_inverse():
    """integrate_frame(dynamic_mask, invert_input_mask=True) == integrate_frame(~dynamic_mask): the reference's
    static_mask = ~dynamic_mask (nvblox_mapping_helpers.py:116-117) read inverted by the native call; both the fused
    path (640x480) and nvblox_integrate on top of it."""
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper, integrate_frame, nvblox_integrate
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    a, b, c = (get_nvblox_mapper(mcfg, feature_channels=16) for _ in range(3))

    class Extractor:  # stands in for FeatureExtractor.compute (the DNN is out of scope): returns the frame's feature image
        def __init__(self):
            self.next = None

        def compute(self, rgb):
            return self.next.unsqueeze(0)

    ex = Extractor()
    for k, i in enumerate([0, 8]):
        f = S.frame(cfg, i, 16)
        dyn = np.zeros(f["depth"].shape, dtype=bool)
        dyn[100 + 5 * k: 220, 200:330] = True
        kw = dict(nvblox_mapping_config=mcfg, depth_frame=dev(f["depth"]), feature_frame=dev(f["features"]),
                  intrinsics=torch.from_numpy(f["K"]), camera_pose=torch.from_numpy(f["T_W_C"]), rgb=dev(f["rgb"]),
                  input_mask_erosion_iterations=3, valid_depth_mask_erosion_iterations=4, mapper_id=MAPPER_TO_ID.STATIC)
        for m in (a, b, c):
            m.decay()
        ia = integrate_frame(mapper=a, input_mask=dev(~dyn), **kw)
        ib = integrate_frame(mapper=b, input_mask=dev(dyn), invert_input_mask=True, **kw)
        assert torch.equal(ia["depth_mask"], ib["depth_mask"]) and torch.equal(ia["feature_mask"], ib["feature_mask"])
        assert torch.equal(ia["input_mask"], ib["input_mask"]) and set(ia.keys()) == set(ib.keys())
        ex.next = dev(f["features"])
        mcfg_c = NvbloxMappingCfg("DRILL_IN_BOX")
        mcfg_c.static_mask_erosion_iterations, mcfg_c.valid_depth_mask_erosion_iterations = 3, 4
        out = nvblox_integrate(c, mcfg_c, ex, dev(f["depth"]), torch.from_numpy(f["K"]), torch.from_numpy(f["T_W_C"]), dev(f["rgb"]),
                               dev(dyn), include_dynamic=False)
        assert torch.equal(out["STATIC"]["depth_mask"], ia["depth_mask"])
    ref = [t.clone() for t in a.feature_layer_view(0).get_all_blocks_split()]
    for m in (b, c):
        got = m.feature_layer_view(0).get_all_blocks_split()
        assert torch.equal(got[2], ref[2]) and torch.equal(got[1], ref[1]) and torch.equal(got[0].view(torch.int16), ref[0].view(torch.int16))
        ta, tb = a.tsdf_layer_view(0).get_all_blocks(), m.tsdf_layer_view(0).get_all_blocks()
        assert torch.equal(ta[0], tb[0]) and torch.equal(ta[1], tb[1])
    assert int((ref[1] > 0).sum()) > 5000


def test_mesh_topology_and_color_mesh(oracle_mod, tmp_path):
    """Triangle connectivity + per-vertex colours (Mapper.get_color_mesh, FeatureMesh.triangles()) against the oracle:
    same triangles in the same order, same colours; the surface is consistently oriented; PLY export round-trips."""
    cfg = small_cfg(4)
    orc, gpu = run_both(oracle_mod, cfg, 16, [0, 6, 12], use_mask=True, color=True, decay=True)
    ov, _ = orc.feature_mesh()
    ot, oc = orc.mesh_topology()
    mesh = gpu.get_feature_mesh(0)
    assert np.array_equal(mesh.vertices().cpu().numpy().view(np.uint32), ov.view(np.uint32))
    tris = mesh.triangles()
    assert tris.dtype == torch.int32 and np.array_equal(tris.cpu().numpy(), ot) and ot.shape[0] > 10000
    gpu.update_color_mesh(0)
    cm = gpu.get_color_mesh(0)
    assert np.array_equal(cm.vertices().cpu().numpy().view(np.uint32), ov.view(np.uint32))
    assert np.array_equal(cm.triangles().cpu().numpy(), ot)
    assert np.array_equal(cm.vertex_colors_u8().cpu().numpy(), oc) and (oc.sum(axis=1) > 0).mean() > 0.5
    assert cm.vertex_colors().dtype == torch.float32 and float(cm.vertex_colors().max()) <= 1.0
    # orientation: every directed edge inside one block is matched by its reverse at most once, never duplicated
    t = ot.astype(np.int64)
    e = np.concatenate([t[:, [0, 1]], t[:, [1, 2]], t[:, [2, 0]]])
    code = e[:, 0] * (ov.shape[0] + 1) + e[:, 1]
    uniq, counts = np.unique(code, return_counts=True)
    assert counts.max() <= 2  # (2 only at pinched ambiguous patterns)
    rev = e[:, 1] * (ov.shape[0] + 1) + e[:, 0]
    assert np.isin(rev, uniq).mean() > 0.8  # interior edges have their opposite; block borders (per-block vertices) / map boundary do not
    # normals point towards free space: for a floor seen from above that is +z on average
    tri = ov[ot]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    cam = S.camera_pose(cfg, 6)[:3, 3]
    towards_cam = ((cam[None, :] - tri.mean(axis=1)) * n).sum(axis=1)
    assert (towards_cam > 0).mean() > 0.8
    path = str(tmp_path / "nvblox_mesh_00000.ply")
    cm.save(path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    assert b"element vertex %d" % ov.shape[0] in head and b"element face %d" % ot.shape[0] in head
    assert len(body) == ov.shape[0] * 15 + ot.shape[0] * 13
    assert np.array_equal(np.frombuffer(body[:12], "<f4"), ov[0])
    # an empty map has no mesh
    gpu.clear()
    assert gpu.get_color_mesh(0) is None


def _fused_vs_oracle(oracle_mod, gpu, orc, cfg, frames, channels, k_in=3, k_depth=4, border=5, invert=False):
    from oracle import image_ops as IO

    for k, i in enumerate(frames):
        f = S.frame(cfg, i, channels) if not callable(i) else i()
        dyn = np.zeros(f["depth"].shape, dtype=bool)
        dyn[2 + k: 2 + k + f["depth"].shape[0] // 4, 3: 3 + f["depth"].shape[1] // 3] = True
        static = ~dyn
        odm, ofm = IO.frame_masks(static, f["depth"], 0.3, k_in, k_depth, border, cfg.height, cfg.width)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
        gpu.decay()
        dm, fm = gpu.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(dyn if invert else static),
                                     torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), 0.3, k_in, k_depth, border, 0,
                                     invert_input_mask=invert)
        assert np.array_equal(dm.cpu().numpy().astype(bool), odm) and np.array_equal(fm.cpu().numpy().astype(bool), ofm)
    compare_tsdf(orc, gpu)
    compare_features(orc, gpu)
    rgb, w, idx = gpu.color_layer_view(0).get_all_blocks_split()
    orgb, ow = orc.all_colors()
    assert np.array_equal(idx.cpu().numpy(), orc.block_indices(1)) and np.array_equal(w.cpu().numpy(), ow)
    assert np.array_equal(rgb.cpu().numpy(), orgb)


@pytest.mark.parametrize("invert", [False, True])
def test_fused_call_falls_back_for_unbounded_workspaces(oracle_mod, invert):
    """No workspace bounds -> view grid of tens of thousands of cells, no dense table: mmf_integrate_frame takes its
    stand-alone-kernel route (three-kernel allocation, hash index, eager decay, inverted mask through scratch)."""
    cfg = small_cfg(4)
    over = dict(workspace_bounds_type=0, max_integration_distance_m=2.5)
    _fused_vs_oracle(oracle_mod, make_mapper(16, **over), make_oracle(oracle_mod, 16, **over), cfg, [0, 6, 40, 46], 16, invert=invert)


@pytest.mark.parametrize("flip", ["raycast_walk_from_camera", "appearance_blend_division"])
def test_switchable_spec_arrangements_match_the_oracle(oracle_mod, flip):
    """The two places where the spec was arranged for the GPU are parameters of both implementations
    (tests/pin_report.py flips them): with either flipped, HIP and oracle still agree bit for bit -- and walking from the camera
    gives the very map the default gives."""
    cfg = small_cfg(4)
    over = {flip: True}
    gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 6, 40, 46, 90], 16)
    if flip == "raycast_walk_from_camera":
        ref = make_oracle(oracle_mod, 16)
        _fused_vs_oracle(oracle_mod, make_mapper(16), ref, cfg, [0, 6, 40, 46, 90], 16)
        assert np.array_equal(ref.block_indices(0), orc.block_indices(0)) and np.array_equal(ref.all_tsdf().view(np.uint32), orc.all_tsdf().view(np.uint32))


ROUND6_FLIPS = ["block_index_by_division", "view_truncation_band_marking", "bilinear_four_weight_sum"]


@pytest.mark.parametrize("route", ["bounded", "hash"])
@pytest.mark.parametrize("flips", [(f,) for f in ROUND6_FLIPS] + [tuple(ROUND6_FLIPS), ("bilinear_four_weight_sum", "fma_contraction"),
                                   tuple(ROUND6_FLIPS) + ("fma_contraction", "appearance_blend_division")])
def test_round6_spec_switches_match_the_oracle(oracle_mod, flips, route):
    """Three more recollection risks made switchable in both implementations (oracle/mmf_oracle.c orc_params):
    the block of a point by division, the second marking pass of the view calculator, bilinear samples as four weighted taps.  With
    any of them set -- alone, together, with the FMA and the blend-division switches on top -- HIP (stand-alone launches) and oracle
    agree bit for bit: block sets in allocation order, TSDF, colours, feature halves; through the fused entry point and, for a
    stream of stand-alone calls, in a bounded workspace and on the hash path."""
    cfg = small_cfg(4)
    over = {f: True for f in flips}
    if route == "hash":
        over.update(workspace_bounds_type=0, max_integration_distance_m=2.5)
    gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 6, 40, 46, 90], 16)
    orc2, gpu2 = run_both(oracle_mod, cfg, 16, [3, 9, 50], **over)
    compare_tsdf(orc2, gpu2)
    compare_features(orc2, gpu2)
    # mesh + vertex features (voxel look-ups under block_index_by_division)
    gpu.update_feature_mesh(0)
    mesh = gpu.get_feature_mesh(0)
    ov, of = orc.feature_mesh()
    assert np.array_equal(mesh.vertices().cpu().numpy(), ov) and np.array_equal(mesh.vertex_features().cpu().numpy().view(np.uint16), of.view(np.uint16))


@pytest.mark.skipif(bool(fusion_common.SPEC_FLIPS), reason="compares a flipped oracle with the default one: under MMF_SPEC_FLIPS the default IS flipped")
def test_round6_flips_are_not_no_ops(oracle_mod):
    """Each flip changes what it says it changes (and only that) on the oracle: the band marking adds blocks to the view, the four-tap
    sum moves values by rounding, the division keeps this stream's block sets (its differences live on block faces: the constructed
    case below)."""
    cfg = small_cfg(4)
    frames = [0, 6, 40, 46, 90]

    def run(**over):
        orc = make_oracle(oracle_mod, 16, **over)
        for i in frames:
            f = S.frame(cfg, i, 16)
            orc.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], None)
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], None)
        return orc

    base = run()
    band = run(view_truncation_band_marking=1)
    b0, b1 = {tuple(r) for r in base.block_indices(0).tolist()}, {tuple(r) for r in band.block_indices(0).tolist()}
    assert b0 < b1  # a strict superset
    w4 = run(bilinear_four_weight_sum=1)
    assert np.array_equal(w4.block_indices(0), base.block_indices(0))
    d = np.abs(w4.all_tsdf() - base.all_tsdf())
    assert 0.0 < float(d.max()) < 1e-5
    # division: a workspace bound whose reciprocal product lands on the other side of a block face.  floor(x * (1 / bs)) and
    # floor(x / bs) differ for SOME float32 x next to a multiple of bs = 0.08f: search one, then the two rules give other block sets.
    bs, inv = np.float32(0.08), np.float32(1.0) / np.float32(0.08)
    ks = np.arange(1, 4000, dtype=np.float32)
    for eps in (0, 1, -1):
        x = np.nextafter(ks * bs, np.float32(np.inf) if eps > 0 else np.float32(-np.inf)).astype(np.float32) if eps else (ks * bs).astype(np.float32)
        diff = np.floor(x * inv) != np.floor(x / bs)
        if diff.any():
            break
    assert diff.any(), "no float32 near a block face where the two rules differ (unexpected)"
    xf = float(x[np.argmax(diff)])
    lo = np.array([-0.37, -0.75, -0.13], dtype=np.float32)
    hi = np.array([xf, 0.75, 0.65], dtype=np.float32)
    a = make_oracle(oracle_mod, 16, ws_min=lo, ws_max=hi)
    b = make_oracle(oracle_mod, 16, ws_min=lo, ws_max=hi, block_index_by_division=1)
    T = np.eye(4, dtype=np.float32)
    T[:3, 2], T[:3, 0], T[:3, 1] = [1.0, 0.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0]
    T[:3, 3] = [xf - 1.0, 0.005, 0.305]
    depth = np.full((cfg.height, cfg.width), 1.0 + 0.01, dtype=np.float32)  # a wall just beyond the bound: rays end in its last blocks
    for o in (a, b):
        o.add_depth_frame(depth, T, cfg.intrinsics(), None)
    assert a.block_indices(0).shape != b.block_indices(0).shape or not np.array_equal(a.block_indices(0), b.block_indices(0))


def test_fused_call_on_images_too_narrow_for_the_bit_packed_masks(oracle_mod):
    """W < 16: the bit-packed mask job does not fit its scratch, the byte kernels run instead."""
    cfg = S.StreamConfig(width=12, height=40, fx=10.0, fy=10.0, cx=5.5, cy=19.5)
    _fused_vs_oracle(oracle_mod, make_mapper(8), make_oracle(oracle_mod, 8), cfg, [0, 3], 8, k_in=1, k_depth=1, border=0, invert=True)


def test_degenerate_frames(oracle_mod):
    """Frames that carry no information: all depth invalid, camera looking away from the workspace, then a normal one."""
    cfg = small_cfg(4)
    base = S.frame(cfg, 0, 16)

    def blind():
        f = dict(base)
        f["depth"] = np.zeros_like(base["depth"])
        return f

    def away():
        f = dict(S.frame(cfg, 3, 16))
        T = f["T_W_C"].copy()
        T[:3, :3] = T[:3, :3] @ np.diag([1.0, -1.0, -1.0]).astype(np.float32)  # turn around: optical axis away from the scene
        f["T_W_C"] = T
        return f

    gpu, orc = make_mapper(16), make_oracle(oracle_mod, 16)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [blind, away], 16)
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() == orc.block_indices(0).shape[0]
    assert gpu.get_color_mesh(0) is None or gpu.get_feature_mesh(0).vertices().shape[0] == orc.feature_mesh()[0].shape[0]
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [5, blind, 7], 16)
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() > 100


@pytest.mark.parametrize("route", ["fused", "hash"])
def test_voxel_centres_on_the_camera_plane(oracle_mod, route):
    """Round-5 advisor finding: the branch-free voxel loop projects EVERY voxel of a candidate block, also those on or behind the
    camera plane (p.z == 0: u = inf / NaN, a float -> int conversion of which is undefined).  A camera INSIDE the workspace, axis
    aligned, its centre exactly on a voxel centre: a whole plane of voxel centres of the camera's own (candidate) blocks has
    p.z == +-0 in float32, the planes beside it p.z = +-1 cm.  The indices are clamped in float before the conversion now; the map must
    equal the oracle's bit for bit and nothing may be read outside the depth image."""
    cfg = small_cfg(4)
    pos = np.array([-0.295, 0.005, 0.305])  # a voxel centre ((k + 0.5) cm on every axis), well inside the DRILL_IN_BOX workspace
    T = np.eye(4)
    T[:3, 2] = [1.0, 0.0, 0.0]   # optical axis = world +x (towards the sphere)
    T[:3, 0] = [0.0, -1.0, 0.0]  # image x = world -y
    T[:3, 1] = [0.0, 0.0, -1.0]  # image y = world -z
    T[:3, 3] = pos
    T = T.astype(np.float32)

    def inside():
        f = dict(S.frame(cfg, 0, 16))
        f["T_W_C"] = T
        f["depth"] = S.render_depth(cfg, T)
        return f

    f = inside()
    assert (f["depth"] > 0.3).mean() > 0.5
    # (the oracle's own arithmetic: voxel centres with p.z == 0 exist for this pose)
    xs = (np.arange(-37, 95, dtype=np.float32) + np.float32(0.5)) * np.float32(0.01)
    assert np.sum(xs == T[0, 3]) == 1
    over = {} if route == "fused" else dict(workspace_bounds_type=0, max_integration_distance_m=2.5)
    gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [inside, 0, inside, 6], 16)
    assert orc.block_indices(0).shape[0] > 50


@pytest.mark.parametrize("route", ["bounded", "hash"])
@pytest.mark.parametrize("shape", [(1, 96), (48, 1), (2, 2)])
def test_one_row_and_one_column_depth_images(oracle_mod, route, shape):
    """Images without a 2 x 2 footprint (one row, one column) and the smallest one with exactly one: the stand-alone call takes them
    (the reference's API does not forbid them).  The fused launches' branch-free voxel loop clamps a 2 x 2 footprint into the image and
    therefore needs two rows and two columns -- thinner images must be routed to the branching form (mmf_api.hip, launch_tsdf_pass_t):
    HIP == oracle bit for bit, with and without a mask, and nothing is read outside the image (a 1 x W image sits at the very end of
    its allocation here)."""
    h, w = shape
    cfg = small_cfg(4)
    f = S.frame(cfg, 2, 16)
    over = {} if route == "bounded" else dict(workspace_bounds_type=0, max_integration_distance_m=2.5)
    K = f["K"].copy()
    K[0, 2], K[1, 2] = (w - 1) / 2.0, (h - 1) / 2.0  # the principal point inside the strip
    r0, c0 = f["depth"].shape[0] // 2, f["depth"].shape[1] // 2
    depth = np.ascontiguousarray(f["depth"][r0 - h // 2: r0 - h // 2 + h, c0 - w // 2: c0 - w // 2 + w]).astype(np.float32)
    assert depth.shape == (h, w) and (depth > 0).any()
    for use_mask in (False, True):
        gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
        mask = (np.arange(h * w).reshape(h, w) % 5 != 0).astype(np.uint8) if use_mask else None
        for _ in range(2):
            orc.add_depth_frame(depth, f["T_W_C"], K, mask)
            gpu.add_depth_frame(dev(depth), torch.from_numpy(f["T_W_C"]), torch.from_numpy(K), None if mask is None else dev(mask), 0)
        assert orc.block_indices(0).shape[0] > 0
        compare_tsdf(orc, gpu)


@pytest.mark.parametrize("route", ["fused", "hash", "no_max_distance"])
def test_non_finite_depth_pixels(oracle_mod, route):
    """What a simulated depth camera delivers for rays that hit nothing: +inf (Isaac Lab's distance_to_image_plane), and NaN /
    -inf / negative values from upstream arithmetic.  The spec (oracle): a tap is valid iff depth > 0 -- NaN, -inf and negative
    pixels are no measurement; +inf is one: the ray is walked to the maximum integration distance and the voxels in front are
    free space wherever the weighting function gives inf a weight; with no maximum distance set (a bounded workspace without
    one: the third route) a +inf pixel casts no ray.  HIP == oracle bit for bit, nothing non-finite in the map."""
    cfg = small_cfg(4)
    rng = np.random.default_rng(7)

    def frame(i):
        def make():
            f = dict(S.frame(cfg, i, 16))
            d = f["depth"].copy()
            h, w = d.shape
            d[: h // 5] = np.inf                                 # sky
            d[h // 2: h // 2 + 6, w // 3: w // 3 + 9] = np.inf    # a hole inside the scene (bilinear taps straddle its rim)
            ys, xs = rng.integers(0, h, 40), rng.integers(0, w, 40)
            d[ys[:20], xs[:20]] = np.nan
            d[ys[20:30], xs[20:30]] = -np.inf
            d[ys[30:], xs[30:]] = -1.0
            f["depth"] = d
            return f
        return make

    over = {"fused": {}, "hash": dict(workspace_bounds_type=0, max_integration_distance_m=2.5), "no_max_distance": dict(max_integration_distance_m=0.0)}[route]
    gpu, orc = make_mapper(16, **over), make_oracle(oracle_mod, 16, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [frame(0), frame(6), frame(40)], 16)
    vox, idx = gpu.tsdf_layer_view(0).get_all_blocks()
    assert idx.shape[0] > 50 and bool(torch.isfinite(vox).all())
    # the stand-alone call and a constant weight (inf gets a weight: free space in front of the hole)
    over2 = dict(over, weighting_mode=0)  # (the oracle's numbering: kConstantWeight)
    gpu, orc = make_mapper(16, **over2), make_oracle(oracle_mod, 16, **over2)
    for i in (0, 6):
        f = frame(i)()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
        gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    compare_tsdf(orc, gpu)
    vox, _ = gpu.tsdf_layer_view(0).get_all_blocks()
    assert bool(torch.isfinite(vox).all())


def test_pool_exhaustion_is_reported():
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import BlockMemoryPoolParams, MapperParams

    mp = MapperParams()
    pool = BlockMemoryPoolParams()
    pool.num_preallocated_blocks = 64  # far fewer than one 160x120 view touches
    mp.set_block_memory_pool_params(pool)
    m = Mapper(voxel_sizes_m=0.01, mapper_parameters=mp, feature_channels=16)
    cfg = small_cfg(4)
    f = S.frame(cfg, 0, 16)
    m.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    with pytest.raises(RuntimeError, match="pool exhausted"):
        m.tsdf_layer_view(0).num_allocated_blocks()


def test_wg_trace_hooks_are_compiled_out_of_the_product_build():
    """mmf_debug_wg_trace: the per-workgroup timeline exists in the instrumented build only (make WG_TRACE=1); the product
    library refuses a buffer (and accepts switching the trace off)."""
    import torch

    from nvblox_mindmap_amd import _lib

    if os.environ.get("MMF_LIB", "libmmfusion.so") != "libmmfusion.so":
        pytest.skip("instrumented library selected")
    buf = torch.zeros(3 * 6 * 8192, dtype=torch.int64, device="cuda")
    with pytest.raises(RuntimeError, match="WG_TRACE"):
        _lib.check(_lib.lib().mmf_debug_wg_trace(_lib.dptr(buf), 6 * 8192), "mmf_debug_wg_trace")
    _lib.check(_lib.lib().mmf_debug_wg_trace(None, 0), "mmf_debug_wg_trace")


def test_lazy_decay_deallocates_in_fused_frames_and_survives_interleaved_standalone_calls(oracle_mod):
    """Strong decay inside fused frames: the deallocations come from the per-block maximum weights (light path, the W *= f rides
    in the TSDF pass); a stand-alone add_depth_frame / eager decay in between marks those maxima stale, the next fused frame
    takes the full decay pass, the one after it the light path again.  Blocks, order and values equal the oracle's throughout."""
    cfg = small_cfg(4)
    over = dict(tsdf_decay_factor=0.2, decayed_weight_threshold=1e-2)
    gpu, orc = make_mapper(8, **over), make_oracle(oracle_mod, 8, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 60, 120, 180, 0], 8)
    n_after_orbit = orc.num_blocks(0)
    f = S.frame(cfg, 30, 8)
    for m in (orc, gpu):
        m.decay()
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
    gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    compare_tsdf(orc, gpu)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [90, 150, 150], 8)
    for _ in range(3):  # and an eager decay (a reader in between) after light frames
        orc.decay()
        gpu.decay()
        assert gpu.tsdf_layer_view(0).num_allocated_blocks() == orc.num_blocks(0)
    compare_tsdf(orc, gpu)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [10, 200], 8)
    assert orc.num_blocks(0) > 0 and n_after_orbit > 0


def test_merged_allocation_and_tsdf_launch_under_pool_exhaustion():
    """k_alloc_tsdf with fewer pool slots than the view needs: the allocation grants what fits, the workgroups that wait for new
    blocks stop at the granted count (no hang), the error is reported, and later frames keep running on the full pool."""
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import BlockMemoryPoolParams, MapperParams, ViewCalculatorParams

    mp = MapperParams()
    pool = BlockMemoryPoolParams()
    pool.num_preallocated_blocks = 96
    mp.set_block_memory_pool_params(pool)
    vc = ViewCalculatorParams()
    vc.workspace_bounds_type = "kBoundingBox"
    for k, v in zip(("x", "y"), (0, 1)):
        setattr(vc, f"workspace_bounds_min_corner_{k}_m", float(REF_PARAMS["ws_min"][v]))
        setattr(vc, f"workspace_bounds_max_corner_{k}_m", float(REF_PARAMS["ws_max"][v]))
    vc.workspace_bounds_min_height_m = float(REF_PARAMS["ws_min"][2])
    vc.workspace_bounds_max_height_m = float(REF_PARAMS["ws_max"][2])
    mp.set_view_calculator_params(vc)
    m = Mapper(voxel_sizes_m=0.01, mapper_parameters=mp, feature_channels=8)
    cfg = small_cfg(4)
    for i in (0, 5, 10):
        f = S.frame(cfg, i, 8)
        m.decay()
        m.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(np.ones(f["depth"].shape, dtype=bool)),
                          torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), 0.3, 1, 1, 0, 0)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="pool exhausted"):
        m.tsdf_layer_view(0).num_allocated_blocks()


def test_merged_launch_with_churn_matches_oracle(oracle_mod):
    """Every frame of this sequence deallocates blocks (strong decay) and allocates others while the camera jumps around the
    orbit: k_alloc_tsdf's pass over the existing blocks runs beside an allocation that reuses the freed slots, its new-block
    workgroups take dozens to hundreds of published blocks.  Block sets, order and values equal the oracle's."""
    cfg = small_cfg(4)
    over = dict(tsdf_decay_factor=0.3, decayed_weight_threshold=5e-2)
    gpu, orc = make_mapper(8, **over), make_oracle(oracle_mod, 8, **over)
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 90, 10, 200, 100, 20, 300, 110, 30], 8)


@pytest.mark.skipif(fusion_common.NOT_DEFAULT_ROUTE, reason=fusion_common.ROUTE_SKIP_REASON)
def test_hand_over_recovery_yields_the_oracle_map(oracle_mod, monkeypatch):
    """MMF_DEBUG_FORCE_ALLOC_TIMEOUT=1: every second waiter workgroup of k_alloc_tsdf "times out" at once and abandons its rounds;
    the workgroup that terminates last sweeps them.  Same churn sequence as above (hundreds of new blocks per frame): block
    sets, order and values equal the oracle's, and the sweeper really did the work."""
    monkeypatch.setenv("MMF_DEBUG_FORCE_ALLOC_TIMEOUT", "1")
    cfg = small_cfg(4)
    over = dict(tsdf_decay_factor=0.3, decayed_weight_threshold=5e-2)
    gpu, orc = make_mapper(8, **over), make_oracle(oracle_mod, 8, **over)
    monkeypatch.delenv("MMF_DEBUG_FORCE_ALLOC_TIMEOUT")
    _fused_vs_oracle(oracle_mod, gpu, orc, cfg, [0, 90, 10, 200, 100, 20, 300, 110, 30], 8)
    assert gpu.debug_alloc_recoveries(0) > 100
    # the stand-alone depth call shares the launch
    f = S.frame(cfg, 150, 8)
    for m in (orc, gpu):
        m.decay()
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
    gpu.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    compare_tsdf(orc, gpu)
    normal = make_mapper(8, **over)
    f = S.frame(cfg, 0, 8)
    normal.add_depth_frame(dev(f["depth"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), None, 0)
    assert normal.debug_alloc_recoveries(0) == 0  # ordinary operation never needs the sweeper


@pytest.mark.skipif(fusion_common.NOT_DEFAULT_ROUTE, reason=fusion_common.ROUTE_SKIP_REASON)
def test_hand_over_failure_is_reported_once_and_cleared(monkeypatch):
    """MMF_DEBUG_FORCE_ALLOC_TIMEOUT=2: the sweeper gives up as well -> the map is incomplete.  The next call on the mapper
    fails with MMF_ERR_BAD_STATE (no integration on top of a broken map), the error is then cleared, clear() gives a usable
    mapper again."""
    monkeypatch.setenv("MMF_DEBUG_FORCE_ALLOC_TIMEOUT", "2")
    gpu = make_mapper(8)
    monkeypatch.delenv("MMF_DEBUG_FORCE_ALLOC_TIMEOUT")
    cfg = small_cfg(4)

    def frame(i):
        f = S.frame(cfg, i, 8)
        gpu.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), dev(np.ones(f["depth"].shape, dtype=bool)),
                            torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), 0.3, 1, 1, 0, 0)

    frame(0)  # asynchronous: the call itself succeeds
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="hand-over"):
        frame(5)
    torch.cuda.synchronize()
    gpu.clear()
    assert gpu.tsdf_layer_view(0).num_allocated_blocks() == 0  # reported once: no error left after clear()
    frame(0)
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="hand-over"):  # also surfaces at a synchronising call
        gpu.update_feature_mesh(0)


def test_grid_tag_wraps_and_merged_path_equals_separate_launches(monkeypatch):
    """270 x (decay, fused frame, stand-alone add_depth_frame of another view): mapper A runs the merged k_alloc_tsdf launch in
    both calls (tagged grid flags: the 8-bit tag wraps twice; light decay); mapper B is created with MMF_NO_ALLOC_TSDF=1 and keeps
    the separate allocation / TSDF launches (cleared grid flags, eager decay before the stand-alone call).  Same blocks, same
    order, same bits."""
    cfg = small_cfg(4)
    over = dict(tsdf_decay_factor=0.9, decayed_weight_threshold=0.3)
    a = make_mapper(8, **over)
    monkeypatch.setenv("MMF_NO_ALLOC_TSDF", "1")
    b = make_mapper(8, **over)
    monkeypatch.delenv("MMF_NO_ALLOC_TSDF")
    ones = dev(np.ones((cfg.height, cfg.width), dtype=bool))
    frames = [S.frame(cfg, i, 8) for i in range(0, 360, 12)]
    for k in range(270):
        f, g = frames[(k * 7) % len(frames)], frames[(k * 11 + 3) % len(frames)]
        for m in (a, b):
            m.decay()
            m.integrate_frame(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), ones, torch.from_numpy(f["T_W_C"]),
                              torch.from_numpy(f["K"]), 0.3, 2, 2, 0, 0)
            if k % 3 == 0:
                m.decay()
            m.add_depth_frame(dev(g["depth"]), torch.from_numpy(g["T_W_C"]), torch.from_numpy(g["K"]), None, 0)
    ta, ia = a.tsdf_layer_view(0).get_all_blocks()[:2]
    tb, ib = b.tsdf_layer_view(0).get_all_blocks()[:2]
    assert ia.shape[0] > 50 and torch.equal(ia, ib) and torch.equal(ta, tb)
    fa, fb = a.feature_layer_view(0).get_all_blocks(), b.feature_layer_view(0).get_all_blocks()
    assert all(torch.equal(x, y) for x, y in zip(fa, fb))


def test_two_mappers_in_one_call_equal_sequential_and_oracle(oracle_mod, monkeypatch):
    """nvblox_integrate(include_dynamic=True) through mmf_integrate_frame_multi (both mappers' frames as roles of the same five
    launches: k_front2 ... k_feature_flat2) against the two integrate_frame calls in sequence -- same maps, same masks, bit for
    bit -- and against the oracle driven mapper by mapper.  Full size (640x480, reference erosions), decay every frame, the
    dynamic region moves, so both mappers allocate and deallocate along the way."""
    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg
    from oracle import image_ops as IO

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    C = 16
    a, b = (H.get_nvblox_mapper(mcfg, feature_channels=C) for _ in range(2))
    orcs = [make_oracle(oracle_mod, C, tsdf_decay_factor=mcfg.tsdf_decay_factor) for _ in range(2)]

    class Extractor:
        def compute(self, rgb):
            return self.next.unsqueeze(0)

    ex = Extractor()
    for k, i in enumerate([0, 6, 12, 40, 46, 52]):
        f = S.frame(cfg, i, C)
        dyn = np.zeros(f["depth"].shape, dtype=bool)
        dyn[80 + 7 * k: 300, 150 + 9 * k: 420] = True
        ex.next = dev(f["features"])
        outs = []
        for m, pair in ((a, True), (b, False)):
            monkeypatch.setattr(H, "PAIR_MAPPERS", pair)
            m.decay()
            outs.append(H.nvblox_integrate(m, mcfg, ex, dev(f["depth"]), torch.from_numpy(f["K"]), torch.from_numpy(f["T_W_C"]),
                                           dev(f["rgb"]), dev(dyn), include_dynamic=True))
        for name in ("STATIC", "DYNAMIC"):
            assert set(outs[0][name].keys()) == set(outs[1][name].keys())
            for key in ("depth_mask", "feature_mask", "input_mask"):
                assert torch.equal(outs[0][name][key], outs[1][name][key]), (name, key)
        for orc, mask, k_in in ((orcs[0], ~dyn, mcfg.static_mask_erosion_iterations), (orcs[1], dyn, mcfg.dynamic_mask_erosion_iterations)):
            odm, ofm = IO.frame_masks(mask, f["depth"], mcfg.min_integration_distance_m, k_in, mcfg.valid_depth_mask_erosion_iterations,
                                      mcfg.feature_mask_border_percent, cfg.height, cfg.width)
            orc.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
    torch.cuda.synchronize()
    for mid, orc in ((MAPPER_TO_ID.STATIC, orcs[0]), (MAPPER_TO_ID.DYNAMIC, orcs[1])):
        ta, tb = a.tsdf_layer_view(mid).get_all_blocks(), b.tsdf_layer_view(mid).get_all_blocks()
        assert ta[1].shape[0] > 20 and torch.equal(ta[0], tb[0]) and torch.equal(ta[1], tb[1])
        fa, fb = a.feature_layer_view(mid).get_all_blocks_split(), b.feature_layer_view(mid).get_all_blocks_split()
        assert all(torch.equal(x, y) for x, y in zip(fa, fb))
        ca, cb = a.color_layer_view(mid).get_all_blocks_split(), b.color_layer_view(mid).get_all_blocks_split()
        assert all(torch.equal(x, y) for x, y in zip(ca, cb))
        assert np.array_equal(ta[1].cpu().numpy(), orc.block_indices(0))
        assert np.abs(ta[0].cpu().numpy() - orc.all_tsdf()).max() <= 1e-5
        fo, wo = orc.all_features()
        assert np.array_equal(fa[2].cpu().numpy(), orc.block_indices(2)) and np.array_equal(fa[1].cpu().numpy(), wo)
        assert np.abs(fa[0].cpu().numpy().astype(np.float32) - fo.astype(np.float32)).max() <= 1e-5
    assert a.stats(MAPPER_TO_ID.DYNAMIC)["feature_voxels_updated"] > 0 and a.stats(MAPPER_TO_ID.STATIC)["feature_voxels_updated"] > 0


def test_integrate_frame_multi_argument_checks():
    from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper
    from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg

    m = get_nvblox_mapper(NvbloxMappingCfg("DRILL_IN_BOX"), feature_channels=8)
    cfg = small_cfg(4)
    f = S.frame(cfg, 0, 8)
    ones = dev(np.ones(f["depth"].shape, dtype=bool))
    job = {"mapper_id": 0, "input_mask": ones, "input_mask_erosion_iterations": 1, "valid_depth_mask_erosion_iterations": 1}
    with pytest.raises(RuntimeError, match="different mapper"):
        m.integrate_frame_multi(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]),
                                0.3, 0, [job, dict(job)])
    # a single job and a trio (pair + single) are fine
    out = m.integrate_frame_multi(dev(f["depth"]), dev(f["rgb"]), dev(f["features"]), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]),
                                  0.3, 0, [job])
    assert len(out) == 1 and out[0][0].shape == f["depth"].shape
