"""CPU tests: the oracle's numpy restatements against the golden vectors generated from the reference,
the C oracle's internal consistency, and the host logic that needs no GPU."""
import numpy as np
import pytest
import torch

from oracle import image_ops as IO

GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden")


def load(name):
    return np.load(f"{GOLD}/{name}.npz")


def unpack(bits, shape):
    n = int(np.prod(shape))
    return np.unpackbits(bits)[:n].reshape(shape).astype(bool)


def test_erode_mask_matches_reference():
    g = load("masks")
    H, W = g["shape"]
    for nm in ("random", "struct"):
        m = unpack(g[nm], (H, W))
        for k in (1, 3, 10, 17, 20):
            assert np.array_equal(IO.erode_mask(m, k), unpack(g[f"erode_{nm}_{k}"], (H, W))), (nm, k)


def test_border_mask_matches_reference():
    g = load("masks")
    for (h, w, pct) in ((48, 64, 5), (512, 512, 5), (480, 640, 5), (10, 12, 5), (100, 30, 7)):
        assert np.array_equal(IO.get_border_mask(h, w, pct), unpack(g[f"border_{h}_{w}_{pct}"], (h, w)))


def test_frame_masks_match_reference_composition():
    g = load("masks")
    for nm in ("same", "up", "down", "odd", "sq"):
        h, w, hf, wf, k_in, k_depth, pct = [int(x) for x in g[f"fm_{nm}_params"]]
        im = unpack(g[f"fm_{nm}_in"], (h, w))
        dm, fm = IO.frame_masks(im, g[f"fm_{nm}_depth"], 0.30, k_in, k_depth, pct, hf, wf)
        assert np.array_equal(dm, unpack(g[f"fm_{nm}_depth_mask"], (h, w))), nm
        assert np.array_equal(fm, unpack(g[f"fm_{nm}_feature_mask"], (hf, wf))), nm


def test_backprojection_matches_reference():
    g = load("backprojection")
    for nm in "abc":
        out = IO.get_camera_pointcloud(g[f"{nm}_K"], g[f"{nm}_depth"], g[f"{nm}_pos"], g[f"{nm}_quat"])
        ref = g[f"{nm}_out"]
        assert out.shape == ref.shape
        # float32 matmul accumulation order differs between BLAS back ends: 1e-5 relative to the depth scale
        assert np.abs(out - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
        assert np.allclose(IO.pose_to_homo(np.concatenate([g[f"{nm}_pos"], g[f"{nm}_quat"]], axis=1)), g[f"{nm}_homo"], atol=1e-6)


def test_feature_upsample_matches_reference():
    g = load("feature_upsample")
    for nm in "abcd":
        ref32, ref16 = g[f"{nm}_out_f32"], g[f"{nm}_out"]
        hf, wf, cpad = ref32.shape
        out = IO.upsample_features(g[f"{nm}_low"], hf, wf, cpad)
        # torch's CPU kernel interpolates rows then columns; a few float32 ulps of difference are expected
        assert np.abs(out - ref32).max() <= 2e-6 * max(1.0, np.abs(ref32).max())
        # after the f16 cast at most one f16 ulp apart (only where the f32 values straddle a rounding boundary)
        d = np.abs(out.astype(np.float16).astype(np.float32) - ref16.astype(np.float32))
        assert d.max() <= 2.0 ** -10 * max(1.0, np.abs(ref32).max())


# ---- host logic of the product that needs no GPU -------------------------------------------------
def test_vertex_sampling_matches_reference():
    from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod, sample_to_n_vertices

    g = load("vertex_sampling")
    for nm in ("down", "equal", "pad"):
        verts, feats, N = torch.from_numpy(g[f"{nm}_verts"]), torch.from_numpy(g[f"{nm}_feats"]), int(g[f"{nm}_N"])
        for method in VertexSamplingMethod:
            v, f, m = sample_to_n_vertices(verts, feats, N, method, seed=7)
            assert np.array_equal(v.numpy(), g[f"{nm}_{method.value}_v"]), (nm, method)
            assert np.array_equal(f.numpy(), g[f"{nm}_{method.value}_f"]), (nm, method)
            assert np.array_equal(m.numpy(), g[f"{nm}_{method.value}_m"]), (nm, method)


def test_vertex_sampling_known_answers():
    """The known-answer checks of the reference's own tests (mindmap/tests/test_vertex_sampling.py)."""
    from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod, sample_to_n_vertices

    v, f = torch.rand(10, 3), torch.rand(10, 4)
    vv, ff, m = sample_to_n_vertices(v, f, 16, VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
    assert vv.shape == (16, 3) and ff.shape == (16, 4) and m.sum() == 10
    assert torch.all(vv[~m] == 0) and torch.all(ff[~m] == 0) and torch.equal(vv[:10], v)
    vv, ff, m = sample_to_n_vertices(v, f, 4, VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, seed=3)
    assert vv.shape == (4, 3) and m.all()
    rows = {tuple(r.tolist()) for r in v}
    assert all(tuple(r.tolist()) in rows for r in vv) and len({tuple(r.tolist()) for r in vv}) == 4
    vv, _, _ = sample_to_n_vertices(v, f, 3, VertexSamplingMethod.LOWEST)
    assert torch.equal(vv[:, 2], torch.sort(v[:, 2], descending=True).values[:3])  # reference keeps the HIGHEST z


def test_pose_to_homo_matches_reference():
    from nvblox_mindmap_amd.geometry.transforms import pose_to_homo, quaternion_wxyz_to_matrix

    g = load("backprojection")
    for nm in "abc":
        poses = torch.from_numpy(np.concatenate([g[f"{nm}_pos"], g[f"{nm}_quat"]], axis=1))
        assert np.array_equal(pose_to_homo(poses).numpy(), g[f"{nm}_homo"])
    assert pose_to_homo(poses[0]).shape == (1, 4, 4)
    q = load("quaternion")
    assert np.allclose(quaternion_wxyz_to_matrix(torch.from_numpy(q["q"])).numpy(), q["R"], atol=1e-6)


def test_downscale_and_border_mask_host():
    from nvblox_mindmap_amd.image_processing.image_mask_operations import downscale_mask, get_border_mask

    g = load("masks")
    m = torch.from_numpy(unpack(g["downscale_in"], (2, 1, 16, 24)))
    assert np.array_equal(downscale_mask(m, 2).numpy(), unpack(g["downscale_out_2"], (2, 1, 8, 12)))
    assert np.array_equal(downscale_mask(m, 4).numpy(), unpack(g["downscale_out_4"], (2, 1, 4, 6)))
    # the known-pattern test of the reference (mindmap/tests/test_image_mask_operations.py:22-38)
    k = torch.zeros((1, 1, 4, 4), dtype=torch.bool)
    k[0, 0] = torch.tensor([[1, 1, 0, 0], [1, 1, 0, 0], [0, 0, 0, 2], [1, 0, 1, 1]], dtype=torch.bool)
    d = downscale_mask(k, 2)
    assert d.shape == (1, 1, 2, 2) and d[0, 0].tolist() == [[True, False], [False, False]]
    for (h, w, pct) in ((48, 64, 5), (512, 512, 5), (10, 12, 5)):
        bm, bh, bw = get_border_mask((h, w), pct, "cpu")
        assert np.array_equal(bm.numpy(), unpack(g[f"border_{h}_{w}_{pct}"], (h, w)))
        assert [bh, bw] == g[f"border_{h}_{w}_{pct}_hw"].tolist()


# ---- C oracle self-consistency ----------------------------------------------------------------------
def test_oracle_half_conversion(oracle_mod):
    L = oracle_mod.lib()
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(4000).astype(np.float32) * s for s in (1e-8, 1e-5, 1e-3, 1.0, 100.0, 7e4)])
    vals = np.concatenate([vals, np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e9, np.inf, -np.inf, 2.0 ** -24, 2.0 ** -25,
                                           2.0 ** -25 * 1.0001, 2.0 ** -14, 6.1e-5], dtype=np.float32)])
    for v in vals:
        h = L.orc_f2h(float(v))
        assert h == int(np.float32(v).astype(np.float16).view(np.uint16)), v
    for h in rng.integers(0, 65536, size=3000):
        f = np.uint16(h).view(np.float16).astype(np.float32)
        o = np.float32(L.orc_h2f(int(h)))
        assert (np.isnan(f) and np.isnan(o)) or f == o


def test_oracle_fusion_geometry(oracle_mod):
    """The restated algorithm reconstructs the analytic scene: surface vertices lie on the ground plane / sphere /
    box within a voxel, synthetic depth agrees with the rendered depth, features come out near their inputs."""
    from nvblox_mindmap_amd import synthetic as S

    from fusion_common import make_oracle, small_cfg

    cfg = small_cfg(4)
    orc = make_oracle(oracle_mod, 8)
    for i in (0, 10, 20):
        f = S.frame(cfg, i, 8)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"])
    v, feats = orc.feature_mesh()
    assert v.shape[0] > 1000
    d_plane = np.abs(v[:, 2])
    d_sphere = np.abs(np.linalg.norm(v - S.SPHERE_C, axis=1) - S.SPHERE_R)
    q = np.maximum(np.maximum(S.BOX_MIN - v, v - S.BOX_MAX), 0)
    inside = np.all((v >= S.BOX_MIN) & (v <= S.BOX_MAX), axis=1)
    d_box = np.where(inside, np.min(np.minimum(v - S.BOX_MIN, S.BOX_MAX - v), axis=1), np.linalg.norm(q, axis=1))
    d = np.minimum(np.minimum(d_plane, d_sphere), d_box)
    assert np.quantile(d, 0.95) < 0.012, np.quantile(d, 0.95)
    sd = orc.synthetic_depth()
    full = f["depth"][2::4, 2::4][: sd.shape[0], : sd.shape[1]]
    ok = (sd > 0) & (full > 0)
    assert ok.mean() > 0.3
    assert np.median(np.abs(sd[ok] - full[ok])) < 0.01
    # idempotence: extracting the mesh twice gives the same vertices
    v2, _ = orc.feature_mesh()
    assert np.array_equal(v, v2)
    # determinism: a second oracle run gives identical bits
    orc2 = make_oracle(oracle_mod, 8)
    for i in (0, 10, 20):
        f = S.frame(cfg, i, 8)
        orc2.decay()
        orc2.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
        orc2.add_feature_frame(f["features"], f["T_W_C"], f["K"])
    assert np.array_equal(orc.all_tsdf().view(np.uint32), orc2.all_tsdf().view(np.uint32))


def test_oracle_edge_cases(oracle_mod):
    """Empty / fully-masked / out-of-workspace input allocates nothing; clear() empties the map."""
    from nvblox_mindmap_amd import synthetic as S

    from fusion_common import make_oracle, small_cfg

    cfg = small_cfg(8)
    orc = make_oracle(oracle_mod, 8)
    f = S.frame(cfg, 0, 8)
    orc.add_depth_frame(np.zeros_like(f["depth"]), f["T_W_C"], f["K"])
    assert orc.num_blocks(0) == 0
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], np.zeros(f["depth"].shape, np.uint8))
    assert orc.num_blocks(0) == 0
    orc.add_feature_frame(f["features"], f["T_W_C"], f["K"])
    assert orc.num_blocks(2) == 0 and orc.feature_mesh()[0].shape[0] == 0
    T_far = f["T_W_C"].copy()
    T_far[:3, 3] += 100.0
    orc.add_depth_frame(f["depth"], T_far, f["K"])
    assert orc.num_blocks(0) == 0
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
    n = orc.num_blocks(0)
    assert n > 0
    idx = orc.block_indices(0)
    lo = np.floor(S.DRILL_IN_BOX_AABB_MIN / np.float32(0.08)).astype(int)
    hi = np.floor(S.DRILL_IN_BOX_AABB_MAX / np.float32(0.08)).astype(int)
    assert np.all(idx >= lo) and np.all(idx <= hi)
    assert len({tuple(r) for r in idx.tolist()}) == n  # no duplicates
    orc.clear()
    assert orc.num_blocks(0) == 0
