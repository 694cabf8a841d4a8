"""GPU parity of the image-side HIP kernels: against the golden vectors generated from the reference's
Python and against the numpy oracle at the full benchmark sizes."""
import os

import numpy as np
import pytest
import torch

from oracle import image_ops as IO

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(f"{GOLD}/{name}.npz")


def unpack(bits, shape):
    return np.unpackbits(bits)[: int(np.prod(shape))].reshape(shape).astype(bool)


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_erode_mask_golden():
    from nvblox_mindmap_amd.image_processing import erode_mask

    g = load("masks")
    H, W = g["shape"]
    for nm in ("random", "struct"):
        m = cu(unpack(g[nm], (H, W)))
        for k in (1, 3, 10, 17, 20):
            out = erode_mask(m, iterations=k)
            assert out.dtype == torch.bool
            assert np.array_equal(out.cpu().numpy(), unpack(g[f"erode_{nm}_{k}"], (H, W))), (nm, k)


def test_frame_masks_golden():
    from nvblox_mindmap_amd.image_processing import depth_mask, feature_mask, frame_masks

    g = load("masks")
    for nm in ("same", "up", "down", "odd", "sq"):
        h, w, hf, wf, k_in, k_depth, pct = [int(x) for x in g[f"fm_{nm}_params"]]
        im, d = cu(unpack(g[f"fm_{nm}_in"], (h, w))), cu(g[f"fm_{nm}_depth"])
        dm, fm = frame_masks(im, d, 0.30, k_in, k_depth, pct, (hf, wf))
        assert np.array_equal(dm.cpu().numpy().astype(bool), unpack(g[f"fm_{nm}_depth_mask"], (h, w))), nm
        assert np.array_equal(fm.cpu().numpy().astype(bool), unpack(g[f"fm_{nm}_feature_mask"], (hf, wf))), nm
        # the stand-alone entry points give the same answers
        assert torch.equal(depth_mask(im, d, 0.30), dm)
        assert torch.equal(feature_mask(im, d, 0.30, k_in, k_depth, pct, (hf, wf)), fm)


@pytest.mark.parametrize("shape", [(480, 640, 480, 640), (512, 512, 512, 512), (480, 640, 240, 320), (33, 47, 40, 50), (100, 4100, 100, 4100)])
def test_frame_masks_vs_oracle_full_size(shape):
    """Benchmark-sized (and awkward) images with the reference's erosion counts (17 / 20) vs the numpy oracle."""
    from nvblox_mindmap_amd.image_processing import frame_masks

    h, w, hf, wf = shape
    rng = np.random.default_rng(h * 7 + w)
    im = rng.uniform(size=(h, w)) > 0.00005
    im[h // 3: h // 3 + 5, w // 4: w // 4 + 9] = False
    d = rng.uniform(0.5, 2.0, size=(h, w)).astype(np.float32)
    d[rng.uniform(size=(h, w)) > 0.99995] = 0.0
    odm, ofm = IO.frame_masks(im, d, 0.30, 17, 20, 5, hf, wf)
    dm, fm = frame_masks(cu(im), cu(d), 0.30, 17, 20, 5, (hf, wf))
    assert np.array_equal(dm.cpu().numpy().astype(bool), odm)
    assert np.array_equal(fm.cpu().numpy().astype(bool), ofm)
    assert 0.02 < ofm.mean() < 0.98 or min(h, w) < 64


def test_backprojection_golden():
    from nvblox_mindmap_amd.image_processing import backproject_depth_to_pointcloud, get_camera_pointcloud

    g = load("backprojection")
    for nm in "abc":
        K, depth, pos, quat = cu(g[f"{nm}_K"]), cu(g[f"{nm}_depth"]), cu(g[f"{nm}_pos"]), cu(g[f"{nm}_quat"])
        out = get_camera_pointcloud(K, depth, pos, quat)
        ref = g[f"{nm}_out"]
        assert tuple(out.shape) == ref.shape
        # north_star's tolerance: 1e-5 ABSOLUTE (metres), float32 against the reference's BLAS-ordered accumulation
        assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5
        # unbatched call signature
        out1 = get_camera_pointcloud(K[0], depth[0], pos[0], quat[0])
        assert out1.shape == ref.shape[1:] and torch.equal(out1, out[0])
        B, H, W = depth.shape
        from nvblox_mindmap_amd.geometry import pose_to_homo

        flat = backproject_depth_to_pointcloud(depth, K, pose_to_homo(torch.cat([pos, quat], dim=1)))
        assert flat.shape == (B, H * W, 3)
        assert torch.equal(flat.permute(0, 2, 1).reshape(B, 3, H, W), out)


def test_backprojection_full_size_vs_oracle():
    from nvblox_mindmap_amd.image_processing import get_camera_pointcloud

    rng = np.random.default_rng(5)
    B, H, W = 4, 512, 512
    depth = rng.uniform(0.0, 4.0, size=(B, H, W)).astype(np.float32)
    K = np.tile(np.array([[586.4, 0, 256.0], [0, 586.4, 256.0], [0, 0, 1]], dtype=np.float32), (B, 1, 1))
    pos = rng.uniform(-1, 1, size=(B, 3)).astype(np.float32)
    quat = rng.standard_normal((B, 4)).astype(np.float32)
    quat /= np.linalg.norm(quat, axis=1, keepdims=True)
    ref = IO.get_camera_pointcloud(K, depth, pos, quat)
    out = get_camera_pointcloud(cu(K), cu(depth), cu(pos), cu(quat)).cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-5  # absolute, metres (coordinates reach ~5 m here)


def test_feature_upsample_golden():
    from nvblox_mindmap_amd.image_processing import upsample_features

    g = load("feature_upsample")
    for nm in "abcd":
        ref16, ref32 = g[f"{nm}_out"], g[f"{nm}_out_f32"]
        hf, wf, cpad = ref16.shape
        out = upsample_features(cu(g[f"{nm}_low"]), (hf, wf), cpad)
        assert out.dtype == torch.float16 and tuple(out.shape) == (hf, wf, cpad)
        o = out.cpu().numpy()
        c = g[f"{nm}_low"].shape[0]
        assert np.all(o[..., c:] == 0)
        # within one f16 ulp of the reference chain (interpolate -> rearrange -> pad -> .to(float16))
        assert np.abs(o.astype(np.float32) - ref32).max() <= 2.0 ** -10 * max(1.0, np.abs(ref32).max())
        assert (o.view(np.uint16) == ref16.view(np.uint16)).mean() > 0.99


def test_feature_upsample_reference_shape_vs_oracle():
    """16x16x384 backbone output -> 512x512x768 f16 (the reference's real shape, C_pad = 768)."""
    from nvblox_mindmap_amd.image_processing import upsample_features

    rng = np.random.default_rng(2)
    low = rng.standard_normal((384, 16, 16)).astype(np.float32)
    out = upsample_features(cu(low), (512, 512), 768).cpu().numpy()
    ref = IO.upsample_features(low, 512, 512, 768)
    assert np.abs(out.astype(np.float32) - ref).max() <= 2.0 ** -10 * np.abs(ref).max()
    assert (out.view(np.uint16) == ref.astype(np.float16).view(np.uint16)).mean() > 0.999
