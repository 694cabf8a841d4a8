"""On-disk formats on the CPU: zstd binding, vertex-feature files, depth / rgb PNG, pose and
intrinsics files, the map container, the per-item transforms."""
import os
import pickle

import numpy as np
import pytest
import torch

from nvblox_mindmap_amd.data_loading.sample_transformer import DepthTransformer, RgbTransformer
from nvblox_mindmap_amd.io import dataset_files as D
from nvblox_mindmap_amd.io import zstd
from nvblox_mindmap_amd.io.map_file import read_map_file, write_map_file


def test_zstd_matches_an_independent_codec():
    pa = pytest.importorskip("pyarrow")  # bundles its own zstd: an independent implementation of the format
    rng = np.random.default_rng(0)
    for data in (b"", b"a", rng.integers(0, 40, 300000, dtype=np.uint8).tobytes(), os.urandom(70000)):
        c = zstd.compress(data, 1)
        assert zstd.decompress(c) == data
        if data:
            assert pa.decompress(c, decompressed_size=len(data), codec="zstd").to_pybytes() == data
            assert zstd.decompress(pa.compress(data, codec="zstd", asbytes=True)) == data
    assert zstd.decompress(zstd.compress(b"abc") + zstd.compress(b"def", 9)) == b"abcdef"  # concatenated frames
    with pytest.raises(ValueError):
        zstd.decompress(b"definitely not zstd")


def test_vertex_feature_file_roundtrip_and_layout(tmp_path):
    v, f = torch.randn(257, 3) * 2, torch.randn(257, 24)
    path = D.frame_path(str(tmp_path), 7, D.VERTEX_FEATURES_FILE_NAME)
    assert os.path.basename(path) == "0007.nvblox_vertex_features.zst"  # {frame_index:04d}.<item name>
    D.write_vertex_features(path, v, f)
    # what the reference's reader does (dataset.py:410-415): zstd stream -> pickle -> dict of f16 CPU tensors
    raw = pickle.loads(zstd.decompress(open(path, "rb").read()))
    assert set(raw) == {"vertices", "features", "channel_length"} and raw["channel_length"] == 24
    assert raw["vertices"].dtype == torch.float16 and raw["features"].dtype == torch.float16
    s = D.load_item(path)
    assert torch.equal(s["vertices"], v.half()) and torch.equal(s["features"], f.half())


def test_vertex_feature_reader_refuses_code_execution(tmp_path):
    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    p = tmp_path / "0000.nvblox_vertex_features.zst"
    p.write_bytes(zstd.compress(pickle.dumps({"vertices": Evil(), "features": 1})))
    with pytest.raises(pickle.UnpicklingError):
        D.read_vertex_features(str(p))


def test_vertex_feature_reader_refuses_a_nested_pickle_in_the_storage_bytes(tmp_path):
    """A tensor's storage is pickled as torch.storage._load_from_bytes(<bytes>), which upstream is an unrestricted
    torch.load of those bytes: a payload one level down must be refused as well (and must not run)."""
    import torch.storage

    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            return (os.system, (f"touch {marker}",))

    class Nested:  # what a crafted file would name in place of a real storage
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (pickle.dumps(Evil()),))

    p = tmp_path / "0000.nvblox_vertex_features.zst"
    p.write_bytes(zstd.compress(pickle.dumps({"vertices": Nested(), "features": 1})))
    with pytest.raises(Exception) as e:
        D.read_vertex_features(str(p))
    assert isinstance(e.value, (pickle.UnpicklingError, RuntimeError))
    assert not marker.exists()


def test_depth_png_is_u16_millimetres_with_the_writers_clamp(tmp_path):
    depth = torch.rand(48, 64) * 70.0  # beyond the 65.535 m a u16 millimetre image can hold
    depth[0, 0], depth[0, 1], depth[0, 2] = float("inf"), -1.0, 1.2345
    path = D.frame_path(str(tmp_path), 3, "pov_depth.png")
    D.write_depth_png(path, depth)
    from PIL import Image

    with Image.open(path) as im:
        assert im.mode in ("I;16", "I") and im.size == (64, 48)
    got = D.load_item(path, torch.float32)
    exp = (torch.clamp(depth, 0.0, 65535 / 1000.0 - 1e-3) * 1000.0).to(torch.int32).to(torch.float32)
    assert torch.equal(got, exp)
    assert got[0, 0] == 65533 and got[0, 1] == 0 and got[0, 2] == 1234  # truncation, not rounding
    metres = DepthTransformer()(got)
    assert metres.dtype == torch.float32 and abs(float(metres[0, 2]) - 1.234) < 1e-6


def test_rgb_pose_intrinsics_files(tmp_path):
    rgb = torch.randint(0, 256, (20, 30, 3), dtype=torch.uint8)
    D.write_rgb_png(D.frame_path(str(tmp_path), 0, "pov_rgb.png"), rgb)
    back = D.load_item(D.frame_path(str(tmp_path), 0, "pov_rgb.png"), torch.float32)
    assert torch.equal(back, rgb.float())
    chw = RgbTransformer()(back)
    assert chw.shape == (3, 20, 30) and chw.dtype == torch.float32 and float(chw.max()) <= 1.0
    assert torch.equal(chw, (rgb.float() / 255.0).permute(2, 0, 1))
    D.write_pose(D.frame_path(str(tmp_path), 0, "pov_pose.npy"), torch.tensor([1.0, 2.0, 3.0]), torch.tensor([1.0, 0.0, 0.0, 0.0]))
    assert D.load_item(D.frame_path(str(tmp_path), 0, "pov_pose.npy")).tolist() == [1, 2, 3, 1, 0, 0, 0]
    K = torch.tensor([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]])
    D.write_intrinsics(D.frame_path(str(tmp_path), 0, "pov_intrinsics.npy"), K)
    assert torch.equal(D.load_item(D.frame_path(str(tmp_path), 0, "pov_intrinsics.npy")), K)
    with pytest.raises(ValueError):
        D.load_item(str(tmp_path / "0000.something.xyz"))


def test_map_container_roundtrip(tmp_path):
    arrays = {"tsdf_idx": np.arange(12, dtype=np.int32).reshape(4, 3), "tsdf": np.random.rand(4, 8, 8, 8, 2).astype(np.float32),
              "feature": (np.random.rand(4, 8, 8, 8, 16) * 10).astype(np.float16), "color_idx": np.zeros((0, 3), np.int32)}
    path = str(tmp_path / "0000.nvblox_map_static.nvblx")
    write_map_file(path, {"voxel_size_m": 0.01, "feature_channels": 16}, dict(arrays))
    meta, got = read_map_file(path)
    assert meta == {"voxel_size_m": 0.01, "feature_channels": 16}
    for k, v in arrays.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape and np.array_equal(np.asarray(got[k]), v)
    (tmp_path / "other.nvblx").write_bytes(b"SQLite format 3\0" + b"\0" * 100)  # what CUDA nvblox writes
    with pytest.raises(ValueError):
        read_map_file(str(tmp_path / "other.nvblx"))


def test_frame_dataset_reads_the_reference_layout(tmp_path):
    from torch.utils.data import DataLoader

    from nvblox_mindmap_amd.data_loading.dataset import MindmapFrameDataset, write_synthetic_demo

    write_synthetic_demo(str(tmp_path / "demo_00000"), 5, image_size=(32, 48), feature_dim=16)
    write_synthetic_demo(str(tmp_path / "demo_00001"), 3, image_size=(32, 48), feature_dim=16, seed=1)
    np.save(str(tmp_path / "demo_00001" / "demo_successful.npy"), np.array(False))  # failed demo: skipped
    os.remove(str(tmp_path / "demo_00000" / "0004.pov_depth.png"))                 # incomplete frame: skipped
    ds = MindmapFrameDataset(str(tmp_path), num_vertices=512, seed=0)
    assert len(ds) == 4
    s = ds[1]
    assert s["rgb_u8"].shape == (1, 32, 48, 3) and s["rgb_u8"].dtype == torch.uint8
    assert s["depth_mm"].shape == (1, 32, 48) and s["depth_mm"].dtype == torch.int16
    assert s["camera_poses"].shape == (1, 7) and s["intrinsics"].shape == (1, 3, 3)
    assert s["vertices"].shape == (512, 3) and s["vertex_features"].shape == (512, 16) and s["vertex_features"].dtype == torch.float16
    assert s["vertices_valid_mask"].dtype == torch.bool and s["gripper_history"].shape == (3, 1, 8)
    raw = D.read_png(str(tmp_path / "demo_00000" / "0001.pov_depth.png"))
    assert torch.equal(s["depth_mm"][0].to(torch.int32) & 0xFFFF, raw)
    batch = next(iter(DataLoader(ds, batch_size=2, shuffle=False, num_workers=0)))
    assert batch["rgb_u8"].shape == (2, 1, 32, 48, 3) and batch["vertices"].shape == (2, 512, 3)
    with pytest.raises(FileNotFoundError):
        MindmapFrameDataset(str(tmp_path / "demo_00001"))


def test_geometry_augmentation_matches_reference():
    """data_loading/sample_transformer.py:76-300 (GeometryAugmentor / GeometryNoiser / random_transform_* /
    apply_random_transform_to_sample), vectors from the imported reference (tests/golden/make_golden_augmentation.py): same
    seeds => same draws => same transforms; quaternions compared up to sign where the reference does not standardise them."""
    import random

    from nvblox_mindmap_amd.data_loading import sample_transformer as ST

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "augmentation.npz"))
    t = lambda k: torch.from_numpy(g[k])  # noqa: E731
    t_range, rpy_range = (g["t_lo"].tolist(), g["t_hi"].tolist()), (g["rpy_lo"].tolist(), g["rpy_hi"].tolist())

    def same_rotation(a, b):
        return np.allclose(np.abs((a * b).sum(-1)), 1.0, atol=1e-6)

    random.seed(11)
    tr, q = ST.random_transform_uniform(t_range, rpy_range)
    assert np.allclose(tr.numpy(), g["uniform_t"], atol=1e-7) and same_rotation(q.numpy(), g["uniform_q"])
    random.seed(12)
    aug = ST.GeometryAugmentor(t_range, rpy_range)
    poses = aug(t("poses"))
    assert np.allclose(poses.numpy(), g["aug_poses"], atol=1e-6)
    assert np.allclose(aug({"vertices": t("vertices")})["vertices"].numpy(), g["aug_vertices"], atol=1e-6)  # the SAME transform
    aug.reset()
    assert np.allclose(aug(t("vertices")).numpy(), g["aug_vertices_after_reset"], atol=1e-6)
    assert not np.allclose(g["aug_vertices"], g["aug_vertices_after_reset"], atol=1e-3)
    torch.manual_seed(21)
    tr, q = ST.random_transform_gaussian(0.01, 2.0, 7)
    assert np.allclose(tr.numpy(), g["gauss_t"], atol=1e-7) and same_rotation(q.numpy(), g["gauss_q"])
    torch.manual_seed(22)
    noiser = ST.GeometryNoiser(0.02, 3.0)
    assert np.allclose(noiser({"vertices": t("vertices")})["vertices"].numpy(), g["noisy_vertices"], atol=1e-6)
    assert np.allclose(noiser(t("flat_poses")).numpy(), g["noisy_flat_poses"], atol=1e-6)
    # a [nhist, ngrippers, 8] history: one transform per history entry, positions move by about the standard deviation
    torch.manual_seed(23)
    hist = noiser(t("poses"))
    assert hist.shape == (3, 2, 8) and 0 < float((hist[..., :3] - t("poses")[..., :3]).abs().max()) < 0.2
    assert torch.equal(hist[..., 7], t("poses")[..., 7])


def test_dataset_applies_the_same_augmentation_to_every_geometric_item(tmp_path):
    import random

    from nvblox_mindmap_amd.data_loading.dataset import MindmapFrameDataset, write_synthetic_demo
    from nvblox_mindmap_amd.data_loading.sample_transformer import GeometryAugmentor, apply_random_transform_to_sample
    from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod

    write_synthetic_demo(str(tmp_path / "demo_00000"), 2, image_size=(32, 32), feature_dim=8, ngrippers=2, vertex_count_range=(40, 41))
    plain = MindmapFrameDataset(str(tmp_path), num_vertices=40, vertex_sampling_method=VertexSamplingMethod.NONE)
    aug = GeometryAugmentor(([0.1, 0.1, 0.1], [0.2, 0.2, 0.2]), ([-20.0, -20.0, -20.0], [20.0, 20.0, 20.0]))
    with pytest.raises(NotImplementedError):  # cameras are not moved: refused unless the caller says the model is mesh-only
        MindmapFrameDataset(str(tmp_path), num_vertices=40, vertex_sampling_method=VertexSamplingMethod.NONE, geometry_augmentor=aug)
    moved = MindmapFrameDataset(str(tmp_path), num_vertices=40, vertex_sampling_method=VertexSamplingMethod.NONE, geometry_augmentor=aug,
                                allow_untransformed_cameras=True)
    a = plain[1]
    random.seed(3)
    b = moved[1]
    tr, q = aug._transform  # the transform drawn for this sample (reset() inside __getitem__)
    for key in ("vertices", "gripper_history", "gt_gripper_pred"):
        assert torch.allclose(b[key], apply_random_transform_to_sample(a[key], tr, q), atol=1e-6), key
        assert not torch.allclose(b[key][..., :3], a[key][..., :3], atol=1e-2)
    assert torch.equal(b["vertex_features"], a["vertex_features"]) and torch.equal(b["rgb_u8"], a["rgb_u8"])
