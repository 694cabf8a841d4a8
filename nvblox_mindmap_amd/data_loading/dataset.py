"""File-fed samples for policy training.

``MindmapFrameDataset`` reads demos stored in the reference's per-frame layout (io/dataset_files.py) the way
``NvbloxMindmapDataset.__getitem__`` does (mindmap/data_loading/dataset.py:425-490): one sample = the items of one frame.
Two deliberate differences, both about where the work happens:
  * images stay in their storage dtype on the host (rgb u8, depth u16 millimetres): the ``RgbTransformer`` /
    ``DepthTransformer`` arithmetic runs on the GPU after the copy (``gpu_unpack``), so the host -> device copy moves
    1 + 2 bytes per pixel instead of 12 + 4 and the loader workers only decode;
  * the depth back-projection of ``unpack_pcd`` (data_loading/batching.py:213-262), a CPU job in the reference, is the HIP
    kernel behind ``training.trainer.unpack_batch``.
The embodiment-specific parts of the reference dataset (robot states -> keyposes / policy states) are out of scope: the
gripper history / prediction targets are read from per-frame ``.npy`` items (``gripper_history.npy``,
``gt_gripper_pred.npy``, the names the reference itself uses for cached samples, dataset.py:233-250).
"""
import glob
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch.utils.data import Dataset

from ..io import dataset_files as D
from ..io import vertex_cache as VC
from ..mapping.nvblox_mapper_constants import DEPTH_SCALE_FACTOR
from .vertex_sampling import VertexSamplingMethod, sample_to_n_vertices, select_vertex_indices


class MindmapFrameDataset(Dataset):
    def __init__(self, dataset_path: str, cameras: Sequence[str] = ("pov",), num_vertices: int = 2048,
                 vertex_sampling_method: VertexSamplingMethod = VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT,
                 with_vertex_features: bool = True, seed: Optional[int] = None, geometry_augmentor=None, geometry_noiser=None,
                 use_raw_vertex_cache: bool = True, allow_untransformed_cameras: bool = False):
        """``geometry_augmentor`` / ``geometry_noiser``: sample_transformer.GeometryAugmentor / GeometryNoiser, wired as the
        reference does (dataset_files_by_encoding_method.py:258-279): ONE random rigid transform per sample applied to the mesh
        vertices, the gripper history and the target poses; independent Gaussian pose noise on the history and the vertices,
        not on the target; both before the vertices are sampled."""
        if geometry_augmentor is not None and not allow_untransformed_cameras:
            # the reference refuses random transforms unless the data type is MESH (dataset_files_by_encoding_method.py:258-279):
            # the rigid transform moves the mesh, the gripper history and the targets, NOT the camera poses / depth images that
            # every sample of this dataset also carries -- an RGBD_AND_MESH model would see a point cloud and a mesh in two frames
            raise NotImplementedError("geometry augmentation transforms the mesh, history and targets but not the cameras: pass "
                                      "allow_untransformed_cameras=True for a model that reads the mesh only (data type MESH)")
        self.augmentor, self.noiser = geometry_augmentor, geometry_noiser
        # ``use_raw_vertex_cache``: where ``io.vertex_cache.convert_dataset`` has left a memory-mappable copy of a frame's vertex
        # features, map it and read only the sampled rows (same selection, same values; 3 MB of page cache instead of 18 MB of
        # zstd + pickle per sample at the reference's shape); likewise raw copies of the two PNGs (same pixels, no inflate)
        self.use_raw_vertex_cache = use_raw_vertex_cache
        self.cameras = list(cameras)
        self.num_vertices = num_vertices
        self.method = vertex_sampling_method
        self.with_vertex_features = with_vertex_features
        self.seed = seed
        self.samples: List[Dict[str, str]] = []
        demos = sorted(d for d in glob.glob(os.path.join(dataset_path, "demo_*")) if os.path.isdir(d)) or [dataset_path]
        for demo in demos:
            ok = os.path.join(demo, "demo_successful.npy")
            if os.path.exists(ok) and not bool(np.load(ok)):  # failed demos are skipped (dataset.py:160-175)
                continue
            for pose in sorted(glob.glob(os.path.join(demo, f"*.{self.cameras[0]}_pose.npy"))):
                frame = os.path.basename(pose).split(".")[0]
                items = {}
                for cam in self.cameras:
                    for kind, ext in (("rgb", "png"), ("depth", "png"), ("pose", "npy"), ("intrinsics", "npy")):
                        items[f"{cam}_{kind}"] = os.path.join(demo, f"{frame}.{cam}_{kind}.{ext}")
                items["gripper_history"] = os.path.join(demo, f"{frame}.gripper_history.npy")
                items["gt_gripper_pred"] = os.path.join(demo, f"{frame}.gt_gripper_pred.npy")
                if with_vertex_features:
                    items["vertex_features"] = os.path.join(demo, f"{frame}.{D.VERTEX_FEATURES_FILE_NAME}")
                if all(os.path.exists(p) for p in items.values()):
                    yaw = os.path.join(demo, f"{frame}.gt_head_yaw.npy")
                    if os.path.exists(yaw):
                        items["gt_head_yaw"] = yaw
                    self.samples.append(items)
        if not self.samples:
            raise FileNotFoundError(f"no complete frames under {dataset_path}")

    def __len__(self) -> int:
        return len(self.samples)

    def _sample_from_raw(self, raw_path: str, geometric, seed, source: str = None, generator=None):
        """``sample_to_n_vertices`` on the mapped file: ALL vertices are read (6 B each) and augmented / noised exactly as on
        the decompressed path (same RNG draws), the selection is drawn on the same V, and only the selected FEATURE rows are
        touched.  Returns what the decompressed path returns."""
        v_map, f_map = VC.open_raw(raw_path, source)  # (StaleRawCopy when the .zst changed after the copy was made)
        vertices = torch.from_numpy(np.array(v_map))  # [V,3] float16 copy
        if self.augmentor is not None or self.noiser is not None:
            vertices = geometric(vertices.to(torch.float32), noisy=True)
        n, want, method = vertices.shape[0], self.num_vertices, self.method
        if method == VertexSamplingMethod.NONE or n == want:
            return vertices.to(torch.float32), torch.from_numpy(np.array(f_map)), torch.ones(n, dtype=torch.bool)
        if n > want:
            sel = select_vertex_indices(n, want, method, "cpu", seed, vertices[:, 2], generator)
            rows = sel.numpy()
            order = np.argsort(rows, kind="stable")  # ascending file offsets for the page cache; undone below
            feats = np.empty((want, f_map.shape[1]), dtype=np.float16)
            feats[order] = f_map[rows[order]]
            return vertices[sel].to(torch.float32), torch.from_numpy(feats), torch.ones(want, dtype=torch.bool)
        pad = want - n
        feats = torch.cat([torch.from_numpy(np.array(f_map)), torch.zeros((pad, f_map.shape[1]), dtype=torch.float16)], dim=0)
        verts = torch.cat([vertices, torch.zeros((pad, 3), dtype=vertices.dtype)], dim=0)
        valid = torch.ones(want, dtype=torch.bool)
        valid[n:] = False
        return verts.to(torch.float32), feats, valid

    def __getitem__(self, idx: int) -> Dict[str, torch.Tensor]:
        return self.get(idx)

    def get(self, idx: int, generator: Optional[torch.Generator] = None, pyrandom=None) -> Dict[str, torch.Tensor]:
        """Sample ``idx``.  Without arguments this is ``__getitem__``: the random draws (vertex selection, pose noise, augmentation)
        come from the process-wide generators and a seeded dataset calls ``torch.manual_seed(seed + idx)``, exactly like the
        reference's dataset inside a DataLoader WORKER PROCESS.  A loader THREAD of the training process (data_loading/pinned_loader.py)
        passes its own ``generator`` (torch, CPU) and ``pyrandom`` (``random.Random``): the same values for the same seed, and the
        generators the trainer draws its diffusion noise from are neither reseeded nor raced."""
        it = self.samples[idx]
        out = {}
        if self.augmentor is not None:
            self.augmentor.reset(pyrandom)  # a new transform for this sample, shared by all its geometric items (dataset.py:451-454)

        def geometric(x, noisy: bool):
            if self.augmentor is not None:
                x = self.augmentor(x)
            if noisy and self.noiser is not None:
                x = self.noiser(x, generator)
            return x

        rgb, depth, pose, intr = [], [], [], []
        for cam in self.cameras:
            rgb.append(D.read_png(it[f"{cam}_rgb"], self.use_raw_vertex_cache))                          # [H,W,3] u8
            depth.append(D.read_png(it[f"{cam}_depth"], self.use_raw_vertex_cache).to(torch.int16))      # u16 bit pattern, 2 B / pixel
            pose.append(torch.as_tensor(np.load(it[f"{cam}_pose"])).to(torch.float32))
            intr.append(torch.as_tensor(np.load(it[f"{cam}_intrinsics"])).to(torch.float32))
        out["rgb_u8"] = torch.stack(rgb)
        out["depth_mm"] = torch.stack(depth)
        out["camera_poses"] = torch.stack(pose)
        out["intrinsics"] = torch.stack(intr)
        out["gripper_history"] = geometric(torch.as_tensor(np.load(it["gripper_history"])).to(torch.float32), noisy=True)
        out["gt_gripper_pred"] = geometric(torch.as_tensor(np.load(it["gt_gripper_pred"])).to(torch.float32), noisy=False)
        if "gt_head_yaw" in it:
            out["gt_head_yaw"] = torch.as_tensor(np.load(it["gt_head_yaw"])).to(torch.float32)
        if self.with_vertex_features:
            seed = None if self.seed is None else self.seed + idx
            raw = VC.raw_path_of(it["vertex_features"]) if self.use_raw_vertex_cache else None
            sampled = None
            if raw is not None and os.path.exists(raw):
                try:
                    sampled = self._sample_from_raw(raw, geometric, seed, it["vertex_features"], generator)
                except VC.StaleRawCopy as e:  # a regenerated dataset with old copies lying around: the .zst is the truth
                    D._warn_stale(str(e))
            if sampled is not None:
                out["vertices"], out["vertex_features"], out["vertices_valid_mask"] = sampled
            else:
                s = D.read_vertex_features(it["vertex_features"])
                if self.augmentor is not None or self.noiser is not None:
                    s["vertices"] = geometric(s["vertices"].to(torch.float32), noisy=True)
                # sample the stored f16 rows, convert afterwards: the same N rows as sampling the float32 copy (selection / padding
                # do no arithmetic), without a float32 copy of the whole [V, C] matrix (37 MB at V = 12 k, C = 768) per sample
                v, f, valid = sample_to_n_vertices(s["vertices"], s["features"], self.num_vertices, self.method, seed, generator)
                out["vertices"], out["vertex_features"], out["vertices_valid_mask"] = v.to(torch.float32), f.to(torch.float16), valid
        return out


_TABLES: Dict[str, Dict[str, torch.Tensor]] = {}


def _tables(device) -> Dict[str, torch.Tensor]:
    """value -> transformed value for every possible pixel value, computed ON THE CPU with the reference's arithmetic
    (image / 255.0, image / DEPTH_SCALE_FACTOR in float32).  The GPU's float division is not guaranteed to round like the
    CPU's; a gather from these tables is, and is as cheap."""
    key = str(torch.device(device))
    if key not in _TABLES:
        _TABLES[key] = {
            "rgb": (torch.arange(256, dtype=torch.float32) / 255.0).type(torch.float32).to(device),
            "depth": (torch.arange(65536, dtype=torch.float32) / DEPTH_SCALE_FACTOR).to(torch.float32).to(device),
        }
    return _TABLES[key]


def gpu_unpack(batch: Dict[str, torch.Tensor], device) -> Dict[str, torch.Tensor]:
    """Collated loader batch (host, storage dtypes) -> the batch format of training.trainer (device): the copies are
    issued first, then RgbTransformer / DepthTransformer (sample_transformer.py:42-73) on the GPU, bit-identical to the
    CPU transformers."""
    dev = {k: v.to(device, non_blocking=True) for k, v in batch.items()}
    out = {k: v for k, v in dev.items() if k not in ("rgb_u8", "depth_mm")}
    tab = _tables(device)
    out["rgbs"] = tab["rgb"][dev["rgb_u8"].long()].permute(0, 1, 4, 2, 3).contiguous()       # [B,ncam,3,H,W] in [0,1]
    out["depths"] = tab["depth"][dev["depth_mm"].to(torch.int32) & 0xFFFF]                    # metres
    if "gt_head_yaw" not in out:
        out["gt_head_yaw"] = torch.zeros(out["gt_gripper_pred"].shape[0], out["gt_gripper_pred"].shape[1], 1, device=device)
    return out


class DevicePrefetcher:
    """Iterates a DataLoader of ``MindmapFrameDataset`` batches and hands out ``gpu_unpack``-ed batches one step ahead: the
    host -> device copies (pinned memory) and the GPU-side transforms of batch i+1 are issued on a side stream while the
    training step of batch i runs on the main one.  (At batch 32 a loader batch is ~150 MB: ~6 ms of copies per step that
    would otherwise sit on the compute stream.)"""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self._it = None
        self._next = None

    def _load(self):
        try:
            host = next(self._it)
        except StopIteration:
            self._next = None
            return
        with torch.cuda.stream(self.stream):
            self._next = gpu_unpack(host, self.device)
            release = getattr(self.loader, "release", None)
            if release is not None and getattr(host, "slot", -1) >= 0:
                # a PinnedBatchLoader batch is a view of one of its slots: the slot goes back to the pool once the copies have run
                ev = torch.cuda.Event()
                ev.record(self.stream)
                release(host, ev)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        self._it = iter(self.loader)
        self._load()
        return self

    def __next__(self):
        if self._next is None:
            raise StopIteration
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(self.stream)
        batch = self._next
        for t in batch.values():
            if torch.is_tensor(t):
                t.record_stream(main)
        self._load()
        return batch


def write_synthetic_demo(directory: str, n_frames: int, image_size=(64, 64), feature_dim: int = 16, num_history: int = 3,
                         prediction_horizon: int = 1, ngrippers: int = 1, camera: str = "pov", seed: int = 0,
                         vertex_count_range=(200, 3000)) -> None:
    """A demo in the reference's on-disk layout with random content (tests / loader benchmarks).  ``vertex_count_range``: the
    reference stores the UNSAMPLED feature mesh of a frame (save_feature_mesh_to_disk calls get_vertices_and_features with
    sample_vertices=False, nvblox_to_disk_helpers.py:40-50), i.e. 10^4 .. 10^5 vertices; the default keeps the tests small."""
    os.makedirs(directory, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    H, W = image_size
    for i in range(n_frames):
        D.write_rgb_png(D.frame_path(directory, i, f"{camera}_rgb.png"), torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8))
        D.write_depth_png(D.frame_path(directory, i, f"{camera}_depth.png"), 0.4 + 1.2 * torch.rand(H, W, generator=g))
        D.write_pose(D.frame_path(directory, i, f"{camera}_pose.npy"), torch.tensor([1.4, 0.0, 0.6]) + 0.05 * torch.randn(3, generator=g),
                     torch.nn.functional.normalize(torch.tensor([0.5, -0.5, 0.5, -0.5]) + 0.01 * torch.randn(4, generator=g), dim=0))
        D.write_intrinsics(D.frame_path(directory, i, f"{camera}_intrinsics.npy"),
                           torch.tensor([[586.4 * W / 512, 0.0, W / 2.0], [0.0, 586.4 * H / 512, H / 2.0], [0.0, 0.0, 1.0]]))
        nv = int(torch.randint(int(vertex_count_range[0]), int(vertex_count_range[1]), (1,), generator=g))
        D.write_vertex_features(D.frame_path(directory, i, D.VERTEX_FEATURES_FILE_NAME), torch.rand(nv, 3, generator=g),
                                torch.randn(nv, feature_dim, generator=g))

        def poses(n):
            q = torch.nn.functional.normalize(torch.randn(n, ngrippers, 4, generator=g), dim=-1)
            return torch.cat([torch.rand(n, ngrippers, 3, generator=g), q, (torch.rand(n, ngrippers, 1, generator=g) > 0.5).float()], dim=-1)

        np.save(D.frame_path(directory, i, "gripper_history.npy"), poses(num_history).numpy())
        np.save(D.frame_path(directory, i, "gt_gripper_pred.npy"), poses(prediction_horizon).numpy())
    np.save(os.path.join(directory, "demo_successful.npy"), np.array(True))
