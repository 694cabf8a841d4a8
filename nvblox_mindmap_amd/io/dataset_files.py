"""Readers / writers of the per-frame files of a mindmap dataset.

File naming and content follow the reference:
  ``NNNN.nvblox_vertex_features.zst``  zstd(level 1) of pickle({"vertices": f16 [N,3], "features": f16 [N,F],
                                       "channel_length": F})      mapping/helpers/nvblox_to_disk_helpers.py:53-65,
                                                                   read by data_loading/dataset.py:410-415,466-468
  ``NNNN.<cam>_depth.png``             16-bit grayscale PNG, millimetres (DEPTH_SCALE_FACTOR = 1000, clamped to the u16
                                       range)                      isaaclab_utils/isaaclab_writer.py:92-109
  ``NNNN.<cam>_rgb.png``               8-bit RGB PNG               isaaclab_writer.py:80-90
  ``NNNN.<cam>_pose.npy``              [x,y,z,qw,qx,qy,qz]         isaaclab_writer.py:72-78
  ``NNNN.<cam>_intrinsics.npy``        3x3                         isaaclab_writer.py:111-123
"""
import io
import os
import pickle
from typing import Dict, Tuple

import numpy as np
import torch

from ..mapping.nvblox_mapper_constants import DEPTH_SCALE_FACTOR
from . import zstd

VERTEX_FEATURES_FILE_NAME = "nvblox_vertex_features.zst"  # data_loading/item_names.py:12


# ---- vertex features -------------------------------------------------------------------------------------------------
def write_vertex_features(path: str, vertices: torch.Tensor, features: torch.Tensor) -> None:
    """The file save_feature_mesh_to_disk writes (nvblox_to_disk_helpers.py:53-65)."""
    assert vertices.shape[0] == features.shape[0] and vertices.shape[1] == 3
    pc_ob = {
        "vertices": vertices.to(torch.float16).cpu(),
        "features": features.to(torch.float16).cpu(),
        "channel_length": features.shape[1],
    }
    with open(path, "wb") as f:
        f.write(zstd.compress(pickle.dumps(pc_ob, protocol=pickle.HIGHEST_PROTOCOL), level=1))


def _safe_load_from_bytes(b: bytes):
    """Stand-in for torch.storage._load_from_bytes, which a pickled tensor's storage names (TypedStorage.__reduce__): the
    original is ``torch.load(..., weights_only=False)``, i.e. a second, UNRESTRICTED unpickle of bytes nested in the file.
    The nested stream is read with torch's weights-only unpickler instead (tensors / storages only, anything else raises)."""
    return torch.load(io.BytesIO(b), weights_only=True)


class _TensorUnpickler(pickle.Unpickler):
    """pickle.load restricted to what a vertex-feature file contains (CPU torch tensors, numpy arrays, builtins): public
    datasets are untrusted input and a plain pickle.load executes whatever the file names -- including, one level down, the
    storage bytes of a tensor (see _safe_load_from_bytes)."""

    _ALLOWED = {
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch", "HalfStorage"),
        ("torch", "FloatStorage"), ("torch", "DoubleStorage"), ("torch", "LongStorage"), ("torch", "IntStorage"),
        ("torch", "BoolStorage"), ("torch", "ByteStorage"), ("torch", "BFloat16Storage"), ("torch.storage", "UntypedStorage"),
        ("torch.storage", "TypedStorage"), ("torch.storage", "_load_from_bytes"), ("torch", "float16"), ("torch", "float32"),
        ("collections", "OrderedDict"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    }

    def find_class(self, module, name):
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            return _safe_load_from_bytes
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"vertex-feature file references {module}.{name}: refused")


def read_vertex_features(path: str) -> Dict:
    """``NvbloxMindmapDataset.unpickle_zst`` (dataset.py:410-415): {"vertices", "features", "channel_length"}."""
    with open(path, "rb") as f:
        raw = zstd.decompress(f.read())
    sample = _TensorUnpickler(io.BytesIO(raw)).load()
    if not isinstance(sample, dict) or "vertices" not in sample or "features" not in sample:
        raise ValueError(f"{path}: not a vertex-feature file")
    return sample


# ---- images ----------------------------------------------------------------------------------------------------------
def depth_to_millimetres(depth_m: torch.Tensor) -> torch.Tensor:
    """Metres -> integer millimetres in the u16 range with the writer's clamp and truncation (isaaclab_writer.py:102-108),
    as int32 on the tensor's device (torch has no arithmetic on uint16)."""
    d = torch.clamp(depth_m.to(torch.float32), min=0.0, max=65535 / DEPTH_SCALE_FACTOR - 1e-3)
    return (d * DEPTH_SCALE_FACTOR).to(torch.int32)


def write_depth_png(path: str, depth_m: torch.Tensor) -> None:
    from PIL import Image

    arr = depth_to_millimetres(depth_m.detach()).cpu().numpy().astype(np.uint16)
    Image.fromarray(arr).save(path)


_STALE_WARNED = [0]


def _warn_stale(what: str) -> None:
    """A raw loader copy (io/vertex_cache.py) that no longer matches its source is skipped, loudly but not once per sample."""
    import warnings

    _STALE_WARNED[0] += 1
    if _STALE_WARNED[0] <= 3:
        warnings.warn(f"stale raw loader copy ignored ({what}); re-run `python -m nvblox_mindmap_amd.io.vertex_cache <dataset>`")


def read_png(path: str, use_raw_cache: bool = False) -> torch.Tensor:
    """``torch.as_tensor(imageio.imread(path))`` (dataset.py:463-465): u16 [H,W] for depth, u8 [H,W,3] for rgb.
    ``use_raw_cache``: read ``<path>.raw`` (io/vertex_cache.py: the same pixels, no inflate) when it exists."""
    arr = None
    if use_raw_cache and os.path.exists(path + ".raw"):
        from .vertex_cache import StaleRawCopy, read_raw_image

        try:
            arr = read_raw_image(path + ".raw", source=path)
        except StaleRawCopy as e:  # the PNG was rewritten after the copy was made: the PNG is the truth
            _warn_stale(str(e))
    if arr is None:
        from PIL import Image

        with Image.open(path) as im:
            arr = np.array(im)
    if arr.dtype == np.int32:  # PIL mode "I" for 16-bit PNGs on some versions
        arr = arr.astype(np.uint16)
    if arr.dtype == np.uint16:
        return torch.from_numpy(arr.astype(np.int32))  # torch has no arithmetic on uint16; values are preserved
    return torch.from_numpy(arr)


def write_rgb_png(path: str, rgb: torch.Tensor) -> None:
    from PIL import Image

    Image.fromarray(rgb.detach().to("cpu").numpy().astype(np.uint8)).save(path)


def write_pose(path: str, translation_W_C: torch.Tensor, rotation_W_C_quat_wxyz: torch.Tensor) -> None:
    np.save(path, torch.cat([translation_W_C, rotation_W_C_quat_wxyz]).cpu().numpy())


def write_intrinsics(path: str, intrinsics: torch.Tensor) -> None:
    np.save(path, intrinsics.cpu().numpy())


def frame_path(directory: str, frame_index: int, item: str) -> str:
    return os.path.join(directory, f"{frame_index:04d}.{item}")


def load_item(path: str, dtype=torch.float32):
    """One dataset item by extension, as NvbloxMindmapDataset.__getitem__ does (dataset.py:457-468)."""
    ext = os.path.basename(path).split(".")[-1]
    if ext == "npy":
        return torch.as_tensor(np.load(path)).to(dtype)
    if ext == "png":
        return read_png(path).to(dtype)
    if ext == "zst":
        return read_vertex_features(path)
    raise ValueError(f"Unsupported item: {path}")
