"""Raw, memory-mappable copy of a frame's vertex-feature file (a loader-side cache, not a format of the reference).

The reference stores a frame's UNSAMPLED feature mesh as ``NNNN.nvblox_vertex_features.zst`` -- zstd of a pickle of two float16
tensors, 10-14 k vertices x 768 channels = ~18 MB (mapping/helpers/nvblox_to_disk_helpers.py:53-65) -- and its loader
decompresses and unpickles all of it to keep ``num_vertices_to_sample`` = 2 048 rows (data_loading/dataset.py:410-415,
sample_transformer.py:150-186).  At 8 GPUs x 343 samples/s that is the decode cost that decides the scaling of the training
step.  ``convert_dataset`` writes, next to every such file, ``NNNN.nvblox_vertex_features.raw``:

    offset 0     8 bytes  magic  b"MMFVTX01"
           8     int64    V   (vertices)
           16    int64    C   (feature channels)
           24    int64    offset of the vertex array   (4096)
           32    int64    offset of the feature array  (page aligned)
    vertices  float16 [V, 3]   row-major
    features  float16 [V, C]   row-major

``open_raw`` maps it; the loader then draws its row selection exactly as before (same RNG draws on the same V) and touches
only the selected feature rows: 3 MB of page-cache reads instead of 18 MB of zstd + pickle per sample.
"""
import glob
import os
import struct
import sys
from typing import Tuple

import numpy as np
import torch

from .dataset_files import VERTEX_FEATURES_FILE_NAME, read_vertex_features

MAGIC = b"MMFVTX01"
RAW_SUFFIX = "nvblox_vertex_features.raw"
_PAGE = 4096


def raw_path_of(zst_path: str) -> str:
    assert zst_path.endswith(VERTEX_FEATURES_FILE_NAME)
    return zst_path[: -len(VERTEX_FEATURES_FILE_NAME)] + RAW_SUFFIX


def write_raw(path: str, vertices: torch.Tensor, features: torch.Tensor) -> None:
    v = np.ascontiguousarray(vertices.detach().to("cpu", torch.float16).numpy())
    f = np.ascontiguousarray(features.detach().to("cpu", torch.float16).numpy())
    assert v.ndim == 2 and v.shape[1] == 3 and f.ndim == 2 and f.shape[0] == v.shape[0]
    off_v = _PAGE
    off_f = (off_v + v.nbytes + _PAGE - 1) // _PAGE * _PAGE
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(MAGIC + struct.pack("<qqqq", v.shape[0], f.shape[1], off_v, off_f))
        fh.seek(off_v)
        fh.write(v.tobytes())
        fh.seek(off_f)
        fh.write(f.tobytes())
    os.replace(tmp, path)


def open_raw(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """(vertices [V,3] float16, features [V,C] float16) as read-only memory maps."""
    with open(path, "rb") as fh:
        head = fh.read(40)
    if len(head) < 40 or head[:8] != MAGIC:
        raise ValueError(f"{path}: not a raw vertex-feature file")
    V, C, off_v, off_f = struct.unpack("<qqqq", head[8:])
    size = os.path.getsize(path)
    if V < 0 or C <= 0 or off_v < 40 or off_f < off_v + V * 6 or off_f + V * C * 2 > size:
        raise ValueError(f"{path}: inconsistent header")
    if V == 0:
        return np.zeros((0, 3), np.float16), np.zeros((0, C), np.float16)
    v = np.memmap(path, dtype=np.float16, mode="r", offset=off_v, shape=(V, 3))
    f = np.memmap(path, dtype=np.float16, mode="r", offset=off_f, shape=(V, C))
    return v, f


def convert_dataset(dataset_path: str, overwrite: bool = False) -> int:
    """Write the raw copy next to every ``*.nvblox_vertex_features.zst`` under ``dataset_path``; returns the number written."""
    n = 0
    for zst in sorted(glob.glob(os.path.join(dataset_path, "**", f"*.{VERTEX_FEATURES_FILE_NAME}"), recursive=True)):
        raw = raw_path_of(zst)
        if os.path.exists(raw) and not overwrite and os.path.getmtime(raw) >= os.path.getmtime(zst):
            continue
        s = read_vertex_features(zst)
        write_raw(raw, s["vertices"], s["features"])
        n += 1
    return n


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m nvblox_mindmap_amd.io.vertex_cache <dataset path> [--overwrite]")
    print(f"{convert_dataset(sys.argv[1], '--overwrite' in sys.argv[2:])} raw vertex-feature files written")
