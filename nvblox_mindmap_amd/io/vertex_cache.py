"""Raw, memory-mappable copy of a frame's vertex-feature file (a loader-side cache, not a format of the reference).

The reference stores a frame's UNSAMPLED feature mesh as ``NNNN.nvblox_vertex_features.zst`` -- zstd of a pickle of two float16
tensors, 10-14 k vertices x 768 channels = ~18 MB (mapping/helpers/nvblox_to_disk_helpers.py:53-65) -- and its loader
decompresses and unpickles all of it to keep ``num_vertices_to_sample`` = 2 048 rows (data_loading/dataset.py:410-415,
sample_transformer.py:150-186).  At 8 GPUs x 343 samples/s that is the decode cost that decides the scaling of the training
step.  ``convert_dataset`` writes, next to every such file, ``NNNN.nvblox_vertex_features.raw``:

    offset 0     8 bytes  magic  b"MMFVTX02"
           8     int64    V   (vertices)
           16    int64    C   (feature channels)
           24    int64    offset of the vertex array   (4096)
           32    int64    offset of the feature array  (page aligned)
           40    int64    size of the .zst it was made from      } the source's stamp: a reader that is given the source path
           48    int64    its modification time in nanoseconds   } refuses a copy whose source has changed since (StaleRawCopy)
           56    int64    2^32 | crc32 of the source's first and last 64 KiB (MMFVTX03; 0 = none): a dataset COPIED without its
                          modification times (cp -r, object-store syncs, image layers) keeps its raw copies -- size + content decide
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

MAGIC = b"MMFVTX03"
MAGIC_V2 = b"MMFVTX02"  # (round-4 files: (size, mtime) stamp, no content hash)
MAGIC_V1 = b"MMFVTX01"  # (round-3 files: no source stamp; accepted only while the copy is not older than its source)
_HASH_SPAN = 65536
STALE_COUNT = [0]  # raw copies refused in this process (surfaced by the loaders' statistics)
RAW_SUFFIX = "nvblox_vertex_features.raw"
_PAGE = 4096


class StaleRawCopy(ValueError):
    """The raw copy does not belong to the source file next to it (regenerated / overwritten dataset): use the source."""


def _stamp(source: str) -> Tuple[int, int]:
    st = os.stat(source)
    return int(st.st_size), int(st.st_mtime_ns)


def _content_hash(source: str) -> int:
    """2^32 | crc32(first 64 KiB + last 64 KiB of the file): cheap (two reads), changes with any regeneration of a compressed file."""
    import zlib

    with open(source, "rb") as fh:
        head = fh.read(_HASH_SPAN)
        size = os.fstat(fh.fileno()).st_size
        tail = b""
        if size > _HASH_SPAN:
            fh.seek(max(size - _HASH_SPAN, _HASH_SPAN))
            tail = fh.read(_HASH_SPAN)
    return (1 << 32) | (zlib.crc32(head + tail) & 0xFFFFFFFF)


def _stamp3(source) -> Tuple[int, int, int]:
    return (*_stamp(source), _content_hash(source)) if source else (0, 0, 0)


def _check_stamp(path: str, source, size: int, mtime_ns: int, content: int = 0) -> None:
    """``source``: the file the copy was made from, or None (no check: the caller vouches for it).  Fresh = same (size, mtime), or
    same size and same content hash (a dataset copied without its modification times keeps its raw copies)."""
    if source is None:
        return
    if not os.path.exists(source):
        return  # a dataset shipped as raw copies only
    if (size, mtime_ns) == (0, 0):  # a copy without a stamp: as fresh as its modification time says
        if os.path.getmtime(path) < os.path.getmtime(source):
            STALE_COUNT[0] += 1
            raise StaleRawCopy(f"{path} is older than {source}")
        return
    now = _stamp(source)
    if (size, mtime_ns) == now:
        return
    if content and size == now[0] and content == _content_hash(source):
        return
    STALE_COUNT[0] += 1
    raise StaleRawCopy(f"{path} was made from another version of {source}")


def raw_path_of(zst_path: str) -> str:
    assert zst_path.endswith(VERTEX_FEATURES_FILE_NAME)
    return zst_path[: -len(VERTEX_FEATURES_FILE_NAME)] + RAW_SUFFIX


def write_raw(path: str, vertices: torch.Tensor, features: torch.Tensor, source: str = None) -> None:
    v = np.ascontiguousarray(vertices.detach().to("cpu", torch.float16).numpy())
    f = np.ascontiguousarray(features.detach().to("cpu", torch.float16).numpy())
    assert v.ndim == 2 and v.shape[1] == 3 and f.ndim == 2 and f.shape[0] == v.shape[0]
    off_v = _PAGE
    off_f = (off_v + v.nbytes + _PAGE - 1) // _PAGE * _PAGE
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(MAGIC + struct.pack("<qqqq", v.shape[0], f.shape[1], off_v, off_f) + struct.pack("<qqq", *_stamp3(source)))
        fh.seek(off_v)
        fh.write(v.tobytes())
        fh.seek(off_f)
        fh.write(f.tobytes())
    os.replace(tmp, path)


def raw_header(path: str, source: str = None) -> Tuple[int, int, int, int]:
    """(V, C, offset of the vertices, offset of the features) of a raw vertex-feature file, validated; ``source``: the .zst the
    copy stands for -- StaleRawCopy if it has changed since the copy was written."""
    with open(path, "rb") as fh:
        head = fh.read(64)
    if len(head) < 40 or head[:8] not in (MAGIC, MAGIC_V2, MAGIC_V1):
        raise ValueError(f"{path}: not a raw vertex-feature file")
    V, C, off_v, off_f = struct.unpack("<qqqq", head[8:40])
    size_src, mtime_src = struct.unpack("<qq", head[40:56]) if (head[:8] != MAGIC_V1 and len(head) >= 56) else (0, 0)
    content = struct.unpack("<q", head[56:64])[0] if (head[:8] == MAGIC and len(head) >= 64) else 0
    _check_stamp(path, source, size_src, mtime_src, content)
    size = os.path.getsize(path)
    if V < 0 or C <= 0 or off_v < 40 or off_f < off_v + V * 6 or off_f + V * C * 2 > size:
        raise ValueError(f"{path}: inconsistent header")
    return V, C, off_v, off_f


def open_raw(path: str, source: str = None) -> Tuple[np.ndarray, np.ndarray]:
    """(vertices [V,3] float16, features [V,C] float16) as read-only memory maps (``source``: see ``raw_header``)."""
    V, C, off_v, off_f = raw_header(path, source)
    if V == 0:
        return np.zeros((0, 3), np.float16), np.zeros((0, C), np.float16)
    v = np.memmap(path, dtype=np.float16, mode="r", offset=off_v, shape=(V, 3))
    f = np.memmap(path, dtype=np.float16, mode="r", offset=off_f, shape=(V, C))
    return v, f


# ---- images: the same idea for the two PNGs of a frame ------------------------------------------------------------------------
# ``NNNN.<cam>_rgb.png`` / ``NNNN.<cam>_depth.png`` (isaaclab_utils/isaaclab_writer.py:80-109) cost ~11 ms of inflate + defilter
# per sample at 512x512 -- after the vertex features the largest term of the loader's per-sample time.  ``<name>.png.raw``:
#     offset 0  8 bytes magic b"MMFIMG02";  int32 H, W, C;  int32 itemsize (1: uint8, 2: uint16);  int64 size, int64 mtime_ns of the
#     PNG it was made from;  int64 2^32 | crc32 of its first / last 64 KiB (0: none, round-4 files);  data at offset 48
#     (MMFIMG01, round 3: no stamp, data at offset 32)
IMG_MAGIC = b"MMFIMG02"
IMG_MAGIC_V1 = b"MMFIMG01"


def write_raw_image(path: str, arr: np.ndarray, source: str = None) -> None:
    a = np.ascontiguousarray(arr)
    assert a.dtype in (np.uint8, np.uint16) and a.ndim in (2, 3)
    H, W = a.shape[:2]
    C = a.shape[2] if a.ndim == 3 else 0
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(IMG_MAGIC + struct.pack("<iiii", H, W, C, a.dtype.itemsize) + struct.pack("<qqq", *_stamp3(source)))
        fh.write(a.tobytes())
    os.replace(tmp, path)


def _image_header(fh, path: str, source) -> Tuple[int, int, int, int, int]:
    head = fh.read(32)
    if len(head) < 32 or head[:8] not in (IMG_MAGIC, IMG_MAGIC_V1):
        raise ValueError(f"{path}: not a raw image file")
    H, W, C, item = struct.unpack("<iiii", head[8:24])
    size_src = mtime_src = content = 0
    data_at = 32
    if head[:8] == IMG_MAGIC:
        size_src, mtime_src = struct.unpack("<qq", head[24:32] + fh.read(8))
        content = struct.unpack("<q", fh.read(8))[0]
        data_at = 48
    _check_stamp(path, source, size_src, mtime_src, content)
    if H <= 0 or W <= 0 or C not in (0, 3, 4) or item not in (1, 2):
        raise ValueError(f"{path}: inconsistent header")
    return H, W, C, item, data_at


def raw_image_header(path: str, source: str = None) -> Tuple[int, int, int, int, int]:
    """(H, W, C (0: no channel axis), bytes per value, offset of the pixel block), validated against the file size and -- with
    ``source`` -- against the PNG the copy stands for (StaleRawCopy)."""
    with open(path, "rb") as fh:
        H, W, C, item, data_at = _image_header(fh, path, source)
        if os.fstat(fh.fileno()).st_size < data_at + H * W * max(C, 1) * item:
            raise ValueError(f"{path}: truncated")
    return H, W, C, item, data_at


def read_raw_image(path: str, source: str = None) -> np.ndarray:
    """The pixel array of the PNG it was made from (uint8 [H,W,3] / uint16 [H,W]), as a fresh array.  ``source``: that PNG --
    StaleRawCopy if it has changed since."""
    with open(path, "rb") as fh:
        H, W, C, item, _ = _image_header(fh, path, source)
        n = H * W * max(C, 1) * item
        data = fh.read(n)
    if len(data) != n:
        raise ValueError(f"{path}: truncated")
    a = np.frombuffer(data, dtype=np.uint8 if item == 1 else np.uint16)
    return a.reshape((H, W, C) if C else (H, W)).copy()


def _fresh(read) -> bool:
    try:
        read()
        return True
    except ValueError:  # stale, truncated or foreign: rewrite it
        return False


def convert_dataset(dataset_path: str, overwrite: bool = False, images: bool = True) -> int:
    """Write the raw copy next to every ``*.nvblox_vertex_features.zst`` (and, with ``images``, next to every ``*.png``) under
    ``dataset_path``; returns the number of files written."""
    n = 0
    if images:
        from PIL import Image

        for png in sorted(glob.glob(os.path.join(dataset_path, "**", "*.png"), recursive=True)):
            raw = png + ".raw"
            if os.path.exists(raw) and not overwrite and _fresh(lambda: read_raw_image(raw, png)):
                continue
            with Image.open(png) as im:
                arr = np.array(im)
            if arr.dtype == np.int32:
                arr = arr.astype(np.uint16)
            write_raw_image(raw, arr, source=png)
            n += 1
    for zst in sorted(glob.glob(os.path.join(dataset_path, "**", f"*.{VERTEX_FEATURES_FILE_NAME}"), recursive=True)):
        raw = raw_path_of(zst)
        if os.path.exists(raw) and not overwrite and _fresh(lambda: open_raw(raw, zst)):
            continue
        s = read_vertex_features(zst)
        write_raw(raw, s["vertices"], s["features"], source=zst)
        n += 1
    return n


if __name__ == "__main__":
    if len(sys.argv) < 2:
        raise SystemExit("usage: python -m nvblox_mindmap_amd.io.vertex_cache <dataset path> [--overwrite]")
    print(f"{convert_dataset(sys.argv[1], '--overwrite' in sys.argv[2:])} raw vertex-feature files written")
