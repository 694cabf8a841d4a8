"""Map checkpoint container behind ``Mapper.save_map`` / ``Mapper.load_from_file``.

The reference saves maps with nvblox's own serializer (``mapper.save_map(path, mapper_id)``,
mapping/helpers/nvblox_to_disk_helpers.py:88-93; loaded again by paper/teaser/convert_maps_usd.py).  That serializer
(an SQLite schema inside upstream nvblox) is not part of the reference tree, so its byte layout cannot be reproduced or
pinned here; files written by this module are NOT readable by CUDA nvblox and vice versa.  What is kept is the contract:
one file per mapper holds everything needed to resume fusion with identical results (tests/test_gpu_io.py).

Layout: ``MMFNVBLX1\\n`` | u64 header length | JSON header | arrays, each 64-byte aligned, little endian.
Header: {"voxel_size_m", "feature_channels", "arrays": [{"name", "dtype", "shape", "offset"}]}; arrays:
``tsdf_idx [n,3] i32``, ``tsdf [n,8,8,8,2] f32``, ``color_idx``, ``color_rgb [n,8,8,8,3] u8``, ``color_w [n,8,8,8] f32``,
``feature_idx``, ``feature [n,8,8,8,C] f16``, ``feature_w [n,8,8,8] f32`` -- blocks in allocation order.
"""
import json
import os
import struct
from typing import Dict

import numpy as np

MAGIC = b"MMFNVBLX1\n"
_ALIGN = 64


def write_map_file(path: str, meta: Dict, arrays: Dict[str, np.ndarray]) -> None:
    entries, offset = [], 0
    for name, a in arrays.items():
        a = np.ascontiguousarray(a)
        arrays[name] = a
        entries.append({"name": name, "dtype": a.dtype.str, "shape": list(a.shape), "offset": offset})
        offset += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    header = dict(meta)
    header["arrays"] = entries
    hb = json.dumps(header).encode("utf-8")
    base = len(MAGIC) + 8 + len(hb)
    pad = (-base) % _ALIGN
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<Q", len(hb) + pad))
        f.write(hb + b" " * pad)
        for e in entries:
            a = arrays[e["name"]]
            f.write(a.tobytes() if a.nbytes < (1 << 20) else memoryview(a).cast("B"))
            f.write(b"\0" * ((-a.nbytes) % _ALIGN))
    os.replace(tmp, path)  # a crash while saving never leaves a truncated checkpoint under the final name


def read_map_file(path: str):
    """-> (meta dict, {name: np.memmap})"""
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError(f"{path} is not a map file written by nvblox_mindmap_amd (CUDA nvblox .nvblx files are not supported)")
        (hl,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(hl).decode("utf-8"))
    base = len(MAGIC) + 8 + hl
    arrays = {}
    for e in header.pop("arrays"):
        shape = tuple(e["shape"])
        if int(np.prod(shape)) == 0:
            arrays[e["name"]] = np.zeros(shape, dtype=np.dtype(e["dtype"]))
        else:
            arrays[e["name"]] = np.memmap(path, dtype=np.dtype(e["dtype"]), mode="c", offset=base + e["offset"], shape=shape)
    return header, arrays
