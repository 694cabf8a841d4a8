"""Prints the in-kernel timeline of the TSDF allocation over a few fused frames: the single workgroup of k_alloc_jobs, or the
LAST of k_alloc_tsdf's allocation workgroups (phases: - | loads | scan | counts of the earlier workgroups | emit + publish)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from nvblox_mindmap_amd import _lib  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg  # noqa: E402

dev = torch.device("cuda", 0)
cfg = S.StreamConfig(hole_mode="patches")
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
frames = B.build_stream(cfg, 40, 64, dev)
m = get_nvblox_mapper(mcfg, feature_channels=64)
for i in range(20):
    B.step(m, mcfg, frames[i])
out = (C.c_int64 * 10)()
_lib.check(_lib.lib().mmf_get_alloc_timeline(m._h, 0, 1, out))
names = ["compaction", "table loads", "scan", "earlier counts / insert", "emit + publish"]
for i in range(20, 32):
    B.step(m, mcfg, frames[i])
    _lib.check(_lib.lib().mmf_get_alloc_timeline(m._h, 0, 1, out))
    t = list(out)
    line = " ".join(f"{n}={(t[k + 1] - t[k]) / 100.0:.1f}us" for k, n in enumerate(names)) + f" total={(t[5] - t[0]) / 100.0:.1f}us"
    if t[6] > 0:  # k_alloc_jobs only: the mask column workgroups sharing the launch
        line += (f" | mask cols: first start {(t[7] - t[0]) / 100.0:+.1f}us, last start {(t[9] - t[0]) / 100.0:+.1f}us, last end "
                 f"{(t[6] - t[0]) / 100.0:+.1f}us relative to alloc start; longest workgroup {t[8] / 100.0:.1f}us")
    print("frame", i, line)
