"""Per-kernel HIP-event brackets of the fused frame at the reference's shape (512x512, 768 channels)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg  # noqa: E402

dev = torch.device("cuda", 0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 768
cfg = S.StreamConfig(width=512, height=512, fx=586.4, fy=586.4, cx=255.5, cy=255.5, hole_mode="patches")
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
frames = B.build_stream(cfg, 4, C, dev)
m = get_nvblox_mapper(mcfg, feature_channels=C)
for i in range(8):
    B.step(m, mcfg, frames[i % 4])
torch.cuda.synchronize()
m.profile_reset()
m.profile_enable(True)
for i in range(24):
    B.step(m, mcfg, frames[i % 4])
torch.cuda.synchronize()
m.profile_enable(False)
print({k: round(v[0] / v[1] * 1e3, 1) for k, v in m.profile().items() if v[1]})
print(m.stats(MAPPER_TO_ID.STATIC))
