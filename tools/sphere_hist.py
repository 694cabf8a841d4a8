import os, sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.argv = ["x"]
import bench as B
from nvblox_mindmap_amd import _lib, synthetic as S
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg
dev = torch.device("cuda", 0)
cfg = S.StreamConfig(hole_mode="patches"); mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
frames = B.build_stream(cfg, 40, 64, dev); m = get_nvblox_mapper(mcfg, feature_channels=64)
for i in range(24): B.step(m, mcfg, frames[i])
torch.cuda.synchronize()
cap = 6 * 8192
buf = torch.zeros(3 * cap, dtype=torch.int64, device=dev)
_lib.check(_lib.lib().mmf_debug_wg_trace(_lib.dptr(buf), cap), "x")
for i in range(24, 27):
    buf.zero_(); torch.cuda.synchronize(); B.step(m, mcfg, frames[i]); torch.cuda.synchronize()
    rec = buf.cpu().numpy().reshape(cap, 3); rec = rec[rec[:, 0] != 0]
    rid = rec[:, 0] & 0xff
    tr = rec[rid == 41]; d = (tr[:, 2] - tr[:, 1]) / 100.0
    print("frame", i, "trace WGs", len(d), "skipped(<1.5us)", int((d < 1.5).sum()), "hist", np.histogram(d, bins=[0, 1.5, 4, 8, 12, 16, 20, 30])[0].tolist(), "max", d.max().round(1), "launch span", ((tr[:, 2].max() - tr[:, 1].min()) / 100.0).round(1))
_lib.check(_lib.lib().mmf_debug_wg_trace(None, 0), "x")
