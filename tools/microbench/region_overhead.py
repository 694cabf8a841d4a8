"""Where does the fixed cost of a timed region of the headline stream go?  (bench.py at the driver's --steps 20: 57.1 us per frame against
52.8 at 200 steps = ~90 us per region.)  Host timestamps and stream events around the region's pieces, for several region lengths."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench as B  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg  # noqa: E402

dev = torch.device("cuda", 0)
cfg = S.StreamConfig(hole_mode="patches")
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
frames = B.build_stream(cfg, 60, 64, dev)
m = get_nvblox_mapper(mcfg, feature_channels=64)
m.set_deferred_feature_rows(True)
for i in range(40):
    B.step(m, mcfg, frames[i % 60])
m.flush()
torch.cuda.synchronize()
k = 40
for n in (1, 2, 5, 10, 20, 40, 100):
    rows = []
    for rep in range(7):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for i in range(n):
            B.step(m, mcfg, frames[(k + i) % 60])
        t_steps = time.perf_counter()
        e1.record()
        m.flush()
        e2.record()
        t_enq = time.perf_counter()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        k += n
        rows.append(((t1 - t0) * 1e6, (t_steps - t0) * 1e6, (t_enq - t0) * 1e6, e0.elapsed_time(e1) * 1e3, e1.elapsed_time(e2) * 1e3, e0.elapsed_time(e2) * 1e3))
    rows.sort()
    r = rows[len(rows) // 2]
    print(f"n={n:4d}: wall {r[0]:8.1f} us ({r[0] / n:6.1f}/frame) | host: steps enqueued at {r[1]:7.1f}, flush enqueued at {r[2]:7.1f} | GPU: first event -> steps' last "
          f"kernel {r[3]:8.1f}, flush (tail + rows) {r[4]:6.1f}, total {r[5]:8.1f} | wall - GPU total = {r[0] - r[5]:6.1f} us")
