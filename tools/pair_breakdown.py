import sys, json, torch, time
sys.path.insert(0, "/root/repo")
sys.argv = ["bench.py"]
import bench
from nvblox_mindmap_amd import synthetic as S
import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import NvbloxMappingCfg
dev = torch.device("cuda:0")
frames = bench.build_stream(S.StreamConfig(hole_mode="patches"), 50, 64, dev)
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
class Ex:
    def compute(self, rgb): return self.next.unsqueeze(0)
ex = Ex()
dyn = torch.zeros_like(frames[0]["dynamic_mask"])
dyn[dyn.shape[0] // 4: 3 * dyn.shape[0] // 4, dyn.shape[1] // 3: 2 * dyn.shape[1] // 3] = True
for pair in (True, False):
    H.PAIR_MAPPERS = pair
    mapper = H.get_nvblox_mapper(mcfg, feature_channels=64)
    def step(i):
        fr = frames[i % 50]; ex.next = fr["features"]; mapper.decay()
        H.nvblox_integrate(mapper, mcfg, ex, fr["depth"], fr["K"], fr["T_W_C"], fr["rgb"], dyn, include_dynamic=True)
    for i in range(30): step(i)
    torch.cuda.synchronize()
    mapper.profile_reset(); mapper.profile_enable(True, kernels=list(bench.KERNEL_OF_CLASS), stride=1)
    for i in range(100): step(30 + i)
    torch.cuda.synchronize(); mapper.profile_enable(False)
    pr = mapper.profile()
    print("pair" if pair else "sequential", {k: (round(v[0] / v[1] * 1e3, 2), v[1]) for k, v in pr.items() if v[1]})
    print({k: mapper.stats(mid) for k, mid in (("static", 0), ("dynamic", 1))})
