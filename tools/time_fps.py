import sys, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fps_restatement import farthest_point_sampling_numpy
from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling
def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for B in (1, 2, 32):
    x = torch.randn(B, 3072, 120, device="cuda")
    print(B, "ms", timed(lambda: farthest_point_sampling(x, 614, 0)))
x = torch.randn(3, 3072, 120, device="cuda")
x[:, 100:200] = 0  # ties
got = farthest_point_sampling(x, 614, 0)
ref = torch.from_numpy(farthest_point_sampling_numpy(x.cpu().numpy(), 614, 0))
print("equal", torch.equal(got.cpu(), ref))
