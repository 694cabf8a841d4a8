import sys, torch
sys.path.insert(0, "/root/repo")
from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling, farthest_point_sampling_reference
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
ref = farthest_point_sampling_reference(x, 614, 0)
print("equal", torch.equal(got, ref))
