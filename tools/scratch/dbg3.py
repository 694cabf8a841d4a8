import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from test_gpu_model_inputs import *
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
mcfg.aabb_min_m, mcfg.aabb_max_m = torch.tensor([-10.0, -10.0, -10.0]), torch.tensor([10.0, 10.0, 10.0])
gpu = build_map(8, scale=2, frames=(0, 9, 18), workspace_bounds_type=0, max_integration_distance_m=3.0)
rv, rf = reference_rows(gpu, mcfg, False, 0)
v, f, valid = get_vertices_and_features(gpu, 0, mcfg, False, 0, sample_vertices=False)
print("unsampled False equal", np.array_equal(v.cpu().numpy(), rv))
torch.manual_seed(3)
s_before = torch.get_rng_state()
v, f, valid = get_vertices_and_features(gpu, 0, mcfg, True, 0, sample_vertices=True, number_of_vertices_to_sample=2048,
                                        vertex_sampling_method=VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
s_after = torch.get_rng_state()
rv, rf = reference_rows(gpu, mcfg, True, 0)
n = rv.shape[0]
print("n ref", n)
torch.set_rng_state(s_before)
sel = torch.randperm(n)[:2048].numpy()
print("state after equal", torch.equal(torch.get_rng_state(), s_after))
print("v == rv[sel]", np.array_equal(v[0].cpu().numpy(), rv[sel]))
ev, ef, _ = reference_sample(rv, rf, 2048, VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, 3)
print("v == ev", np.array_equal(v[0].cpu().numpy(), ev), "ev == rv[sel]", np.array_equal(ev, rv[sel]))
