import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from test_gpu_model_inputs import *
from nvblox_mindmap_amd.data_loading.vertex_sampling import randperm_prefix
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
mcfg.aabb_min_m, mcfg.aabb_max_m = torch.tensor([-10.0, -10.0, -10.0]), torch.tensor([10.0, 10.0, 10.0])
gpu = build_map(8, scale=2, frames=(0, 9, 18), workspace_bounds_type=0, max_integration_distance_m=3.0)
for n in (48062, 30000, 40000):
    torch.manual_seed(3); a = torch.randperm(n)[:2048]; sa = torch.get_rng_state()
    torch.manual_seed(3); b = randperm_prefix(n, 2048); sb = torch.get_rng_state()
    print(n, "randperm equal", torch.equal(a, b), torch.equal(sa, sb), torch.get_num_threads())
n = gpu.model_inputs_prepare(0, mcfg.aabb_min_host, mcfg.aabb_max_host, 8, True)
print("n", n)
torch.manual_seed(3)
rows = torch.randperm(n)[:2048].cuda()
v, f, valid = gpu.model_inputs_gather(0, rows, 2048, 2048, torch.float16)
va, fa, _ = gpu.model_inputs_gather(0, None, n, n, torch.float16)
print("gather rows == all[rows]", torch.equal(v, va[rows]), torch.equal(f, fa[rows]))
rv, rf = reference_rows(gpu, mcfg, True, 0)
print("all == ref", np.array_equal(va.cpu().numpy(), rv))
torch.manual_seed(3)
v2, f2, valid2 = get_vertices_and_features(gpu, 0, mcfg, True, 0, sample_vertices=True, number_of_vertices_to_sample=2048,
                                        vertex_sampling_method=VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
print("helper == gather rows", torch.equal(v2[0], v))
