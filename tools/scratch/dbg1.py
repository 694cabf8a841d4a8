import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests")); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import numpy as np, torch
from test_gpu_model_inputs import *
mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
mcfg.aabb_min_m, mcfg.aabb_max_m = torch.tensor([-10.0, -10.0, -10.0]), torch.tensor([10.0, 10.0, 10.0])
gpu = build_map(8, scale=2, frames=(0, 9, 18), workspace_bounds_type=0, max_integration_distance_m=3.0)
print("blocks", gpu.tsdf_layer_view(0).num_allocated_blocks(), gpu.feature_layer_view(0).num_allocated_blocks())
for rz in (False, True):
    rv, rf = reference_rows(gpu, mcfg, rz, 0)
    v, f, valid = get_vertices_and_features(gpu, 0, mcfg, rz, 0, sample_vertices=False)
    print(rz, "ref n", rv.shape[0], "mine", v.shape[0])
    if rv.shape[0] == v.shape[0]:
        print(" equal v", np.array_equal(v.cpu().numpy(), rv), "f", np.array_equal(f.cpu().numpy().view(np.uint16), rf.view(np.uint16)))
    else:
        vm = v.cpu().numpy()
        k = 0
        while k < min(len(vm), len(rv)) and np.array_equal(vm[k], rv[k]): k += 1
        print(" first mismatch at", k, vm[k], rv[k])
# timing of the facade's stages at REF
import profile_model_inputs as P
device = torch.device("cuda:0")
cfg, C, frames, samples, ex, facade = P.build("ref", device, 4)
from nvblox_mindmap_amd.mapping.helpers.nvblox_input_helpers import frame_inputs_from_sample
for i in range(8):
    fr, smp = frames[i % 4], samples[i % 4]
    ex.next, ex.low = fr["features"], fr["lowres"]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    facade.decay()
    t1 = time.perf_counter()
    inp = frame_inputs_from_sample(smp, 0)
    t2 = time.perf_counter()
    facade._update_reconstruction(*inp, "pov")
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(i, "decay %.3f inputs %.3f update(host) %.3f drain %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t4-t3)*1e3))
