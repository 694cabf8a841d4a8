import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig, fused_ops as FO
from nvblox_mindmap_amd.training import build_model, synthetic_batch
from nvblox_mindmap_amd.training.trainer import unpack_batch
cfg = DiffuserActorConfig(); torch.manual_seed(0)
model = build_model(cfg, device="cuda").eval()
s = unpack_batch(cfg, synthetic_batch(cfg, 1, "cuda", seed=1))
DiffuserActor.enable_fused_inference(True)
def infer():
    with torch.no_grad():
        return model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None, s["gripper_history"], run_inference=True)[0]
for fuse in (True, False, True, False):
    FO.FUSE_OUT_FFN_QKV = fuse
    model.enable_graph_sampling(True)
    for _ in range(3): infer()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): infer()
    torch.cuda.synchronize(); print("fuse", fuse, (time.perf_counter()-t)/10*1e3, "ms", flush=True)
