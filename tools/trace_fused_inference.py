"""Only the fused-ops + HIP-graph inference configuration (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig  # noqa: E402
from nvblox_mindmap_amd.training import build_model, synthetic_batch  # noqa: E402
from nvblox_mindmap_amd.training.trainer import unpack_batch  # noqa: E402

cfg = DiffuserActorConfig()
torch.manual_seed(0)
model = build_model(cfg, device="cuda").eval()
s = unpack_batch(cfg, synthetic_batch(cfg, 1, "cuda", seed=1))
DiffuserActor.enable_fused_inference(True)
model.enable_graph_sampling(True)
for _ in range(4):
    with torch.no_grad():
        model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None,
              s["gripper_history"], run_inference=True)
torch.cuda.synchronize()
