"""Latency of one policy inference (B=1: encoder once + 100 denoising steps), eager vs HIP-graph replay of the loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig  # noqa: E402
from nvblox_mindmap_amd.training import build_model, synthetic_batch  # noqa: E402
from nvblox_mindmap_amd.training.trainer import unpack_batch  # noqa: E402


def main():
    cfg = DiffuserActorConfig()  # RGBD_AND_MESH, 512x512, 2048 vertices x 768, 100 diffusion steps
    torch.manual_seed(0)
    model = build_model(cfg, device="cuda").eval()
    s = unpack_batch(cfg, synthetic_batch(cfg, 1, "cuda", seed=1))

    def infer():
        with torch.no_grad():
            return model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"],
                         s["vertices_valid_mask"], None, s["gripper_history"], run_inference=True)[0]

    def timed(n=5):
        infer()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            infer()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    eager = timed()
    model.enable_graph_sampling(True)
    t0 = time.perf_counter()
    infer()
    torch.cuda.synchronize()
    capture = (time.perf_counter() - t0) * 1e3
    graphed = timed()
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor

    DiffuserActor.enable_fused_inference(True)
    model.enable_graph_sampling(False)
    fused_eager = timed()
    model.enable_graph_sampling(True)
    fused_graph = timed()
    print(f"policy inference B=1, {cfg.diffusion_timesteps} steps: eager {eager:.1f} ms, graph replay {graphed:.1f} ms "
          f"(first call incl. capture {capture:.0f} ms); fused ops: eager {fused_eager:.1f} ms, graph replay {fused_graph:.1f} ms")


if __name__ == "__main__":
    main()
