"""Where a batch-1 policy inference spends its time (bench.py's policy_inference leg gives the total): the encoder by part
(frozen backbone, image / vertex tokens, FPS, gripper history), the step-invariant prelude of the head, the denoising loop.

    python tools/time_inference_parts.py [--backbone-dtype float16]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone-dtype", default="float16x3")
    a = ap.parse_args()
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, synthetic_batch
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    dev = torch.device("cuda:0")
    cfg = DiffuserActorConfig(backbone_matmul_dtype=a.backbone_dtype)
    torch.manual_seed(0)
    model = build_model(cfg, device=dev).eval()
    s = unpack_batch(cfg, synthetic_batch(cfg, 1, dev, seed=1))
    model.enable_graph_sampling(True)
    DiffuserActor.enable_fused_inference(True)

    def infer():
        with torch.no_grad():
            return model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"], s["vertices_valid_mask"],
                         None, s["gripper_history"], run_inference=True)

    def timed(fn, n=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    print(f"whole inference: {timed(infer, 5):.2f} ms (backbone matmuls in {a.backbone_dtype})")
    enc = model.encoder
    with torch.no_grad():
        print(f"  backbone_features: {timed(lambda: enc.backbone_features(s['rgbs'])):.2f} ms")
        bf = enc.backbone_features(s["rgbs"])
        print(f"  encode_images (given backbone features): {timed(lambda: enc.encode_images(s['rgbs'], s['pcds'], s['pcd_valid_mask'], bf)):.2f} ms")
        traj, yaw, _, e, _ = infer()
        noise = torch.randn((1 + cfg.diffusion_timesteps, 1, cfg.prediction_horizon, cfg.ngrippers, 9), device=dev)
        print(f"  graph-sampled denoising (prelude + loop): {timed(lambda: model._graph_sampler.run(e, noise), 5):.2f} ms")
        head = model.prediction_head
        print(f"  prepare_context (eager): {timed(lambda: head.prepare_context(e)):.2f} ms")


if __name__ == "__main__":
    main()
