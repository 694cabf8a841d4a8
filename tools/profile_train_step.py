"""Breaks one policy training step into its stages with torch.cuda events (backbone, back-projection + encode, head + loss,
backward, optimizer).  Run on the GPU box: `python tools/profile_train_step.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig  # noqa: E402
from nvblox_mindmap_amd.training import build_model, build_optimizer, synthetic_batch, train_one_step  # noqa: E402


def timed(fn, n=5):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    torch.manual_seed(0)
    cfg = DiffuserActorConfig()
    model = build_model(cfg, device="cuda")
    opt = build_optimizer(model)
    batch = synthetic_batch(cfg, 32, "cuda", seed=1)
    from nvblox_mindmap_amd.diffuser_actor.fps import farthest_point_sampling

    xf = torch.randn(32, 3072, 120, device="cuda")
    print("fps B=32 N=3072 C=120 -> 614: ms", timed(lambda: farthest_point_sampling(xf, 614, 0)))
    print("full step ms", timed(lambda: train_one_step(cfg, model, opt, batch)))
    rgb = torch.rand(32, 3, 512, 512, device="cuda")
    with torch.no_grad():
        print("backbone fwd ms", timed(lambda: model.encoder.backbone(rgb)))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            print("backbone fwd bf16-autocast ms", timed(lambda: model.encoder.backbone(rgb)))
    from torch.profiler import ProfilerActivity, profile

    # stage times: forward pieces under no_grad (the backward of a stage costs about twice its forward)
    from nvblox_mindmap_amd.training.trainer import unpack_batch
    s = unpack_batch(cfg, batch)
    enc_mod = model.encoder
    with torch.no_grad():
        print("fwd encode_inputs ms", timed(lambda: model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"],
                                                           s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None,
                                                           s["gripper_history"])))
    print("fwd (autograd on) ms", timed(lambda: model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"],
                                                      s["vertex_features"], s["vertices"], s["vertices_valid_mask"], None,
                                                      s["gripper_history"])[0][0]))

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"],
              s["vertices_valid_mask"], None, s["gripper_history"])[0][0].backward()

    print("fwd + bwd ms", timed(fwd_bwd))
    print("optimizer.step ms", timed(opt.step))

    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            train_one_step(cfg, model, opt, batch)
        torch.cuda.synchronize()
    rows = [e for e in prof.key_averages() if e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.self_device_time_total)
    total = sum(e.self_device_time_total for e in rows)
    print(f"device time per step {total / 2e3:.2f} ms over {sum(e.count for e in rows) // 2} kernels")
    for e in rows[:60]:
        print(f"{e.self_device_time_total / 2e3:9.3f} ms {e.count // 2:5d} x {e.self_device_time_total / e.count:9.1f} us  {e.key[:110]}")


if __name__ == "__main__":
    main()
