"""Which operators (and input shapes) carry the GPU time of the TRAINABLE half of a policy training step: torch.profiler over the
eager form of training.GraphedTrainStep (use_graphs=False, no backbone overlap -- the same kernels as the captured step, issued one
by one), grouped by operator + input shapes, and by the module that issued them (record_function ranges of nn.Module.forward).
Usage (GPU box): python tools/train_op_profile.py [rows] > gpurun_out/train_op_profile.txt"""
import os
import sys

os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig  # noqa: E402
from nvblox_mindmap_amd.training import GraphedTrainStep, build_model, synthetic_batch  # noqa: E402


def main(rows=60):
    torch.manual_seed(0)
    cfg = DiffuserActorConfig(backbone_matmul_dtype="float16x3")
    model = build_model(cfg, device="cuda")
    batches = [synthetic_batch(cfg, 32, "cuda", seed=i) for i in range(2)]
    g = GraphedTrainStep(cfg, model, batches[0], use_graphs=False, overlap_backbone=False)
    for i in range(3):
        g.step(batches[i % 2], batches[(i + 1) % 2])
    torch.cuda.synchronize()
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_modules=True) as prof:
        for i in range(n):
            g.step(batches[i % 2], batches[(i + 1) % 2])
        torch.cuda.synchronize()
    ev = prof.key_averages(group_by_input_shape=True)
    tot = sum(e.self_device_time_total for e in ev)
    print(f"steps profiled: {n}; self device time per step: {tot / n / 1e3:.2f} ms")
    print(f"{'ms/step':>8s} {'calls/step':>10s} {'avg us':>8s}  op  [input shapes]")
    for e in sorted(ev, key=lambda e: -e.self_device_time_total)[:rows]:
        if e.self_device_time_total <= 0:
            continue
        print(f"{e.self_device_time_total / n / 1e3:8.3f} {e.count / n:10.1f} {e.self_device_time_total / max(e.count, 1):8.1f}  {e.key}  {str(e.input_shapes)[:150]}")
    # by operator only
    ev2 = prof.key_averages()
    print("\nby operator:")
    for e in sorted(ev2, key=lambda e: -e.self_device_time_total)[:40]:
        if e.self_device_time_total <= 0:
            continue
        print(f"{e.self_device_time_total / n / 1e3:8.3f} {e.count / n:10.1f} {e.self_device_time_total / max(e.count, 1):8.1f}  {e.key}")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
