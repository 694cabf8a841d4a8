"""Training step with and without the frozen-backbone prefetch (training.trainer.BackbonePrefetcher), fp32 and fp16 backbone
matmuls.  Run on the GPU box: `python tools/time_train_overlap.py`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print("stream priority range", torch.cuda.Stream.priority_range())
    for dt in ("float32", "float16"):
        for pre, prio in ((False, 0), (True, 0), (True, -1)):
            os.environ["BENCH_PREFETCH_PRIORITY"] = str(prio)
            r = bench.run_training(dev, 1, steps=12, warmup=4, backbone_matmul_dtype=dt, prefetch_backbone=pre)
            print(f"backbone {dt:8s} prefetch={pre!s:5s} priority={prio:2d}: {r['ms_per_step']:7.2f} ms/step  {r['step_per_s']:6.2f} step/s",
                  flush=True)


if __name__ == "__main__":
    main()
