"""Process-group plumbing (counterpart of mindmap/model_utils/multi_gpu.py:16-40 and distributed_training.py:16-57).

One process per GPU, env:// rendezvous (torchrun).  ``backend="nccl"`` IS RCCL on PyTorch-ROCm; collectives used by the
training path: DDP's bucketed gradient all-reduce (10.9 MB -> one 25 MB bucket), an all-gather of a small metrics
object at evaluation time, and barriers.  The fusion path uses no data-path collective (replicas only)."""
import os
from typing import Any, List, Optional

import torch
import torch.distributed as dist


def get_rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class ProcessGroup:
    """``with ProcessGroup(): ...`` -- initialises the default group from the torchrun environment (no-op for a single
    process without RANK/WORLD_SIZE), binds the rank to its GPU, destroys the group on exit."""

    def __init__(self, backend: Optional[str] = None):
        self.backend = backend
        self.started = False

    def __enter__(self):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1 and not dist.is_initialized():
            backend = self.backend or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, init_method="env://")
            self.started = True
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        return self

    def __exit__(self, *exc):
        if self.started:
            dist.destroy_process_group()
        return False


def all_gather_objects(obj: Any) -> List[Any]:
    """Every rank's picklable object, in rank order (evaluation metrics; two small all-gathers underneath)."""
    if get_world_size() == 1:
        return [obj]
    out = [None] * get_world_size()
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (the benchmark's timing reduction)."""
    if get_world_size() == 1:
        return float(value)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if get_world_size() > 1:
        dist.barrier()
