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


_FORCED = None  # None: follow the environment (MMF_FORCE_COLLECTIVES); True / False: set by force_collectives()


def force_collectives(on: Optional[bool] = None) -> bool:
    """Forced-collective mode: with a process group of ONE rank, every collective of the training path (the weight broadcast, the
    flat-gradient all-reduce, the timing max-reduce, barriers, DistributedDataParallel's wrapping) is still issued instead of being
    skipped as the identity it is.  It exists so that ``backend="nccl"`` -- RCCL -- executes on a one-GPU box exactly the calls
    an 8-GPU launch makes (mindmap/model_utils/multi_gpu.py:21-34, mindmap/run_training.py:608-613); results are bit-identical to
    the plain step (a sum over one rank, a scale by 1.0).  ``force_collectives(True / False)`` sets it for the process,
    ``force_collectives()`` reads it (default: the environment variable MMF_FORCE_COLLECTIVES=1)."""
    global _FORCED
    if on is not None:
        _FORCED = bool(on)
    if _FORCED is not None:
        return _FORCED
    return os.environ.get("MMF_FORCE_COLLECTIVES", "0") == "1"


def collectives_active() -> bool:
    """Do the training path's collectives run?  A process group exists AND (it has more than one rank OR the forced mode is on)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or force_collectives()


class ProcessGroup:
    """``with ProcessGroup(): ...`` -- initialises the default group from the torchrun environment (no-op for a single
    process without RANK/WORLD_SIZE), binds the rank to its GPU, destroys the group on exit."""

    def __init__(self, backend: Optional[str] = None, force: Optional[bool] = None):
        self.backend = backend
        self.force = force  # True: a one-rank group is created too and the collectives run on it (force_collectives)
        self.started = False
        self._forced_before = None

    def __enter__(self):
        world = int(os.environ.get("WORLD_SIZE", "1"))
        forced = force_collectives() if self.force is None else bool(self.force)
        if (world > 1 or forced) and not dist.is_initialized():
            backend = self.backend or ("nccl" if torch.cuda.is_available() else "gloo")
            if torch.cuda.is_available():
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))  # before the communicator is made: RCCL binds to the current device
            if world == 1:
                # a group of one (forced mode): explicit rank / size over a TCP store on a port of our own -- the process environment
                # is left alone (code that looks for a launcher's WORLD_SIZE must not find ours)
                dist.init_process_group(backend=backend, init_method=f"tcp://127.0.0.1:{free_port()}", world_size=1, rank=0)
            else:
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29500")
                dist.init_process_group(backend=backend, init_method="env://")
            self.started = True
        if self.force is not None:
            self._forced_before = _FORCED
            force_collectives(self.force)
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        return self

    def __exit__(self, *exc):
        global _FORCED
        if self.started:
            dist.destroy_process_group()
        if self.force is not None:
            _FORCED = self._forced_before
        return False


def free_port() -> int:
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def all_gather_objects(obj: Any) -> List[Any]:
    """Every rank's picklable object, in rank order (evaluation metrics; two small all-gathers underneath)."""
    if not collectives_active():
        return [obj]
    out = [None] * get_world_size()
    dist.all_gather_object(out, obj)
    return out


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a scalar (the benchmark's timing reduction)."""
    if not collectives_active():
        return float(value)
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier() -> None:
    if collectives_active():
        dist.barrier()
