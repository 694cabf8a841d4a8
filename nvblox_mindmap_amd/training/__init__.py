"""Data-parallel training of the policy (BASELINE config 5): one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) / gloo on CPU, DistributedDataParallel gradient all-reduce."""
from .checkpoint import load_inference_checkpoint, load_train_checkpoint, save_checkpoint  # noqa: F401
from .distributed import ProcessGroup, all_gather_objects, barrier, get_rank, get_world_size, max_over_ranks  # noqa: F401
from .graphed import GraphedTrainStep, configure_runtime_for_graphs  # noqa: F401
from .loop import evaluate_nsteps, run_training  # noqa: F401
from .sampler import DistributedWeightedSampler  # noqa: F401
from .trainer import BackbonePrefetcher, build_model, build_optimizer, synthetic_batch, train_one_step, wrap_ddp  # noqa: F401
