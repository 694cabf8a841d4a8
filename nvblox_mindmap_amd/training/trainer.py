"""Training step of the policy (counterpart of mindmap/run_training.py:140-217,597-642)."""
from typing import Dict, Optional

import torch
import torch.nn as nn
from torch.nn.parallel import DistributedDataParallel

from ..diffuser_actor import DiffuserActor, DiffuserActorConfig
from ..image_processing.backprojection import get_camera_pointcloud
from ..mapping.nvblox_mapper_constants import get_workspace_bounds
from ..nvblox_torch.timer import Timer
from .distributed import collectives_active


def build_model(cfg: Optional[DiffuserActorConfig] = None, task: str = "DRILL_IN_BOX", device="cuda") -> DiffuserActor:
    cfg = cfg or DiffuserActorConfig()
    return DiffuserActor(cfg, get_workspace_bounds(task)).to(device)


def wrap_ddp(model: nn.Module, device) -> nn.Module:
    """DDP exactly as the reference wraps it (run_training.py:608-613): find_unused_parameters because the instruction
    branch is unused when use_instruction = 0, no buffer broadcast.  Single process: the bare model (unless the forced-collective
    mode asks for the wrapper over a one-rank group, training/distributed.py: force_collectives)."""
    if not collectives_active():
        return model
    dev = torch.device(device)
    ids = [dev.index] if dev.type == "cuda" else None
    return DistributedDataParallel(model, device_ids=ids, broadcast_buffers=False, find_unused_parameters=True)


def build_optimizer(model: nn.Module, lr: float = 1e-4, weight_decay: float = 5e-4) -> torch.optim.Optimizer:
    """AdamW with the reference's two groups and its exact rule (run_training.py:140-153): a parameter whose NAME contains
    "bias", "LayerNorm.weight" or "LayerNorm.bias" gets no weight decay, everything else 5e-4.  The reference's LayerNorm
    modules are attributes called ``norm*`` (layers.py:331,358), so -- as there -- LayerNorm gains ARE decayed; the same holds
    here (``norm`` attributes).  Only the frozen stand-in backbone is left out: the reference's extractor is not part of its
    module tree (feature_extraction.py:132-160), hence not in its optimizer either."""
    no_decay_names = ["bias", "LayerNorm.weight", "LayerNorm.bias"]
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if any(nd in name for nd in no_decay_names) else decay).append(p)
    return torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0, "lr": lr}, {"params": decay, "weight_decay": weight_decay, "lr": lr}])


def build_lr_scheduler(optimizer, train_iters: int = 100000, convergence_percentage: float = 0.75, end_factor: float = 0.5):
    return torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=1.0, end_factor=end_factor,
                                             total_iters=int(train_iters * convergence_percentage))


def synthetic_batch(cfg: DiffuserActorConfig, batch_size: int, device, num_vertices: int = 2048, seed: int = 0,
                    task: str = "DRILL_IN_BOX") -> Dict[str, torch.Tensor]:
    """A cached-sample-shaped batch as the data loader hands it over BEFORE unpack_batch:
    rgb in [0,1], metric depth, camera intrinsics / pose, sampled map vertices + f16 features, gripper states."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    wb = get_workspace_bounds(task)
    lo, hi = wb[0], wb[1]
    H, W = cfg.image_size
    ncam = 2 if cfg.add_external_cam else 1
    B = batch_size

    def poses(*shape):
        q = torch.nn.functional.normalize(torch.randn(*shape, 4, generator=g), dim=-1)
        p = lo + (hi - lo) * torch.rand(*shape, 3, generator=g)
        return torch.cat([p, q, (torch.rand(*shape, 1, generator=g) > 0.5).float()], dim=-1)

    cam_pos = torch.tensor([1.4, 0.0, 0.6]) + 0.05 * torch.randn(B, ncam, 3, generator=g)
    cam_quat = torch.tensor([0.5, -0.5, 0.5, -0.5]).expand(B, ncam, 4) + 0.01 * torch.randn(B, ncam, 4, generator=g)
    batch = {
        "rgbs": torch.rand(B, ncam, 3, H, W, generator=g),
        "depths": 0.4 + 1.2 * torch.rand(B, ncam, H, W, generator=g),
        "intrinsics": torch.tensor([[586.4, 0.0, W / 2.0], [0.0, 586.4, H / 2.0], [0.0, 0.0, 1.0]]).expand(B, ncam, 3, 3).contiguous(),
        "camera_poses": torch.cat([cam_pos, cam_quat], dim=-1),
        "vertices": lo + (hi - lo) * torch.rand(B, num_vertices, 3, generator=g),
        "vertex_features": torch.randn(B, num_vertices, cfg.feature_dim, generator=g).to(torch.float16),
        "vertices_valid_mask": torch.rand(B, num_vertices, generator=g) > 0.05,
        "gripper_history": poses(B, cfg.num_history, cfg.ngrippers),
        "gt_gripper_pred": poses(B, cfg.prediction_horizon, cfg.ngrippers),
        "gt_head_yaw": torch.rand(B, cfg.prediction_horizon, 1, generator=g) - 0.5,
    }
    return {k: v.to(device) for k, v in batch.items()}


def unpack_batch(cfg: DiffuserActorConfig, batch: Dict[str, torch.Tensor], min_depth: float = 0.0) -> Dict[str, torch.Tensor]:
    """Loader batch -> model inputs (mindmap/data_loading/batching.py:213-262,304-326,360-417).  The reference back-projects
    the depth on the CPU before ``.to(device)`` (its "CPU back-projection path"); here it is the HIP kernel on the GPU."""
    out = dict(batch)
    uses_images = cfg.data_type in ("rgbd", "rgbd_and_mesh")
    if uses_images:
        ncam = batch["depths"].shape[1]
        with Timer("step/train/unpack_pcd"):
            out["pcds"] = torch.stack([
                get_camera_pointcloud(batch["intrinsics"][:, c], batch["depths"][:, c], batch["camera_poses"][:, c, :3],
                                      batch["camera_poses"][:, c, 3:]) for c in range(ncam)], dim=1)
        out["pcd_valid_mask"] = batch["depths"] > min_depth
    else:
        out["rgbs"], out["pcds"], out["pcd_valid_mask"] = None, None, None
    if cfg.data_type in ("mesh", "rgbd_and_mesh"):
        out["vertex_features"] = batch["vertex_features"].to(torch.float32)
    else:
        out["vertex_features"], out["vertices"], out["vertices_valid_mask"] = None, None, None
    return out


class BackbonePrefetcher:
    """Evaluates the frozen image backbone of the NEXT batch on a second HIP stream while the trainable part of the current
    batch (encoder + diffusion head forward, backward, optimizer) runs on the main stream.

    The backbone is frozen (feature_extraction.py: the extractor is never trained), so its output for batch t+1 does not
    depend on the optimizer step of batch t: the result is identical to the serial order.  Its fp32 GEMMs are MFMA-bound while
    the trainable part is a long tail of small HBM-bound kernels, so the two fill each other's gaps.  Every step still runs
    one backbone forward and one full trainable pass.

        pre = BackbonePrefetcher(model)
        feats = pre.submit(first_batch)
        for batch, next_batch in ...:
            nxt = pre.submit(next_batch)             # enqueued on the side stream, returns at once
            train_one_step(cfg, model, opt, batch, backbone_feats=pre.wait(feats))
            feats = nxt
    """

    def __init__(self, model: nn.Module, priority: int = 0):
        self.model = model.module if hasattr(model, "module") else model
        self.stream = torch.cuda.Stream(priority=priority)

    def submit(self, batch: Dict[str, torch.Tensor]):
        rgbs = batch["rgbs"]
        self.stream.wait_stream(torch.cuda.current_stream())  # the batch was produced on the main stream
        with torch.cuda.stream(self.stream):
            feats = self.model.encoder.backbone_features(rgbs)
            done = torch.cuda.Event()
            done.record(self.stream)
        rgbs.record_stream(self.stream)
        return feats, done

    @staticmethod
    def wait(handle) -> torch.Tensor:
        feats, done = handle
        torch.cuda.current_stream().wait_event(done)
        feats.record_stream(torch.cuda.current_stream())
        return feats


def train_one_step(cfg: DiffuserActorConfig, model: nn.Module, optimizer, batch: Dict[str, torch.Tensor], scheduler=None,
                   backbone_feats: Optional[torch.Tensor] = None, unpack=None):
    """unpack -> forward (losses) -> backward (DDP all-reduce overlaps) -> AdamW step.  Returns the detached losses.
    ``backbone_feats``: the frozen backbone's output for this batch when a BackbonePrefetcher computed it ahead."""
    with Timer("step/train/unpack_batch"):
        s = (unpack or unpack_batch)(cfg, batch)
    optimizer.zero_grad(set_to_none=True)
    with Timer("step/train/compute_losses"):
        losses, _, _ = model(s["gt_gripper_pred"], s["gt_head_yaw"], s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"],
                             s["vertices"], s["vertices_valid_mask"], None, s["gripper_history"], backbone_feats=backbone_feats)
    with Timer("step/train/backprop"):
        losses[0].backward()
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return tuple(None if x is None else x.detach() for x in losses)
