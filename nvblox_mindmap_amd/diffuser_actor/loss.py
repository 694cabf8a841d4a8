"""Training loss of the policy (behaviour of mindmap/model_utils/loss.py:33-80): L1 on predicted noise of position and
6-D rotation, BCE-with-logits on gripper openness, MSE on head yaw; weights 30 / 10 / 1 / 1."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class LossWeights:
    pos_loss: float = 30.0
    rot_loss: float = 10.0
    gripper_loss: float = 1.0
    head_yaw_loss: float = 1.0


def compute_loss(pred, head_yaw_pred, target, gt_openness, gt_head_yaw, weights: LossWeights, predict_head_yaw: bool):
    """pred (...,10) = pos3 + rot6 + openness logit; target (...,9); gt_openness (...,1).
    Returns (total, pos, rot, gripper, head_yaw) -- only `total` carries gradients."""
    assert pred.shape[-1] == target.shape[-1] + gt_openness.shape[-1]
    pos = F.l1_loss(pred[..., :3], target[..., :3])
    rot = F.l1_loss(pred[..., 3:9], target[..., 3:9])
    grip = F.binary_cross_entropy_with_logits(pred[..., 9:10], gt_openness) if gt_openness.numel() else pred.new_zeros(())
    total = weights.pos_loss * pos + weights.rot_loss * rot + weights.gripper_loss * grip
    yaw = None
    if predict_head_yaw:
        yaw_l = F.mse_loss(head_yaw_pred, gt_head_yaw)
        total = total + weights.head_yaw_loss * yaw_l
        yaw = yaw_l.detach()
    return total, pos.detach(), rot.detach(), grip.detach(), yaw


def _rotation_angle(q: torch.Tensor) -> torch.Tensor:
    """Rotation angle (radians, >= 0) of quaternions (..., 4), real part first: the norm of pytorch3d's
    ``quaternion_to_axis_angle`` (geometry/pytorch3d_transforms.py:546-574), small-angle series included."""
    vec = q[..., 1:]
    norms = vec.norm(dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angle = 2 * half
    ratio = torch.where(angle.abs() < 1e-6, 0.5 - angle * angle / 48, torch.sin(half) / angle)  # sin(a/2) / a
    return (vec / ratio).norm(dim=-1)


def compute_metrics(pred, head_yaw_pred, target, gt_head_yaw, predict_head_yaw: bool):
    """Proxy metrics of checkpoint evaluation (mindmap/model_utils/loss.py:83-139; run_training.py:342-369 averages them over
    batches and ranks).  pred / target (..., ngrippers, 8): position, quaternion (real part first), openness.  Returns a dict of
    0-d tensors + ``bias`` (3,): mean / std of the Euclidean and of the per-axis absolute position error, the signed mean
    position error, the (historical) L1 between quaternions, the rotation error in degrees, the openness L1 and, optionally,
    the absolute head-yaw error in degrees."""
    from .relative_conversions import quaternion_invert, quaternion_multiply

    assert pred.shape[:-1] == target.shape[:-1] and pred.shape[-1] == 8 and target.shape[-1] == 8
    dp = pred[..., :3] - target[..., :3]
    per_axis = (dp ** 2).sqrt()  # |error| per axis, written the way the reference computes it
    euclid = (dp ** 2).sum(-1).sqrt()
    m = {"distance_m": euclid.mean(), "distance_m_std": euclid.std()}
    for k, name in enumerate("xyz"):
        m[f"distance_m_{name}"] = per_axis[..., k].mean()
        m[f"distance_m_std_{name}"] = per_axis[..., k].std()
    m["bias"] = dp.mean(dim=(0, 1, 2))
    m["rot_l1"] = (pred[..., 3:7] - target[..., 3:7]).abs().sum(-1).mean()
    delta = quaternion_multiply(pred[..., 3:7], quaternion_invert(target[..., 3:7]))
    m["rot_error_deg"] = (_rotation_angle(delta) * 180 / torch.pi).mean()
    m["openness_l1"] = (pred[..., 7:] - target[..., 7:]).abs().sum(-1).mean()
    if predict_head_yaw:
        m["head_yaw_error_deg"] = (head_yaw_pred - gt_head_yaw).abs().mean() * 180 / torch.pi
    return m
