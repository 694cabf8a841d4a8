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
