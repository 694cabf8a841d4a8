"""Gripper-relative model inputs (mirror of mindmap/model_utils/relative_conversions.py:15-133; the ``relative`` option of
the reference DiffuserActor, diffuser_actor.py:46,509,554-566).

Conventions as in the reference: poses are [x, y, z, qw, qx, qy, qz(, openness)]; the "current pose" is the LAST entry of
the gripper history; point clouds and the history are only TRANSLATED (the reference leaves their rotation alone), the
predicted trajectory is translated AND rotated (R_EE_P = inv(R_W_EE) * R_W_P) and comes back through the inverse.
Quaternion products are returned with a non-negative real part (pytorch3d's ``quaternion_multiply``,
geometry/pytorch3d_transforms.py:425-439)."""
import torch


def _hamilton(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz,
                        aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw), dim=-1)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a * b (real part first, broadcasting), sign chosen so that the real part is >= 0."""
    q = _hamilton(a, b)
    return torch.where(q[..., :1] < 0, -q, q)


def quaternion_invert(q: torch.Tensor) -> torch.Tensor:
    """Conjugate (= inverse of a unit quaternion)."""
    return q * q.new_tensor([1.0, -1.0, -1.0, -1.0])


def get_current_pose_from_gripper_history(gripper_history: torch.Tensor) -> torch.Tensor:
    """(B, nhist, ngrippers, X) -> (B, ngrippers, X): the newest pose."""
    return gripper_history[:, -1]


def to_relative_pcd(pcd: torch.Tensor, current_pose: torch.Tensor) -> torch.Tensor:
    """(B, ncam, 3, H, W) world points minus the current position.  Like the reference's function (:29-44) this takes a
    per-sample pose (B, X): with a per-gripper pose (B, ngrippers, X) there is no single origin and the reference's
    ``view(B, 1, 3, 1, 1)`` raises -- so does this."""
    if current_pose.dim() != 2:
        raise RuntimeError(f"to_relative_pcd needs a (batch, X) pose, got {tuple(current_pose.shape)}")
    return pcd - current_pose[:, :3].reshape(-1, 1, 3, 1, 1)


def to_relative_gripper_history(gripper_history: torch.Tensor, current_pose: torch.Tensor) -> torch.Tensor:
    """(B, nhist, ngrippers, X): positions minus each gripper's current position; rotations untouched; a new tensor."""
    out = gripper_history.clone()
    out[..., :3] -= current_pose[:, None, :, :3]
    return out


def _split8(trajectory: torch.Tensor):
    assert trajectory.shape[-1] == 8, "trajectory rows are position + quaternion + gripper state"
    return trajectory[..., :3], trajectory[..., 3:7], trajectory[..., 7:8]


def to_relative_trajectory(trajectory: torch.Tensor, current_pose: torch.Tensor) -> torch.Tensor:
    """(B, n, ngrippers, 8) world poses -> poses in the current gripper frame (gripper state passed through)."""
    assert trajectory.shape[0] == current_pose.shape[0]
    pos, quat, state = _split8(trajectory)
    here, facing = current_pose[..., :3].unsqueeze(1), current_pose[..., 3:7].unsqueeze(1)
    return torch.cat([pos - here, quaternion_multiply(quaternion_invert(facing), quat), state], dim=-1)


def to_absolute_trajectory(trajectory: torch.Tensor, current_pose: torch.Tensor) -> torch.Tensor:
    """Inverse of to_relative_trajectory (up to the sign of the quaternion)."""
    pos, quat, state = _split8(trajectory)
    here, facing = current_pose[..., :3].unsqueeze(1), current_pose[..., 3:7].unsqueeze(1)
    return torch.cat([pos + here, quaternion_multiply(facing, quat), state], dim=-1)
