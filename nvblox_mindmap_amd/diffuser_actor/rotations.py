"""Pose parametrisations and workspace normalisation used by the policy
(mirrors the behaviour of mindmap/model_utils/normalization.py:55-234 and mindmap/geometry/utils.py:60-104,164-260;
own formulas, pinned against the reference by tests/golden/policy_math.npz)."""
import torch
import torch.nn.functional as F


def normalise_quat(q: torch.Tensor) -> torch.Tensor:
    return q / q.norm(dim=-1, keepdim=True).clamp_min(1e-10)


def quat_wxyz_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """(...,4) real-first quaternion (not necessarily unit) -> (...,3,3)."""
    w, x, y, z = q.unbind(-1)
    s = 2.0 / (q * q).sum(-1)
    R = torch.stack([
        1 - s * (y * y + z * z), s * (x * y - z * w), s * (x * z + y * w),
        s * (x * y + z * w), 1 - s * (x * x + z * z), s * (y * z - x * w),
        s * (x * z - y * w), s * (y * z + x * w), 1 - s * (x * x + y * y),
    ], dim=-1)
    return R.reshape(q.shape[:-1] + (3, 3))


def matrix_to_quat_wxyz(R: torch.Tensor) -> torch.Tensor:
    """(...,3,3) rotation -> (...,4) real-first unit quaternion.  Of the four algebraically equivalent candidates
    (one per component used as pivot) the one with the largest pivot is taken: best conditioned."""
    m = R.reshape(R.shape[:-2] + (9,))
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(-1)
    four_sq = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], dim=-1)
    pivot = four_sq.argmax(dim=-1)
    cand = torch.stack([
        torch.stack([four_sq[..., 0], m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, four_sq[..., 1], m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, four_sq[..., 2], m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, four_sq[..., 3]], dim=-1),
    ], dim=-2)  # (...,4 candidates,4): each is 4*q_pivot*q
    sel = torch.gather(cand, -2, pivot[..., None, None].expand(pivot.shape + (1, 4))).squeeze(-2)
    return F.normalize(sel, dim=-1)


def matrix_to_ortho6d(R: torch.Tensor) -> torch.Tensor:
    """First two COLUMNS of R, concatenated: (...,3,3) -> (...,6)."""
    return torch.cat([R[..., :, 0], R[..., :, 1]], dim=-1)


def ortho6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """Gram-Schmidt of the two 3-vectors, columns (x, y, z = x cross y'): (...,6) -> (...,3,3)."""
    a, b = d6[..., :3], d6[..., 3:6]
    x = a / a.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    z = torch.cross(x, b, dim=-1)
    z = z / z.norm(dim=-1, keepdim=True).clamp_min(1e-8)
    y = torch.cross(z, x, dim=-1)
    return torch.stack([x, y, z], dim=-1)


def normalize_pos(pos: torch.Tensor, workspace_bounds: torch.Tensor):
    """Affine map of the workspace AABB to [-1,1]^3; also returns which points lie inside (inclusive)."""
    lo = workspace_bounds[0].to(pos.device, torch.float32)
    hi = workspace_bounds[1].to(pos.device, torch.float32)
    inside = ((pos >= lo) & (pos <= hi)).all(dim=-1)
    return (pos - lo) / (hi - lo) * 2.0 - 1.0, inside


def unnormalize_pos(pos: torch.Tensor, workspace_bounds: torch.Tensor) -> torch.Tensor:
    lo = workspace_bounds[0].to(pos.device, torch.float32)
    hi = workspace_bounds[1].to(pos.device, torch.float32)
    return (pos + 1.0) / 2.0 * (hi - lo) + lo


def normalize_pointcloud(pcd: torch.Tensor, workspace_bounds: torch.Tensor):
    """(B,ncam,3,H,W) world points -> normalised points (same layout) + in-bounds mask (B,ncam,H,W)."""
    p, inside = normalize_pos(pcd.permute(0, 1, 3, 4, 2), workspace_bounds)
    return p.permute(0, 1, 4, 2, 3), inside


def normalize_trajectory(traj: torch.Tensor, workspace_bounds: torch.Tensor, quaternion_format: str = "wxyz") -> torch.Tensor:
    """(...,7) = xyz + quaternion  ->  (...,9) = normalised xyz + 6-D rotation."""
    assert traj.shape[-1] == 7
    pos, _ = normalize_pos(traj[..., :3], workspace_bounds)
    q = normalise_quat(traj[..., 3:7])
    if quaternion_format == "xyzw":
        q = q[..., (3, 0, 1, 2)]
    return torch.cat([pos, matrix_to_ortho6d(quat_wxyz_to_matrix(q))], dim=-1)


def unnormalize_trajectory(traj: torch.Tensor, workspace_bounds: torch.Tensor, quaternion_format: str = "wxyz",
                           rotation_parametrization: str = "6D_from_query") -> torch.Tensor:
    """(...,9[+1]) normalised xyz + 6-D rotation [+ openness logit] -> (...,7[+1]) xyz + quaternion [+ openness prob].

    Reference quirk kept for parity: unless ``rotation_parametrization`` is exactly "6D", the reference first L2-normalises
    channels 3:7 as if they held a quaternion (mindmap/model_utils/normalization.py, ``if rotation_parametrization !=
    "6D"``) -- with its default "6D_from_query" that rescales the first four of the six rotation channels before the
    Gram-Schmidt step."""
    pos = unnormalize_pos(traj[..., :3], workspace_bounds)
    d6 = traj[..., 3:9]
    if rotation_parametrization != "6D":
        d6 = torch.cat([normalise_quat(d6[..., :4]), d6[..., 4:]], dim=-1)
    q = matrix_to_quat_wxyz(ortho6d_to_matrix(d6))
    if quaternion_format == "xyzw":
        q = q[..., (1, 2, 3, 0)]
    out = [pos, q]
    if traj.shape[-1] > 9:
        out.append(traj[..., 9:10].sigmoid())
        if traj.shape[-1] > 10:
            out.append(traj[..., 10:])
    return torch.cat(out, dim=-1)
