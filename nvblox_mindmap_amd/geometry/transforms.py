"""Pose helpers of the path (mindmap/image_processing/backprojection.py:16-48, mindmap/geometry/utils.py:164).

``pose_to_homo`` in the reference converts quaternions with ``transforms3d.quat2mat`` on the CPU in a
per-element Python loop (backprojection.py:34-41, with a TODO to stay on the GPU).  Here it is a batched
torch expression on whatever device the poses live on; the formula is transforms3d's (float64
arithmetic, then cast to float32, like ``torch.tensor(quat2mat(q), dtype=float32)``).
"""
import torch


def quaternion_wxyz_to_matrix(q: torch.Tensor) -> torch.Tensor:
    """[...,4] (w,x,y,z) -> [...,3,3] float64.  transforms3d.quaternions.quat2mat semantics:
    Nq = |q|^2; if Nq < float64 eps -> identity; s = 2/Nq."""
    q = q.to(torch.float64)
    w, x, y, z = q.unbind(-1)
    Nq = w * w + x * x + y * y + z * z
    eps = torch.finfo(torch.float64).eps
    safe = Nq >= eps
    s = torch.where(safe, 2.0 / torch.where(safe, Nq, torch.ones_like(Nq)), torch.zeros_like(Nq))
    X, Y, Z = x * s, y * s, z * s
    wX, wY, wZ = w * X, w * Y, w * Z
    xX, xY, xZ = x * X, x * Y, x * Z
    yY, yZ, zZ = y * Y, y * Z, z * Z
    R = torch.stack(
        [
            torch.stack([1.0 - (yY + zZ), xY - wZ, xZ + wY], dim=-1),
            torch.stack([xY + wZ, 1.0 - (xX + zZ), yZ - wX], dim=-1),
            torch.stack([xZ - wY, yZ + wX, 1.0 - (xX + yY)], dim=-1),
        ],
        dim=-2,
    )
    eye = torch.eye(3, dtype=torch.float64, device=q.device).expand_as(R)
    return torch.where(safe[..., None, None], R, eye)


def _pose_to_homo_host(pose7) -> "np.ndarray":
    """One pose on the host: transforms3d's quat2mat in float64 (what the reference itself runs per pose, on the CPU), cast to
    float32, translation copied."""
    import numpy as np

    x, y, z, w, qx, qy, qz = [float(v) for v in pose7]
    Nq = w * w + qx * qx + qy * qy + qz * qz
    T = np.eye(4, dtype=np.float32)
    if Nq >= np.finfo(np.float64).eps:
        s = 2.0 / Nq
        X, Y, Z = qx * s, qy * s, qz * s
        wX, wY, wZ = w * X, w * Y, w * Z
        xX, xY, xZ = qx * X, qx * Y, qx * Z
        yY, yZ, zZ = qy * Y, qy * Z, qz * Z
        T[:3, :3] = np.array([[1.0 - (yY + zZ), xY - wZ, xZ + wY], [xY + wZ, 1.0 - (xX + zZ), yZ - wX], [xZ - wY, yZ + wX, 1.0 - (xX + yY)]])
    T[:3, 3] = pose7[:3]
    return T


def pose_to_homo(poses: torch.Tensor) -> torch.Tensor:
    """[...,7] = (x, y, z, qw, qx, qy, qz) -> [B,4,4] float32 homogeneous transforms
    (B = 1 for an unbatched pose, like the reference)."""
    assert poses.ndim >= 1 and poses.shape[-1] == 7
    if poses.numel() == 7 and poses.is_cuda:
        # A single pose on the GPU (the per-frame case of the mapping path): the reference converts it on the host anyway
        # (backprojection.py:34-36).  One 28-byte copy + ~40 scalar operations instead of ~25 kernel launches for one
        # quaternion; the same float64 formula as the batched path below, so the same float32 matrix.
        host = torch.from_numpy(_pose_to_homo_host(poses.detach().reshape(7).to("cpu", torch.float32).numpy()))
        return host.to(poses.device, non_blocking=True).unsqueeze(0)
    flat = poses.reshape(-1, 7)
    R = quaternion_wxyz_to_matrix(flat[:, 3:]).to(torch.float32)
    T = torch.eye(4, device=poses.device, dtype=torch.float32).repeat(flat.shape[0], 1, 1)
    T[:, :3, :3] = R
    T[:, :3, 3] = flat[:, :3].to(torch.float32)
    return T
