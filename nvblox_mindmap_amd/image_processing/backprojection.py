"""Depth back-projection on MI355X (mirror of mindmap/image_processing/backprojection.py:51-146).

Same names, argument meaning and output layout as the reference; the work is one HIP kernel
(``mmf_backproject_depth``: 4 B read + 12 B written per pixel) instead of meshgrid + inverse + two
batched matmuls + cat.  GPU tensors only -- there is no CPU fallback in the product path.
"""
import torch

from .. import _lib
from ..geometry.transforms import pose_to_homo


def _backproject_chw(depth_image: torch.Tensor, intrinsics: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """[B,H,W], [B,3,3], [B,4,4] -> [B,3,H,W] float32, non-finite values mapped to 0."""
    assert depth_image.ndim == 3 and intrinsics.ndim == 3 and transform.ndim == 3
    assert depth_image.shape[0] == intrinsics.shape[0] == transform.shape[0]
    if not depth_image.is_cuda:
        raise RuntimeError("backprojection runs on the GPU only (no CPU fallback)")
    dev = depth_image.device
    B, H, W = depth_image.shape
    d = depth_image.to(torch.float32).contiguous()
    K = intrinsics.to(device=dev, dtype=torch.float32).contiguous()
    T = transform.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().mmf_backproject_depth(_lib.dptr(d), _lib.dptr(K), _lib.dptr(T), B, H, W, _lib.dptr(out), _lib.stream_ptr(dev)),
               "mmf_backproject_depth")
    return out


def backproject_depth_to_pointcloud(depth_image: torch.Tensor, intrinsics: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """(B,H,W), (B,3,3), (B,4,4) -> (B, H*W, 3) world points, u = column / v = row integer pixel coordinates
    (backprojection.py:74-99).  Unlike the reference this already maps NaN/inf to 0 (the reference does it one
    call later, in get_camera_pointcloud)."""
    B, H, W = depth_image.shape
    return _backproject_chw(depth_image, intrinsics, transform).reshape(B, 3, H * W).permute(0, 2, 1)


def get_camera_pointcloud(intrinsics: torch.Tensor, depth: torch.Tensor, position: torch.Tensor, orientation: torch.Tensor) -> torch.Tensor:
    """(3,3)|(B,3,3), (H,W)|(B,H,W), (3,)|(B,3), (4,)|(B,4) wxyz -> (3,H,W)|(B,3,H,W) (backprojection.py:104-146)."""
    added_batch_dim = False
    if depth.ndim == 2:
        added_batch_dim = True
        intrinsics, depth = intrinsics.unsqueeze(0), depth.unsqueeze(0)
        position, orientation = position.unsqueeze(0), orientation.unsqueeze(0)
    assert intrinsics.ndim == 3 and depth.ndim == 3 and position.ndim == 2 and orientation.ndim == 2
    transform = pose_to_homo(torch.concatenate([position, orientation], dim=1))
    pointcloud = _backproject_chw(depth, intrinsics, transform)
    return pointcloud.squeeze(0) if added_batch_dim else pointcloud
