"""nvblox_torch.indexing (import site: mindmap/paper/utils/utils.py:16)."""
import torch


def get_voxel_center_grids(block_indices: torch.Tensor, voxel_size_m: float, device=None) -> torch.Tensor:
    """Centres of all voxels of the given blocks: [n,3] int block indices -> [n,8,8,8,3] float32.

    centre = block_index * (8*voxel_size) + (voxel_index + 0.5) * voxel_size.
    """
    device = device if device is not None else block_indices.device
    b = block_indices.to(device=device, dtype=torch.float32)
    v = torch.arange(8, device=device, dtype=torch.float32)
    gx, gy, gz = torch.meshgrid(v, v, v, indexing="ij")
    local = (torch.stack([gx, gy, gz], dim=-1) + 0.5) * voxel_size_m
    return b[:, None, None, None, :] * (8.0 * voxel_size_m) + local[None]
