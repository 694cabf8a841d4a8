"""nvblox_torch.layer (imported, unused, at mindmap/mapping/helpers/nvblox_output_helpers.py:13;
used by mindmap/paper/utils/utils.py:18).  Thin aliases over the layer views of ``mapper``."""
import torch

from .mapper import FeatureLayerView as FeatureLayer  # noqa: F401
from .mapper import TsdfLayerView as Layer  # noqa: F401


def convert_layer_to_dense_tensor(layer, aabb_min_m, aabb_max_m, unobserved_value: float = 0.0) -> torch.Tensor:
    """Dense [X,Y,Z,F] tensor of the layer's voxels inside an AABB (block-aligned), channels = the
    block payload (TSDF: distance, weight; feature: C features + weight)."""
    blocks, idx = layer.get_all_blocks()
    vs = layer.voxel_size()
    bs = 8.0 * vs
    lo = torch.floor(torch.as_tensor(aabb_min_m, dtype=torch.float32) / bs).to(torch.int64)
    hi = torch.floor(torch.as_tensor(aabb_max_m, dtype=torch.float32) / bs).to(torch.int64)
    dims = (hi - lo + 1).tolist()
    F = blocks.shape[-1]
    dense = torch.full((dims[0] * 8, dims[1] * 8, dims[2] * 8, F), float(unobserved_value), dtype=blocks.dtype, device=blocks.device)
    # one scatter for all blocks (a Python loop with a device slice per block was ~3 000 launches for a task's map)
    b = idx.to(torch.int64).to(blocks.device) - lo.to(blocks.device)
    inside = ((b >= 0) & (b < torch.tensor(dims, device=blocks.device))).all(dim=1)
    b, kept = b[inside], blocks[inside]
    if b.shape[0]:
        view = dense.view(dims[0], 8, dims[1], 8, dims[2], 8, F)  # [bx, vx, by, vy, bz, vz, F]: a block is view[bx, :, by, :, bz, :]
        view[b[:, 0], :, b[:, 1], :, b[:, 2], :, :] = kept
    return dense
