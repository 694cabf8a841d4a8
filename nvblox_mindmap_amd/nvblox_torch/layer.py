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
    idx64 = idx.to(torch.int64).cpu()
    for i in range(idx64.shape[0]):
        b = idx64[i] - lo
        if bool(((b >= 0) & (b < torch.tensor(dims))).all()):
            x, y, z = (b * 8).tolist()
            dense[x:x + 8, y:y + 8, z:z + 8] = blocks[i]
    return dense
