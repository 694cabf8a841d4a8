"""Map -> model input tensors (mirror of mindmap/mapping/helpers/nvblox_output_helpers.py:22-91)."""
from typing import Optional, Tuple

import torch

from ...data_loading.vertex_sampling import VertexSamplingMethod, sample_to_n_vertices, select_vertex_indices
from ...nvblox_torch.mapper import Mapper
from ..nvblox_mapper_constants import NvbloxMappingCfg


def get_vertices_and_features(mapper: Mapper, mapper_id: int, nvblox_mapping_config: NvbloxMappingCfg,
                              remove_zero_features: bool, num_excess_features: int, sample_vertices: bool,
                              number_of_vertices_to_sample: Optional[int] = None,
                              vertex_sampling_method: Optional[VertexSamplingMethod] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Surface vertices + features, AABB-filtered (strict), zero-pad channels stripped, all-zero rows dropped,
    then sampled / padded to N."""
    mapper.update_feature_mesh(mapper_id)
    mesh = mapper.get_feature_mesh(mapper_id)
    vertices = mesh.vertices()
    features = mesh.vertex_features()
    assert vertices.shape[0] == features.shape[0]
    assert vertices.shape[0] != 0, "No vertices found in the mesh."
    assert vertices.is_cuda and features.is_cuda

    # One combined row filter over the ORIGINAL mesh rows, then ONE gather -- same rows, same order and same RNG draws as the
    # reference's chain of boolean-mask copies (nvblox_output_helpers.py:57-74: AABB, strip the pad channels, drop all-zero
    # rows), without copying the whole [V, C] feature matrix twice (80 MB at V = 26 k, C = 768).
    lo = nvblox_mapping_config.aabb_min_m.to(vertices.device)
    hi = nvblox_mapping_config.aabb_max_m.to(vertices.device)
    keep = ((vertices > lo) & (vertices < hi)).all(dim=1)  # strict on both sides
    used = features[..., :-num_excess_features] if num_excess_features > 0 else features
    if remove_zero_features:
        keep &= (used != 0).any(dim=1)
    rows = torch.nonzero(keep).squeeze(1)

    if not sample_vertices:
        # the reference hands back un-batched rows with a [1, V] mask in this case (:76-80)
        return vertices[rows], used[rows], torch.ones((1, rows.shape[0]), dtype=torch.bool, device=vertices.device)

    n, want = int(rows.shape[0]), number_of_vertices_to_sample
    if n > want and vertex_sampling_method != VertexSamplingMethod.NONE:
        z = vertices[rows, 2] if vertex_sampling_method == VertexSamplingMethod.LOWEST else None
        rows = rows[select_vertex_indices(n, want, vertex_sampling_method, vertices.device, z=z)]
    # n <= want (pad + mask), NONE, or already exactly `want` rows: sample_to_n_vertices finishes the job without drawing
    v, f, valid_mask = sample_to_n_vertices(vertices[rows], used[rows], want, vertex_sampling_method)
    return v.unsqueeze(0), f.unsqueeze(0), valid_mask.unsqueeze(0)
