"""Map -> model input tensors (mirror of mindmap/mapping/helpers/nvblox_output_helpers.py:22-91)."""
from typing import Optional, Tuple

import torch

from ...data_loading.vertex_sampling import VertexSamplingMethod, sample_to_n_vertices
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

    aabb_min_m = nvblox_mapping_config.aabb_min_m.to(vertices.device)
    aabb_max_m = nvblox_mapping_config.aabb_max_m.to(vertices.device)
    mask = torch.all(torch.logical_and(vertices > aabb_min_m, vertices < aabb_max_m), dim=1)
    vertices = vertices[mask]
    features = features[mask]
    if num_excess_features > 0:
        features = features[..., :-num_excess_features]
    if remove_zero_features:
        zero_feature_mask = torch.all(features == 0, dim=1)
        vertices = vertices[~zero_feature_mask]
        features = features[~zero_feature_mask]
    if not sample_vertices:
        valid_mask = torch.ones(vertices.shape[0], dtype=torch.bool, device=vertices.device).unsqueeze(0)
    else:
        vertices, features, valid_mask = sample_to_n_vertices(vertices, features, number_of_vertices_to_sample, vertex_sampling_method)
        if valid_mask.ndim == 1:
            vertices, features, valid_mask = vertices.unsqueeze(0), features.unsqueeze(0), valid_mask.unsqueeze(0)
    return vertices, features, valid_mask
