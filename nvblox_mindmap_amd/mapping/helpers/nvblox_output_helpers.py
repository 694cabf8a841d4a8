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

    aabb_min_m = nvblox_mapping_config.aabb_min_m.to(vertices.device)
    aabb_max_m = nvblox_mapping_config.aabb_max_m.to(vertices.device)
    mask = torch.all(torch.logical_and(vertices > aabb_min_m, vertices < aabb_max_m), dim=1)
    if (sample_vertices and number_of_vertices_to_sample is not None
            and vertex_sampling_method not in (None, VertexSamplingMethod.NONE)):
        # Same rows, same order, same RNG draws as the step-by-step form below -- but ONE gather of the N sampled rows instead
        # of two boolean-mask copies of the whole [V, C] feature matrix (80 MB at V = 26 k, C = 768): the two filters are
        # combined into one mask over the ORIGINAL rows and the sampled indices are mapped through the kept-row list.
        used = features if num_excess_features <= 0 else features[..., :-num_excess_features]
        keep = mask & torch.any(used != 0, dim=1) if remove_zero_features else mask
        kept = torch.nonzero(keep).squeeze(1)
        n = int(kept.shape[0])
        if n > number_of_vertices_to_sample:
            sel = select_vertex_indices(n, number_of_vertices_to_sample, vertex_sampling_method, vertices.device,
                                        z=vertices[kept, 2] if vertex_sampling_method == VertexSamplingMethod.LOWEST else None)
            rows = kept[sel]
            valid_mask = torch.ones(number_of_vertices_to_sample, device=vertices.device, dtype=torch.bool)
            return vertices[rows].unsqueeze(0), used[rows].unsqueeze(0), valid_mask.unsqueeze(0)
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
