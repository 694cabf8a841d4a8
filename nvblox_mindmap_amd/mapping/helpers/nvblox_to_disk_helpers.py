"""Map outputs to disk (mirror of mindmap/mapping/helpers/nvblox_to_disk_helpers.py:21-96)."""
import os

from ...io.dataset_files import VERTEX_FEATURES_FILE_NAME, write_vertex_features
from ...nvblox_torch.mapper import Mapper
from ..nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg
from .nvblox_output_helpers import get_vertices_and_features


def save_feature_mesh_to_disk(mapper: Mapper, mapping_config: NvbloxMappingCfg, num_excess_features: int, frame_index: int,
                              save_directory: str, include_dynamic: bool):
    """Save the feature vertices of the STATIC mapper as ``NNNN.nvblox_vertex_features.zst`` (:21-67)."""
    assert not include_dynamic, "Dynamics are not supported for mesh encoding yet."
    vertices, features, _ = get_vertices_and_features(mapper, MAPPER_TO_ID.STATIC, mapping_config, remove_zero_features=True,
                                                      num_excess_features=num_excess_features, sample_vertices=False)
    assert vertices.shape[0] == features.shape[0]
    assert vertices.shape[1] == 3
    write_vertex_features(os.path.join(save_directory, f"{frame_index:04d}.{VERTEX_FEATURES_FILE_NAME}"), vertices, features)
    return vertices, features


def save_serialized_nvblox_map_to_disk(mapper: Mapper, save_directory: str, index: int, include_dynamic: bool) -> None:
    """``NNNN.nvblox_map_static.nvblx`` (and ``..._dynamic.nvblx``) via Mapper.save_map (:70-96)."""
    mapper.save_map(os.path.join(save_directory, f"{index:04d}.nvblox_map_static.nvblx"), MAPPER_TO_ID.STATIC)
    if include_dynamic:
        mapper.save_map(os.path.join(save_directory, f"{index:04d}.nvblox_map_dynamic.nvblx"), MAPPER_TO_ID.DYNAMIC)
