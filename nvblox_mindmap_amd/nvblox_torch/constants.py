"""nvblox_torch.constants (import site: mindmap/image_processing/feature_extraction.py:18,159-162,215).

Upstream ``feature_array_num_elements()`` is a compile-time constant of the nvblox build
(768 in the reference, docker/install_nvblox.sh:24-25).  Here the channel count is a run-time
property of each Mapper; this module holds the default new Mappers are created with.
"""
import os

_FEATURE_ARRAY_NUM_ELEMENTS = int(os.environ.get("MMF_FEATURE_ARRAY_NUM_ELEMENTS", "768"))


class constants:  # noqa: N801  (the reference calls constants.constants.feature_array_num_elements())
    @staticmethod
    def feature_array_num_elements() -> int:
        return _FEATURE_ARRAY_NUM_ELEMENTS

    @staticmethod
    def set_feature_array_num_elements(n: int) -> None:
        """Extension: choose the feature channel count (multiple of 8) for Mappers created afterwards."""
        global _FEATURE_ARRAY_NUM_ELEMENTS
        n = int(n)
        if n <= 0 or n % 8 != 0:
            raise ValueError("feature_array_num_elements must be a positive multiple of 8")
        _FEATURE_ARRAY_NUM_ELEMENTS = n

    @staticmethod
    def voxels_per_side() -> int:
        return 8


feature_array_num_elements = constants.feature_array_num_elements
set_feature_array_num_elements = constants.set_feature_array_num_elements
