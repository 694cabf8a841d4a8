"""Per-item transforms of the data loader (mirror of mindmap/data_loading/sample_transformer.py:28-73).  They are plain
tensor ops and run wherever the sample lives -- on the GPU when the loader hands device tensors (section 8(f) N4)."""
import torch

from ..mapping.nvblox_mapper_constants import DEPTH_SCALE_FACTOR


class SampleTransformer:
    def reset(self):
        pass

    def __call__(self, sample: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


class RgbTransformer(SampleTransformer):
    """[H,W,3] in [0,255] -> [3,H,W] float32 in [0,1] (image_processing/image_conversions.py:13-38)."""

    def __call__(self, image):
        assert image.dim() == 3 and image.shape[-1] == 3
        return (image / 255.0).permute(2, 0, 1).type(torch.float32)


class DepthTransformer(SampleTransformer):
    """u16 millimetres -> float32 metres (:62-73)."""

    def __call__(self, image):
        return (image / DEPTH_SCALE_FACTOR).to(torch.float32)
