"""Feature-map upsample + channel pad + fp16 cast in one HIP kernel.

Replaces, for the mapper's input, the chain  F.interpolate(bilinear, align_corners=False)
(mindmap/image_processing/feature_extraction.py:126-128,188-189) -> "b c h w -> b h w c" (:191) ->
zero-pad torch.cat (:198-210) -> .contiguous().to(float16) (mapping/helpers/nvblox_mapping_helpers.py:256),
which at the reference shape (512x512x768) moves ~3.6 GB per camera frame; the fused kernel writes the
403 MB f16 result once.
"""
from typing import Optional, Tuple

import torch

from .. import _lib


def upsample_features(features_chw: torch.Tensor, output_hw: Tuple[int, int], pad_to_channels: int, fma_contraction: Optional[bool] = None) -> torch.Tensor:
    """[C,h,w] (or [1,C,h,w]) float32 backbone output -> [Hf,Wf,pad_to_channels] float16, channels >= C zero.
    ``fma_contraction``: the spec switch of the same name (mmf_params) applied to this op's arithmetic (None: the process default,
    MMF_FMA_CONTRACTION -- what the mappers of the process are built with)."""
    if fma_contraction is None:
        from ..nvblox_torch.mapper_params import FMA_CONTRACTION_DEFAULT as fma_contraction
    if features_chw.ndim == 4:
        assert features_chw.shape[0] == 1
        features_chw = features_chw[0]
    if not features_chw.is_cuda:
        raise RuntimeError("upsample_features runs on the GPU only (no CPU fallback)")
    Cin, h, w = features_chw.shape
    Hf, Wf = int(output_hw[0]), int(output_hw[1])
    low = features_chw.to(torch.float32).permute(1, 2, 0).contiguous()  # tiny: h*w*C
    out = torch.empty((Hf, Wf, int(pad_to_channels)), dtype=torch.float16, device=features_chw.device)
    _lib.check(_lib.lib().mmf_upsample_features_spec(_lib.dptr(low), h, w, Cin, _lib.dptr(out), Hf, Wf, int(pad_to_channels),
                                                     1 if fma_contraction else 0, _lib.stream_ptr(features_chw.device)), "mmf_upsample_features")
    return out
