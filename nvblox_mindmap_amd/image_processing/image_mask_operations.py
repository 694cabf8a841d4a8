"""Mask algebra of the fusion path on MI355X (mirror of mindmap/image_processing/image_mask_operations.py).

``erode_mask`` replaces k iterations of bool->float->max_pool2d->bool (image_mask_operations.py:35-41) by one
separable (2k+1)x(2k+1) dilation of the inverted mask (two HIP kernels); ``feature_mask`` fuses the whole
feature-mask computation of integrate_frame (nvblox_mapping_helpers.py:201-253) into two kernels.
"""
from typing import Tuple

import torch

from .. import _lib


def _u8(mask: torch.Tensor) -> torch.Tensor:
    if not mask.is_cuda:
        raise RuntimeError("mask operations run on the GPU only (no CPU fallback)")
    if mask.dtype == torch.bool:
        return mask.contiguous().view(torch.uint8)  # torch bools are 0/1 bytes: reinterpret, no copy
    return (mask if mask.dtype == torch.uint8 else mask.to(torch.uint8)).contiguous()


def frame_masks(input_mask: torch.Tensor, depth: torch.Tensor, min_depth_m: float, input_mask_erosion_iterations: int,
                valid_depth_mask_erosion_iterations: int, border_percent: int, feature_hw: Tuple[int, int]
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(depth_mask uint8 (H,W), feature_mask uint8 (Hf,Wf)) of integrate_frame in one library call
    (nvblox_mapping_helpers.py:201-204 and :222-253)."""
    H, W = depth.shape
    Hf, Wf = int(feature_hw[0]), int(feature_hw[1])
    m = _u8(input_mask)
    d = depth if (depth.dtype == torch.float32 and depth.is_contiguous()) else depth.to(torch.float32).contiguous()
    dm = torch.empty((H, W), dtype=torch.uint8, device=depth.device)
    fm = torch.empty((Hf, Wf), dtype=torch.uint8, device=depth.device)
    tmp = torch.empty((H * W + 8,), dtype=torch.uint8, device=depth.device)
    _lib.check(_lib.lib().mmf_frame_masks(_lib.dptr(m), _lib.dptr(d), H, W, float(min_depth_m), int(input_mask_erosion_iterations),
                                          int(valid_depth_mask_erosion_iterations), int(border_percent), Hf, Wf, _lib.dptr(dm),
                                          _lib.dptr(fm), _lib.dptr(tmp), _lib.stream_ptr(depth.device)), "mmf_frame_masks")
    return dm, fm


def erode_mask(mask: torch.Tensor, kernel_size: int = 3, iterations: int = 1) -> torch.Tensor:
    """Erodes a (H,W) bool mask: zeros grow by `iterations` pixels in all 8 directions (image border does not erode)."""
    assert mask.dim() == 2, "Mask must be 2D"
    assert kernel_size % 2 == 1, "Kernel size must be odd."
    assert mask.dtype == torch.bool, "Mask must be of type bool"
    H, W = mask.shape
    radius = iterations * ((kernel_size - 1) // 2)
    m = _u8(mask)
    out = torch.empty_like(m)
    tmp = torch.empty_like(m)
    _lib.check(_lib.lib().mmf_erode_mask(_lib.dptr(m), _lib.dptr(out), _lib.dptr(tmp), H, W, int(radius), _lib.stream_ptr(mask.device)),
               "mmf_erode_mask")
    return out.to(torch.bool)


def get_border_mask(mask_shape, mask_border_percent: float, device) -> Tuple[torch.Tensor, int, int]:
    """True everywhere except a border of int(percent*0.01*size) pixels (image_mask_operations.py:44-68)."""
    height, width = mask_shape[:2]
    mask = torch.full((height, width), True, dtype=torch.bool, device=device)
    border_h = int(mask_border_percent * 0.01 * height)
    border_w = int(mask_border_percent * 0.01 * width)
    if border_h > 0 and border_w > 0:
        mask[:border_h, :] = False
        mask[-border_h:, :] = False
        mask[:, :border_w] = False
        mask[:, -border_w:] = False
    return mask, border_h, border_w


def downscale_mask(mask: torch.Tensor, downscale_factor: int) -> torch.Tensor:
    """AND-pooling of a (B,1,H,W) bool mask (image_mask_operations.py:71-101; model side, not on the fusion path)."""
    assert downscale_factor > 0, "Downscale factor must be positive"
    assert mask.dim() == 4, "Mask must be 4D"
    assert mask.dtype == torch.bool, "Mask must be of type bool"
    assert mask.shape[2] % downscale_factor == 0 and mask.shape[3] % downscale_factor == 0
    f = downscale_factor
    view = mask.view(mask.shape[0], mask.shape[1], mask.shape[2] // f, f, mask.shape[3] // f, f)
    return torch.all(torch.all(view, dim=-1), dim=-2)


def depth_mask(input_mask: torch.Tensor, depth: torch.Tensor, min_depth_m: float) -> torch.Tensor:
    """uint8 (H,W): input_mask & (depth > min_depth_m)   (nvblox_mapping_helpers.py:201-204)."""
    H, W = depth.shape
    m = _u8(input_mask)
    d = depth.to(torch.float32).contiguous()
    out = torch.empty((H, W), dtype=torch.uint8, device=depth.device)
    _lib.check(_lib.lib().mmf_depth_mask(_lib.dptr(m), _lib.dptr(d), H, W, float(min_depth_m), _lib.dptr(out), _lib.stream_ptr(depth.device)),
               "mmf_depth_mask")
    return out


def feature_mask(input_mask: torch.Tensor, depth: torch.Tensor, min_depth_m: float, input_mask_erosion_iterations: int,
                 valid_depth_mask_erosion_iterations: int, border_percent: int, feature_hw: Tuple[int, int]) -> torch.Tensor:
    """uint8 (Hf,Wf): border & nearest_upsample(erode(input_mask,k1) & erode(depth > min_d, k2))
    (nvblox_mapping_helpers.py:222-253)."""
    H, W = depth.shape
    Hf, Wf = int(feature_hw[0]), int(feature_hw[1])
    m = _u8(input_mask)
    d = depth.to(torch.float32).contiguous()
    out = torch.empty((Hf, Wf), dtype=torch.uint8, device=depth.device)
    tmp = torch.empty((H, W), dtype=torch.uint8, device=depth.device)
    _lib.check(_lib.lib().mmf_feature_mask(_lib.dptr(m), _lib.dptr(d), H, W, float(min_depth_m), int(input_mask_erosion_iterations),
                                           int(valid_depth_mask_erosion_iterations), int(border_percent), Hf, Wf, _lib.dptr(out),
                                           _lib.dptr(tmp), _lib.stream_ptr(depth.device)), "mmf_feature_mask")
    return out
