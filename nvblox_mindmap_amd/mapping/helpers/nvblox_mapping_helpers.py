"""Frame integration on MI355X (mirror of mindmap/mapping/helpers/nvblox_mapping_helpers.py:30-273).

Same function names, arguments and returned dictionaries as the reference.  Differences, all on the
performance side: the mask algebra is two fused HIP kernels instead of 37 max-pools + casts, and the
feature intrinsics are a scaled COPY (the reference scales the caller's tensor in place,
nvblox_mapping_helpers.py:233-234 -- harmless there only because the factor is 1.0).
"""
import os
from typing import Dict, Optional

import torch

from ...image_processing.image_mask_operations import frame_masks as _frame_masks
from ...nvblox_torch.mapper import Mapper
from ...nvblox_torch.mapper_params import (
    BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams)
from ...nvblox_torch.projective_integrator_types import ProjectiveIntegratorType
from ...nvblox_torch.timer import Timer
from ..nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg


# nvblox_integrate(include_dynamic=True): both mappers' frames as roles of the SAME five launches, one native call
# (Mapper.integrate_frame_multi).  On by default; MMF_PAIR_MAPPERS=0 issues the two integrate_frame calls one after the other.
PAIR_MAPPERS = os.environ.get("MMF_PAIR_MAPPERS", "1") != "0"
# A feature extractor may offer ``compute_lowres(rgb) -> (low [h, w, C] float32, (Hf, Wf))`` next to the reference's
# ``compute``: the backbone's own output and the size ``compute`` would resize it to.  nvblox_integrate then hands the LOW-RES
# map to the native call, which evaluates f16(bilinear(low)) at each tap itself -- bit-identical to integrating compute()'s
# image (tests/test_gpu_facade.py), without ever materialising it (403 MB at 512x512x768).  MMF_LOWRES_FEATURES=0: always compute().
LOWRES_FEATURES = os.environ.get("MMF_LOWRES_FEATURES", "1") != "0"

class _IntegrationImages(dict):
    """The images dictionary integrate_frame returns (:263-271).  Entries only the visualiser consumes are computed on
    first access instead of on every frame: ``rgb_frame`` (CHW float in [0,1]) and, when the native call was given the
    dynamic mask to use inverted, ``input_mask`` (= ~dynamic_mask)."""

    def __init__(self, items, rgb, lazy=None):
        super().__init__(items)
        self._lazy = {"rgb_frame": lambda: rgb.permute(2, 0, 1) / 255.0}
        if lazy:
            self._lazy.update(lazy)

    def __missing__(self, key):
        if key in self._lazy:
            value = self._lazy[key]()
            self[key] = value
            return value
        raise KeyError(key)

    def __contains__(self, key):
        return key in self._lazy or super().__contains__(key)

    def _materialise(self):
        for k in self._lazy:
            self[k]

    def keys(self):
        self._materialise()
        return super().keys()

    def items(self):
        self._materialise()
        return super().items()

    def values(self):
        self._materialise()
        return super().values()


def get_nvblox_mapper(mapper_config: NvbloxMappingCfg, feature_channels: Optional[int] = None) -> Mapper:
    """Two-mapper Mapper (STATIC=0, DYNAMIC=1) with the reference's parameter choices (:40-76)."""
    pi = ProjectiveIntegratorParams()
    pi.projective_integrator_max_integration_distance_m = mapper_config.projective_integrator_max_integration_distance_m
    pi.projective_appearance_integrator_measurement_weight = mapper_config.projective_appearance_integrator_measurement_weight
    de = TsdfDecayIntegratorParams()
    de.tsdf_decay_factor = mapper_config.tsdf_decay_factor
    vc = ViewCalculatorParams()
    vc.raycast_subsampling_factor = 1
    vc.workspace_bounds_type = "kBoundingBox"
    vc.workspace_bounds_min_corner_x_m = float(mapper_config.aabb_min_m[0])
    vc.workspace_bounds_min_corner_y_m = float(mapper_config.aabb_min_m[1])
    vc.workspace_bounds_min_height_m = float(mapper_config.aabb_min_m[2])
    vc.workspace_bounds_max_corner_x_m = float(mapper_config.aabb_max_m[0])
    vc.workspace_bounds_max_corner_y_m = float(mapper_config.aabb_max_m[1])
    vc.workspace_bounds_max_height_m = float(mapper_config.aabb_max_m[2])
    pool = BlockMemoryPoolParams()
    pool.expansion_factor = 1.0
    pool.num_preallocated_blocks = 0
    mp = MapperParams()
    mp.set_projective_integrator_params(pi)
    mp.set_tsdf_decay_integrator_params(de)
    mp.set_view_calculator_params(vc)
    mp.set_block_memory_pool_params(pool)
    return Mapper(
        voxel_sizes_m=[mapper_config.voxel_size_m, mapper_config.voxel_size_m],
        integrator_types=[ProjectiveIntegratorType.TSDF, ProjectiveIntegratorType.TSDF],
        mapper_parameters=mp,
        feature_channels=feature_channels,
    )


def integrate_frame(mapper: Mapper, nvblox_mapping_config: NvbloxMappingCfg, depth_frame: torch.Tensor,
                    feature_frame: torch.Tensor, intrinsics: torch.Tensor, camera_pose: torch.Tensor, rgb: torch.Tensor,
                    input_mask: torch.Tensor, input_mask_erosion_iterations: int,
                    valid_depth_mask_erosion_iterations: int, mapper_id: int,
                    invert_input_mask: bool = False) -> Dict[str, torch.Tensor]:
    """Depth + colour + feature integration of one frame into `mapper_id` (:162-273).
    Extension: ``invert_input_mask=True`` integrates with ``~input_mask`` (how nvblox_integrate derives the static mask
    from the dynamic one, :116-117); the native call reads the mask inverted, no inverted copy is made."""
    assert input_mask.dtype == torch.bool
    cfg = nvblox_mapping_config
    H, W = depth_frame.shape
    Hf, Wf = feature_frame.shape[0], feature_frame.shape[1]

    feat16 = feature_frame if feature_frame.dtype == torch.float16 else feature_frame.to(torch.float16)
    if (Hf, Wf) == (H, W):
        # the reference's only configuration (512 -> 512): the whole function body is one native call
        depth_mask_u8, feature_mask = mapper.integrate_frame(
            depth_frame, rgb, feat16, input_mask, camera_pose, intrinsics, cfg.min_integration_distance_m,
            input_mask_erosion_iterations, valid_depth_mask_erosion_iterations, cfg.feature_mask_border_percent, mapper_id,
            invert_input_mask=invert_input_mask)
        depth_mask = depth_mask_u8.view(torch.bool)
        items = {
            "depth_frame": depth_frame,
            "depth_mask": depth_mask,
            "rgb_mask": depth_mask,
            "feature_frame": feature_frame,
            "feature_mask": feature_mask,
        }
        if invert_input_mask:
            return _IntegrationImages(items, rgb, {"input_mask": lambda: ~input_mask})
        items["input_mask"] = input_mask
        return _IntegrationImages(items, rgb)

    if invert_input_mask:
        input_mask = ~input_mask

    # depth_mask = input_mask & (depth > min_integration_distance)  (:201-204) and
    # feature_mask = border & nearest_upsample(erode(input_mask, k1) & erode(depth > min_d, k2))  (:222-253):
    # both from one library call (two kernels)
    depth_mask_u8, feature_mask = _frame_masks(input_mask, depth_frame, cfg.min_integration_distance_m,
                                               input_mask_erosion_iterations, valid_depth_mask_erosion_iterations,
                                               cfg.feature_mask_border_percent, (Hf, Wf))
    pose_host = camera_pose.detach().to("cpu", torch.float32)
    k_host = intrinsics.detach().to("cpu", torch.float32)

    mapper.add_depth_frame(depth_frame, pose_host, k_host, depth_mask_u8, mapper_id)
    mapper.add_color_frame(rgb.contiguous(), pose_host, k_host, mask_frame=depth_mask_u8, mapper_id=mapper_id)

    # intrinsics of the feature image: first two rows scaled (:229-234) -- per axis, so non-square images work
    feature_intrinsics = k_host.clone()
    feature_intrinsics[0, :] *= Wf / W
    feature_intrinsics[1, :] *= Hf / H

    mapper.add_feature_frame(feat16.contiguous(), pose_host, feature_intrinsics, feature_mask, mapper_id)

    depth_mask = depth_mask_u8.view(torch.bool)  # 0/1 bytes: reinterpret, no copy
    return _IntegrationImages({
        "depth_frame": depth_frame,
        "depth_mask": depth_mask,
        "rgb_mask": depth_mask,
        "feature_frame": feature_frame,
        "feature_mask": feature_mask,
        "input_mask": input_mask,
    }, rgb)


def _release(feature_extractor) -> None:
    """End of a frame: an extractor that held its network output between compute_lowres and compute lets go of it."""
    rel = getattr(feature_extractor, "release_lowres", None)
    if rel is not None:
        rel()


def follow_mapper_arithmetic(feature_extractor, mapper: Mapper) -> None:
    """An extractor that up-samples with this library's kernel (``fma_contraction`` attribute, e.g. BackboneFeatureExtractor) takes the
    spec switch from the mapper it feeds: ``compute()`` + add_feature_frame and the fused low-res route then agree bit for bit also
    when the mapper's parameter was set explicitly and differs from the process default (round-5 advisor finding)."""
    if hasattr(feature_extractor, "fma_contraction") and feature_extractor.fma_contraction != mapper.fma_contraction:
        feature_extractor.fma_contraction = mapper.fma_contraction


def nvblox_integrate(mapper: Mapper, nvblox_mapping_config: NvbloxMappingCfg, feature_extractor, depth_frame: torch.Tensor,
                     intrinsics: torch.Tensor, camera_pose: torch.Tensor, rgb: torch.Tensor, dynamic_mask: torch.Tensor,
                     include_dynamic: bool) -> Dict[str, Dict[str, torch.Tensor]]:
    """Extract features and integrate the frame into the STATIC (and optionally DYNAMIC) mapper (:79-159)."""
    assert dynamic_mask.dtype == torch.bool
    cfg = nvblox_mapping_config
    out = {}
    follow_mapper_arithmetic(feature_extractor, mapper)
    # static_mask = ~dynamic_mask (:116-117): the native call reads the dynamic mask inverted instead
    use_dyn = bool(cfg.use_dynamic_mask)

    def jobs_for(masks_static, masks_dynamic):
        jobs = [{"mapper_id": MAPPER_TO_ID.STATIC, "input_mask": masks_static, "invert_input_mask": use_dyn,
                 "input_mask_erosion_iterations": cfg.static_mask_erosion_iterations,
                 "valid_depth_mask_erosion_iterations": cfg.valid_depth_mask_erosion_iterations}]
        if include_dynamic:
            jobs.append({"mapper_id": MAPPER_TO_ID.DYNAMIC, "input_mask": masks_dynamic, "invert_input_mask": False,
                         "input_mask_erosion_iterations": cfg.dynamic_mask_erosion_iterations,
                         "valid_depth_mask_erosion_iterations": cfg.valid_depth_mask_erosion_iterations})
        return jobs

    def images_of(jobs, masks, feature_frame, lazy_feature_frame=None):
        for job, (dm_u8, fm) in zip(jobs, masks):
            dm = dm_u8.view(torch.bool)
            items = {"depth_frame": depth_frame, "depth_mask": dm, "rgb_mask": dm, "feature_mask": fm}
            lazy = {}
            if lazy_feature_frame is not None:
                lazy["feature_frame"] = lazy_feature_frame
            else:
                items["feature_frame"] = feature_frame
            im = job["input_mask"]
            if job["invert_input_mask"]:
                lazy["input_mask"] = lambda im=im: ~im
            else:
                items["input_mask"] = im
            out[MAPPER_TO_ID(job["mapper_id"]).name] = _IntegrationImages(items, rgb, lazy)
        return out

    static_in = dynamic_mask if use_dyn else torch.ones_like(dynamic_mask)
    rgb1 = rgb.unsqueeze(0)  # ONE tensor object for compute_lowres and the fallback compute: the extractor's hand-over is by identity
    if LOWRES_FEATURES and depth_frame.is_cuda and hasattr(feature_extractor, "compute_lowres"):
        with Timer("nvblox_mapper/compute_features"):
            low, feature_size = feature_extractor.compute_lowres(rgb=rgb1)
        if low is not None and tuple(feature_size) == tuple(depth_frame.shape):
            # (the feature image of the returned dictionaries -- read by the visualiser only -- is materialised on first access)
            jobs = jobs_for(static_in, dynamic_mask)
            if len(jobs) == 1:  # (the single-frame entry point: less host work than a one-element job list)
                masks = [mapper.integrate_frame_lowres(depth_frame, rgb, low, static_in, camera_pose, intrinsics,
                                                       cfg.min_integration_distance_m, cfg.static_mask_erosion_iterations,
                                                       cfg.valid_depth_mask_erosion_iterations, cfg.feature_mask_border_percent,
                                                       MAPPER_TO_ID.STATIC, invert_input_mask=use_dyn)]
            else:
                masks = mapper.integrate_frame_multi(depth_frame, rgb, None, camera_pose, intrinsics, cfg.min_integration_distance_m,
                                                     cfg.feature_mask_border_percent, jobs, lowres_features=low)
            _release(feature_extractor)
            return images_of(jobs, masks, None, lambda: feature_extractor.compute(rgb=rgb.unsqueeze(0)).squeeze(0))
    with Timer("nvblox_mapper/compute_features"):
        feature_frame = feature_extractor.compute(rgb=rgb1).squeeze(0)  # (takes compute_lowres's held network output, if any)
    _release(feature_extractor)

    def static_half():
        return integrate_frame(
            mapper=mapper, nvblox_mapping_config=cfg, depth_frame=depth_frame, feature_frame=feature_frame, intrinsics=intrinsics,
            camera_pose=camera_pose, rgb=rgb, input_mask=dynamic_mask if use_dyn else torch.ones_like(dynamic_mask),
            input_mask_erosion_iterations=cfg.static_mask_erosion_iterations,
            valid_depth_mask_erosion_iterations=cfg.valid_depth_mask_erosion_iterations, mapper_id=MAPPER_TO_ID.STATIC,
            invert_input_mask=use_dyn)

    def dynamic_half():
        return integrate_frame(
            mapper=mapper, nvblox_mapping_config=cfg, depth_frame=depth_frame, feature_frame=feature_frame, intrinsics=intrinsics,
            camera_pose=camera_pose, rgb=rgb, input_mask=dynamic_mask,
            input_mask_erosion_iterations=cfg.dynamic_mask_erosion_iterations,
            valid_depth_mask_erosion_iterations=cfg.valid_depth_mask_erosion_iterations, mapper_id=MAPPER_TO_ID.DYNAMIC)

    Hf, Wf = feature_frame.shape[0], feature_frame.shape[1]
    if include_dynamic and PAIR_MAPPERS and depth_frame.is_cuda and (Hf, Wf) == tuple(depth_frame.shape):
        # Both mappers in ONE native call: the two frames are roles of the same five launches (one latency chain, one enqueue;
        # bit-identical to the two integrate_frame calls below).
        feat16 = feature_frame if feature_frame.dtype == torch.float16 else feature_frame.to(torch.float16)
        jobs = jobs_for(static_in, dynamic_mask)
        masks = mapper.integrate_frame_multi(depth_frame, rgb, feat16, camera_pose, intrinsics, cfg.min_integration_distance_m,
                                             cfg.feature_mask_border_percent, jobs)
        return images_of(jobs, masks, feature_frame)
    out[MAPPER_TO_ID.STATIC.name] = static_half()
    if include_dynamic:
        out[MAPPER_TO_ID.DYNAMIC.name] = dynamic_half()
    return out
