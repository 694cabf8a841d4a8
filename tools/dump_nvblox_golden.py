#!/usr/bin/env python3
"""nvblox PIN KIT -- dump golden vectors of the REAL integrator for this repository's parity tests.

Why: CUDA nvblox (`nvblox_torch`) is an empty submodule in the reference tree, so the integrator oracle of this repository
(oracle/mmf_oracle.c) restates nvblox from its published algorithm and is "parity unpinned" (DESIGN.md section 6).  This
script turns that into a one-command check for anybody with an NVIDIA box that has upstream `nvblox_torch` installed:

    python tools/dump_nvblox_golden.py --config bl --frames 24          # -> tests/golden/nvblox_bl_patches.npz
    python tools/dump_nvblox_golden.py --config ref --frames 12         # needs an nvblox built with 768 feature channels
    python tools/dump_nvblox_golden.py --config small --hole-mode pixels

Commit the .npz files; `pytest tests/test_gpu_nvblox_golden.py -m gpu` (MI355X) and `pytest tests/test_cpu_nvblox_golden.py`
(CPU oracle) then compare against them instead of skipping.

What it depends on: ONLY `nvblox_torch` (+ torch, numpy) and one pure-numpy file of this repository loaded by path
(nvblox_mindmap_amd/synthetic.py: the SURVEY 8(d) stream).  The integrate_frame mask algebra is restated below in numpy
(frame_masks; tests/test_cpu_nvblox_golden.py checks it against the golden masks generated from the reference's own
functions).  It does not import libmmfusion, the oracle or the mindmap package.

What it replays: the reference's call sequence, statement by statement --
    mapper.decay()                                                     closed_loop/policies/nvblox_diffuser_actor_policy.py:77
    depth_mask = input_mask & (depth > min_integration_distance_m)     mapping/helpers/nvblox_mapping_helpers.py:201-204
    mapper.add_depth_frame(depth, T_W_C.cpu(), K.cpu(), depth_mask_u8, id)            :207-209
    mapper.add_color_frame(rgb, T, K, mask_frame=depth_mask_u8, mapper_id=id)          :212-218
    feature_mask = border & nearest_up(erode(input_mask,k1) & erode(valid_depth,k2))   :222-253
    mapper.add_feature_frame(feat_f16, T, K_feat, feature_mask_u8, id)                 :255-261
  then   mapper.update_feature_mesh(id); mesh = mapper.get_feature_mesh(id)            nvblox_output_helpers.py:49-52
with the parameters get_nvblox_mapper sets (:40-70) for the DRILL_IN_BOX task (nvblox_mapper_constants.py:33-70).

`--backend mmf` runs the same script against THIS repository's drop-in (`nvblox_mindmap_amd.nvblox_torch`) on an MI355X: that
is how tests/test_gpu_nvblox_golden.py checks the kit itself (same code path, same file format) and it proves that the
kit touches nothing outside the API surface of SURVEY.md section 8(b).

File format (npz; every array little-endian):
  meta                json string: config, stream parameters, mapper parameters, frame indices, backend + version strings
  tsdf_indices        [n,3] int32  allocated TSDF block indices, sorted lexicographically
  feature_indices     [m,3] int32  same for the feature layer
  blocks_per_frame    [F] int32    allocated TSDF blocks after each frame (localises a divergence in time)
  tsdf_sample_idx     [K,3] int32  block indices of the K sampled TSDF blocks (every n/K-th of the sorted list)
  tsdf_sample         [K,8,8,8,2] f32   {distance, weight} per voxel, voxel index [x][y][z]
  feature_sample_idx  [K,3] int32
  feature_sample      [K,8,8,8,Cs+1] f32  Cs sampled channels (meta.feature_channel_sample) + weight (last)
  n_vertices          int          number of feature-mesh vertices
  vertices            [Vs,3] f32   every (V/Vs)-th vertex after a lexicographic sort (meta.vertex_stride)
  vertex_features     [Vs,Cs] f32  the same vertices' features, sampled channels
"""
import argparse
import importlib.util
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_by_path(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


S = _load_by_path("mmf_synthetic", "nvblox_mindmap_amd/synthetic.py")


def _square_dilate(a: np.ndarray, k: int) -> np.ndarray:
    """True where any pixel of the (2k+1)^2 window (clipped at the image border) is True: two running-sum passes."""
    if k <= 0:
        return a.copy()
    H, W = a.shape
    c = np.zeros((H, W + 1), dtype=np.int32)
    np.cumsum(a, axis=1, out=c[:, 1:])
    lo, hi = np.maximum(np.arange(W) - k, 0), np.minimum(np.arange(W) + k + 1, W)
    rows = (c[:, hi] - c[:, lo]) > 0
    c = np.zeros((H + 1, W), dtype=np.int32)
    np.cumsum(rows, axis=0, out=c[1:, :])
    lo, hi = np.maximum(np.arange(H) - k, 0), np.minimum(np.arange(H) + k + 1, H)
    return (c[hi, :] - c[lo, :]) > 0


def frame_masks(input_mask, depth, min_depth_m, k_input, k_depth, border_percent):
    """The two masks of integrate_frame for a feature image of the depth image's size (nvblox_mapping_helpers.py:201-253):
    depth_mask = input & (depth > min); feature_mask = erode(input, k_input) & erode(depth > min, k_depth) & border, where
    erode(m, k) = k rounds of a 3x3 max-pool on ~m (image_mask_operations.py:16-41) = NOT dilate_{(2k+1)^2}(NOT m), and the
    border mask clears int(percent/100 * size) pixels on every side (:44-68)."""
    valid = depth > np.float32(min_depth_m)
    eroded = ~_square_dilate(~input_mask, k_input) & ~_square_dilate(~valid, k_depth)
    H, W = depth.shape
    bh, bw = int(border_percent * 0.01 * H), int(border_percent * 0.01 * W)
    border = np.ones((H, W), dtype=bool)
    if bh > 0 and bw > 0:
        border[:bh] = border[-bh:] = False
        border[:, :bw] = border[:, -bw:] = False
    return input_mask & valid, eroded & border

# DRILL_IN_BOX row of TASK_TO_NVBLOX_MAPPER_CFG + COMMON_NVBLOX_MAPPER_CFG (mapping/nvblox_mapper_constants.py:33-41,62-70)
TASK = dict(voxel_size_m=0.01, projective_integrator_max_integration_distance_m=5.0, projective_appearance_integrator_measurement_weight=1.0,
            tsdf_decay_factor=0.98, min_integration_distance_m=0.30, static_mask_erosion_iterations=17,
            valid_depth_mask_erosion_iterations=20, feature_mask_border_percent=5,
            aabb_min_m=[-0.37, -0.75, -0.13], aabb_max_m=[0.95, 0.75, 0.65])

CONFIGS = {
    # name: (StreamConfig kwargs, feature channels)
    "bl": (dict(), 64),                                                                      # BASELINE configs[2]: 640x480, C=64
    "ref": (dict(width=512, height=512, fx=586.4, fy=586.4, cx=255.5, cy=255.5), 768),       # the reference's real shape
    "small": (dict(width=160, height=120, fx=131.25, fy=131.25, cx=79.5, cy=59.5), 16),      # seconds on the CPU oracle
}

# The medium / low-confidence items of the recalled spec.  Each is a NAMED parameter of this
# repository's integrator (mmf_params / oracle OrcParams); tests/pin_report.py flips them one at a time against a dumped
# file and reports which setting reproduces nvblox.  Items that are code-level choices rather than parameters are listed
# with flip=None so that the report prints them as "to check by hand".
SPEC_ITEMS = [
    dict(item="depth sampling rule", param="lin_interp_max_diff_vox", ours=2.0, flips=[0.0, 1e9],
         note="0 = nearest tap only, 1e9 = always bilinear, 2 = bilinear where the four taps agree within 2 voxels"),
    dict(item="TSDF measurement weight (upstream's WeightingFunctionType)", param="weighting_mode", ours=1, flips=[0, 2, 3, 4, 5],
         note="0 constant, 1 inverse-square, 2 constant-dropoff, 3 inverse-square-dropoff, 4 inverse-square-TSDF-distance-penalty, 5 linear-with-max"),
    dict(item="TSDF max weight", param="max_weight", ours=5.0, flips=[100.0, 1e4]),
    dict(item="truncation distance (voxels)", param="truncation_distance_vox", ours=4.0, flips=[2.0, 8.0]),
    dict(item="appearance max weight", param="appearance_max_weight", ours=5.0, flips=[100.0, 1e4]),
    dict(item="decayed-weight deallocation threshold", param="decayed_weight_threshold", ours=1e-3, flips=[1e-4, 1e-2]),
    dict(item="deallocate fully decayed blocks", param="deallocate_decayed_blocks", ours=1, flips=[0]),
    dict(item="mesh minimum weight", param="mesh_min_weight", ours=1e-4, flips=[1e-6, 1e-2]),
    dict(item="sphere-tracing ray subsampling (occlusion test)", param="st_subsampling", ours=4, flips=[1, 2]),
    dict(item="sphere-tracing surface epsilon (voxels)", param="st_surface_eps_vox", ours=0.1, flips=[0.5]),
    dict(item="sphere-tracing maximum steps", param="st_max_steps", ours=100, flips=[25, 400]),
    dict(item="sphere-tracing maximum ray length (m)", param="st_max_ray_length_m", ours=15.0, flips=[4.0]),
    dict(item="raycast marks blocks up to depth + truncation (0: up to the depth)", param="raycast_to_truncation", ours=1, flips=[0]),
    dict(item="decay leaves the colour / feature layers untouched (1: their weights fade too)", param="decay_appearance_layers", ours=0,
         flips=[1]),
    # the two places where the spec was arranged for the GPU: switchable, so that the pin decides
    dict(item="appearance blend: one reciprocal per voxel (1: a division per channel)", param="appearance_blend_division", ours=0, flips=[1],
         note="<= 1 ulp of the stored type either way; with 1 frames take the stand-alone appearance kernels"),
    dict(item="raycast block walk starts where the ray enters the workspace bounds (1: at the camera)", param="raycast_walk_from_camera",
         ours=0, flips=[1], note="the same block sets by construction (tests/test_cpu_raycast_walk.py): a flip that changes anything is a bug"),
    # the arithmetic mode: this spec never fuses a multiply with an add; nvcc does by default (-fmad=true)
    dict(item="multiply-adds contracted into FMAs (projection, bilinear samples, TSDF / appearance blend numerators)", param="fma_contraction",
         ours=0, flips=[1], note="a contraction-only difference shows up in (nearly) every voxel at the last bits: if flipping this one item "
                                 "removes it, upstream's build contracts where this restatement assumes it does"),
    # three more recollection risks (round 6): each switchable in the oracle and in the HIP library alike
    dict(item="block (and voxel) of a point by floor(p * (1 / size)) (1: floor(p / size))", param="block_index_by_division", ours=0, flips=[1],
         note="differs exactly for points within an ulp of a block / voxel face: workspace bounds, ray end points, the tracer's and the mesh's voxel look-ups"),
    dict(item="blocks in view: the traversed blocks of each pixel's ray (1: plus every block within the truncation distance of its surface point)",
         param="view_truncation_band_marking", ours=0, flips=[1], note="SURVEY.md App. A.2's second marking kernel: adds blocks, never removes one"),
    dict(item="bilinear samples as nested lerps (1: four weighted taps in upstream's order of terms)", param="bilinear_four_weight_sum", ours=0,
         flips=[1], note="depth, synthetic depth, colour and feature taps; a rounding-level difference in (nearly) every observed voxel"),
    dict(item="feature storage rounding (RNE)", param=None, flips=None, note="code: __float2half_rn"),
    dict(item="feature-mesh vertex takes the feature of the voxel containing it", param=None, flips=None, note="code: k_mesh_emit"),
]


def load_backend(name):
    if name == "nvblox":
        import nvblox_torch  # noqa: F401  (upstream)
        from nvblox_torch.mapper import Mapper
        from nvblox_torch.mapper_params import (BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams,
                                                TsdfDecayIntegratorParams, ViewCalculatorParams)
        from nvblox_torch.projective_integrator_types import ProjectiveIntegratorType
        from nvblox_torch.constants import constants

        version = getattr(nvblox_torch, "__version__", "unknown")
        return dict(Mapper=Mapper, BlockMemoryPoolParams=BlockMemoryPoolParams, MapperParams=MapperParams,
                    ProjectiveIntegratorParams=ProjectiveIntegratorParams, TsdfDecayIntegratorParams=TsdfDecayIntegratorParams,
                    ViewCalculatorParams=ViewCalculatorParams, TSDF=ProjectiveIntegratorType.TSDF,
                    feature_channels=int(constants.feature_array_num_elements()), version=f"nvblox_torch {version}", mapper_kwargs={})
    if name == "mmf":
        sys.path.insert(0, ROOT)
        from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
        from nvblox_mindmap_amd.nvblox_torch.mapper_params import (BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams,
                                                                 TsdfDecayIntegratorParams, ViewCalculatorParams)
        from nvblox_mindmap_amd.nvblox_torch.projective_integrator_types import ProjectiveIntegratorType

        return dict(Mapper=Mapper, BlockMemoryPoolParams=BlockMemoryPoolParams, MapperParams=MapperParams,
                    ProjectiveIntegratorParams=ProjectiveIntegratorParams, TsdfDecayIntegratorParams=TsdfDecayIntegratorParams,
                    ViewCalculatorParams=ViewCalculatorParams, TSDF=ProjectiveIntegratorType.TSDF, feature_channels=None,
                    version="nvblox_mindmap_amd (this repository's drop-in)", mapper_kwargs=None)
    raise ValueError(name)


def make_mapper(B, channels):
    """get_nvblox_mapper (nvblox_mapping_helpers.py:30-76) with ONE mapper (the static one)."""
    pi = B["ProjectiveIntegratorParams"]()
    pi.projective_integrator_max_integration_distance_m = TASK["projective_integrator_max_integration_distance_m"]
    pi.projective_appearance_integrator_measurement_weight = TASK["projective_appearance_integrator_measurement_weight"]
    de = B["TsdfDecayIntegratorParams"]()
    de.tsdf_decay_factor = TASK["tsdf_decay_factor"]
    vc = B["ViewCalculatorParams"]()
    vc.raycast_subsampling_factor = 1
    vc.workspace_bounds_type = "kBoundingBox"
    vc.workspace_bounds_min_corner_x_m, vc.workspace_bounds_min_corner_y_m, vc.workspace_bounds_min_height_m = TASK["aabb_min_m"]
    vc.workspace_bounds_max_corner_x_m, vc.workspace_bounds_max_corner_y_m, vc.workspace_bounds_max_height_m = TASK["aabb_max_m"]
    pool = B["BlockMemoryPoolParams"]()
    pool.expansion_factor = 1.0
    pool.num_preallocated_blocks = 0
    mp = B["MapperParams"]()
    mp.set_projective_integrator_params(pi)
    mp.set_tsdf_decay_integrator_params(de)
    mp.set_view_calculator_params(vc)
    mp.set_block_memory_pool_params(pool)
    kwargs = dict(voxel_sizes_m=[TASK["voxel_size_m"]], integrator_types=[B["TSDF"]], mapper_parameters=mp)
    if B["mapper_kwargs"] is None:  # this repository's Mapper takes the channel count at run time
        kwargs["feature_channels"] = channels
    return B["Mapper"](**kwargs)


def _np(x):
    import torch

    if isinstance(x, (list, tuple)):
        x = torch.stack([torch.as_tensor(v) for v in x]) if len(x) else torch.zeros((0,))
    return torch.as_tensor(x).detach().to("cpu").numpy()


def sort_rows(idx):
    idx = np.asarray(idx).reshape(-1, 3)
    return np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))


def replay(B, cfg_name, hole_mode, n_frames, use_decay, use_masks, n_block_samples=6, n_vertex_samples=4096, n_channel_samples=16,
           device="cuda", frame_indices=None):
    """Run the stream through backend B (see load_backend) and collect the arrays of the file format.  `device` / `frame_indices`
    exist for the consumers (tests run the same function on this repository's implementations and compare dict to dict)."""
    import torch

    kw, channels = CONFIGS[cfg_name]
    scfg = S.StreamConfig(hole_mode=hole_mode, **kw)
    if B.get("feature_channels") is not None and B["feature_channels"] != channels:
        raise SystemExit(f"this nvblox build has {B['feature_channels']} feature channels, config {cfg_name!r} needs {channels} "
                         f"(rebuild with -DNVBLOX_FEATURE_ARRAY_NUM_ELEMENTS={channels}, docker/install_nvblox.sh:24-25)")
    mapper = B["make_mapper"](channels) if "make_mapper" in B else make_mapper(B, channels)
    dev = torch.device(device)
    stride = max(scfg.num_poses // n_frames, 1)
    indices = list(frame_indices) if frame_indices is not None else [(k * stride) % scfg.num_poses for k in range(n_frames)]
    H, W = scfg.height, scfg.width
    K = torch.from_numpy(scfg.intrinsics())
    blocks_per_frame = []
    for idx in indices:
        f = S.frame(scfg, idx, channels)
        depth = f["depth"]
        input_mask = np.ones((H, W), dtype=bool)  # static mask of a frame without dynamic objects
        if use_masks:
            dm, fm = frame_masks(input_mask, depth, TASK["min_integration_distance_m"], TASK["static_mask_erosion_iterations"],
                                 TASK["valid_depth_mask_erosion_iterations"], TASK["feature_mask_border_percent"])
        else:
            dm, fm = depth > np.float32(TASK["min_integration_distance_m"]), np.ones((H, W), dtype=bool)
        T = torch.from_numpy(f["T_W_C"])
        dm_u8 = torch.from_numpy(dm.astype(np.uint8)).to(dev)
        if use_decay:
            mapper.decay()
        mapper.add_depth_frame(torch.from_numpy(depth).to(dev), T, K, dm_u8, 0)
        mapper.add_color_frame(torch.from_numpy(f["rgb"]).to(dev).contiguous(), T, K, mask_frame=dm_u8, mapper_id=0)
        mapper.add_feature_frame(torch.from_numpy(f["features"]).to(dev).contiguous(), T, K.clone(),
                                 torch.from_numpy(fm.astype(np.uint8)).to(dev), 0)
        blocks_per_frame.append(int(mapper.tsdf_layer_view(0).num_allocated_blocks()))

    out = {}
    tsdf_blocks, tsdf_idx = mapper.tsdf_layer_view(0).get_all_blocks()
    tsdf_blocks, tsdf_idx = _np(tsdf_blocks), _np(tsdf_idx).astype(np.int32).reshape(-1, 3)
    order = sort_rows(tsdf_idx)
    out["tsdf_indices"] = tsdf_idx[order]
    pick = order[:: max(len(order) // n_block_samples, 1)][:n_block_samples]
    out["tsdf_sample_idx"] = tsdf_idx[pick]
    out["tsdf_sample"] = tsdf_blocks[pick].astype(np.float32).reshape(len(pick), 8, 8, 8, 2)

    feat_blocks, feat_idx = mapper.feature_layer_view(0).get_all_blocks()
    feat_idx = _np(feat_idx).astype(np.int32).reshape(-1, 3)
    forder = sort_rows(feat_idx)
    out["feature_indices"] = feat_idx[forder]
    chan = np.unique(np.linspace(0, channels - 1, n_channel_samples).astype(np.int64))
    fpick = forder[:: max(len(forder) // n_block_samples, 1)][:n_block_samples]
    out["feature_sample_idx"] = feat_idx[fpick]
    sample = []
    for i in fpick:  # one block at a time: a 768-channel layer is GBs as float32
        blk = _np(feat_blocks[int(i)]).astype(np.float32).reshape(8, 8, 8, channels + 1)
        sample.append(np.concatenate([blk[..., chan], blk[..., -1:]], axis=-1))
    out["feature_sample"] = np.stack(sample) if sample else np.zeros((0, 8, 8, 8, len(chan) + 1), np.float32)
    out["blocks_per_frame"] = np.asarray(blocks_per_frame, dtype=np.int32)

    mapper.update_feature_mesh(0)
    mesh = mapper.get_feature_mesh(0)
    v = _np(mesh.vertices()).astype(np.float32).reshape(-1, 3)
    vf = mesh.vertex_features()
    vorder = sort_rows(np.round(v * 1e6).astype(np.int64)) if len(v) else np.zeros((0,), np.int64)
    vstride = max(len(v) // n_vertex_samples, 1)
    vpick = vorder[::vstride]
    out["n_vertices"] = np.asarray(len(v), dtype=np.int64)
    out["vertices"] = v[vpick]
    out["vertex_features"] = _np(vf[torch.as_tensor(vpick, device=vf.device)][:, torch.as_tensor(chan, device=vf.device)]).astype(np.float32) \
        if len(v) else np.zeros((0, len(chan)), np.float32)
    meta = dict(format=1, config=cfg_name, hole_mode=hole_mode, frames=n_frames, frame_indices=indices, stream=dict(
        width=W, height=H, fx=scfg.fx, fy=scfg.fy, cx=scfg.cx, cy=scfg.cy, num_poses=scfg.num_poses, radius_m=scfg.radius_m,
        height_m=scfg.height_m), feature_channels=channels, feature_channel_sample=[int(c) for c in chan], vertex_stride=int(vstride),
        decay=bool(use_decay), masks=bool(use_masks), task=TASK, backend=B["version"], torch=torch.__version__,
        spec_items=SPEC_ITEMS)
    out["meta"] = np.array(json.dumps(meta))
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", choices=sorted(CONFIGS), default="bl")
    ap.add_argument("--hole-mode", choices=["patches", "pixels"], default="patches",
                    help="SURVEY 8(d) prescribes 'pixels'; with the reference's 20-pixel valid-depth erosion those erase the "
                         "whole feature mask, so bench.py and the feature parity tests use 'patches' (DESIGN.md 3.1). Dump both.")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--no-decay", action="store_true", help="skip mapper.decay() before each frame")
    ap.add_argument("--no-masks", action="store_true", help="integrate with depth>min only (no erosion / border)")
    ap.add_argument("--backend", choices=["nvblox", "mmf"], default="nvblox")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    B = load_backend(args.backend)
    out = replay(B, args.config, args.hole_mode, args.frames, not args.no_decay, not args.no_masks)
    path = args.out or os.path.join(ROOT, "tests", "golden", f"nvblox_{args.config}_{args.hole_mode}.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out['tsdf_indices'])} TSDF blocks, {len(out['feature_indices'])} feature blocks, "
          f"{int(out['n_vertices'])} vertices ({os.path.getsize(path) / 1e6:.2f} MB)")
    return path


if __name__ == "__main__":
    main()
