#!/usr/bin/env python3
"""Headline benchmark: RGB-D + 64-channel feature frames/s fused at 1 cm voxels on MI355X.

Workload = BASELINE.json configs[2] ("TSDF + 64-ch DINO feature-layer fusion, 640x480 stream, one
MI355X"), the configuration the metric is quoted on.  One step = the hot path over one synthetic frame,
exactly the call sequence of the reference's policy loop (closed_loop/policies/nvblox_diffuser_actor_policy.py:77-83):
    mapper.decay(); integrate_frame(depth, rgb, features)   ->  mask algebra, add_depth_frame,
    add_color_frame, add_feature_frame   (mapping/helpers/nvblox_mapping_helpers.py:162-273)
All inputs (depth, rgb, f16 HWC feature image, pose, K) are resident in HBM before the timed region.

Per-frame fusion does not shard: with --gpus N every rank runs an independent
replica (its own map, its own copy of the stream), no data-path collective; value = frames of all ranks /
max-over-ranks time ("weak" scaling).

Prints ONE JSON line (rank 0).  `roofline` is for k_feature_flat, the HBM-bound kernel that moves the feature rows of
the voxels a frame updates (the dominant kernel at the reference's 512x512x768 shape; at C=64 the frame is six
latency-bound launches, all listed in `kernel_us_per_launch`), timed with HIP events on the launch stream inside the
timed region; `cpu_baseline` is the CPU oracle ("port") timed on the host cores on a bounded sample of the same frames
(reported baseline, not the target).  Extra legs in the same line: `reference_shape` (512x512x768, with and without the
fused low-res feature path), `train` (policy training step/s, DDP over RCCL when launched with N > 1) and
`backprojection` (GPU kernel + torch-CPU baseline).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.image_processing.feature_resize import upsample_features  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper, integrate_frame  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3e12 achievable


def lowres_features(index: int, channels: int, lowres: int = 16) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(1000003 * (index + 1)))
    return rng.standard_normal((channels, lowres, lowres), dtype=np.float32)


def build_stream(cfg: S.StreamConfig, n_frames: int, channels: int, device):
    """Pre-generate the frames on the device: depth f32, rgb u8, features f16 HWC (via the HIP upsample
    kernel, the path's own K11 replacement), dynamic mask, pose, K."""
    frames = []
    stride = max(cfg.num_poses // n_frames, 1)
    for k in range(n_frames):
        idx = (k * stride) % cfg.num_poses
        T = S.camera_pose(cfg, idx)
        depth = S.render_depth(cfg, T)
        rgb = S.render_rgb(cfg, idx)
        low = torch.from_numpy(lowres_features(idx, channels)).to(device)
        feat = upsample_features(low, (cfg.height, cfg.width), channels)
        frames.append({
            "index": idx,
            "lowres": low.permute(1, 2, 0).contiguous(),  # [h,w,C] f32: the backbone output the image was made from
            "depth": torch.from_numpy(depth).to(device),
            "rgb": torch.from_numpy(rgb).to(device),
            "features": feat,
            "dynamic_mask": torch.zeros((cfg.height, cfg.width), dtype=torch.bool, device=device),
            "T_W_C": torch.from_numpy(T),
            "K": torch.from_numpy(cfg.intrinsics()),
        })
    torch.cuda.synchronize(device)
    return frames


def step(mapper, mcfg, fr):
    """decay + the STATIC-mapper half of nvblox_integrate (nvblox_mapping_helpers.py:116-141): static mask = ~dynamic mask
    (read inverted by the native call), depth + colour + feature integration."""
    mapper.decay()
    integrate_frame(mapper=mapper, nvblox_mapping_config=mcfg, depth_frame=fr["depth"], feature_frame=fr["features"],
                    intrinsics=fr["K"], camera_pose=fr["T_W_C"], rgb=fr["rgb"], input_mask=fr["dynamic_mask"],
                    input_mask_erosion_iterations=mcfg.static_mask_erosion_iterations,
                    valid_depth_mask_erosion_iterations=mcfg.valid_depth_mask_erosion_iterations,
                    mapper_id=MAPPER_TO_ID.STATIC, invert_input_mask=True)


def cpu_baseline(cfg, mcfg, frames, channels, n_sample):
    """Same steps on the CPU oracle (test infrastructure used here only as the reported baseline)."""
    from oracle import oracle as O

    O.build()
    p = O.default_params(
        voxel_size=mcfg.voxel_size_m, max_integration_distance_m=mcfg.projective_integrator_max_integration_distance_m,
        raycast_subsampling=1, workspace_bounds_type=2, ws_min=mcfg.aabb_min_m.tolist(), ws_max=mcfg.aabb_max_m.tolist(),
        tsdf_decay_factor=mcfg.tsdf_decay_factor,
        appearance_measurement_weight=mcfg.projective_appearance_integrator_measurement_weight, feature_channels=channels)
    orc = O.OracleMapper(p)
    host = []
    from nvblox_mindmap_amd.image_processing.image_mask_operations import depth_mask, feature_mask
    for fr in frames[:n_sample]:
        sm = ~fr["dynamic_mask"]
        dm = depth_mask(sm, fr["depth"], mcfg.min_integration_distance_m)
        fm = feature_mask(sm, fr["depth"], mcfg.min_integration_distance_m, mcfg.static_mask_erosion_iterations,
                          mcfg.valid_depth_mask_erosion_iterations, mcfg.feature_mask_border_percent, fr["features"].shape[:2])
        host.append((fr["depth"].cpu().numpy(), fr["rgb"].cpu().numpy(), fr["features"].cpu().numpy(), dm.cpu().numpy(),
                     fm.cpu().numpy(), fr["T_W_C"].numpy(), fr["K"].numpy()))
    t0 = time.perf_counter()
    for depth, rgb, feat, dm, fm, T, K in host:
        orc.decay()
        orc.add_depth_frame(depth, T, K, dm)
        orc.add_color_frame(rgb, T, K, dm)
        orc.add_feature_frame(feat, T, K, fm)
    dt = time.perf_counter() - t0
    return {
        "value": len(host) / dt,
        "unit": "frames/s",
        "cores": O.num_threads(),
        "kind": "port",
        "sample": f"first {len(host)} frames of the same stream, CPU oracle (C, OpenMP on the per-block loops; "
                  f"raycast single-threaded), masks precomputed",
    }


def flat_bytes_per_voxel(C: int) -> int:
    """Algorithmic bytes k_feature_flat moves per updated voxel: the voxel's f16 channel row read and written (2 x 2C), its
    four bilinear taps of the f16 feature image (4 x 2C) and its 20-byte survivor record."""
    return 2 * 2 * C + 4 * 2 * C + 20


def run_reference_shape(device, steps=24, warmup=6, n_frames=4):
    """Short untimed-for-the-headline run at the shape the reference really uses: 512x512 images,
    fx = 586.4 px, 768 feature channels (403 MB f16 feature image per frame)."""
    cfg = S.StreamConfig(width=512, height=512, fx=586.4, fy=586.4, cx=255.5, cy=255.5, hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    C = 768
    frames = build_stream(cfg, n_frames, C, device)
    mapper = get_nvblox_mapper(mcfg, feature_channels=C)
    for i in range(warmup):
        step(mapper, mcfg, frames[i % n_frames])
    torch.cuda.synchronize(device)
    mapper.reset_stats(MAPPER_TO_ID.STATIC)
    mapper.profile_reset()
    mapper.profile_enable(True, kernels=["feature", "feature_flat"])
    t0 = time.perf_counter()
    for i in range(steps):
        step(mapper, mcfg, frames[(warmup + i) % n_frames])
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    mapper.profile_enable(False)
    ms, n = mapper.profile()["feature_flat"]
    gate_ms, gate_n = mapper.profile()["feature"]
    st = mapper.stats(MAPPER_TO_ID.STATIC)
    fb = st["feature_blocks_updated"] / max(st["feature_frames"], 1)
    cb = st["color_blocks_updated"] / max(st["color_frames"], 1)
    vox = st["feature_voxels_updated"] / max(st["feature_frames"], 1)
    nbytes = vox * flat_bytes_per_voxel(C)
    out = {"image": [cfg.height, cfg.width], "feature_channels": C, "frames_per_s": steps / dt, "ms_per_step": dt / steps * 1e3,
           "feature_blocks_per_frame": fb, "feature_voxels_updated_per_frame": vox}
    if n:
        out["k_feature_flat_us"] = ms / n * 1e3
        out["k_feature_flat_algorithmic_bytes"] = nbytes
        out["k_feature_flat_algorithmic_GBps"] = nbytes / (ms / n * 1e-3) / 1e9
        out["k_feature_flat_frac_of_hbm_peak"] = nbytes / (ms / n * 1e-3) / HBM_PEAK_BYTES_PER_S
    if gate_n:
        out["k_app_frame_gating_us"] = gate_ms / gate_n * 1e3

    # The whole per-frame pipeline from the backbone's 16x16xC output (what the reference's FeatureExtractor hands over
    # before its own resize, feature_extraction.py:188-191): (a) up-sample to [512,512,768] f16 then integrate (two steps,
    # 403 MB image written and gathered); (b) the fused low-res path (mmf_integrate_frame_lowres), same results.
    def timed(fn):
        for i in range(warmup):
            fn(frames[i % n_frames])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            fn(frames[(warmup + i) % n_frames])
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / steps

    def with_upsample(fr):
        fr2 = dict(fr)
        fr2["features"] = upsample_features(fr["lowres"].permute(2, 0, 1), (cfg.height, cfg.width), C)
        step(mapper, mcfg, fr2)

    def fused_lowres(fr):
        mapper.decay()
        mapper.integrate_frame_lowres(fr["depth"], fr["rgb"], fr["lowres"], fr["dynamic_mask"], fr["T_W_C"], fr["K"],
                                      mcfg.min_integration_distance_m, mcfg.static_mask_erosion_iterations,
                                      mcfg.valid_depth_mask_erosion_iterations, mcfg.feature_mask_border_percent,
                                      MAPPER_TO_ID.STATIC, invert_input_mask=True)

    mapper.clear()
    dt_up = timed(with_upsample)
    mapper.clear()
    mapper.profile_reset()
    mapper.profile_enable(True, kernels=["feature_flat"])
    dt_low = timed(fused_lowres)
    mapper.profile_enable(False)
    ms, n = mapper.profile()["feature_flat"]
    out["from_backbone_output"] = {
        "upsample_then_integrate_frames_per_s": 1.0 / dt_up, "upsample_then_integrate_ms": dt_up * 1e3,
        "fused_lowres_frames_per_s": 1.0 / dt_low, "fused_lowres_ms": dt_low * 1e3,
        "fused_lowres_k_feature_flat_us": (ms / n * 1e3) if n else None,
        "upsampled_image_MB_avoided": cfg.height * cfg.width * C * 2 / 1e6}
    del mapper, frames
    torch.cuda.empty_cache()
    return out


def run_backprojection(device, cpu=True):
    """Depth back-projection: one HIP kernel, 4 B read + 12 B written per pixel.
    Timed for the 640x480 single frame and the training batch [32,512,512]; the CPU figure is the reference's op sequence
    on torch CPU tensors (oracle/image_ops.py:backproject_torch_cpu) with all host threads.  All GPU timing happens first
    (after a warm-up long enough to bring the clocks back up), the CPU legs afterwards."""
    from nvblox_mindmap_amd.image_processing.backprojection import _backproject_chw

    out, host = {}, {}
    for name, (B, H, W) in {"single_640x480": (1, 480, 640), "batch_32x512x512": (32, 512, 512)}.items():
        g = torch.Generator().manual_seed(B)
        depth = (torch.rand((B, H, W), generator=g) * 2.0 + 0.3)
        K = torch.tensor([[525.0, 0, W / 2 - 0.5], [0, 525.0, H / 2 - 0.5], [0, 0, 1]]).expand(B, 3, 3).contiguous()
        T = torch.eye(4).expand(B, 4, 4).clone()
        T[:, :3, 3] = torch.rand((B, 3), generator=g)
        host[name] = (depth, K, T)
        d_d, K_d, T_d = depth.to(device), K.to(device), T.to(device)
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            _backproject_chw(d_d, K_d, T_d)
        torch.cuda.synchronize(device)
        n = 300
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            _backproject_chw(d_d, K_d, T_d)  # includes the output allocation from torch's caching allocator
        b.record()
        torch.cuda.synchronize(device)
        ms = a.elapsed_time(b) / n
        px = B * H * W
        out[name] = {"frames_per_s": B / (ms * 1e-3), "ms_per_call": ms, "algorithmic_GBps": px * 16 / (ms * 1e-3) / 1e9,
                     "frac_of_hbm_peak": px * 16 / (ms * 1e-3) / HBM_PEAK_BYTES_PER_S, "bytes_per_pixel": 16}
    if cpu:
        from oracle.image_ops import backproject_torch_cpu

        torch.set_num_threads(os.cpu_count() or 1)
        for name, (depth, K, T) in host.items():
            B = depth.shape[0]
            backproject_torch_cpu(depth, K, T)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                backproject_torch_cpu(depth, K, T)
            dt = (time.perf_counter() - t0) / reps
            out[name]["cpu_frames_per_s"] = B / dt
            out[name]["cpu_threads"] = torch.get_num_threads()
    return out


def run_two_mappers(device, frames, channels, steps=100, warmup=20):
    """The reference's full nvblox_integrate (nvblox_mapping_helpers.py:79-159): decay of both mappers, then the frame into the
    STATIC mapper (mask = ~dynamic) and into the DYNAMIC one (mask = dynamic).  Timed with the two mappers' chains in sequence
    on one stream (the helper's default) and with the dynamic mapper's chain on a second stream (MMF_OVERLAP_MAPPERS=1)."""
    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H

    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")

    class Extractor:  # the backbone is out of scope here: the stream's feature image stands in for its output
        def compute(self, rgb):
            return self.next.unsqueeze(0)

    ex = Extractor()
    out = {}
    dyn = torch.zeros_like(frames[0]["dynamic_mask"])
    dyn[dyn.shape[0] // 4: 3 * dyn.shape[0] // 4, dyn.shape[1] // 3: 2 * dyn.shape[1] // 3] = True  # a sixth of the image is "dynamic"
    saved = H.OVERLAP_MAPPERS
    try:
        for name, overlap in (("sequential", False), ("two_streams", True)):
            H.OVERLAP_MAPPERS = overlap
            mapper = get_nvblox_mapper(mcfg, feature_channels=channels)
            if mapper.num_mappers() < 2:
                return None

            def step(i):
                fr = frames[i % len(frames)]
                ex.next = fr["features"]
                mapper.decay()
                H.nvblox_integrate(mapper, mcfg, ex, fr["depth"], fr["K"], fr["T_W_C"], fr["rgb"], dyn, include_dynamic=True)

            for i in range(warmup):
                step(i)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(steps):
                step(warmup + i)
            torch.cuda.synchronize(device)
            out[name + "_ms_per_frame"] = (time.perf_counter() - t0) / steps * 1e3
            del mapper
    finally:
        H.OVERLAP_MAPPERS = saved
    out["workload"] = "decay + nvblox_integrate(include_dynamic=True): static and dynamic mapper, a sixth of the image dynamic, 640x480, C=%d" % channels
    return out


def run_tsdf_only(device, steps=200, warmup=20):
    """BASELINE configs[1]: TSDF-only integration (decay + add_depth_frame: raycast, allocation, TSDF update) of the 640x480
    stream at 1 cm voxels, through the reference's stand-alone Mapper calls."""
    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    n = 50
    stride = cfg.num_poses // n
    frames = []
    for k in range(n):
        T = S.camera_pose(cfg, k * stride)
        frames.append((torch.from_numpy(S.render_depth(cfg, T)).to(device), torch.from_numpy(T), torch.from_numpy(cfg.intrinsics())))
    mapper = get_nvblox_mapper(mcfg, feature_channels=64)

    def step(i):
        d, T, K = frames[i % n]
        mapper.decay()
        mapper.add_depth_frame(d, T, K, None, MAPPER_TO_ID.STATIC)

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    del mapper
    return {"frames_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "workload": "decay + add_depth_frame, 640x480, 1 cm voxels"}


def run_closed_loop(device, steps=5):
    """BASELINE configs[3]: one control step of the closed loop on one GPU, end to end -- decay + fused RGB-D/feature frame
    (512x512, 768 feature channels, the reference's shape) -> surface vertices + features sampled to 2048
    (get_vertices_and_features) -> depth back-projection -> policy inference (encoder + 100 denoising steps, fused ops + HIP
    graph).  The backbone's feature extraction for the frame is part of the policy encoder here (random-init ViT-B/16)."""
    from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig
    from nvblox_mindmap_amd.image_processing.backprojection import get_camera_pointcloud
    from nvblox_mindmap_amd.mapping.helpers.nvblox_output_helpers import get_vertices_and_features
    from nvblox_mindmap_amd.training import build_model, synthetic_batch

    C = 768
    cfg = S.StreamConfig(width=512, height=512, fx=586.4, fy=586.4, cx=255.5, cy=255.5, hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    frames = build_stream(cfg, 4, C, device)
    mapper = get_nvblox_mapper(mcfg, feature_channels=C)
    pcfg = DiffuserActorConfig()
    torch.manual_seed(0)
    model = build_model(pcfg, device=device).eval()
    DiffuserActor.enable_fused_inference(True)
    model.enable_graph_sampling(True)
    hist = synthetic_batch(pcfg, 1, device, seed=3)["gripper_history"]
    parts = {"fusion": 0.0, "map_to_model_input": 0.0, "policy_inference": 0.0}

    def control_step(i, record):
        fr = frames[i % 4]
        t = [time.perf_counter()]
        step(mapper, mcfg, fr)
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        v, f, valid = get_vertices_and_features(mapper, MAPPER_TO_ID.STATIC, mcfg, remove_zero_features=True, num_excess_features=0,
                                                sample_vertices=True, number_of_vertices_to_sample=2048,
                                                vertex_sampling_method=VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT)
        q = torch.tensor([[0.5, -0.5, 0.5, -0.5]], device=device)
        pcd = get_camera_pointcloud(fr["K"].to(device)[None], fr["depth"][None], fr["T_W_C"][:3, 3].to(device)[None], q)
        rgb = (fr["rgb"].permute(2, 0, 1).float() / 255.0)[None, None]
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        with torch.no_grad():
            traj = model(None, None, rgb, pcd[:, None], (fr["depth"] > 0)[None, None], f.float(), v, valid, None, hist, run_inference=True)[0]
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        if record:
            for name, a, b in zip(parts, t[:-1], t[1:]):
                parts[name] += (b - a) * 1e3
        return traj

    try:
        for i in range(3):
            control_step(i, False)  # warm-up: fills the map, captures the graph
        t0 = time.perf_counter()
        for i in range(steps):
            control_step(3 + i, True)
        total = (time.perf_counter() - t0) / steps * 1e3
    finally:
        DiffuserActor.enable_fused_inference(False)
    out = {"ms_per_control_step": total, "control_steps_per_s": 1e3 / total, "breakdown_ms": {k: v / steps for k, v in parts.items()},
           "shape": "512x512 RGB-D, 768 feature channels, 2048 sampled vertices, 100 denoising steps, batch 1"}
    del mapper, model, frames
    torch.cuda.empty_cache()
    return out


def run_policy_inference(device, reps=3):
    """Closed-loop serving latency of the policy: batch 1, encoder once + 100 denoising steps of the
    diffusion head, eager and with the denoising loop replayed as one captured HIP graph (same results bit for bit)."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, synthetic_batch
    from nvblox_mindmap_amd.training.trainer import unpack_batch

    cfg = DiffuserActorConfig()
    torch.manual_seed(0)
    model = build_model(cfg, device=device).eval()
    s = unpack_batch(cfg, synthetic_batch(cfg, 1, device, seed=1))

    def infer():
        with torch.no_grad():
            return model(None, None, s["rgbs"], s["pcds"], s["pcd_valid_mask"], s["vertex_features"], s["vertices"],
                         s["vertices_valid_mask"], None, s["gripper_history"], run_inference=True)[0]

    def timed():
        infer()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            infer()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / reps * 1e3

    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor

    eager = timed()
    model.enable_graph_sampling(True)
    graphed = timed()  # its warm-up call captures the graph
    DiffuserActor.enable_fused_inference(True)  # fused rotary / AdaLN / attention / scheduler kernels, cached context K/V
    try:
        fused = timed()
    finally:
        DiffuserActor.enable_fused_inference(False)
        model.enable_graph_sampling(False)
    out = {"batch": 1, "diffusion_steps": cfg.diffusion_timesteps, "eager_ms": eager, "hip_graph_ms": graphed,
           "fused_ops_hip_graph_ms": fused, "inferences_per_s": 1e3 / fused, "dtype": "f32",
           "note": "hip_graph: same kernels, bit-identical; fused_ops: agrees to float rounding (tests/test_gpu_policy.py)"}
    del model
    torch.cuda.empty_cache()
    return out


def measure_d2d_copy(device, mib=1024, iters=10):
    """Device-to-device copy rate of the box (the measured counterpart of the 8 TB/s spec peak): bytes read +
    bytes written per second of a large torch copy."""
    src = torch.empty(mib * 1024 * 1024, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize(device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        dst.copy_(src)
    b.record()
    torch.cuda.synchronize(device)
    sec = a.elapsed_time(b) / iters * 1e-3
    del src, dst
    torch.cuda.empty_cache()
    return 2.0 * mib * 1024 * 1024 / sec / 1e9


def run_training(device, world, steps=8, warmup=3, per_gpu_batch=32, backbone_matmul_dtype="float32", prefetch_backbone=False):
    """Policy training step/s (second half of the BASELINE metric; config 5): diffuser_actor, RGBD_AND_MESH, per-GPU batch 32,
    one 512x512 camera, 2048 vertices x 768 features, frozen ViT-B/16-shaped backbone (random-init stand-in for RADIO v2.5-B),
    fp32, synthetic cached-sample-shaped batches resident on the GPU; DDP (RCCL all-reduce) when world > 1."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import (BackbonePrefetcher, build_model, build_optimizer, synthetic_batch, train_one_step,
                                             wrap_ddp)
    from nvblox_mindmap_amd.training.distributed import barrier, max_over_ranks

    torch.manual_seed(0)
    cfg = DiffuserActorConfig(backbone_matmul_dtype=backbone_matmul_dtype)
    model = build_model(cfg, device=device)
    n_train = sum(p.numel() for p in model.parameters() if p.requires_grad)
    n_frozen = sum(p.numel() for p in model.parameters() if not p.requires_grad)
    ddp = wrap_ddp(model, device)
    opt = build_optimizer(ddp)
    batches = [synthetic_batch(cfg, per_gpu_batch, device, seed=1000 * int(os.environ.get("RANK", "0")) + i) for i in range(2)]
    pre = BackbonePrefetcher(ddp, priority=int(os.environ.get("BENCH_PREFETCH_PRIORITY", "0"))) if prefetch_backbone else None

    def run(n, first, feats):
        # with the prefetcher: the frozen backbone of batch i+1 runs on a second stream next to the trainable pass of batch i;
        # every timed step executes exactly one backbone forward and one trainable forward/backward/optimizer step
        for i in range(first, first + n):
            nxt = pre.submit(batches[(i + 1) % 2]) if pre else None
            train_one_step(cfg, ddp, opt, batches[i % 2], backbone_feats=pre.wait(feats) if pre else None)
            feats = nxt
        return feats

    feats = run(warmup, 0, pre.submit(batches[0]) if pre else None)
    barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(steps, warmup, feats)
    torch.cuda.synchronize(device)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0, device if torch.distributed.get_backend() == "nccl" else None) if world > 1 else \
        time.perf_counter() - t0
    out = {"step_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "samples_per_s": steps * per_gpu_batch * world / dt,
           "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * world, "steps": steps, "warmup": warmup,
           "trainable_params": n_train, "frozen_backbone_params": n_frozen, "dtype": "f32", "backbone_prefetch": bool(prefetch_backbone),
           "allreduce_payload_MB": n_train * 4 / 1e6, "parallelism": f"dp{world}" if world > 1 else "single",
           "model": "diffuser_actor RGBD_AND_MESH, 1 cam 512x512, 2048 vertices x 768, frozen ViT-B/16-shaped backbone (random init)"}
    del model, ddp, opt, batches
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--frames", type=int, default=0, help="distinct pre-generated frames (default min(steps, 200))")
    ap.add_argument("--cpu-sample", type=int, default=12, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket kernels with HIP events")
    ap.add_argument("--no-ref-shape", action="store_true", help="skip the short run at the reference's real shape (512x512x768)")
    ap.add_argument("--no-train", action="store_true", help="skip the policy training-step measurement")
    ap.add_argument("--train-steps", type=int, default=8)
    ap.add_argument("--no-infer", action="store_true", help="skip the policy inference latency leg")
    ap.add_argument("--no-backproj", action="store_true", help="skip the back-projection leg (GPU kernel + torch-CPU baseline)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the fusion path has no CPU fallback")
    # one rank per GPU.  (BENCH_DIST_BACKEND=gloo lets the multi-rank control flow be exercised on a box with fewer GPUs than
    # ranks -- ranks then share devices modulo the device count; RCCL itself refuses two ranks on one GPU.)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist  # RCCL via backend "nccl": barrier + max-reduce of the timing, DDP all-reduce

        dist.init_process_group(backend=backend, init_method="env://")

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    n_frames = args.frames or min(max(args.steps, 1), 200)
    frames = build_stream(cfg, n_frames, args.channels, device)
    mapper = get_nvblox_mapper(mcfg, feature_channels=args.channels)

    for i in range(args.warmup):
        step(mapper, mcfg, frames[i % n_frames])
    torch.cuda.synchronize(device)
    mapper.reset_stats(MAPPER_TO_ID.STATIC)
    mapper.profile_reset()
    if not args.no_profile:
        # only the roofline kernel is timed inside the timed region, every 8th frame (the events cost host time)
        mapper.profile_enable(True, kernels=["feature_flat"], stride=8)

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(mapper, mcfg, frames[(args.warmup + i) % n_frames])
    t_enqueued = time.perf_counter() - t0  # host time to enqueue all steps (GPU may lag behind)
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    mapper.profile_enable(False)
    prof = mapper.profile()
    stats = mapper.stats(MAPPER_TO_ID.STATIC)
    # untimed pass with every kernel class bracketed: per-kernel breakdown for the report
    breakdown = {}
    if not args.no_profile:
        mapper.profile_reset()
        mapper.profile_enable(True)
        for i in range(min(args.steps, 50)):
            step(mapper, mcfg, frames[(args.warmup + args.steps + i) % n_frames])
        torch.cuda.synchronize(device)
        mapper.profile_enable(False)
        breakdown = {k: (v[0] / v[1] * 1e3 if v[1] else None) for k, v in mapper.profile().items()}

    ref_shape = None
    if rank == 0 and not args.no_ref_shape:
        ref_shape = run_reference_shape(device)
    train = None
    if not args.no_train:  # every rank takes part (DDP)
        if dist is not None:
            dist.barrier()
        train = run_training(device, world, steps=args.train_steps)  # strictly serial order: the figure compared across N
        if world == 1:
            # single GPU only (measured there; not exercised next to RCCL): the frozen backbone of the next batch on a second
            # stream beside the trainable pass of the current one (training.BackbonePrefetcher, bit-identical results)
            pre = run_training(device, world, steps=args.train_steps, prefetch_backbone=True)
            train["with_backbone_prefetch"] = {"step_per_s": pre["step_per_s"], "ms_per_step": pre["ms_per_step"]}
        # secondary figure, not the headline: the frozen backbone's matmuls with float16 inputs / fp32 accumulation -- the
        # mantissa width of the TF32 mode the reference runs its backbone in (feature_extraction.py:322); gfx950 has no TF32
        t16 = run_training(device, world, steps=args.train_steps, backbone_matmul_dtype="float16", prefetch_backbone=(world == 1))
        train["fp16_backbone_matmuls"] = {"step_per_s": t16["step_per_s"], "ms_per_step": t16["ms_per_step"],
                                          "note": "frozen backbone under float16 autocast (10-bit mantissa like the reference's TF32 "
                                                  "backbone, fp32 accumulate); everything trainable stays float32"}
    infer = run_policy_inference(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    closed_loop = run_closed_loop(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    tsdf_only = run_tsdf_only(device) if (rank == 0 and not args.no_ref_shape) else None
    two_mappers = run_two_mappers(device, frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    backproj = run_backprojection(device) if (rank == 0 and not args.no_backproj) else None  # has CPU legs: after every GPU measurement

    if rank == 0:
        C = args.channels
        n_feat_frames = max(stats["feature_frames"], 1)
        feat_blocks_per_frame = stats["feature_blocks_updated"] / n_feat_frames
        tsdf_blocks_per_frame = stats["tsdf_blocks_updated"] / max(stats["depth_frames"], 1)
        col_blocks_per_frame = stats["color_blocks_updated"] / max(stats["color_frames"], 1)
        feat_voxels_per_frame = stats["feature_voxels_updated"] / n_feat_frames
        # (1) SURVEY.md section 8(d) model: the whole feature image once + every voxel of every candidate block
        #     read and written (plus the colour image / voxels: the two updates are one launch, k_app_integrate2)
        model_bytes = (cfg.height * cfg.width * (2 * C + 1) + feat_blocks_per_frame * 512 * 2 * (2 * C + 4)
                       + cfg.height * cfg.width * (3 + 1) + col_blocks_per_frame * 512 * 16)
        # (2) roofline kernel = k_feature_flat, the balanced feature-row update that carries the path's bulk data.  Unit of
        #     work = a voxel the frame actually updates (device counter; the gating launch k_app_frame decides which).
        bytes_per_launch = feat_voxels_per_frame * flat_bytes_per_voxel(C)
        feat_ms, feat_n = prof["feature_flat"]
        roofline = None
        traffic = None
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/)
            with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
                pmc = json.load(f)["k_feature_flat"]
            traffic = (2.0 * pmc["FETCH_SIZE_KB"] + pmc["WRITE_SIZE_KB"]) * 1024.0  # FETCH_SIZE x2: gfx950 correction
        except Exception:
            traffic = None
        if feat_n > 0 and feat_ms > 0:
            avg_s = feat_ms / feat_n * 1e-3
            achieved = bytes_per_launch / avg_s
            roofline = {
                "bound": "hbm",
                "kernel": "k_feature_flat (balanced feature-row update of the frame's surviving voxels)",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK_BYTES_PER_S / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_BYTES_PER_S,
                "traffic": traffic,
                "avg_launch_us": avg_s * 1e6,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "bytes_per_unit": flat_bytes_per_voxel(C),
                "unit_of_work": "feature voxel updated (passed the occlusion/mask gate)",
                "feature_voxels_updated_per_launch": feat_voxels_per_frame,
                "feature_blocks_per_launch": feat_blocks_per_frame,
                "measured_d2d_copy_GBps": measure_d2d_copy(device),  # read + write rate of a 1 GiB copy on this box
                "note": "at C=64 the frame is 5 latency-bound launches of 7-17 us (kernel_us_per_launch); this kernel is "
                        "the HBM-bound one and dominates at the reference shape (reference_shape.k_feature_flat_*)",
                "survey_8d_model_bytes_per_frame": model_bytes,
            }
        frame_bytes = cfg.height * cfg.width * (4 + 1) + tsdf_blocks_per_frame * 512 * 16 + model_bytes
        cpu = None
        if args.cpu_sample > 0:
            cpu = cpu_baseline(cfg, mcfg, frames, C, min(args.cpu_sample, n_frames))
        fps = world * args.steps / elapsed
        out = {
            "metric": "RGB-D+feature frames/s fused @1 cm voxels",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: decay + depth(TSDF) + colour + 64-ch f16 feature fusion, 640x480 stream, "
                            "1 cm voxels, DRILL_IN_BOX workspace; replicas only for n_gpus>1",
                "image": [cfg.height, cfg.width],
                "feature_channels": C,
                "voxel_size_m": mcfg.voxel_size_m,
                "distinct_frames": n_frames,
                "tsdf_blocks_per_frame": tsdf_blocks_per_frame,
                "feature_blocks_per_frame": feat_blocks_per_frame,
                "survey_8d_model_bytes_per_frame": frame_bytes,
                "survey_8d_model_whole_frame_GBps": frame_bytes * fps / world / 1e9,
            },
            "roofline": roofline,
            "cpu_baseline": cpu,
            "kernel_us_per_launch": breakdown,
            "reference_shape": ref_shape,
            "policy_inference": infer,
            "closed_loop": closed_loop,
            "tsdf_only": tsdf_only,
            "two_mappers": two_mappers,
            "backprojection": backproj,
            "train": train,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()  # rank 0 runs its single-GPU legs after the timed regions: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
