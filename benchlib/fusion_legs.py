"""The fusion-side legs of bench.py beside the headline: CPU baseline (the oracle, `cpu_baseline` only), the reference's real
shape, back-projection, two mappers per frame, N replicas per launch, TSDF only, map -> model inputs, the hash path (unbounded
workspace) and SURVEY 8(d)'s prescribed pixel holes.  Each returns a plain dict; bench.py assembles the record."""
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import ROOT, S, MAPPER_TO_ID, NvbloxMappingCfg, get_nvblox_mapper, integrate_frame, upsample_features  # noqa: F401

def cpu_baseline(cfg, mcfg, frames, channels, n_sample, budget_s=25.0):
    """Same steps on the CPU oracle (test infrastructure used here only as the reported baseline): C + OpenMP (raycast rows,
    TSDF / colour / feature blocks and sphere-traced rows in parallel).  Swept over thread counts -- on a many-core host the
    best setting is rarely "all" -- each setting on a fresh map and the same first frames of the stream; the best is reported."""
    from oracle import oracle as O

    O.build()
    host = []
    from nvblox_mindmap_amd.image_processing.image_mask_operations import depth_mask, feature_mask
    for fr in frames[:n_sample]:
        sm = ~fr["dynamic_mask"]
        dm = depth_mask(sm, fr["depth"], mcfg.min_integration_distance_m)
        fm = feature_mask(sm, fr["depth"], mcfg.min_integration_distance_m, mcfg.static_mask_erosion_iterations,
                          mcfg.valid_depth_mask_erosion_iterations, mcfg.feature_mask_border_percent, fr["features"].shape[:2])
        host.append((fr["depth"].cpu().numpy(), fr["rgb"].cpu().numpy(), fr["features"].cpu().numpy(), dm.cpu().numpy(),
                     fm.cpu().numpy(), fr["T_W_C"].numpy(), fr["K"].numpy()))

    def run(nthreads, n):
        O.set_num_threads(nthreads)
        orc = O.OracleMapper(O.default_params(
            voxel_size=mcfg.voxel_size_m, max_integration_distance_m=mcfg.projective_integrator_max_integration_distance_m,
            raycast_subsampling=1, workspace_bounds_type=2, ws_min=mcfg.aabb_min_m.tolist(), ws_max=mcfg.aabb_max_m.tolist(),
            tsdf_decay_factor=mcfg.tsdf_decay_factor,
            appearance_measurement_weight=mcfg.projective_appearance_integrator_measurement_weight, feature_channels=channels))
        t0 = time.perf_counter()
        for depth, rgb, feat, dm, fm, T, K in host[:n]:
            orc.decay()
            orc.add_depth_frame(depth, T, K, dm)
            orc.add_color_frame(rgb, T, K, dm)
            orc.add_feature_frame(feat, T, K, fm)
        return n / (time.perf_counter() - t0)

    settings = thread_settings()
    sweep = {}
    t_start = time.perf_counter()
    for nt in settings:
        if time.perf_counter() - t_start > budget_s and sweep:
            break
        sweep[nt] = run(nt, len(host))
    best = max(sweep, key=sweep.get)
    return {
        "value": sweep[best],
        "unit": "frames/s",
        "cores": best,
        "kind": "port",
        "sample": f"first {len(host)} frames of the same stream on a fresh map, CPU oracle (C + OpenMP), masks precomputed; "
                  f"best of the thread sweep",
        "thread_sweep_frames_per_s": {str(k): v for k, v in sweep.items()},
        "host_threads": os.cpu_count(),
        "cpu_quota": cpu_quota(),
    }

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
    # whole-frame roofline at this shape, same byte model as the headline (frame_byte_model): here the feature rows dominate
    n_live = int(mapper.tsdf_layer_view(MAPPER_TO_ID.STATIC).num_allocated_blocks())
    model = frame_byte_model(cfg, C, n_live, st["tsdf_blocks_updated"] / max(st["depth_frames"], 1), cb, vox)
    b_frame = sum(model.values())
    out["whole_frame"] = {"algorithmic_bytes_per_frame": b_frame, "per_launch_bytes": model, "achieved_GBps": b_frame / (dt / steps) / 1e9,
                          "frac_of_hbm_peak": b_frame / (dt / steps) / HBM_PEAK_BYTES_PER_S, "tsdf_live_blocks": n_live}
    if n:
        out["k_feature_flat_us"] = ms / n * 1e3
        out["k_feature_flat_algorithmic_bytes"] = nbytes
        out["k_feature_flat_algorithmic_GBps"] = nbytes / (ms / n * 1e-3) / 1e9
        out["k_feature_flat_frac_of_hbm_peak"] = nbytes / (ms / n * 1e-3) / HBM_PEAK_BYTES_PER_S
    if gate_n:
        out["k_app_frame_gating_us"] = gate_ms / gate_n * 1e3
    # the same stream software-pipelined (mmf_set_deferred_feature_rows: the 313 MB row stream of frame N beside the sphere trace of
    # frame N + 1, its gating beside the raycast); flushed inside the timed region
    mapper.set_deferred_feature_rows(True)
    for i in range(warmup):
        step(mapper, mcfg, frames[i % n_frames])
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(steps):
        step(mapper, mcfg, frames[(warmup + i) % n_frames])
    mapper.flush()
    torch.cuda.synchronize(device)
    dtp = time.perf_counter() - t0
    mapper.set_deferred_feature_rows(False)
    out["pipelined"] = {"frames_per_s": steps / dtp, "ms_per_step": dtp / steps * 1e3,
                        "frac_of_hbm_peak": b_frame / (dtp / steps) / HBM_PEAK_BYTES_PER_S}

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
        mapper.flush()  # (a deferred tail, when the mode is on)
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
    mapper.clear()
    mapper.set_deferred_feature_rows(True)  # the same two pipelines with consecutive frames software-pipelined
    dt_low_p = timed(fused_lowres)
    mapper.clear()
    dt_up_p = timed(with_upsample)
    mapper.set_deferred_feature_rows(False)
    out["from_backbone_output_pipelined"] = {"fused_lowres_frames_per_s": 1.0 / dt_low_p, "fused_lowres_ms": dt_low_p * 1e3,
                                             "upsample_then_integrate_frames_per_s": 1.0 / dt_up_p, "upsample_then_integrate_ms": dt_up_p * 1e3}
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

        saved = torch.get_num_threads()
        for name, (depth, K, T) in host.items():
            B = depth.shape[0]
            sweep = {}
            for nt in thread_settings():
                torch.set_num_threads(nt)
                backproject_torch_cpu(depth, K, T)
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    backproject_torch_cpu(depth, K, T)
                sweep[nt] = B / ((time.perf_counter() - t0) / reps)
            best = max(sweep, key=sweep.get)
            out[name]["cpu_frames_per_s"] = sweep[best]
            out[name]["cpu_threads"] = best
            out[name]["cpu_thread_sweep_frames_per_s"] = {str(k): v for k, v in sweep.items()}
        torch.set_num_threads(saved)
    return out

def run_two_mappers(device, frames, channels, steps=100, warmup=20):
    """The reference's full nvblox_integrate (nvblox_mapping_helpers.py:79-159): decay of both mappers, then the frame into the
    STATIC mapper (mask = ~dynamic) and into the DYNAMIC one (mask = dynamic).  Timed as ONE native call whose five launches
    carry both frames (mmf_integrate_frame_multi, the helper's default) and as two calls in sequence (MMF_PAIR_MAPPERS=0).
    (Round 1's two-stream overlap of the two chains was a net loss -- 0.150 vs 0.124 ms -- and has been removed.)"""
    import nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers as H

    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")

    class Extractor:  # the backbone is out of scope here: the stream's feature image stands in for its output
        def compute(self, rgb):
            return self.next.unsqueeze(0)

    ex = Extractor()
    out = {}
    dyn = torch.zeros_like(frames[0]["dynamic_mask"])
    dyn[dyn.shape[0] // 4: 3 * dyn.shape[0] // 4, dyn.shape[1] // 3: 2 * dyn.shape[1] // 3] = True  # a sixth of the image is "dynamic"
    saved = H.PAIR_MAPPERS
    try:
        for name, pair in (("one_call", True), ("sequential", False), ("one_call_pipelined", True)):
            H.PAIR_MAPPERS = pair
            mapper = get_nvblox_mapper(mcfg, feature_channels=channels)
            if mapper.num_mappers() < 2:
                return None
            # pipelined: consecutive camera frames software-pipelined on both mappers (mmf_set_deferred_feature_rows)
            mapper.set_deferred_feature_rows(name.endswith("pipelined"))

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
            mapper.flush()
            t_enq = time.perf_counter() - t0
            torch.cuda.synchronize(device)
            out[name + "_ms_per_frame"] = (time.perf_counter() - t0) / steps * 1e3
            out[name + "_host_enqueue_ms_per_frame"] = t_enq / steps * 1e3
            del mapper
    finally:
        H.PAIR_MAPPERS = saved
    out["workload"] = "decay + nvblox_integrate(include_dynamic=True): static and dynamic mapper, a sixth of the image dynamic, 640x480, C=%d" % channels
    return out

def run_frames_in_flight(device, frames, channels, counts=(1, 2, 4, 8), steps=150, warmup=30):
    """N independent replicas of the headline step on ONE GPU -- N Mapper objects, each fed its own stream -- issued as ONE
    native call per round (``mmf_integrate_frame_batch``: the N frames are roles of the same five launches).  Per-frame fusion
    does not shard, but a single frame's five dependent launches leave half the chip idle; replicas (data generation over several
    demos, several environments per GPU: SURVEY 8(e)) can use it.  The headline metric stays the single stream."""
    from nvblox_mindmap_amd.nvblox_torch.mapper import integrate_frames_batch

    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    out = {}
    for n, pipelined in [(c, True) for c in counts] + [(counts[-1], False)]:
        # pipelined: every replica's stream software-pipelined (mmf_set_deferred_feature_rows: launches 4 and 5 of a replica's frame
        # are roles of launches 1 and 3 of the next round); the last count also unpipelined (five launches per round)
        mappers = [get_nvblox_mapper(mcfg, feature_channels=channels) for _ in range(n)]
        for m in mappers:
            m.set_deferred_feature_rows(pipelined)

        def one(i):
            entries = []
            for q, m in enumerate(mappers):
                fr = frames[(i + 13 * q) % len(frames)]
                m.decay()
                entries.append(dict(mapper=m, mapper_id=MAPPER_TO_ID.STATIC, depth_frame=fr["depth"], color_frame=fr["rgb"],
                                    feature_frame=fr["features"], input_mask=fr["dynamic_mask"], t_w_c=fr["T_W_C"], intrinsics=fr["K"],
                                    min_depth_m=mcfg.min_integration_distance_m,
                                    input_mask_erosion_iterations=mcfg.static_mask_erosion_iterations,
                                    valid_depth_mask_erosion_iterations=mcfg.valid_depth_mask_erosion_iterations,
                                    border_percent=mcfg.feature_mask_border_percent, invert_input_mask=True))
            integrate_frames_batch(entries)

        for i in range(warmup):
            one(i)
        torch.cuda.synchronize(device)
        mappers[0].profile_reset()
        mappers[0].profile_enable(True, kernels=list(KERNEL_OF_CLASS), stride=4)
        t0 = time.perf_counter()
        for i in range(steps):
            one(warmup + i)
        for m in mappers:
            m.flush()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t0
        mappers[0].profile_enable(False)
        prof = mappers[0].profile()
        out[str(n) if pipelined else f"{n}_unpipelined"] = {"aggregate_frames_per_s": n * steps / dt, "ms_per_round": dt / steps * 1e3, "host_enqueue_ms_per_round": t_enq / steps * 1e3,
                       "launch_us": {KERNEL_OF_CLASS[c]: (ms / k * 1e3 if k else None) for c, (ms, k) in prof.items() if c in KERNEL_OF_CLASS}}
        del mappers
        torch.cuda.empty_cache()
    out["workload"] = ("N x (decay + fused frame, 640x480, C=%d, DRILL_IN_BOX), one mmf_integrate_frame_batch call per round; every replica's "
                       "stream software-pipelined (launch_us: k_front / k_sphere_alloc then carry the previous round's k_app_frame / "
                       "k_feature_flat), flushed inside the timed region" % channels)
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

def build_facade(shape: str, device, n_frames: int):
    """The object the reference's policy drives (IsaacLabNvbloxMapper) + a loader-shaped sample stream, at the reference's
    shape ("ref": 512x512, 768 channels) or the benchmark shape ("bl": 640x480, 64 channels)."""
    from nvblox_mindmap_amd.mapping.isaaclab_nvblox_mapper import IsaacLabNvbloxMapper
    from scipy.spatial.transform import Rotation

    if shape == "ref":
        C = 768
        cfg = S.StreamConfig(width=512, height=512, fx=586.4, fy=586.4, cx=255.5, cy=255.5, hole_mode="patches")
    else:
        C = 64
        cfg = S.StreamConfig(hole_mode="patches")
    frames = build_stream(cfg, n_frames, C, device)

    class Extractor:  # the DNN is out of scope: hands the stream's pre-computed backbone output (or feature image) over
        next = low = None

        def compute(self, rgb):
            return self.next.unsqueeze(0)

        def compute_lowres(self, rgb):  # the hand-over nvblox_integrate prefers: the 16x16xC map, sampled inside the kernel
            return self.low, (cfg.height, cfg.width)

        def num_excess_features(self):
            return 0

    ex = Extractor()
    facade = IsaacLabNvbloxMapper("rgbd_and_mesh", None, device, feature_extractor=ex, task="DRILL_IN_BOX", feature_channels=C,
                                  num_vertices_to_sample=2048)
    samples = []
    for fr in frames:  # what the loader / simulator hands the policy: [1, ncam, ...] tensors on the device
        T = fr["T_W_C"].numpy().astype(np.float64)
        q = Rotation.from_matrix(T[:3, :3]).as_quat()
        pose7 = torch.tensor(np.concatenate([T[:3, 3], [q[3], q[0], q[1], q[2]]]), dtype=torch.float32, device=device)
        samples.append({"depths": fr["depth"][None, None], "intrinsics": fr["K"].to(device)[None, None], "camera_poses": pose7[None, None],
                        "rgbs": (fr["rgb"].permute(2, 0, 1).float() / 255.0)[None, None].contiguous(),
                        "segmentation_masks": fr["dynamic_mask"][None, None]})
    return cfg, C, frames, samples, ex, facade

def run_model_inputs(device, shape: str, iters=40, n_frames=12):
    """The OUTPUT half of the hot path, alone (SURVEY 8(a) A11 + A12): ``IsaacLabNvbloxMapper.get_nvblox_model_inputs`` =
    mesh extraction + AABB / zero-row filters + sampling to 2048 rows, on a map fused from `n_frames` frames, and the facade's
    per-frame fusion call beside it.  Two native launches (k_mesh_keep, k_model_inputs_gather) and one synchronisation.
    Algorithmic bytes per call: every live TSDF block read once (8 B/voxel) + one 128 B line of each in-box vertex's feature row
    (the zero test stops at the first non-zero piece) + the kept-vertex list written and the sampled entries read (16 B) +
    per sampled row 2C read, 12 + 4C written."""
    import gc

    cfg, C, frames, samples, ex, facade = build_facade(shape, device, n_frames)

    def fuse(i):
        fr, smp = frames[i % n_frames], samples[i % n_frames]
        ex.next, ex.low = fr["features"], fr["lowres"]
        facade.decay()
        facade.update_reconstruction_from_sample(smp, "pov")

    for i in range(n_frames):
        fuse(i)
    torch.cuda.synchronize(device)
    gc.collect()
    gc.freeze()  # (a full collection of the interpreter's heap costs tens of ms with torch + scipy loaded)
    per = []
    for i in range(iters):
        t0 = time.perf_counter()
        fuse(i)
        torch.cuda.synchronize(device)
        per.append((time.perf_counter() - t0) * 1e3)
    fusion_ms = statistics.median(per)
    m = facade.mapper
    torch.manual_seed(0)
    facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
    m.profile_reset()
    m.profile_enable(True, kernels=["mesh"])
    per = []
    for i in range(iters):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        out = facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
        torch.cuda.synchronize(device)
        per.append((time.perf_counter() - t0) * 1e3)
    m.profile_enable(False)
    ms, n = m.profile()["mesh"]
    gc.unfreeze()
    kernels_us = ms / max(n // 2, 1) * 1e3  # two bracketed launches per call
    n_live = int(m.tsdf_layer_view(MAPPER_TO_ID.STATIC).num_allocated_blocks())
    V = int(m.update_feature_mesh(MAPPER_TO_ID.STATIC))
    kept = int(m.model_inputs_prepare(MAPPER_TO_ID.STATIC, facade.mapping_config.aabb_min_host, facade.mapping_config.aabb_max_host, C, True))
    N = int(out["vertices"].shape[1])
    alg = n_live * 512 * 8 + V * 128 + kept * 16 + N * (16 + 2 * C + 12 + 4 * C)
    res = {"shape": f"{cfg.height}x{cfg.width}x{C}", "ms_per_call": statistics.median(per), "facade_fusion_ms_per_frame": fusion_ms,
           "kernels_us_per_call": kernels_us, "launches_per_call": 2, "mesh_vertices": V, "kept_rows": kept, "sampled_rows": N,
           "live_tsdf_blocks": n_live, "algorithmic_bytes": alg, "achieved_GBps": alg / (kernels_us * 1e-6) / 1e9 if kernels_us else None,
           "frac": alg / (kernels_us * 1e-6) / HBM_PEAK_BYTES_PER_S if kernels_us else None,
           "bound": "latency (one pass over the live blocks' lattices + a host round trip for the RNG draw)",
           "through": "IsaacLabNvbloxMapper.get_nvblox_model_inputs(STATIC, remove_zero_features=True)"}
    del facade, frames, samples
    torch.cuda.empty_cache()
    return res

def get_unbounded_mapper(mcfg, channels):
    """get_nvblox_mapper (nvblox_mapping_helpers.py:30-76) with nvblox's default view-calculator setting instead of the task's
    bounding box: ``workspace_bounds_type = kUnbounded`` -- the block index is then the open-addressing HASH."""
    from nvblox_mindmap_amd.nvblox_torch.mapper import Mapper
    from nvblox_mindmap_amd.nvblox_torch.mapper_params import (
        BlockMemoryPoolParams, MapperParams, ProjectiveIntegratorParams, TsdfDecayIntegratorParams, ViewCalculatorParams)
    from nvblox_mindmap_amd.nvblox_torch.projective_integrator_types import ProjectiveIntegratorType

    pi = ProjectiveIntegratorParams()
    pi.projective_integrator_max_integration_distance_m = mcfg.projective_integrator_max_integration_distance_m
    pi.projective_appearance_integrator_measurement_weight = mcfg.projective_appearance_integrator_measurement_weight
    de = TsdfDecayIntegratorParams()
    de.tsdf_decay_factor = mcfg.tsdf_decay_factor
    vc = ViewCalculatorParams()
    vc.raycast_subsampling_factor = 1
    vc.workspace_bounds_type = "kUnbounded"
    pool = BlockMemoryPoolParams()
    pool.expansion_factor = 1.0
    # the library's pools do not grow: 262 144 blocks per layer (TSDF 1 GB, 64-channel features 17 GB) hold what the orbit sees
    # out to 5 m at 1 cm voxels; the run reports the live count and fails loudly on exhaustion
    pool.num_preallocated_blocks = 262144
    mp = MapperParams()
    mp.set_projective_integrator_params(pi)
    mp.set_tsdf_decay_integrator_params(de)
    mp.set_view_calculator_params(vc)
    mp.set_block_memory_pool_params(pool)
    return Mapper(voxel_sizes_m=[mcfg.voxel_size_m], integrator_types=[ProjectiveIntegratorType.TSDF], mapper_parameters=mp,
                  feature_channels=channels)

UNBOUNDED_CLASSES = {"decay": "k_live_compact_big (a launch of its own only with MMF_NO_BIG_MERGE=1; else a role of the first launch)",
                     "raycast": "k_front_compact_big (the light decay's list compaction: deallocation, tombstones | raycast | mask rows)",
                     "alloc": "k_alloc_big (TSDF: hash lookups + CAS insertion | mask columns)",
                     "tsdf": "k_tsdf_classify + k_tsdf_pass<lazy> (the frame's blocks: missed decays, integration; appearance-candidate flags)",
                     "sphere": "k_sphere_alloc_big (colour | feature allocation | sphere trace)",
                     "feature": "k_app_frame (colour update + feature gating)", "feature_flat": "k_feature_flat"}

def run_unbounded(device, frames, channels, steps=100, warmup=30):
    """The headline step (decay + fused depth / colour / feature frame, same 640x480 stream, same masks) in an UNBOUNDED workspace:
    the block index is the open-addressing hash (CAS insertion by the allocation launch, tombstones from the decay's
    deallocations, amortised rebuild) instead of the dense table of the task's bounding box, the view grid is the whole
    frustum out to the 5 m integration distance, allocation and TSDF pass are separate launches.  North star: "voxel-block hash
    allocation ... wavefront ballot/prefix-sum for hash insertion"."""
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    mapper = get_unbounded_mapper(mcfg, channels)
    n_frames = len(frames)

    def one(i):
        step(mapper, mcfg, frames[i % n_frames])

    for i in range(warmup):
        one(i)
    torch.cuda.synchronize(device)
    mapper.reset_stats(0)
    mapper.profile_reset()
    mapper.profile_enable(True, kernels=list(UNBOUNDED_CLASSES), stride=4)
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    mapper.profile_enable(False)
    prof = mapper.profile()
    stats = mapper.stats(0)
    hs = mapper.hash_state(0)
    hs_cells = hs["view_grid"][0] * hs["view_grid"][1] * hs["view_grid"][2]
    hs["view_grid_cells"] = hs_cells
    n_live = hs["live_blocks"]
    if n_live >= 262144:
        raise RuntimeError("unbounded leg: block pool exhausted")
    nf = max(stats["feature_frames"], 1)
    cfg = S.StreamConfig(hole_mode="patches")
    n_upd = stats["tsdf_blocks_updated"] / max(stats["depth_frames"], 1)
    n_new = stats["tsdf_blocks_allocated"] / max(stats["depth_frames"], 1)
    model = frame_byte_model(cfg, channels, n_live, n_upd, stats["color_blocks_updated"] / max(stats["color_frames"], 1),
                             stats["feature_voxels_updated"] / nf)
    # hash traffic of the allocation launch: one 16 B probe per candidate block (+ one CAS + value store per new block)
    ncand = stats["color_blocks_updated"] / max(stats["color_frames"], 1)
    base = model
    model = {
        # live entry + wmax / wmin read and written (the lazy decay: one multiplication per live BLOCK) + slot key; erase / free push per dead block
        UNBOUNDED_CLASSES["decay"]: n_live * (4 + 8 + 8 + 1 + 8) + n_new * 40,
        UNBOUNDED_CLASSES["raycast"]: base["k_front"],
        # view-grid flags read + cleared, one 16 B probe per candidate (twice: count, assign), 13 B of candidate list, CAS + value per new block
        UNBOUNDED_CLASSES["alloc"]: hs_cells * 2 + (n_upd + 2 * ncand) * (2 * 16 + 13) + n_new * 24 + 2 * n_live,
        # lazy decay: the pass reads and writes the blocks the frame integrates (a near-surface block it only
        # looks at -- appearance flag -- is read; not counted: their number is not in the statistics); 10 B of list / stamp / band
        # words per live block for the classification
        UNBOUNDED_CLASSES["tsdf"]: n_upd * 512 * 16 + cfg.height * cfg.width * 4 + 10 * n_live + 36 * n_upd,
        "k_sphere_trace": base["k_sphere_alloc"],
        UNBOUNDED_CLASSES["feature"]: base["k_app_frame"],
        "k_feature_flat": base["k_feature_flat"],
    }
    per = []
    for cls, name in UNBOUNDED_CLASSES.items():
        ms, n = prof.get(cls, (0.0, 0))
        us = ms / n * 1e3 if n else None
        b = model.get(name, 0.0)
        calls = 1  # bracketed launches of the class per frame
        per.append({"kernel": name, "avg_us_per_frame": us * calls if us else None, "launches_timed": n, "algorithmic_bytes": b,
                    "frac": (b / (us * calls * 1e-6) / HBM_PEAK_BYTES_PER_S) if (us and b) else None})
    b_frame = sum(model.values())
    out = {"frames_per_s": 1.0 / dt, "ms_per_step": dt * 1e3, "hash": hs,
           "tsdf_blocks_integrated_per_frame": n_upd, "tsdf_blocks_allocated_per_frame": n_new,
           "tsdf_blocks_deallocated_per_frame": n_new,  # steady state on the orbit: as many leave as arrive
           "algorithmic_bytes_per_frame": b_frame, "frac": b_frame / dt / HBM_PEAK_BYTES_PER_S, "per_kernel": per,
           # k_front_compact_big, k_alloc_big, k_tsdf_classify, k_tsdf_pass, k_sphere_alloc_big, k_app_frame, k_feature_flat + the conditional
           # rebuild pair behind every 16th compaction (round 4: 11 launches -- the compaction, the pair and the appearance allocation apart)
           "launches_per_frame": 7 + 2.0 / 16.0,
           "workload": "decay + integrate_frame (depth, colour, %d-ch features), 640x480, 1 cm voxels, workspace_bounds_type=kUnbounded, "
                       "max integration distance 5 m" % channels}
    # the same stream software-pipelined (mmf_set_deferred_feature_rows: the scalable launches host the previous frame's gating and rows
    # since round 5), untimed per launch, flushed inside the region
    mapper.set_deferred_feature_rows(True)
    for i in range(8):
        one(warmup + steps + i)
    mapper.flush()
    torch.cuda.synchronize(device)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        for i in range(steps):
            one(warmup + steps + 8 + rep * steps + i)
        mapper.flush()
        torch.cuda.synchronize(device)
        dtp = (time.perf_counter() - t0) / steps
        best = dtp if best is None else min(best, dtp)
    out["pipelined"] = {"frames_per_s": 1.0 / best, "ms_per_step": best * 1e3, "frac": b_frame / best / HBM_PEAK_BYTES_PER_S,
                        "launches_per_frame": 5 + 2.0 / 16.0}
    del mapper
    torch.cuda.empty_cache()
    return out

def run_pixel_holes(device, channels, steps=100, warmup=20, n_frames=40):
    """SURVEY.md section 8(d)'s stream AS PRESCRIBED: invalid depth at the 1 %-density single pixels (u * 73856093 ^ v * 19349663) % 97 == 0.
    With the reference's 20-pixel valid-depth erosion those holes erase the whole feature mask (a pixel survives iff its 41 x 41
    window holds no hole: (1 - 1/97)^1681 = 3e-8), so this stream has TSDF and colour work and NO feature work -- which is why the
    headline uses 16 x 16 hole patches instead (config.workload).  Same call sequence, same pipelining, own mapper; two regions of
    `steps` frames, the faster one reported, with the algorithmic bytes of what the frames actually did."""
    cfg = S.StreamConfig(hole_mode="pixels")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    frames = build_stream(cfg, n_frames, channels, device)
    mapper = get_nvblox_mapper(mcfg, feature_channels=channels)
    mapper.set_deferred_feature_rows(True)
    for i in range(warmup):
        step(mapper, mcfg, frames[i % n_frames])
    mapper.flush()
    torch.cuda.synchronize(device)
    mapper.reset_stats(MAPPER_TO_ID.STATIC)
    regions, k = [], warmup
    for _ in range(2):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            step(mapper, mcfg, frames[(k + i) % n_frames])
        mapper.flush()
        torch.cuda.synchronize(device)
        regions.append(time.perf_counter() - t0)
        k += steps
    st = mapper.stats(MAPPER_TO_ID.STATIC)
    n_live = int(mapper.tsdf_layer_view(MAPPER_TO_ID.STATIC).num_allocated_blocks())
    nf = max(st["depth_frames"], 1)
    surv = st["feature_voxels_updated"] / max(st["feature_frames"], 1)
    model = frame_byte_model(cfg, channels, n_live, st["tsdf_blocks_updated"] / nf, st["color_blocks_updated"] / max(st["color_frames"], 1), surv)
    t = min(regions) / steps
    out = {"frames_per_s": 1.0 / t, "ms_per_step": t * 1e3, "steps": steps, "hole_mode": "pixels", "feature_voxels_updated_per_frame": surv,
           "tsdf_blocks_per_frame": st["tsdf_blocks_updated"] / nf, "algorithmic_bytes_per_frame": sum(model.values()),
           "frac_of_hbm_peak": sum(model.values()) / t / HBM_PEAK_BYTES_PER_S,
           "note": "SURVEY 8(d)'s prescribed holes: the 20-pixel erosion leaves no feature pixel, the frame is TSDF + colour work"}
    del mapper, frames
    torch.cuda.empty_cache()
    return out
