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

Output (rank 0): the FULL record (every leg, tens of KB) is written to `bench_full.json` (repo root, and `gpurun_out/` when
present) and printed on stderr; the LAST STDOUT LINE is the compact record built by `bench_record.compact_line` -- <= 4 KB, always:
the driver's contract keys + `roofline` (whole pipelined frame = 3 launches against the 8 TB/s HBM peak: algorithmic bytes of
`frame_byte_model()` / measured frame time, counter traffic, the dominant launch with its own time / bytes / fraction, the other
shapes as `legs`) + `cpu_baseline` (the C + OpenMP oracle, kind "port", on a bounded sample of the same frames; the reference's
torch-CPU back-projection beside the HIP kernel) + `train` (captured policy training step; steady-state file-fed ratio).  Per-launch
durations are HIP-event stamps on the launch stream, taken in a region of their own behind the headline regions.
"""
import argparse
import json
import os
import sys
import time

import statistics
import subprocess

# The captured training step (training/graphed.py) is one HIP graph of ~1 300 kernel nodes (2 400 before the trainable-side kernels): with the runtime's default AQL ring
# (16 384 packets) hipGraphLaunch blocks the host until the previous step has drained enough of it (measured: 26 of 73 ms per step;
# 0.7 ms with the larger ring).  Read by the HIP runtime when it initialises, i.e. at the first GPU call of this process.
os.environ.setdefault("ROC_AQL_QUEUE_SIZE", "65536")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def spawn_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks ourselves, one process per GPU, the way the reference
    starts training (torchrun --standalone --nnodes 1 --nproc_per_node N, mindmap_osmo/tasks/training_task.py:38).  Fresh child
    processes (this one has not touched the GPU and never does); rank 0's JSON line goes to our stdout."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def _gpus_from_argv(argv) -> int:
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _gpus_from_argv(sys.argv[1:]) > 1:
    # before anything that could initialise the GPU in this process (the children are fresh processes, not a re-exec)
    sys.exit(spawn_ranks(_gpus_from_argv(sys.argv[1:]), sys.argv[1:]))

if os.environ.get("BENCH_HANG_DUMP_S"):
    # diagnostics: every thread's Python stack on stderr after this many seconds, then exit (a multi-rank hang otherwise says nothing)
    import faulthandler

    faulthandler.dump_traceback_later(float(os.environ["BENCH_HANG_DUMP_S"]), exit=True)

import bench_record  # noqa: E402
from nvblox_mindmap_amd import synthetic as S  # noqa: E402
from nvblox_mindmap_amd.image_processing.feature_resize import upsample_features  # noqa: E402
from nvblox_mindmap_amd.mapping.helpers.nvblox_mapping_helpers import get_nvblox_mapper, integrate_frame  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID, NvbloxMappingCfg  # noqa: E402

HBM_PEAK_BYTES_PER_S = 8.0e12  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3e12 achievable
CPU_THREAD_SWEEP = (8, 16, 32, 64)  # + all host threads; the best setting is the reported CPU baseline


def emit(full: dict) -> None:
    """full record -> bench_full.json (+ gpurun_out/) and stderr; compact record (<= 4 KB) = the last stdout line."""
    text = json.dumps(full)
    written = None
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, "bench_full.json"), "w") as f:
                    f.write(text + "\n")
                written = written or os.path.relpath(os.path.join(d, "bench_full.json"), ROOT)
            except OSError:
                pass
    full["full_record"] = written
    print(text, file=sys.stderr, flush=True)
    print(bench_record.compact_line(full), flush=True)


def dry_run(args, world: int, rank: int) -> None:
    """Control flow of a multi-rank run without a GPU (tests/test_cpu_bench.py: `--gpus 2 --dry-run` over gloo): rendezvous,
    warm-up, barrier-bracketed regions, max over ranks, ONE JSON line from rank 0.  The "step" is a sleep."""
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend=os.environ.get("BENCH_DIST_BACKEND", "gloo"), init_method="env://")
    for _ in range(args.warmup):
        time.sleep(1e-4)
    regions = []
    for _ in range(args.repeats):
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(1e-4)
        if dist is not None:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions.append(float(t.item()))
    elapsed = statistics.median(regions)
    if rank == 0:
        # the same shape as a real run's last line (bench_record.compact_line), from a record with the legs absent
        print(bench_record.compact_line({
            "metric": "RGB-D+feature frames/s fused @1 cm voxels", "value": world * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
            "config": {"workload": "dry run: no GPU work, control flow only"}, "dry_run": True,
            "roofline": {"bound": "hbm", "kernel": "none (dry run)", "achieved": None, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                         "frac": None, "traffic": None},
            "cpu_baseline": {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "none (dry run)"},
            "train": {"parallelism": f"dp{world}" if world > 1 else "single"}}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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


def thread_settings():
    ncpu = os.cpu_count() or 1
    return sorted({t for t in CPU_THREAD_SWEEP if t < ncpu} | {ncpu})


def cpu_quota():
    """CPUs this process may actually use: the cgroup's CPU bandwidth quota (a container on a 256-thread host is often capped
    well below os.cpu_count()), else the affinity mask.  Every CPU-side figure of the line (cpu_baseline, the loader) is
    bounded by it."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    try:
        return float(len(os.sched_getaffinity(0)))
    except AttributeError:
        return float(os.cpu_count() or 1)


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


def run_closed_loop(device, steps=8):
    """BASELINE configs[3]: one control step of the closed loop on one GPU, end to end, through the object the reference's policy
    drives (mapping/isaaclab_nvblox_mapper.py; closed_loop/policies/nvblox_diffuser_actor_policy.py:77-83,206-211):
    mapper.decay() + update_reconstruction_from_sample (input helpers: pose 7-vector -> 4x4, rgb float -> u8, back-projection;
    then the fused RGB-D/feature frame, 512x512, 768 feature channels) -> get_nvblox_model_inputs (surface vertices + features
    sampled to 2048) -> policy inference (encoder + 100 denoising steps, fused ops + HIP graph).  The image backbone runs once,
    inside the policy encoder (random-init ViT-B/16); the mapper's extractor hands the stream's pre-computed backbone output
    (16x16x768) over, which the fused frame samples itself (mapping/helpers/nvblox_mapping_helpers.py: compute_lowres)."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActor, DiffuserActorConfig
    from nvblox_mindmap_amd.image_processing.backprojection import get_camera_pointcloud
    from nvblox_mindmap_amd.mapping.isaaclab_nvblox_mapper import IsaacLabNvbloxMapper
    from nvblox_mindmap_amd.training import build_model, synthetic_batch

    cfg, C, frames, samples, ex, facade = build_facade("ref", device, 4)
    pcfg = DiffuserActorConfig()
    torch.manual_seed(0)
    model = build_model(pcfg, device=device).eval()
    DiffuserActor.enable_fused_inference(True)
    model.enable_graph_sampling(True)
    hist = synthetic_batch(pcfg, 1, device, seed=3)["gripper_history"]
    parts = {"fusion": 0.0, "map_to_model_input": 0.0, "policy_inference": 0.0}

    def control_step(i, record):
        fr, smp = frames[i % 4], samples[i % 4]
        ex.next, ex.low = fr["features"], fr["lowres"]
        t = [time.perf_counter()]
        facade.decay()
        facade.update_reconstruction_from_sample(smp, "pov")
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        inp = facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
        pcd = get_camera_pointcloud(smp["intrinsics"][0], smp["depths"][0], smp["camera_poses"][0, :, :3], smp["camera_poses"][0, :, 3:])
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        with torch.no_grad():
            traj = model(None, None, smp["rgbs"], pcd[:, None], (smp["depths"] > 0), inp["vertex_features"], inp["vertices"],
                         inp["vertices_valid_mask"], None, hist, run_inference=True)[0]
        torch.cuda.synchronize(device)
        t.append(time.perf_counter())
        if record:
            for name, a, b in zip(parts, t[:-1], t[1:]):
                parts[name] += (b - a) * 1e3
        return traj

    try:
        for i in range(3):
            control_step(i, False)  # warm-up: fills the map, captures the graph
        import gc

        gc.collect()
        gc.freeze()  # (a generation-2 collection inside a 0.3 ms phase of a handful of steps is the whole phase)
        t0 = time.perf_counter()
        for i in range(steps):
            control_step(3 + i, True)
        total = (time.perf_counter() - t0) / steps * 1e3
    finally:
        DiffuserActor.enable_fused_inference(False)
    out = {"ms_per_control_step": total, "control_steps_per_s": 1e3 / total, "breakdown_ms": {k: v / steps for k, v in parts.items()},
           "shape": "512x512 RGB-D, 768 feature channels, 2048 sampled vertices, 100 denoising steps, batch 1",
           "through": "IsaacLabNvbloxMapper.update_reconstruction_from_sample / get_nvblox_model_inputs"}
    del facade, model, frames, samples
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


def run_training(device, world, steps=8, warmup=3, per_gpu_batch=32, backbone_matmul_dtype="float16x3", prefetch_backbone=False):
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
           "backbone_matmuls": backbone_matmul_dtype,
           "allreduce_payload_MB": n_train * 4 / 1e6, "parallelism": f"dp{world}" if world > 1 else "single",
           "model": "diffuser_actor RGBD_AND_MESH, 1 cam 512x512, 2048 vertices x 768, frozen ViT-B/16-shaped backbone (random init)"}
    del model, ddp, opt, batches
    torch.cuda.empty_cache()
    return out


def run_training_graphed(device, world, steps=20, warmup=5, per_gpu_batch=32, backbone_matmul_dtype="float16x3", overlap_backbone=True):
    """The same training step as run_training (same model, batch, dtype, optimizer rule), arranged so that the GPU -- not the
    interpreter -- bounds it (training.GraphedTrainStep): forward + backward as ONE captured HIP graph with the next batch's frozen
    backbone as a parallel branch, gradients in ONE flat buffer, ONE explicit RCCL all-reduce of it between the graphs (world > 1),
    AdamW over the flat segments as a second graph.  Every timed step runs one backbone forward, one trainable forward / backward,
    one all-reduce and one optimizer step.  Reports what the host and the collective cost: host time to enqueue a step, the
    all-reduce's duration (HIP events on the stream it is issued from, a second short region), the number of ranks an all-reduce of
    ones reaches, and every rank's own step time."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import GraphedTrainStep, all_gather_objects, build_model, synthetic_batch
    from nvblox_mindmap_amd.training.distributed import barrier, max_over_ranks

    torch.manual_seed(0)
    cfg = DiffuserActorConfig(backbone_matmul_dtype=backbone_matmul_dtype)
    model = build_model(cfg, device=device)
    rank = int(os.environ.get("RANK", "0"))
    batches = [synthetic_batch(cfg, per_gpu_batch, device, seed=1000 * rank + i) for i in range(2)]
    t_c = time.perf_counter()
    g = GraphedTrainStep(cfg, model, batches[0], overlap_backbone=overlap_backbone)
    torch.cuda.synchronize(device)
    capture_s = time.perf_counter() - t_c
    observed = g.observed_world()

    def run(n, first):
        for i in range(first, first + n):
            g.step(batches[i % 2], batches[(i + 1) % 2])

    run(warmup, 0)
    barrier()
    torch.cuda.synchronize(device)
    g.host_enqueue_s = g.host_cpu_s = 0.0
    t0 = time.perf_counter()
    run(steps, warmup)
    host_s, host_cpu_s = g.host_enqueue_s, g.host_cpu_s
    torch.cuda.synchronize(device)
    mine = time.perf_counter() - t0
    barrier()
    el = time.perf_counter() - t0
    nccl = world > 1 and torch.distributed.get_backend() == "nccl"
    dt = max_over_ranks(el, device if nccl else None) if world > 1 else el
    per_rank = [r["ms"] for r in all_gather_objects({"ms": mine / steps * 1e3})]
    allreduce = None
    if world > 1:  # a second, short region with HIP events around the collective (kept out of the headline region)
        g.time_allreduce = True
        run(6, warmup + steps)
        ms = g.collect_allreduce_ms()
        g.time_allreduce = False
        allreduce = {"mean_ms": sum(ms) / len(ms), "min_ms": min(ms), "max_ms": max(ms), "payload_MB": g.flat_grad.numel() * 4 / 1e6,
                     "timed_with": "HIP events recorded on the issuing stream around dist.all_reduce(flat_grad) (includes the wait for "
                                   "the slowest rank's backward)"}
    out = {"step_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "samples_per_s": steps * per_gpu_batch * world / dt,
           "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * world, "steps": steps, "warmup": warmup, "dtype": "f32",
           "backbone_matmuls": backbone_matmul_dtype, "backbone_overlap": bool(g.overlap),
           "host_enqueue_ms_per_step": host_s / steps * 1e3, "host_enqueue_frac": host_s / mine,
           "host_cpu_ms_per_step": host_cpu_s / steps * 1e3,
           "rccl_world_observed": observed, "allreduce": allreduce, "allreduce_payload_MB": g.flat_grad.numel() * 4 / 1e6,
           "per_rank_ms_per_step": {"min": min(per_rank), "max": max(per_rank), "all": per_rank},
           "trainable_params_in_flat_buffer": int(g.n_total), "unused_parameter_tensors": len(g.unused_names),
           "capture_s": capture_s, "tuned_gemms": bool(g.tuned_gemms), "runtime_env": {"ROC_AQL_QUEUE_SIZE": os.environ.get("ROC_AQL_QUEUE_SIZE")}, "parallelism": f"dp{world}" if world > 1 else "single",
           "how": "training.GraphedTrainStep: forward+backward = one captured HIP graph (next batch's frozen backbone as a parallel "
                  "branch), one flat gradient buffer, one explicit all-reduce, AdamW over two flat segments as a second graph",
           "trainable_side_kernels": "libmmfusion, float32: attention forward + backward on the f32 matrix cores (8 heads x 15 channels, read "
                                     "from the projections), rotary / LayerNorm(a + b) / AdaLN forward + backward, Linear dW + db by a row-split "
                                     "matrix-core kernel (deterministic sums); MMF_TRAIN_ATTENTION=0 MMF_TRAIN_LAYERNORM=0 MMF_FUSED_ROTARY=0 "
                                     "give torch's operators back"}
    del g, model, batches
    torch.cuda.empty_cache()
    return out


def run_training_file_fed(device, compute_bound_step_per_s, per_gpu_batch=32, n_frames=32, steps=48, threads=3, slots=4,
                          vertex_count_range=(10000, 14000), reference_loader=True):
    """Is the training step loader-bound IN STEADY STATE?  (the risk to ">= 0.9x linear over 8 GPUs" is the loader
    keeping the GPUs fed, not the 8.6 MB all-reduce.)  A demo in the reference's on-disk layout -- 512x512 rgb + u16 depth PNGs,
    pose / intrinsics .npy, one UNSAMPLED vertex-feature .zst per frame (10-14 k vertices x 768 f16 channels, ~20 MB: what
    save_feature_mesh_to_disk writes) with the raw copies of io/vertex_cache.py beside them -- feeds the captured step
    (training.GraphedTrainStep) at per-GPU batch 32 through data_loading.PinnedBatchLoader: `threads` loader threads write every
    sample straight into its row of one of `slots` pinned batch buffers (one copy per byte), DevicePrefetcher copies a batch
    ahead.  Steady state: the timed region starts after more batches than the pipeline can hold have been consumed and spans
    >= 3x its capacity ((slots + 1) x batch samples), so a queue filled during graph capture cannot carry it (round 4's 0.99
    was that artefact).  CPU the loader BURNS = the worker threads' own CPU clocks (thread_time: user + kernel, page faults
    included) per sample.  Beside it: eight such loaders at once in eight processes (what an 8-GPU node asks of its host, under
    this box's CPU quota), and the reference-shaped torch DataLoader (worker processes, default_collate, pin thread) on the
    same files, loader only."""
    import shutil
    import tempfile

    from nvblox_mindmap_amd.data_loading.dataset import DevicePrefetcher, MindmapFrameDataset, write_synthetic_demo
    from nvblox_mindmap_amd.data_loading.pinned_loader import PinnedBatchLoader
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.io import vertex_cache
    from nvblox_mindmap_amd.training import GraphedTrainStep, build_model

    cfg = DiffuserActorConfig()
    ncpu = os.cpu_count() or 1
    # the step's host side is graph launches: a handful of intra-op threads is plenty, and the default (one per hardware thread:
    # 128-256 on the GPU box) spins the container's 16-CPU quota away from the loader
    host_threads_before = torch.get_num_threads()
    torch.set_num_threads(2)
    root = tempfile.mkdtemp(prefix="mmf_file_fed_")
    out = {}
    try:
        t0 = time.perf_counter()
        write_synthetic_demo(os.path.join(root, "demo_00000"), n_frames, image_size=cfg.image_size, feature_dim=cfg.feature_dim,
                             num_history=cfg.num_history, prediction_horizon=cfg.prediction_horizon, ngrippers=cfg.ngrippers,
                             vertex_count_range=vertex_count_range)
        t_write = time.perf_counter() - t0
        t0 = time.perf_counter()
        n_raw = vertex_cache.convert_dataset(root)
        t_convert = time.perf_counter() - t0

        ds = MindmapFrameDataset(root, num_vertices=2048, use_raw_vertex_cache=True)
        mb = sum(os.path.getsize(p) for smp in ds.samples for p in smp.values()) / len(ds) / 1e6
        # the frames on disk are revisited (page-cache reads; a real dataset adds storage latency on top of what is measured here)
        ds.samples = ds.samples * max(1, -(-(steps + 4 * slots + 8) * per_gpu_batch // len(ds.samples)))
        loader = PinnedBatchLoader(ds, per_gpu_batch, shuffle=True, drop_last=True, threads=threads, slots=slots)

        # loader only: batches handed out as fast as the threads fill them
        n = 0
        for i, b in enumerate(loader):
            if i >= 2 * slots:
                break
        loader.reset_stats()
        t0 = time.perf_counter()
        for i, b in enumerate(loader):
            n += b["rgb_u8"].shape[0]
            if i >= 24:
                break
        loader_sps = n / (time.perf_counter() - t0)
        loader_only_cpu_ms = loader.stats()["cpu_ms_per_sample"]

        torch.manual_seed(0)
        model = build_model(cfg, device=device)

        def batches():  # device batches, copies + GPU-side transforms one step ahead on a side stream
            while True:
                for b in DevicePrefetcher(loader, device):
                    yield b

        it = batches()
        cur = next(it)
        g = GraphedTrainStep(cfg, model, cur, data_parallel=False)  # a rank-0-only leg: no collective, the other ranks are not here

        def fed_steps(k, cur):
            for _ in range(k):
                nxt = next(it)
                g.step(cur, nxt)
                cur = nxt
            return cur

        capacity = (slots + 1) * per_gpu_batch  # samples the pipeline can hold: the slots + the batch already on the device
        # untimed: MORE batches than the pipeline holds (filled while the graphs were captured) are consumed first
        cur = fed_steps(2 * (slots + 1) + 2, cur)
        torch.cuda.synchronize(device)
        loader.reset_stats()
        c0, t0 = time.process_time(), time.perf_counter()
        cur = fed_steps(steps, cur)
        torch.cuda.synchronize(device)
        wall = time.perf_counter() - t0
        process_cpu = time.process_time() - c0  # every thread of this process: loader threads + the step's host side
        st = loader.stats()
        fed = steps / wall
        # the comparator under the SAME conditions: this process, this model, the loader's threads alive but idle, two batches
        # resident on the device (the training leg's figure comes from another model instance: +-3 % between runs)
        pair = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in cur.items()}, {k: (v.clone() if torch.is_tensor(v) else v) for k, v in next(it).items()}]
        for i in range(3):
            g.step(pair[i % 2], pair[(i + 1) % 2])
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(3, 3 + steps):
            g.step(pair[i % 2], pair[(i + 1) % 2])
        torch.cuda.synchronize(device)
        resident = steps / (time.perf_counter() - t0)
        del it, model, g, pair, cur
        loader.close()
        need = resident * per_gpu_batch
        cores = st["cpu_ms_per_sample"] * 1e-3 * fed * per_gpu_batch
        out = {"steady_state_step_per_s": fed, "compute_bound_step_per_s": resident,
               "steady_state_over_compute_bound": fed / resident, "file_fed_over_compute_bound": fed / resident,
               "samples_timed": steps * per_gpu_batch, "prefetch_capacity_samples": capacity,
               "samples_consumed_before_the_timed_region": (2 * (slots + 1) + 3) * per_gpu_batch,
               "training_leg_step_per_s": compute_bound_step_per_s,
               "loader": "data_loading.PinnedBatchLoader: rows written in place into pinned batch buffers by a thread pool",
               "threads": threads, "slots": slots, "loader_only_samples_per_s": loader_sps, "samples_per_s_needed_by_one_gpu": need,
               "loader_headroom": loader_sps / need, "loader_cpu_ms_per_sample": st["cpu_ms_per_sample"],
               "loader_only_cpu_ms_per_sample": loader_only_cpu_ms, "loader_cpu_cores_used": cores,
               "eight_gpus_loader_cores": 8 * cores, "process_cpu_cores_used": process_cpu / wall,
               "slow_path_samples": st["slow_path_samples"], "stale_raw_copies": st["stale_raw_copies"],
               "bound": "loader" if (loader_sps < need or fed < 0.95 * resident) else "gpu",
               "host_threads": ncpu, "cpu_quota": cpu_quota(), "per_gpu_batch": per_gpu_batch, "MB_on_disk_per_sample": mb,
               "frames_on_disk": n_frames, "vertices_per_frame": list(vertex_count_range), "dataset_write_s": t_write,
               "raw_cache_files_written": n_raw, "raw_cache_convert_s": t_convert}
        torch.cuda.empty_cache()

        # eight loaders at once, one process each (no GPU in them): what an 8-GPU node asks of this host under this quota
        procs = [subprocess.Popen([sys.executable, "-m", "nvblox_mindmap_amd.data_loading.pinned_loader", root, "4", "2"], cwd=ROOT,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(8)]
        eight = []
        for p in procs:
            so, _ = p.communicate(timeout=300)
            lines = [ln for ln in so.splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                eight.append(json.loads(lines[-1]))
        if len(eight) == 8:
            agg = sum(e["samples_per_s"] for e in eight)
            out["eight_loaders"] = {"aggregate_samples_per_s": agg, "needed_by_8_gpus": 8 * need, "headroom": agg / (8 * need),
                                    "threads_each": 2, "cpu_ms_per_sample": sum(e["cpu_ms_per_sample"] for e in eight) / 8,
                                    "cores_burnt_at_the_needed_rate": 8 * need * sum(e["cpu_ms_per_sample"] for e in eight) / 8 * 1e-3}

        if reference_loader:
            # the reference-shaped loader on the same files, loader only: torch DataLoader, worker processes, default_collate, pin
            # thread -- with the raw copies (round 4's loader) and without (decompress everything: the reference's own path)
            from torch.utils.data import DataLoader

            def rate(raw, nw):
                d2 = MindmapFrameDataset(root, num_vertices=2048, use_raw_vertex_cache=raw)
                d2.samples = d2.samples * max(1, -(-3 * nw * per_gpu_batch // len(d2.samples)))
                dl = DataLoader(d2, batch_size=per_gpu_batch, shuffle=True, num_workers=nw, pin_memory=True, persistent_workers=True, prefetch_factor=4)
                for i, b in enumerate(dl):  # page cache + worker start-up, untimed
                    if i >= nw:
                        break
                t0 = time.perf_counter()
                k = 0
                for b in dl:
                    k += b["rgb_u8"].shape[0]
                r = k / (time.perf_counter() - t0)
                del dl
                return r

            out["torch_dataloader_loader_only_samples_per_s"] = {
                "raw_copies_10_workers": rate(True, 10), "zst_png_20_workers_reference_path": rate(False, max(1, min(20, ncpu - 2)))}
    finally:
        shutil.rmtree(root, ignore_errors=True)
        torch.set_num_threads(host_threads_before)
    torch.cuda.empty_cache()
    return out


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


def frame_byte_model(cfg, C, n_live, n_tsdf_upd, n_cand, n_surv, with_decay=True):
    """ALGORITHMIC bytes per launch of the fused frame = what THIS implementation's algorithm has to move between HBM and the
    chip, counted from the run's own device counters (DESIGN.md section 5 states the same formulas):
      k_front        depth f32 + input mask u8 read, masked depth f32 + two bit-row planes written
      k_alloc_tsdf   every LIVE TSDF block read (8 B/voxel: the pass flags appearance candidates on the voxels it holds) and,
                     with a decay pending, written back (else only the blocks the frame integrates); masked depth read once;
                     bit rows read, depth mask + feature mask u8 written; 32 B of list / stamp / summary words per live block
      k_sphere_alloc near-surface (candidate) blocks' TSDF voxels read once + the 1/4-resolution synthetic depth written
                     + the two candidate lists (16 B per candidate and layer)
      k_app_frame    candidate blocks: colour voxels (8 B) + feature weights (4 B) read and written; rgb u8x3 + two masks +
                     synthetic depth read; 20 B survivor record per surviving voxel
      k_feature_flat per surviving voxel: f16 row read + written (2 x 2C), four bilinear taps (4 x 2C), 20 B record
    """
    HW = cfg.height * cfg.width
    synth = (cfg.height // 4) * (cfg.width // 4) * 4
    return {
        "k_front": HW * (4 + 1 + 4) + 2 * HW / 8,
        "k_alloc_tsdf": n_live * 512 * 8 + (n_live if with_decay else n_tsdf_upd) * 512 * 8 + HW * 4 + 2 * HW / 8 + 2 * HW + 32 * n_live,
        "k_sphere_alloc": n_cand * 512 * 8 + synth + 2 * 16 * n_cand,
        "k_app_frame": n_cand * 512 * (8 + 4) * 2 + HW * (3 + 1 + 1) + synth + 20 * n_surv,
        "k_feature_flat": n_surv * flat_bytes_per_voxel(C),
    }


KERNEL_OF_CLASS = {"raycast": "k_front", "tsdf": "k_alloc_tsdf", "sphere": "k_sphere_alloc", "feature": "k_app_frame",
                   "feature_flat": "k_feature_flat"}
# What bounds each launch, with the counter evidence it rests on (tools/profile_sq.sh -> profiles/*_sq_summary.json: SQ counters of
# the same bench command, fractions of SQ_WAVE_CYCLES).  valu_issue: the achieved rate is VALU wave-instructions/s against the
# chip's issue peak (1 024 SIMDs x clock / 4 cycles per wave64 instruction); latency: most wave-cycles are parked in s_waitcnt on
# dependent loads at full occupancy; hbm: bytes/s against the HBM peak.
VALU_ISSUE_PEAK_PER_S = 1024 * 2.4e9 / 4.0
BOUND_OF_KERNEL = {"k_front": "valu_issue", "k_alloc_tsdf": "latency", "k_sphere_alloc": "valu_issue", "k_app_frame": "latency",
                   "k_feature_flat": "hbm", "k_front_app": "valu_issue", "k_sphere_alloc_flat": "valu_issue"}
# The launches of a frame: (profile class, kernel, the roles whose algorithmic bytes it moves).  Deferred mode (the headline:
# mmf_set_deferred_feature_rows): the colour update + feature gating and the row update of frame N are roles of launches 1 and 3
# of frame N + 1 -- three launches per frame in a stream.
LAUNCHES_EAGER = [("raycast", "k_front", ["k_front"]), ("tsdf", "k_alloc_tsdf", ["k_alloc_tsdf"]), ("sphere", "k_sphere_alloc", ["k_sphere_alloc"]),
                  ("feature", "k_app_frame", ["k_app_frame"]), ("feature_flat", "k_feature_flat", ["k_feature_flat"])]
LAUNCHES_DEFERRED = [("raycast", "k_front_app", ["k_front", "k_app_frame"]), ("tsdf", "k_alloc_tsdf", ["k_alloc_tsdf"]),
                     ("sphere", "k_sphere_alloc_flat", ["k_sphere_alloc", "k_feature_flat"])]


def counters_stamp(path):
    """The `__csrc_sha16__` a counter summary under profiles/ carries (the native sources it was collected on), or None (an older
    summary without a stamp)."""
    try:
        with open(path) as fh:
            return json.load(fh).get("__csrc_sha16__")
    except Exception:
        return None


def sq_evidence():
    """Per-kernel SQ summary of the latest committed counter run (a replayed constant like roofline.traffic: labelled)."""
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_summary.json"))
    if not files:
        return {}, None
    with open(os.path.join(ROOT, "profiles", files[-1])) as fh:
        raw = json.load(fh)
    out = {}
    for name, v in raw.items():
        if not isinstance(v, dict) or "SQ_WAVES" not in v:
            continue
        base = name.split("<")[0]
        if base in BOUND_OF_KERNEL and base not in out:
            out[base] = {k: v.get(k) for k in ("SQ_WAVES", "valu_per_wave", "salu_per_wave", "frac_parked", "frac_issuing", "frac_issue_stall")}
    return out, f"profiles/{files[-1]} ({raw.get('__source__', '')})"


def pmc_traffic():
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this same command (profiles/latest_pmc.json):
    2 x FETCH_SIZE (gfx950 correction of the guide) + WRITE_SIZE.  A replayed constant from the builder's profile run, not a
    live measurement -- the line says so (`traffic_source`)."""
    try:
        with open(os.path.join(ROOT, "profiles", "latest_pmc.json")) as f:
            pmc = json.load(f)
        src = pmc.get("__source__", "profiles/latest_pmc.json (builder's rocprofv3 --pmc passes of this command)")
        out = {}
        for k, v in pmc.items():  # keys carry the template arguments (k_front<true>): fold them onto the kernel's base name
            if isinstance(v, dict) and "FETCH_SIZE_KB" in v:
                out.setdefault(k.split("<")[0], (2.0 * v["FETCH_SIZE_KB"] + v["WRITE_SIZE_KB"]) * 1024.0)
        return out, src
    except Exception:
        return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=5, help="the timed region of exactly --steps steps is run this many times; the median is reported")
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--frames", type=int, default=0, help="distinct pre-generated frames (default min(steps, 200))")
    ap.add_argument("--cpu-sample", type=int, default=12, help="frames timed on the CPU oracle per thread setting (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="no per-launch timing inside the timed region")
    ap.add_argument("--no-ref-shape", action="store_true", help="skip the short run at the reference's real shape (512x512x768)")
    ap.add_argument("--no-train", action="store_true", help="skip the policy training-step measurement")
    ap.add_argument("--train-steps", type=int, default=16)
    ap.add_argument("--no-infer", action="store_true", help="skip the policy inference latency leg")
    ap.add_argument("--no-backproj", action="store_true", help="skip the back-projection leg (GPU kernel + torch-CPU baseline)")
    ap.add_argument("--no-file-fed", action="store_true", help="skip the file-fed training leg (loader-bound vs compute-bound step/s)")
    ap.add_argument("--only-fusion", action="store_true", help="headline fusion measurement only (what the rocprofv3 passes run)")
    ap.add_argument("--ref-shape-only", action="store_true", help="run only the 512x512x768 leg (rocprofv3 passes at the reference shape)")
    ap.add_argument("--file-fed-only", action="store_true", help="run only the file-fed training leg (loader-bound vs compute-bound step/s)")
    ap.add_argument("--in-flight-only", action="store_true", help="run only the frames-in-flight leg (N replicas in one set of launches)")
    ap.add_argument("--unbounded-only", action="store_true", help="run only the unbounded-workspace (hash path) leg (rocprofv3 passes)")
    ap.add_argument("--eager-rows", action="store_true", help="headline without the deferred feature-row update (every frame runs its five launches before the next starts)")
    ap.add_argument("--train-only", action="store_true", help="run only the captured training step (rocprofv3 passes of the training half)")
    ap.add_argument("--with-file-fed", action="store_true", help="with --train-only: also run the (rank-0-only) file-fed leg behind it")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise the multi-rank control flow only")
    args = ap.parse_args()
    if args.only_fusion:
        args.no_ref_shape = args.no_train = args.no_infer = args.no_backproj = args.no_file_fed = True
        args.cpu_sample = 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and "WORLD_SIZE" in os.environ:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, world, rank)
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

    if args.file_fed_only:
        out = run_training_file_fed(device, compute_bound_step_per_s=None)
        if rank == 0:
            print(json.dumps({"file_fed": out}), flush=True)
        return
    if args.train_only:
        out = run_training_graphed(device, world, steps=args.train_steps,
                                   backbone_matmul_dtype=os.environ.get("BENCH_BACKBONE_MATMULS", "float16x3"),
                                   overlap_backbone=os.environ.get("BENCH_TRAIN_OVERLAP", "1") != "0")
        if args.with_file_fed:
            # the rank-0-only leg behind the data-parallel ones, as in the full run (tests/test_gpu_bench_ranks.py: it must not issue
            # a collective the other ranks are not in)
            if dist is not None:
                dist.barrier()
            if rank == 0:
                out["file_fed"] = run_training_file_fed(device, compute_bound_step_per_s=out["step_per_s"], steps=max(args.train_steps // 2, 2), reference_loader=False)
        if rank == 0:
            print(json.dumps({"train": out, "n_gpus": world}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.ref_shape_only:
        out = run_reference_shape(device)
        if rank == 0:
            print(json.dumps({"reference_shape": out}), flush=True)
        return

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    n_frames = args.frames or min(max(args.steps, 1), 200)
    frames = build_stream(cfg, n_frames, args.channels, device)
    if args.in_flight_only:
        out = run_frames_in_flight(device, frames, args.channels)
        if rank == 0:
            print(json.dumps({"frames_in_flight": out}), flush=True)
        return
    if args.unbounded_only:
        out = run_unbounded(device, frames, args.channels, steps=args.steps, warmup=args.warmup)
        if rank == 0:
            print(json.dumps({"unbounded_workspace": out}), flush=True)
        return
    mapper = get_nvblox_mapper(mcfg, feature_channels=args.channels)
    # consecutive frames software-pipelined (mmf_set_deferred_feature_rows): a frame's last launch rides in the next frame's
    # sphere-trace launch; the frames of this stream are resident and never modified (the contract), and every timed region
    # ends with a flush -- all the work of its frames is inside it
    mapper.set_deferred_feature_rows(not args.eager_rows)

    for i in range(args.warmup):
        step(mapper, mcfg, frames[i % n_frames])
    torch.cuda.synchronize(device)
    mapper.reset_stats(MAPPER_TO_ID.STATIC)
    mapper.profile_reset()
    # (per-launch durations are taken in a region of their own behind the headline regions: the headline is not event-stamped)

    # K timed regions of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides; max over ranks per
    # region; the MEDIAN region is the reported one (a 14 ms region is at the mercy of one scheduler hiccup)
    regions, enqueue = [], []
    k = args.warmup
    for _ in range(max(args.repeats, 1)):
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(mapper, mcfg, frames[(k + i) % n_frames])
        mapper.flush()  # the last frame's deferred row update
        t_enq = time.perf_counter() - t0  # host time to enqueue all steps (the GPU may lag behind)
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        el = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        regions.append(el)
        enqueue.append(t_enq)
        k += args.steps
    elapsed = statistics.median(regions)
    t_enqueued = statistics.median(enqueue)

    prof = {}
    if not args.no_profile:
        # one more region of the same steps, NOT part of the headline: every launch of every 2nd frame is stamped with its dispatch's
        # own begin / end times (extension-launch events: no marker packets) -> roofline.per_kernel / kernels_busy_us
        mapper.profile_reset()
        mapper.profile_enable(True, kernels=list(KERNEL_OF_CLASS), stride=2)
        for i in range(args.steps):
            step(mapper, mcfg, frames[(k + i) % n_frames])
        mapper.flush()
        torch.cuda.synchronize(device)
        k += args.steps
        mapper.profile_enable(False)
        prof = mapper.profile()
    undeferred = None
    if not args.eager_rows:
        # the same stream with every frame's five launches run before the next frame starts (what a caller gets who cannot keep
        # a frame's images untouched until the next call): two regions, untimed per launch
        mapper.set_deferred_feature_rows(False)
        und = []
        for _ in range(2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(mapper, mcfg, frames[(k + i) % n_frames])
            torch.cuda.synchronize(device)
            und.append(time.perf_counter() - t0)
            k += args.steps
        undeferred = {"frames_per_s": args.steps / min(und), "ms_per_step": min(und) / args.steps * 1e3,
                      "launches_per_frame": 5}
    stats = mapper.stats(MAPPER_TO_ID.STATIC)
    n_live = int(mapper.tsdf_layer_view(MAPPER_TO_ID.STATIC).num_allocated_blocks())

    ref_shape = None
    if rank == 0 and not args.no_ref_shape:
        ref_shape = run_reference_shape(device)
    train = None
    if not args.no_train:  # every rank takes part (data parallel)
        if dist is not None:
            dist.barrier()
        # headline: the captured step (HIP graphs, flat gradient buffer, one explicit all-reduce); f32 accuracy throughout
        train = run_training_graphed(device, world, steps=args.train_steps)
        train["model"] = "diffuser_actor RGBD_AND_MESH, 1 cam 512x512, 2048 vertices x 768, frozen ViT-B/16-shaped backbone (random init)"
        if dist is not None:
            dist.barrier()
        # the reference-shaped step beside it: eager PyTorch, DistributedDataParallel(find_unused_parameters=True), AdamW over
        # the individual parameters (mindmap/run_training.py:155-217,608-613) -- host-bound on this machine
        eager = run_training(device, world, steps=max(args.train_steps // 2, 4))
        train["eager_ddp_reference_shaped"] = {k: eager[k] for k in ("step_per_s", "ms_per_step", "trainable_params", "frozen_backbone_params",
                                                                      "backbone_matmuls", "parallelism")}
        # secondary figure, not the headline: the frozen backbone's matmuls with float16 inputs / fp32 accumulation -- the
        # mantissa width of the TF32 mode the reference runs its backbone in (feature_extraction.py:322); gfx950 has no TF32
        t16 = run_training_graphed(device, world, steps=args.train_steps, backbone_matmul_dtype="float16")
        train["fp16_backbone_matmuls"] = {"step_per_s": t16["step_per_s"], "ms_per_step": t16["ms_per_step"],
                                          "host_enqueue_ms_per_step": t16["host_enqueue_ms_per_step"],
                                          "note": "frozen backbone under float16 autocast (10-bit mantissa like the reference's TF32 "
                                                  "backbone, fp32 accumulate); everything trainable stays float32"}
        if rank == 0 and not args.no_file_fed:
            train["file_fed"] = run_training_file_fed(device, compute_bound_step_per_s=train["step_per_s"])
    infer = run_policy_inference(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    closed_loop = run_closed_loop(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    model_inputs = None
    if rank == 0 and not args.no_ref_shape:
        model_inputs = {"ref": run_model_inputs(device, "ref"), "bl": run_model_inputs(device, "bl")}
    tsdf_only = run_tsdf_only(device) if (rank == 0 and not args.no_ref_shape) else None
    two_mappers = run_two_mappers(device, frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    unbounded = run_unbounded(device, frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    in_flight = run_frames_in_flight(device, frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    pixel_holes = run_pixel_holes(device, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    backproj = run_backprojection(device) if (rank == 0 and not args.no_backproj) else None  # has CPU legs: after every GPU measurement

    if rank == 0:
        C = args.channels
        n_feat_frames = max(stats["feature_frames"], 1)
        feat_blocks_per_frame = stats["feature_blocks_updated"] / n_feat_frames
        tsdf_blocks_per_frame = stats["tsdf_blocks_updated"] / max(stats["depth_frames"], 1)
        col_blocks_per_frame = stats["color_blocks_updated"] / max(stats["color_frames"], 1)
        feat_voxels_per_frame = stats["feature_voxels_updated"] / n_feat_frames
        fps = world * args.steps / elapsed
        t_frame = elapsed / args.steps
        model = frame_byte_model(cfg, C, n_live, tsdf_blocks_per_frame, col_blocks_per_frame, feat_voxels_per_frame)
        b_frame = sum(model.values())  # (the five roles; the launches that carry them are named below)
        traffic, traffic_src = pmc_traffic()
        sq, sq_src = sq_evidence()
        per_kernel, busy_us = [], 0.0
        launches = LAUNCHES_EAGER if args.eager_rows else LAUNCHES_DEFERRED
        for cls, kname, roles in launches:
            ms, n = prof.get(cls, (0.0, 0))
            us = ms / n * 1e3 if n else None
            busy_us += us or 0.0
            model[kname] = sum(model[r] for r in roles)
            hbm_frac = (model[kname] / (us * 1e-6) / HBM_PEAK_BYTES_PER_S) if us else None
            bound = BOUND_OF_KERNEL[kname]
            ev = sq.get(kname)
            frac_of_bound = hbm_frac
            if bound == "valu_issue" and ev and us:
                frac_of_bound = ev["SQ_WAVES"] * ev["valu_per_wave"] / (us * 1e-6) / VALU_ISSUE_PEAK_PER_S
            elif bound == "latency" and ev:
                frac_of_bound = None  # no throughput peak to divide by: the evidence is the parked fraction
            per_kernel.append({
                "kernel": kname, "roles": roles, "avg_us": us, "launches_timed": n, "algorithmic_bytes": model[kname],
                "achieved_GBps": (model[kname] / (us * 1e-6) / 1e9) if us else None,
                "frac": hbm_frac, "bound": bound, "frac_of_bound": frac_of_bound, "sq_counters": ev,
                "traffic": traffic.get(kname)})
        timed = [k_ for k_ in per_kernel if k_["avg_us"]]
        dominant = max(timed, key=lambda k_: k_["avg_us"])["kernel"] if timed else None
        # SURVEY.md section 8(d) model, kept for comparison (it charges the whole feature image and every voxel of every candidate block)
        survey_bytes = (cfg.height * cfg.width * (4 + 1) + tsdf_blocks_per_frame * 512 * 16 + cfg.height * cfg.width * (2 * C + 1)
                        + feat_blocks_per_frame * 512 * 2 * (2 * C + 4) + cfg.height * cfg.width * (3 + 1) + col_blocks_per_frame * 512 * 16)
        if undeferred is not None:  # the same algorithmic bytes over the unpipelined frame time (what earlier rounds' lines report)
            undeferred["frac_of_hbm_peak"] = b_frame / (undeferred["ms_per_step"] * 1e-3) / HBM_PEAK_BYTES_PER_S
        roofline = {
            "bound": "hbm",
            "kernel": ("whole frame = 5 launches (k_front, k_alloc_tsdf, k_sphere_alloc, k_app_frame, k_feature_flat)" if args.eager_rows else
                       "whole frame = 3 launches of a software-pipelined stream (k_front_app = raycast | mask rows | decay | colour update + "
                       "feature gating of the previous frame; k_alloc_tsdf; k_sphere_alloc_flat = sphere trace | appearance allocation | "
                       "feature rows of the previous frame)"),
            "achieved": b_frame / t_frame / 1e9,
            "peak": HBM_PEAK_BYTES_PER_S / 1e9,
            "unit": "GB/s",
            "frac": b_frame / t_frame / HBM_PEAK_BYTES_PER_S,
            "traffic": sum(traffic.get(k_, 0.0) for _, k_, _ in launches) if traffic else None,
            "traffic_source": traffic_src,
            "sq_counters_source": sq_src,
            "algorithmic_bytes_per_frame": b_frame,
            "formula": "sum over the five roles of frame_byte_model() (bench.py; DESIGN.md section 5), counts from this run",
            "frame_us": t_frame * 1e6,
            "kernels_busy_us": busy_us if timed else None,
            "launches_per_frame": len(launches),
            "dominant_launch": dominant,
            "per_kernel": per_kernel,
            "counts_per_frame": {"tsdf_live_blocks": n_live, "tsdf_blocks_integrated": tsdf_blocks_per_frame,
                                 "appearance_candidate_blocks": col_blocks_per_frame, "feature_blocks_updated": feat_blocks_per_frame,
                                 "feature_voxels_updated": feat_voxels_per_frame},
            "measured_d2d_copy_GBps": measure_d2d_copy(device),  # read + write rate of a 1 GiB copy on this box
            "note": "a frame's five roles are latency- or issue-bound but the feature rows (bandwidth-bound; they dominate at the "
                    "reference shape: reference_shape.k_feature_flat_*); in a stream the two appearance roles of frame N run beside "
                    "the raycast and the sphere trace of frame N + 1",
            "survey_8d_model_bytes_per_frame": survey_bytes,
            "survey_8d_model_frac": survey_bytes / t_frame / HBM_PEAK_BYTES_PER_S,
        }
        # which build do the replayed counters (traffic, sq_counters) belong to?
        from nvblox_mindmap_amd._lib import source_hash

        build = source_hash()
        pmc_stamp = counters_stamp(os.path.join(ROOT, "profiles", "latest_pmc.json"))
        sq_files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_summary.json"))
        sq_stamp = counters_stamp(os.path.join(ROOT, "profiles", sq_files[-1])) if sq_files else None
        roofline["build_csrc_sha16"] = build
        roofline["traffic_csrc_sha16"] = pmc_stamp
        roofline["sq_counters_csrc_sha16"] = sq_stamp
        roofline["counters_stale"] = bool(pmc_stamp != build or sq_stamp != build)  # True: collected on other sources than the timed build
        # the other shapes / paths of the same hot path, compactly (their full records are the top-level keys of the same names)
        legs = {}
        if ref_shape:
            legs["reference_shape_512x512x768"] = {
                "frames_per_s": ref_shape["frames_per_s"], "whole_frame_frac": ref_shape["whole_frame"]["frac_of_hbm_peak"],
                "pipelined_frames_per_s": ref_shape["pipelined"]["frames_per_s"], "pipelined_whole_frame_frac": ref_shape["pipelined"]["frac_of_hbm_peak"],
                "k_feature_flat_us": ref_shape.get("k_feature_flat_us"), "k_feature_flat_frac": ref_shape.get("k_feature_flat_frac_of_hbm_peak"),
                "fused_lowres_ms": ref_shape["from_backbone_output"]["fused_lowres_ms"],
                "fused_lowres_k_feature_flat_us": ref_shape["from_backbone_output"]["fused_lowres_k_feature_flat_us"],
                "fused_lowres_pipelined_ms": ref_shape["from_backbone_output_pipelined"]["fused_lowres_ms"]}
        if unbounded:
            legs["unbounded_workspace_hash_path"] = {"frames_per_s": unbounded["frames_per_s"], "frac": unbounded["frac"],
                                                     "pipelined_frames_per_s": (unbounded.get("pipelined") or {}).get("frames_per_s"),
                                                     "live_blocks": unbounded["hash"]["live_blocks"], "launches_per_frame": unbounded.get("launches_per_frame")}
        if pixel_holes:
            legs["survey_8d_pixel_holes"] = {k_: pixel_holes[k_] for k_ in ("frames_per_s", "ms_per_step", "frac_of_hbm_peak",
                                                                           "feature_voxels_updated_per_frame", "algorithmic_bytes_per_frame")}
        if undeferred:
            legs["undeferred_5_launches"] = {"frames_per_s": undeferred["frames_per_s"], "frac": undeferred.get("frac_of_hbm_peak")}
        if train:
            legs["train_step"] = {k_: train.get(k_) for k_ in ("step_per_s", "ms_per_step", "host_enqueue_ms_per_step", "host_enqueue_frac",
                                                               "rccl_world_observed", "allreduce", "per_rank_ms_per_step", "parallelism",
                                                               "backbone_matmuls", "tuned_gemms")}
            legs["train_step"]["eager_ddp_ms_per_step"] = (train.get("eager_ddp_reference_shaped") or {}).get("ms_per_step")
            legs["train_step"]["fp16_backbone_ms_per_step"] = (train.get("fp16_backbone_matmuls") or {}).get("ms_per_step")
        if closed_loop:
            legs["closed_loop_ms"] = closed_loop.get("ms_per_control_step") if isinstance(closed_loop, dict) else None
        roofline["legs"] = legs
        cpu = None
        if args.cpu_sample > 0:
            cpu = cpu_baseline(cfg, mcfg, frames, C, min(args.cpu_sample, n_frames))
        if cpu is not None and backproj:
            cpu["backprojection"] = backproj  # the reference's CPU back-projection path beside the HIP kernel
        out = {
            "metric": "RGB-D+feature frames/s fused @1 cm voxels",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": len(regions),
            "region_ms": [r * 1e3 for r in regions],
            "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: decay + depth(TSDF) + colour + 64-ch f16 feature fusion, 640x480 stream, "
                            "1 cm voxels, DRILL_IN_BOX workspace, reference mask algebra (17/20-pixel erosions, 5 % border); "
                            "hole_mode=patches: 16x16 invalid-depth patches instead of SURVEY 8(d)'s 1 % single-pixel holes "
                            "(those + the 20-pixel erosion erase the whole feature mask: no feature work at all); "
                            "replicas only for n_gpus>1",
                "hole_mode": cfg.hole_mode,
                "pipelined": not args.eager_rows,
                "image": [cfg.height, cfg.width],
                "feature_channels": C,
                "voxel_size_m": mcfg.voxel_size_m,
                "distinct_frames": n_frames,
                "tsdf_blocks_per_frame": tsdf_blocks_per_frame,
                "feature_blocks_per_frame": feat_blocks_per_frame,
                "frame_pipelining": ("off: every frame's five launches run before the next frame's" if args.eager_rows else
                                     "mmf_set_deferred_feature_rows: launches 4 and 5 of frame N (colour update + feature gating, feature "
                                     "rows) run as roles of launches 1 and 3 of frame N + 1; the stream's frames are resident and "
                                     "unmodified (the mode's contract); every timed region ends with mmf_flush, so all the work of "
                                     "its frames is inside it; maps bit-identical to the unpipelined sequence "
                                     "(tests/test_gpu_deferred_rows.py); the unpipelined rate is `undeferred`"),
            },
            "undeferred": undeferred,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "reference_shape": ref_shape,
            "policy_inference": infer,
            "closed_loop": closed_loop,
            "model_inputs": model_inputs,
            "tsdf_only": tsdf_only,
            "two_mappers": two_mappers,
            "unbounded_workspace": unbounded,
            "frames_in_flight": in_flight,
            "backprojection": backproj,
            "pixel_holes": pixel_holes,
            "train": train,
        }
        emit(out)
    if dist is not None:
        dist.barrier()  # rank 0 runs its single-GPU legs after the timed regions: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
