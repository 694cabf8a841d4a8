"""The policy-side legs of bench.py: closed loop (fusion + map -> model inputs + 100-step inference), policy inference, the training
step (eager DDP-shaped and captured) and the file-fed training step in steady state.  Each returns a plain dict."""
import json
import os
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

from .common import *  # noqa: F401,F403
from .common import ROOT, S, MAPPER_TO_ID, NvbloxMappingCfg, get_nvblox_mapper, integrate_frame  # noqa: F401
from .fusion_legs import build_facade  # noqa: F401

def run_closed_loop(device, steps=20):
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
    parts = {"fusion": [], "map_to_model_input": [], "policy_inference": []}

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
                parts[name].append((b - a) * 1e3)
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
        # the fusion phase alone, steps back to back (no inference in between), synchronised after each: what the phase costs when
        # the host's caches are warm and the GPU has not idled -- inside the loop the same code follows 24 ms of inference
        # (profiles/r06c_facade_closed_loop_between.txt: the first launches after it are slow whatever they are)
        b2b = []
        for i in range(steps):
            fr, smp = frames[i % 4], samples[i % 4]
            ex.next, ex.low = fr["features"], fr["lowres"]
            t0 = time.perf_counter()
            facade.decay()
            facade.update_reconstruction_from_sample(smp, "pov")
            torch.cuda.synchronize(device)
            b2b.append((time.perf_counter() - t0) * 1e3)
        back_to_back = statistics.median(b2b)
    finally:
        DiffuserActor.enable_fused_inference(False)
    # breakdown: the MEDIAN over the control steps of each phase (a 0.3 ms phase measured 20 times is at the mercy of one scheduler
    # hiccup: means of 0.36-0.71 ms were seen for the same build); the means are kept beside it
    out = {"ms_per_control_step": total, "control_steps_per_s": 1e3 / total,
           "breakdown_ms": {k: statistics.median(v) for k, v in parts.items()},
           "breakdown_mean_ms": {k: sum(v) / len(v) for k, v in parts.items()},
           "fusion_back_to_back_ms": back_to_back, "steps": steps, "facade_frame_pipelining": bool(facade.frame_pipelining),
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

def run_training(device, world, steps=8, warmup=3, per_gpu_batch=32, backbone_matmul_dtype="float16x3", prefetch_backbone=False):
    """Policy training step/s (second half of the BASELINE metric; config 5): diffuser_actor, RGBD_AND_MESH, per-GPU batch 32,
    one 512x512 camera, 2048 vertices x 768 features, frozen ViT-B/16-shaped backbone (random-init stand-in for RADIO v2.5-B),
    fp32, synthetic cached-sample-shaped batches resident on the GPU; DDP (RCCL all-reduce) when world > 1."""
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import (BackbonePrefetcher, build_model, build_optimizer, synthetic_batch, train_one_step,
                                             wrap_ddp)
    from nvblox_mindmap_amd.training.distributed import barrier, collectives_active, max_over_ranks

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
    dt = max_over_ranks(time.perf_counter() - t0, device if torch.distributed.get_backend() == "nccl" else None) if collectives_active() else \
        time.perf_counter() - t0
    out = {"step_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "samples_per_s": steps * per_gpu_batch * world / dt,
           "per_gpu_batch": per_gpu_batch, "global_batch": per_gpu_batch * world, "steps": steps, "warmup": warmup,
           "trainable_params": n_train, "frozen_backbone_params": n_frozen, "dtype": "f32", "backbone_prefetch": bool(prefetch_backbone),
           "backbone_matmuls": backbone_matmul_dtype,
           "allreduce_payload_MB": n_train * 4 / 1e6, "parallelism": f"dp{world}" if world > 1 else "single",
           "ddp_wrapped": ddp is not model,
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
    from nvblox_mindmap_amd.training.distributed import barrier, collectives_active, max_over_ranks

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
    coll = collectives_active()  # world > 1, or the forced one-rank group (BENCH_FORCE_DIST=1)
    nccl = coll and torch.distributed.get_backend() == "nccl"
    dt = max_over_ranks(el, device if nccl else None) if coll else el
    per_rank = [r["ms"] for r in all_gather_objects({"ms": mine / steps * 1e3})]
    allreduce = None
    if g.collective:  # a second, short region with HIP events around the collective (kept out of the headline region)
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
           "rccl_world_observed": observed, "collective_backend": torch.distributed.get_backend() if coll else None,
           "collectives_forced_on_one_rank": bool(coll and world == 1), "allreduce": allreduce, "allreduce_payload_MB": g.flat_grad.numel() * 4 / 1e6,
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
