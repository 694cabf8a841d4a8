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

The legs beside the headline live in `benchlib/` (common.py: stream, step, byte model, launch tables; fusion_legs.py; policy_legs.py);
this file keeps the driver's contract: ranks, the barrier-bracketed headline regions, the record and its compact last line.

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
from benchlib.common import *  # noqa: E402,F401,F403  (re-exported: tools use bench.build_stream / bench.step / bench.frame_byte_model ...)
from benchlib.common import S, MAPPER_TO_ID, NvbloxMappingCfg, get_nvblox_mapper, integrate_frame  # noqa: E402,F401
from benchlib.fusion_legs import *  # noqa: E402,F401,F403
from benchlib.policy_legs import *  # noqa: E402,F401,F403


def _flush_c_stdio() -> None:
    """Flush the C library's stdio buffers: native libraries of this process print through them -- RCCL its five-line version banner
    at communicator creation -- and a pipe makes them fully buffered, i.e. written at process EXIT, behind everything Python printed."""
    try:
        import ctypes

        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def silence_stdout() -> None:
    """File descriptor 1 of THIS process -> /dev/null, for good (Python's sys.stdout and every native library's printf alike)."""
    sys.stdout.flush()
    _flush_c_stdio()
    fd = os.open(os.devnull, os.O_WRONLY)
    os.dup2(fd, 1)
    os.close(fd)


def print_last_line(text: str) -> None:
    """The driver parses the LAST line of the job's stdout, which every rank shares.  Round 6 found RCCL's banner (buffered C stdio,
    flushed at exit) landing BEHIND rank 0's record once `backend="nccl"` really ran -- on an 8-GPU launch eight times.  So: ranks > 0
    never own stdout (`silence_stdout` at start-up), rank 0 flushes whatever native code has buffered, prints the record, and closes its
    stdout behind it: nothing can follow the record."""
    sys.stdout.flush()
    _flush_c_stdio()
    print(text, flush=True)
    silence_stdout()


def dist_timeout_s(dist):
    """The default process group's collective timeout in seconds (what a rank waiting in a barrier for rank 0's solo legs must stay under)."""
    try:
        import torch.distributed.distributed_c10d as c10d

        return float(c10d._get_default_timeout(dist.get_backend()).total_seconds())
    except Exception:
        return None


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
    print_last_line(bench_record.compact_line(full))


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
        print_last_line(bench_record.compact_line({
            "metric": "RGB-D+feature frames/s fused @1 cm voxels", "value": world * args.steps / elapsed, "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "none (dry run)",
            "config": {"workload": "dry run: no GPU work, control flow only"}, "dry_run": True,
            "roofline": {"bound": "hbm", "kernel": "none (dry run)", "achieved": None, "peak": HBM_PEAK_BYTES_PER_S / 1e9, "unit": "GB/s",
                         "frac": None, "traffic": None},
            "cpu_baseline": {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "none (dry run)"},
            "train": {"parallelism": f"dp{world}" if world > 1 else "single"}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


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
    if rank != 0:
        silence_stdout()  # the job's stdout carries ONE record, rank 0's, as its last line (print_last_line)
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
    forced = world == 1 and os.environ.get("BENCH_FORCE_DIST", "0") == "1"
    if world > 1:
        import torch.distributed as dist  # RCCL via backend "nccl": barrier + max-reduce of the timing, DDP all-reduce

        dist.init_process_group(backend=backend, init_method="env://")
    elif forced:
        # BENCH_FORCE_DIST=1 on one GPU: a process group of ONE rank over RCCL, and every collective an N-rank run issues is issued
        # (barriers around the timed regions, the device-tensor max-reduce, the weight broadcast, the flat-gradient all-reduce between
        # the two HIP graphs, DDP's wrapper) -- the 8-GPU launch must not be the first time backend="nccl" runs in this code base
        import torch.distributed as dist
        from nvblox_mindmap_amd.training.distributed import force_collectives, free_port

        dist.init_process_group(backend=backend, init_method=f"tcp://127.0.0.1:{free_port()}", world_size=1, rank=0)
        force_collectives(True)

    if args.file_fed_only:
        out = run_training_file_fed(device, compute_bound_step_per_s=None)
        if rank == 0:
            print_last_line(json.dumps({"file_fed": out}))
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
            print_last_line(json.dumps({"train": out, "n_gpus": world}))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    if args.ref_shape_only:
        out = run_reference_shape(device)
        if rank == 0:
            print_last_line(json.dumps({"reference_shape": out}))
        return

    cfg = S.StreamConfig(hole_mode="patches")
    mcfg = NvbloxMappingCfg("DRILL_IN_BOX")
    n_frames = args.frames or min(max(args.steps, 1), 200)
    frames = build_stream(cfg, n_frames, args.channels, device)
    if args.in_flight_only:
        out = run_frames_in_flight(device, frames, args.channels)
        if rank == 0:
            print_last_line(json.dumps({"frames_in_flight": out}))
        return
    if args.unbounded_only:
        out = run_unbounded(device, frames, args.channels, steps=args.steps, warmup=args.warmup)
        if rank == 0:
            print_last_line(json.dumps({"unbounded_workspace": out}))
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

    # rank 0 runs its single-GPU legs alone while the other ranks sit in the next collective (a barrier): how long they wait there is
    # reported (`rank0_only_s`), next to the backend's collective timeout that wait must stay under (RCCL's watchdog: 10 min by default)
    t_solo = time.perf_counter()
    ref_shape = None
    if rank == 0 and not args.no_ref_shape:
        ref_shape = run_reference_shape(device)
    solo_before_train = time.perf_counter() - t_solo
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
    t_solo = time.perf_counter()
    infer = run_policy_inference(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    closed_loop = run_closed_loop(device) if (rank == 0 and not args.no_infer and not args.no_train) else None
    model_inputs = None
    if rank == 0 and not args.no_ref_shape:
        model_inputs = {"ref": run_model_inputs(device, "ref"), "bl": run_model_inputs(device, "bl")}
    tsdf_only = run_tsdf_only(device) if (rank == 0 and not args.no_ref_shape) else None
    # the secondary legs run their own step counts: they get the default 200-frame stream whatever --steps is (at the
    # driver's --steps 20 the headline's 20-frame stream would make them measure a shorter orbit than the numbers quoted for them)
    leg_frames = frames
    if rank == 0 and not args.no_ref_shape and n_frames < 200:
        leg_frames = build_stream(cfg, 200, args.channels, device)
    two_mappers = run_two_mappers(device, leg_frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    unbounded = run_unbounded(device, leg_frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    in_flight = run_frames_in_flight(device, leg_frames, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    del leg_frames
    pixel_holes = run_pixel_holes(device, args.channels) if (rank == 0 and not args.no_ref_shape) else None
    # (has CPU legs -- the reference's torch-CPU back-projection on every host thread: after every GPU measurement, and like the CPU
    # baseline a figure of the N = 1 run)
    backproj = run_backprojection(device) if (rank == 0 and not args.no_backproj and world == 1) else None

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
        sq_file = latest_sq_summary()
        sq_stamp = counters_stamp(os.path.join(ROOT, "profiles", sq_file)) if sq_file else None
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
        if args.cpu_sample > 0 and world == 1:
            cpu = cpu_baseline(cfg, mcfg, frames, C, min(args.cpu_sample, n_frames))
        elif world > 1:
            # the CPU baseline is a figure of the N = 1 run (rank 0, one GPU, the host to itself): with N ranks alive the other ranks'
            # host threads sit in a barrier on the same cores
            cpu = {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": "not measured at n_gpus > 1: the CPU baseline is taken by the N = 1 run"}
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
            "rank0_only_s": {"before_the_training_barrier": solo_before_train, "before_the_final_barrier": time.perf_counter() - t_solo,
                             "collective_timeout_s": (dist_timeout_s(dist) if dist is not None else None),
                             "note": "ranks > 0 wait this long in a barrier while rank 0 runs its single-GPU legs"},
        }
        emit(out)
    if dist is not None:
        dist.barrier()  # rank 0 runs its single-GPU legs after the timed regions: nobody tears the group down before it is done
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
