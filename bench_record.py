"""The ONE final stdout line of bench.py: a compact record the driver can parse (<= 4 KB, always).

bench.py measures 14 legs; the full record (tens of KB) goes to `bench_full.json` (repo root, and `gpurun_out/` when that
directory exists) and to stderr.  The last stdout line is built here from the full record: the contract keys of the
driver (metric, value, unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data,
config) + `roofline` + `cpu_baseline` + `train`, numbers rounded to 5 significant digits, no prose longer than 120
characters.  No torch / numpy import: tests/test_cpu_bench.py builds a line from a stub record on any box.
"""
import json
import math

MAX_LINE_BYTES = 4096
MAX_PROSE_CHARS = 120
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")
CPU_BASELINE_KEYS = ("value", "unit", "cores", "kind", "sample")


def _num(x, digits=5):
    """floats to `digits` significant digits (a 17-digit double is 20 bytes of the line)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None
        if x == 0.0:
            return 0.0
        r = round(x, digits - 1 - int(math.floor(math.log10(abs(x)))))
        return int(r) if abs(r) >= 10 ** digits else r
    return x


def _get(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or d.get(p) is None:
            return default
        d = d[p]
    return d


def _short(s, n=MAX_PROSE_CHARS):
    if not isinstance(s, str) or len(s) <= n:
        return s
    return s[: n - 1] + "~"


def _clean(o):
    """drop None-valued keys of nested dicts (the contract's top-level None stays), round floats, cut prose."""
    if isinstance(o, dict):
        return {k: _clean(v) for k, v in o.items() if v is not None}
    if isinstance(o, (list, tuple)):
        return [_clean(v) for v in o]
    if isinstance(o, str):
        return _short(o)
    return _num(o)


def compact_record(full: dict) -> dict:
    """the compact record from bench.py's full record (missing legs are simply absent)."""
    cfg = full.get("config") or {}
    roof = full.get("roofline") or {}
    per_kernel = roof.get("per_kernel") or []
    dom = next((k for k in per_kernel if k.get("kernel") == roof.get("dominant_launch")), None)
    legs_in = roof.get("legs") or {}
    ref = legs_in.get("reference_shape_512x512x768") or {}
    unb = legs_in.get("unbounded_workspace_hash_path") or {}
    holes = legs_in.get("survey_8d_pixel_holes") or {}
    und = legs_in.get("undeferred_5_launches") or legs_in.get("undeferred") or {}
    legs = {
        "ref_k_feature_flat_us": ref.get("k_feature_flat_us"), "ref_k_feature_flat_frac": ref.get("k_feature_flat_frac"),
        "ref_frame_frac": ref.get("whole_frame_frac"), "ref_pipelined_frame_frac": ref.get("pipelined_whole_frame_frac"),
        "ref_fps": ref.get("frames_per_s"), "ref_pipelined_fps": ref.get("pipelined_frames_per_s"),
        "lowres_ms": ref.get("fused_lowres_ms"), "lowres_k_feature_flat_us": ref.get("fused_lowres_k_feature_flat_us"),
        "unbounded_fps": unb.get("frames_per_s"), "unbounded_frac": unb.get("frac"), "unbounded_launches": unb.get("launches_per_frame"),
        "unbounded_pipelined_fps": unb.get("pipelined_frames_per_s"),
        "undeferred_fps": und.get("frames_per_s"), "undeferred_frac": und.get("frac"), "undeferred_launches": und.get("launches_per_frame"),
        "pixel_holes_fps": holes.get("frames_per_s"), "pixel_holes_frac": holes.get("frac_of_hbm_peak"),
        "closed_loop_ms": legs_in.get("closed_loop_ms"),
        # the fusion phase of a control step through the facade: inside the loop (behind the 23 ms inference) and steps back to back
        "closed_loop_fusion_ms": _get(full, "closed_loop", "breakdown_ms", "fusion"),
        "closed_loop_fusion_back_to_back_ms": _get(full, "closed_loop", "fusion_back_to_back_ms"),
        "in_flight_1_fps": _get(full, "frames_in_flight", "1", "aggregate_frames_per_s"),
        "in_flight_4_fps": _get(full, "frames_in_flight", "4", "aggregate_frames_per_s"),
    }
    roofline = {k: roof.get(k) for k in ROOFLINE_KEYS}
    roofline["kernel"] = _short(roofline.get("kernel"))
    roofline.update({
        "algorithmic_bytes_per_frame": roof.get("algorithmic_bytes_per_frame"), "frame_us": roof.get("frame_us"),
        "kernels_busy_us": roof.get("kernels_busy_us"), "launches_per_frame": roof.get("launches_per_frame"),
        "dominant_launch": roof.get("dominant_launch"),
        "dominant_us": (dom or {}).get("avg_us"), "dominant_bytes": (dom or {}).get("algorithmic_bytes"),
        "dominant_frac": (dom or {}).get("frac"), "dominant_traffic": (dom or {}).get("traffic"),
        "launch_us": {k["kernel"]: k.get("avg_us") for k in per_kernel} or None,
        "build_csrc_sha16": roof.get("build_csrc_sha16"), "counters_stale": roof.get("counters_stale"),
        "measured_d2d_copy_GBps": roof.get("measured_d2d_copy_GBps"),
        "survey_8d_model_frac": roof.get("survey_8d_model_frac"),
        "legs": legs,
    })
    cpu_in = full.get("cpu_baseline")
    cpu = None
    if cpu_in:
        cpu = {k: cpu_in.get(k) for k in CPU_BASELINE_KEYS}
        bp = cpu_in.get("backprojection") or full.get("backprojection") or {}
        big = bp.get("batch_32x512x512") or {}
        cpu["backprojection"] = {"shape": "32x512x512", "gpu_fps": big.get("frames_per_s"), "gpu_frac": big.get("frac_of_hbm_peak"),
                                 "cpu_fps": big.get("cpu_frames_per_s"), "cpu_threads": big.get("cpu_threads")}
    tr_in = full.get("train")
    train = None
    if tr_in:
        ff = tr_in.get("file_fed") or {}
        train = {
            "step_per_s": tr_in.get("step_per_s"), "ms_per_step": tr_in.get("ms_per_step"), "per_gpu_batch": tr_in.get("per_gpu_batch"),
            "dtype": tr_in.get("dtype"), "parallelism": tr_in.get("parallelism"),
            "host_enqueue_frac": tr_in.get("host_enqueue_frac"), "rccl_world_observed": tr_in.get("rccl_world_observed"),
            "allreduce_ms": _get(tr_in, "allreduce", "mean_ms"), "collective_backend": tr_in.get("collective_backend"),
            "collectives_forced_on_one_rank": tr_in.get("collectives_forced_on_one_rank"), "allreduce_payload_MB": tr_in.get("allreduce_payload_MB"),
            "per_rank_ms_min": _get(tr_in, "per_rank_ms_per_step", "min"), "per_rank_ms_max": _get(tr_in, "per_rank_ms_per_step", "max"),
            "eager_ddp_ms": _get(tr_in, "eager_ddp_reference_shaped", "ms_per_step"),
            "fp16_backbone_ms": _get(tr_in, "fp16_backbone_matmuls", "ms_per_step"),
            "file_fed_steady_ratio": ff.get("steady_state_over_compute_bound"),
            "file_fed_step_per_s": ff.get("steady_state_step_per_s"),
            "file_fed_samples_timed": ff.get("samples_timed"), "file_fed_prefetch_capacity": ff.get("prefetch_capacity_samples"),
            "loader_cpu_ms_per_sample": ff.get("loader_cpu_ms_per_sample"), "loader_cores_used": ff.get("loader_cpu_cores_used"),
            "loader_only_samples_per_s": ff.get("loader_only_samples_per_s"), "loader_bound": ff.get("bound"),
            "eight_loaders_samples_per_s": _get(ff, "eight_loaders", "aggregate_samples_per_s"),
            "eight_gpus_need_samples_per_s": _get(ff, "eight_loaders", "needed_by_8_gpus"),
            # the reference's own loader path on the same files and host cores (torch DataLoader, 20 workers, zstd + PNG), loader only
            "reference_loader_samples_per_s": _get(ff, "torch_dataloader_loader_only_samples_per_s", "zst_png_20_workers_reference_path"),
        }
    out = {
        "metric": full.get("metric"), "value": full.get("value"), "unit": full.get("unit"), "n_gpus": full.get("n_gpus"),
        "steps": full.get("steps"), "warmup": full.get("warmup"), "ms_per_step": full.get("ms_per_step"),
        # the same stream without frame pipelining (every frame complete before the next starts: the plain Mapper's default contract)
        "value_undeferred": _get(full, "undeferred", "frames_per_s"), "ms_per_step_undeferred": _get(full, "undeferred", "ms_per_step"),
        "higher_is_better": full.get("higher_is_better", True), "scaling": full.get("scaling", "weak"),
        "vs_baseline": full.get("vs_baseline"), "dtype": full.get("dtype"), "data": full.get("data"),
        "region_ms": (full.get("region_ms") or [])[:9] or None, "host_enqueue_ms_per_step": full.get("host_enqueue_ms_per_step"),
        "config": {"workload": _short(cfg.get("workload")), "hole_mode": cfg.get("hole_mode"), "image": cfg.get("image"),
                   "feature_channels": cfg.get("feature_channels"), "voxel_size_m": cfg.get("voxel_size_m"),
                   "pipelined": cfg.get("pipelined")},
        "roofline": roofline, "cpu_baseline": cpu, "train": train,
        "rank0_only_s": ({k: full["rank0_only_s"].get(k) for k in ("before_the_training_barrier", "before_the_final_barrier", "collective_timeout_s")}
                         if full.get("rank0_only_s") else None),
        "full_record": full.get("full_record"),
    }
    if full.get("dry_run"):
        out["dry_run"] = True
    top = {k: (_num(v) if not isinstance(v, (dict, list)) else _clean(v)) for k, v in out.items()}
    for k in ("roofline", "cpu_baseline", "train"):  # a leg that did not run stays an explicit null
        top.setdefault(k, None)
    for leg, keys in (("roofline", ROOFLINE_KEYS), ("cpu_baseline", CPU_BASELINE_KEYS)):
        if isinstance(top.get(leg), dict):
            for k in keys:  # e.g. `traffic` may be null by the contract; the key stays
                top[leg].setdefault(k, None)
    return top


def compact_line(full: dict) -> str:
    """the final stdout line; sheds the optional parts, least important first, should it ever exceed the bound."""
    rec = compact_record(full)
    line = json.dumps(rec, separators=(",", ":"))
    for shed in (("roofline", "launch_us"), ("roofline", "legs"), ("train",), ("cpu_baseline", "backprojection")):
        if len(line.encode()) <= MAX_LINE_BYTES:
            break
        d = rec
        for p in shed[:-1]:
            d = d.get(p) or {}
        if isinstance(d, dict):
            d.pop(shed[-1], None)
        line = json.dumps(rec, separators=(",", ":"))
    if len(line.encode()) > MAX_LINE_BYTES:
        raise ValueError(f"compact bench line is {len(line.encode())} bytes > {MAX_LINE_BYTES}")
    return line
