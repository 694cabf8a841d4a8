"""bench.py's multi-rank control flow without a GPU: `--gpus 2 --dry-run` must start two ranks by itself (no launcher, no
WORLD_SIZE in the environment), rendezvous over gloo on 127.0.0.1 and print ONE JSON line with n_gpus == 2 -- the contract the
driver relies on (reference: torchrun --standalone --nnodes 1 --nproc_per_node N, mindmap_osmo/tasks/training_task.py:38)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_the_ranks():
    out = run(["--gpus", "2", "--dry-run", "--steps", "5", "--warmup", "1", "--repeats", "3"])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1 and out["dry_run"] is True
    assert out["train"]["parallelism"] == "dp2" and out["scaling"] == "weak" and out["value"] > 0


def test_single_rank_dry_run_and_launcher_environment():
    out = run(["--dry-run", "--steps", "3", "--warmup", "0"])
    assert out["n_gpus"] == 1 and out["train"]["parallelism"] == "single"
    # under a launcher (WORLD_SIZE set) bench.py must not spawn again
    out = run(["--gpus", "1", "--dry-run", "--steps", "3", "--warmup", "0"],
              {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert out["n_gpus"] == 1


def _stub_full_record(prose=4000, kernels=5):
    """a record shaped like bench.py's full one, with every leg present and absurdly long prose / float digits everywhere."""
    long = "x" * prose
    pk = [{"kernel": f"k_{i}", "roles": ["a", "b"], "avg_us": 18.123456789012345 + i, "launches_timed": 100, "algorithmic_bytes": 33481234.123456,
           "achieved_GBps": 1771.123456789, "frac": 0.22123456789, "bound": "latency", "sq_counters": {"SQ_WAVES": 8385, "note": long},
           "traffic": 33480000.5} for i in range(kernels)]
    legs = {"reference_shape_512x512x768": {"frames_per_s": 8640.123456, "whole_frame_frac": 0.38, "pipelined_frames_per_s": 11533.9,
                                            "pipelined_whole_frame_frac": 0.51, "k_feature_flat_us": 45.8, "k_feature_flat_frac": 0.85,
                                            "fused_lowres_ms": 0.11, "fused_lowres_k_feature_flat_us": 51.1, "fused_lowres_pipelined_ms": 0.09},
            "unbounded_workspace_hash_path": {"frames_per_s": 3932.1889322128777, "frac": 0.17276751618839786, "live_blocks": 88533, "launches_per_frame": 9},
            "survey_8d_pixel_holes": {"frames_per_s": 19684.1, "ms_per_step": 0.05, "frac_of_hbm_peak": 0.108},
            "undeferred_5_launches": {"frames_per_s": 15679.875887452481, "frac": 0.1489337299738036, "launches_per_frame": 5},
            "closed_loop_ms": 24.936669004091527, "train_step": {"note": long}}
    return {
        "metric": "RGB-D+feature frames/s fused @1 cm voxels", "value": 18893.123456789, "unit": "frames/s", "n_gpus": 1, "steps": 200, "warmup": 20,
        "ms_per_step": 0.052929123456789, "repeats": 5, "region_ms": [10.5] * 5, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": long, "hole_mode": "patches", "image": [480, 640], "feature_channels": 64, "voxel_size_m": 0.01, "pipelined": True,
                   "frame_pipelining": long},
        "roofline": {"bound": "hbm", "kernel": long, "achieved": 1435.7123456789, "peak": 8000.0, "unit": "GB/s", "frac": 0.17946123456789,
                     "traffic": 78529000.123, "algorithmic_bytes_per_frame": 75987000.123, "frame_us": 52.929123456789, "kernels_busy_us": 52.496,
                     "launches_per_frame": 3, "dominant_launch": "k_1", "per_kernel": pk, "note": long, "formula": long, "build_csrc_sha16": "1c8285496003d2f1",
                     "counters_stale": False, "measured_d2d_copy_GBps": 5203.5, "survey_8d_model_frac": 0.41, "legs": legs},
        "cpu_baseline": {"value": 36.96944400254281, "unit": "frames/s", "cores": 16, "kind": "port", "sample": long,
                         "thread_sweep_frames_per_s": {str(t): 25.9 for t in (8, 16, 32, 64, 256)},
                         "backprojection": {"batch_32x512x512": {"frames_per_s": 1469382.6145731679, "frac_of_hbm_peak": 0.770379672229337,
                                                                 "cpu_frames_per_s": 328.4920258667571, "cpu_threads": 8}}},
        "train": {"step_per_s": 26.28344812337423, "ms_per_step": 38.04675837454852, "per_gpu_batch": 32, "dtype": "f32", "parallelism": "dp8",
                  "host_enqueue_frac": 0.013853063540732976, "rccl_world_observed": 8, "allreduce": {"mean_ms": 0.31234567, "min_ms": 0.3, "max_ms": 0.33, "payload_MB": 8.57}, "collective_backend": "nccl",
                  "allreduce_payload_MB": 8.573324, "per_rank_ms_per_step": {"min": 38.0, "max": 38.2, "all": [38.0] * 8}, "how": long,
                  "eager_ddp_reference_shaped": {"ms_per_step": 40.3}, "fp16_backbone_matmuls": {"ms_per_step": 25.4},
                  "file_fed": {"steady_state_over_compute_bound": 0.97, "steady_state_step_per_s": 25.5, "samples_timed": 4096,
                               "prefetch_capacity_samples": 1280, "loader_cpu_ms_per_sample": 1.4, "loader_cpu_cores_used": 1.2,
                               "loader_only_samples_per_s": 2500.0, "bound": "gpu", "note": long}},
        "reference_shape": {"note": long}, "unbounded_workspace": {"per_kernel": pk * 4}, "frames_in_flight": {"1": {"aggregate_frames_per_s": 17062.0},
                                                                                                                "4": {"aggregate_frames_per_s": 26568.0}},
    }


def test_compact_line_is_bounded_and_carries_the_contract():
    """the driver keeps an 8 KB tail of stdout: the last line must be <= 4 KB whatever the legs hold (BENCH_r04's parsed was null)."""
    sys.path.insert(0, ROOT)
    import bench_record

    for prose, kernels in ((10, 3), (4000, 5), (100000, 40)):
        line = bench_record.compact_line(_stub_full_record(prose, kernels))
        assert len(line.encode()) <= 4096 and "\n" not in line
        rec = json.loads(line)
        for k in bench_record.CONTRACT_KEYS:
            assert k in rec, k
        for k in bench_record.ROOFLINE_KEYS:
            assert k in rec["roofline"], k
        for k in bench_record.CPU_BASELINE_KEYS:
            assert k in rec["cpu_baseline"], k
        assert rec["config"]["workload"] and set(rec["config"]) >= {"workload", "hole_mode", "feature_channels", "voxel_size_m"}
        assert rec["roofline"]["frac"] == 0.17946 and rec["roofline"]["dominant_launch"] == "k_1" and rec["roofline"]["dominant_us"] is not None
        assert rec["train"]["allreduce_ms"] == 0.31235 and rec["train"]["collective_backend"] == "nccl" and rec["train"]["file_fed_steady_ratio"] == 0.97 and rec["train"]["per_rank_ms_max"] == 38.2
        assert rec["cpu_baseline"]["backprojection"]["cpu_fps"] == 328.49

        def longest(o):
            if isinstance(o, dict):
                return max([longest(v) for v in o.values()] + [0])
            if isinstance(o, list):
                return max([longest(v) for v in o] + [0])
            return len(o) if isinstance(o, str) else 0

        assert longest(rec) <= bench_record.MAX_PROSE_CHARS


def test_dry_run_prints_the_compact_shape():
    out = run(["--dry-run", "--steps", "3", "--warmup", "0"])
    sys.path.insert(0, ROOT)
    import bench_record

    for k in bench_record.CONTRACT_KEYS:
        assert k in out, k
    assert set(out["roofline"]) >= set(bench_record.ROOFLINE_KEYS) and set(out["cpu_baseline"]) >= set(bench_record.CPU_BASELINE_KEYS)


def test_the_record_is_the_last_stdout_line_whatever_native_code_prints():
    """Round 6, found by running RCCL for the first time: its version banner goes through C stdio, which a pipe makes fully buffered --
    it was written at process exit, BEHIND rank 0's record (and the driver parses the last stdout line).  `bench.print_last_line`
    flushes what native code has buffered, prints the record and closes the process's stdout behind it."""
    import subprocess
    import sys

    code = ("import ctypes, sys; sys.argv = ['bench.py']; import bench; libc = ctypes.CDLL(None); "
            "libc.printf(b'native banner (buffered)\\n'); bench.print_last_line('{\"record\": 1}'); "
            "libc.printf(b'late native line\\n'); print('late python line')")
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert lines[-1] == '{"record": 1}' and "native banner (buffered)" in lines[:-1], lines
    assert not any("late" in ln for ln in lines)
