"""bench.py's training legs under a MULTI-RANK launch on a real GPU: two ranks (gloo rendezvous, both on cuda:0 -- the box has one
GPU; RCCL itself needs one GPU per rank) run the captured data-parallel step, then rank 0 alone runs the file-fed leg.  Round 4
found the hang this guards against: the rank-0-only leg built its GraphedTrainStep on the default process group and waited in an
all-reduce nobody else entered (reference: torchrun --nproc_per_node N run_training.py, mindmap_osmo/tasks/training_task.py:38)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_train_then_rank0_file_fed_leg():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    env["BENCH_HANG_DUMP_S"] = "240"  # a hang prints every thread's stack and exits instead of eating the suite's time
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--train-only", "--with-file-fed", "--train-steps", "4"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    t = out["train"]
    assert out["n_gpus"] == 2 and t["parallelism"] == "dp2" and t["rccl_world_observed"] == 2 and t["global_batch"] == 64
    assert len(t["per_rank_ms_per_step"]["all"]) == 2 and t["allreduce"]["payload_MB"] > 1.0
    assert t["file_fed"]["steady_state_step_per_s"] > 0 and t["file_fed"]["slow_path_samples"] == 0


@pytest.mark.gpu
def test_eight_ranks_dress_rehearsal_on_one_gpu():
    """`bench.py --gpus 8` as the driver's SCALE run launches it -- eight ranks, here all on cuda:0 over gloo (RCCL wants a GPU per
    rank) -- through the WHOLE default flow with short legs: barrier-bracketed headline regions on eight replicas, the data-parallel
    training legs (captured step + the reference-shaped DDP step + the fp16-backbone variant) with eight ranks in every collective,
    then rank 0's single-GPU legs while the others wait in the final barrier.  Asserts: rc 0, ONE line of <= 4 KB on stdout,
    n_gpus 8, eight per-rank step times, an all-reduce that reached eight ranks, and the wait of the other ranks (printed in the
    line) far below the backend's collective timeout."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["BENCH_DIST_BACKEND"] = "gloo"
    env["BENCH_HANG_DUMP_S"] = "540"
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--train-steps", "4",
                        "--no-file-fed"], env=env, capture_output=True, text=True, timeout=600)
    wall = time.perf_counter() - t0
    assert p.returncode == 0, p.stderr[-4000:]
    every = [ln for ln in p.stdout.splitlines() if ln.strip()]
    lines = [ln for ln in every if ln.startswith("{")]
    # ONE record, rank 0's, and it is the LAST stdout line (gloo itself prints a "[Gloo] Rank i is connected ..." line per rank first)
    assert len(lines) == 1 and every[-1] is lines[0], p.stdout[-2000:]
    assert len(lines[0].encode()) <= 4096
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 20 and out["scaling"] == "weak" and out["value"] > 0
    assert out["value_undeferred"] > 0 and out["roofline"]["frac"] > 0
    assert out["cpu_baseline"]["value"] is None and "N = 1" in out["cpu_baseline"]["sample"]  # (the contract: rank 0 at N = 1 only)
    t = out["train"]
    assert t["parallelism"] == "dp8" and t["rccl_world_observed"] == 8 and t["collective_backend"] == "gloo"
    assert t["per_rank_ms_min"] > 0 and t["per_rank_ms_max"] >= t["per_rank_ms_min"] and t["allreduce_ms"] > 0
    full = json.load(open(os.path.join(ROOT, "bench_full.json")))
    assert len(full["train"]["per_rank_ms_per_step"]["all"]) == 8 and full["train"]["global_batch"] == 8 * 32
    solo = out["rank0_only_s"]
    assert solo["collective_timeout_s"] >= 600  # (gloo: 30 min; RCCL's default watchdog: 10 min)
    assert max(solo["before_the_training_barrier"], solo["before_the_final_barrier"]) < 0.5 * 600, solo  # half of RCCL's default
    assert wall < 600, wall
    print(f"8 ranks on one GPU: wall {wall:.0f} s, rank-0-only legs {solo}")
