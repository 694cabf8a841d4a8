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
