"""RCCL executes on the one-GPU box: a process group of ONE rank over ``backend="nccl"`` and the training path's collectives forced on
(training/distributed.py: force_collectives) -- the calls an 8-GPU launch makes (mindmap/model_utils/multi_gpu.py:21-34
init_process_group(backend="nccl"); mindmap/run_training.py:608-613 DistributedDataParallel), here with nobody to exchange with, so
the results must equal the plain step's bit for bit.  The work happens in a fresh child process (tests/rccl_single_rank_child.py)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


@pytest.mark.gpu
def test_forced_collectives_over_rccl_equal_the_plain_step_bit_for_bit():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_single_rank_child.py")], env=_clean_env(), capture_output=True,
                       text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["backend"] == "nccl" and out["observed_world"] == 1
    # captured step: broadcast + scale + all-reduce between the two graphs change nothing, to the bit
    assert out["graphed_losses_equal"] and out["graphed_weights_equal"], out
    assert out["weights_moved_by"] > 1e-4, out
    assert out["allreduce_ms"]["payload_MB"] > 1.0 and out["allreduce_ms"]["min"] > 0.0
    assert out["max_over_ranks_device"] == 1.25 and out["all_gather_objects"] == [{"rank": 0}]
    # reference-shaped step: DistributedDataParallel over the one-rank RCCL group
    assert out["ddp_wrapped"] == "DistributedDataParallel"
    assert out["ddp_max_loss_diff"] <= 1e-5 and out["ddp_max_weight_diff"] <= 1e-5, out


@pytest.mark.gpu
def test_bench_train_leg_under_forced_rccl():
    """`BENCH_FORCE_DIST=1 bench.py --gpus 1 --train-only`: the training leg's barrier-bracketed region, the device-tensor
    max-over-ranks and the timed all-reduce run through RCCL; `rccl_world_observed` comes from a real all-reduce."""
    env = _clean_env()
    env["BENCH_FORCE_DIST"] = "1"
    env["BENCH_HANG_DUMP_S"] = "300"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--train-only", "--train-steps", "4"],
                       env=env, capture_output=True, text=True, timeout=420)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    # the record is the LAST stdout line: RCCL's version banner (buffered C stdio, flushed at exit) used to land behind it
    assert [ln for ln in p.stdout.splitlines() if ln.strip()][-1] == lines[0], p.stdout[-1500:]
    t = json.loads(lines[0])["train"]
    assert t["collective_backend"] == "nccl" and t["collectives_forced_on_one_rank"] is True and t["rccl_world_observed"] == 1
    assert t["allreduce"]["payload_MB"] > 1.0 and t["allreduce"]["mean_ms"] > 0.0 and t["parallelism"] == "single"
