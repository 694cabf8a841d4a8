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
