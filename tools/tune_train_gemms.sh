#!/bin/bash
# Regenerates nvblox_mindmap_amd/training/tunableop_gfx950.csv on an MI355X: PyTorch's TunableOp times every hipBLASLt / rocBLAS
# solution for each GEMM shape of the policy's training step (batch 32, the bench's model) and records the fastest.
# Usage (through gpurun): bash tools/tune_train_gemms.sh   -> gpurun_out/tunableop_gfx9500.csv (copy it over the tracked file)
set -u
cd "$GRAFT_REPO_ROOT"
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=gpurun_out/tunableop_gfx950.csv
export PYTORCH_TUNABLEOP_MAX_TUNING_ITERATIONS=30 PYTORCH_TUNABLEOP_MAX_WARMUP_ITERATIONS=2
BENCH_TRAIN_OVERLAP=0 timeout 2000 python3 bench.py --train-only --train-steps 8
ls -la gpurun_out/tunableop_gfx950*
