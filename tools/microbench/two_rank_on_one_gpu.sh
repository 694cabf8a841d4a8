export BENCH_DIST_BACKEND=gloo BENCH_HANG_DUMP_S=150
echo "== train only"
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --train-only > gpurun_out/r04_two_rank_train.log 2> gpurun_out/r04_two_rank_train.err; echo rc=$?
echo "== no train"
timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 20 --warmup 5 --no-train > gpurun_out/r04_two_rank_fusion.log 2> gpurun_out/r04_two_rank_fusion.err; echo rc=$?
