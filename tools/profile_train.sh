cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof_r04zb_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04zb_train -o tr -- python3 bench.py --train-only --train-steps 12 > gpurun_out/prof_r04zb_train/log.txt 2>&1
echo rc=$?
f=$(find gpurun_out/prof_r04zb_train -name "*kernel_trace.csv" | head -1)
python3 tools/train_step_trace.py "$f" 8 > gpurun_out/r04zb_train_step_trace.txt 2>&1
cp "$(find gpurun_out/prof_r04zb_train -name '*kernel_stats.csv' | head -1)" gpurun_out/r04zb_train_kernel_stats.csv
find gpurun_out/prof_r04zb_train -name "*.csv" -size +3M -delete; find gpurun_out/prof_r04zb_train -name "*.db" -delete
head -12 gpurun_out/r04zb_train_step_trace.txt
python3 tools/train_op_profile.py 60 > gpurun_out/r04zb_train_op_profile.txt 2>/dev/null
