# numerics + timing of the training attention kernels, then their per-kernel durations (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 tools/microbench/train_attention_check.py 2>&1 | grep -v amdgpu | tail -8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ta -o ta -- python3 tools/microbench/train_attention_check.py > gpurun_out/ta.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ta/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("tattn", "attn_fwd", "bwd_kernel")):
        print(r["Name"][:60], r["Calls"], "avg us %.1f max us %.1f" % (float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
find gpurun_out/ta -name "*.csv" -size +1M -delete; find gpurun_out/ta -name "*.db" -delete
