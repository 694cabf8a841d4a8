#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box: kernel trace + two PMC passes of the same bench command.
# Usage (through gpurun): bash tools/profile_round.sh r01c   -> gpurun_out/prof_<tag>/...
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
mkdir -p $out
args="bench.py --steps 100 --warmup 10 --cpu-sample 0 --no-profile --no-train --no-infer --no-backproj --no-ref-shape"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python3 $args > $out/trace.log 2>&1
echo "trace rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o $tag -- python3 $args > $out/fetch.log 2>&1
echo "fetch rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o $tag -- python3 $args > $out/write.log 2>&1
echo "write rc=$?"
python3 tools/pmc_summary.py $out/pmc_fetch_write.json $out/fetch $out/write
find $out -name "*kernel_stats.csv" | head -2
find $out -name "*.csv" -size +3M -delete   # per-dispatch traces are large; the summaries are what gets committed
ls -la $out $out/trace 2>/dev/null | head -30
