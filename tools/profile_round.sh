#!/bin/bash
# Collects the per-round rocprofv3 evidence on the GPU box: kernel trace + two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate
# passes, kernel-trace/stats only, as the guide prescribes) of the same bench command, for the benchmark shape (640x480x64:
# `bench.py --only-fusion`) and for the reference's shape (512x512x768: `bench.py --ref-shape-only`).
# Usage (through gpurun): bash tools/profile_round.sh r02a [bl|ref|both|unbounded]   -> gpurun_out/prof_<tag>/{bl,ref}/...
set -u
tag=${1:-rXX}
which=${2:-both}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run_passes() {  # $1 = sub-directory, rest = bench arguments
  local sub=$1; shift
  local out=gpurun_out/prof_$tag/$sub
  mkdir -p $out
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python3 bench.py "$@" > $out/trace.log 2>&1
  echo "$sub trace rc=$?"
  timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o $tag -- python3 bench.py "$@" > $out/fetch.log 2>&1
  echo "$sub fetch rc=$?"
  timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o $tag -- python3 bench.py "$@" > $out/write.log 2>&1
  echo "$sub write rc=$?"
  python3 tools/pmc_summary.py $out/pmc_fetch_write.json "$tag $sub: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of 'bench.py $*'" $out/fetch $out/write
  cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/kernel_stats.csv 2>/dev/null
  find $out -name "*.csv" -size +3M -delete   # per-dispatch traces are large; the summaries are what gets committed
  find $out -name "*.db" -delete
}
if [ "$which" = "bl" ] || [ "$which" = "both" ]; then
  run_passes bl --only-fusion --no-profile --steps 400 --warmup 10 --repeats 1
fi
if [ "$which" = "ref" ] || [ "$which" = "both" ]; then
  run_passes ref --ref-shape-only
fi
if [ "$which" = "unbounded" ]; then  # the hash path: same stream, workspace_bounds_type = kUnbounded
  run_passes unbounded --unbounded-only --steps 100 --warmup 60
fi
ls -la gpurun_out/prof_$tag/* | head -40
