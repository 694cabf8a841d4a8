#!/bin/bash
# A/B of library builds on the headline stream: per-kernel average durations (rocprofv3 --kernel-trace --stats) of
# `bench.py --only-fusion --no-profile` for each MMF_LIB given, and the bench's own frames/s.  Usage (gpurun):
#   bash tools/ab_kernel_times.sh <tag> libmmfusion.so libmmfusion_v1.so [-- extra bench args]
set -u
tag=$1; shift
libs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done
[ $# -gt 0 ] && shift
extra="$*"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/ab_$tag
mkdir -p $out
for lib in "${libs[@]}"; do
  export MMF_LIB=$lib
  d=$out/${lib%.so}
  mkdir -p $d
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python3 bench.py --only-fusion --no-profile --steps 400 --warmup 20 --repeats 1 $extra > $d/trace.log 2>&1
  echo "$lib trace rc=$?"
  for i in 1 2 3; do
    python3 bench.py --only-fusion --no-profile $extra 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$lib', round(d['value']), 'frames/s', round(d['ms_per_step']*1e3,2), 'us; undeferred', round(d['roofline']['legs'].get('undeferred_fps',0)))"
  done
  python3 - "$d" "$lib" <<'PY'
import csv, glob, sys
d, lib = sys.argv[1], sys.argv[2]
for f in glob.glob(d + "/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        n = row["Name"].replace("void ", "").replace("mmf::", "")
        if n.startswith("k_"):
            print(f"  {lib:24s} {n[:70]:70s} calls {row['Calls']:>6s} avg {float(row['AverageNs'])/1e3:8.2f} us")
PY
done
unset MMF_LIB
find $out -name "*.csv" -size +3M -delete; find $out -name "*.db" -delete
