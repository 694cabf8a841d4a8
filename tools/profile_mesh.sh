#!/bin/bash
# rocprofv3 evidence for the map -> model-input half of the path (tools/profile_model_inputs.py): kernel trace + separate
# FETCH_SIZE / WRITE_SIZE passes, at the reference's shape (512x512x768) and the benchmark shape (640x480x64).
# Usage (through gpurun): bash tools/profile_mesh.sh r03a   -> gpurun_out/prof_<tag>/mesh_{ref,bl}/...
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for shape in ref bl; do
  out=gpurun_out/prof_$tag/mesh_$shape
  mkdir -p $out
  python3 tools/profile_model_inputs.py --shape $shape > $out/wall.json 2> $out/wall.err
  cat $out/wall.json
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o $tag -- python3 tools/profile_model_inputs.py --shape $shape > $out/trace.log 2>&1
  echo "$shape trace rc=$?"
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o $tag -- python3 tools/profile_model_inputs.py --shape $shape > $out/fetch.log 2>&1
  echo "$shape fetch rc=$?"
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o $tag -- python3 tools/profile_model_inputs.py --shape $shape > $out/write.log 2>&1
  echo "$shape write rc=$?"
  python3 tools/pmc_summary.py $out/pmc_fetch_write.json "$tag mesh_$shape: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of 'tools/profile_model_inputs.py --shape $shape'" $out/fetch $out/write
  cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" $out/kernel_stats.csv 2>/dev/null
  find $out -name "*.csv" -size +3M -delete
  find $out -name "*.db" -delete
done
