#!/bin/bash
# SQ (shader sequencer) counters of the fused frame's launches: evidence for the NON-HBM bounds the bench line claims
# (instruction issue / latency).  Separate rocprofv3 passes (8 SQ counters fit one pass at most; 4 per pass here), kernel trace
# only, as the guide prescribes.  Usage (through gpurun): bash tools/profile_sq.sh r03b -> gpurun_out/prof_<tag>/sq/sq_summary.json
# What the summary derives per kernel (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint):
#   valu_per_wave / salu_per_wave     instructions issued per wave
#   frac_parked                       SQ_WAIT_ANY / SQ_WAVE_CYCLES        waves parked in s_waitcnt / barriers   (latency)
#   frac_issuing                      SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES waves issuing
#   frac_issue_stall                  SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   waves ready but waiting for an issue slot (issue-bound)
#   waves_per_simd_avg                SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / (4 SIMDs x 256 CUs / ... ) -- occupancy while busy
set -u
tag=${1:-rXX}
shift || true
args=${*:---only-fusion --no-profile --steps 120 --warmup 20 --repeats 1 --cpu-sample 0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag/sq
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -o sq -- python3 bench.py $args > $out/pass$i.log 2>&1
  echo "pass$i ($set) rc=$?"
done
python3 tools/sq_summary.py $out/sq_summary.json "$tag: rocprofv3 --pmc SQ_* passes of 'bench.py $args'" $out
find $out -name "*.csv" -size +3M -delete
find $out -name "*.db" -delete
