#!/bin/bash
# SQ counters of the training attention kernels (four passes of 4): where their wave-cycles go.
# Usage (through gpurun): bash tools/microbench/train_attention_pmc.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_tattn
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pass$i -o a -- python3 tools/microbench/train_attention_check.py > $out/pass$i.log 2>&1
  echo "pass$i rc=$?"
done
python3 - <<'PY'
import csv, glob, collections
for KERNEL in ('k_tattn_fwd', 'k_tattn_dq', 'k_tattn_dkv'):
 print('====', KERNEL)
 acc = collections.defaultdict(lambda: [0.0, 0])
 for f in glob.glob("gpurun_out/prof_tattn/**/*counter_collection.csv", recursive=True):
     for r in csv.DictReader(open(f)):
         if KERNEL in r["Kernel_Name"]:
             a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
 v = {k: t / n for k, (t, n) in acc.items()}
 for k in sorted(v): print(f"{k:28s} {v[k]:16.1f}")
 w = v.get("SQ_WAVES", 1); wc = v.get("SQ_WAVE_CYCLES", 1)
 print("per wave: VALU", v.get("SQ_INSTS_VALU", 0) / w, "LDS", v.get("SQ_INSTS_LDS", 0) / w, "SALU", v.get("SQ_INSTS_SALU", 0) / w)
 print("fractions of wave-cycles: parked", v.get("SQ_WAIT_ANY", 0) / wc, "issue-stall", v.get("SQ_WAIT_INST_ANY", 0) / wc, "(LDS issue stall", v.get("SQ_WAIT_INST_LDS", 0) / wc, ") issuing", v.get("SQ_ACTIVE_INST_ANY", 0) / wc)
 print("MFMA busy cycles / (busy cycles x 4 SIMD-equivalents?):", v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), v.get("SQ_BUSY_CYCLES", 0))
 print("LDS bank conflict / active:", v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_LDS_IDX_ACTIVE", 0))
PY
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
