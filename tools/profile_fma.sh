#!/bin/bash
# The spec switch fma_contraction measured: kernel durations (rocprofv3 --kernel-trace --stats) and SQ instruction
# counts (one --pmc pass, kernel trace only) of the two VALU-bound kernels -- k_feature_flat<LOW> at the reference shape
# (bench.py --ref-shape-only) and k_tsdf_pass<LAZY> on the hash path (bench.py --unbounded-only) -- with the switch off and on
# (MMF_FMA_CONTRACTION: the default of every mapper of the process).  Usage (gpurun): bash tools/profile_fma.sh r05b
set -u
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_${tag}_fma
mkdir -p $out
for fma in 0 1; do
  export MMF_FMA_CONTRACTION=$fma
  for leg in ref unbounded; do
    if [ $leg = ref ]; then args="--ref-shape-only"; else args="--unbounded-only --steps 100 --warmup 60"; fi
    d=$out/${leg}_fma$fma
    mkdir -p $d
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python3 bench.py $args > $d/trace.log 2>&1
    echo "$leg fma=$fma trace rc=$?"
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $d/sq -o sq -- python3 bench.py $args > $d/sq.log 2>&1
    echo "$leg fma=$fma sq rc=$?"
  done
done
unset MMF_FMA_CONTRACTION
python3 - "$out" <<'PY'
import csv, glob, json, os, re, sys
from collections import defaultdict
out = sys.argv[1]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*_fma[01]"))):
    name = os.path.basename(d)
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace("mmf::", "")
            a = dur[k]; a[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3; a[1] += 1
    cnt = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(d, "sq", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace("mmf::", "")
            a = cnt[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    rows = {}
    for k, (t, n) in dur.items():
        if not any(s in k for s in ("k_feature_flat", "k_tsdf_pass", "k_app_frame", "k_tsdf_classify")):
            continue
        c = {c_: v[0] / v[1] for c_, v in cnt.get(k, {}).items()}
        w = c.get("SQ_WAVES")
        rows[k] = {"avg_us": t / n, "dispatches": n, "waves": w, "valu_per_wave": (c.get("SQ_INSTS_VALU", 0) / w) if w else None,
                   "salu_per_wave": (c.get("SQ_INSTS_SALU", 0) / w) if w else None, "valu_insts": c.get("SQ_INSTS_VALU")}
    res[name] = rows
sys.path.insert(0, os.getcwd())
from nvblox_mindmap_amd._lib import source_hash
res["__csrc_sha16__"] = source_hash()
res["__source__"] = "tools/profile_fma.sh: rocprofv3 --kernel-trace --stats and --pmc SQ_* passes of bench.py --ref-shape-only / --unbounded-only with MMF_FMA_CONTRACTION=0|1"
json.dump(res, open(os.path.join(out, "fma_summary.json"), "w"), indent=1)
for name, rows in res.items():
    if name.startswith("__"): continue
    for k, r in sorted(rows.items()):
        print(f"{name:16s} {k[:60]:60s} {r['avg_us']:8.2f} us  x{r['dispatches']:<5d} valu/wave {r['valu_per_wave'] and round(r['valu_per_wave'])}")
PY
find $out -name "*.csv" -size +3M -delete
find $out -name "*.db" -delete
