cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_sq
mkdir -p $out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_INSTS_LDS" "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -o sq -- python3 bench.py --only-fusion --no-profile --steps 60 --warmup 10 --repeats 1 --cpu-sample 0 > $out/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/prof_sq/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").split("::")[-1]
        k = re.sub(r"<.*$", "", k)
        if not k.startswith("k_"): continue
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
PY
