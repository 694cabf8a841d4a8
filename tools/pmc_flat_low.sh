# SQ / cache counters of the reference-shape leg (512x512x768), kept apart by template argument: k_feature_flat<true> is the
# low-res-sampling row update, <false> the materialised one.  Run on the GPU box: bash tools/pmc_flat_low.sh
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_flat_low
mkdir -p $out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/$tag -o sq -- python3 bench.py --ref-shape-only > $out/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/prof_flat_low/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("void ", "")
        if "k_feature_flat" not in k and "k_app_frame" not in k: continue
        k = re.sub(r"\(.*$", "", k).split("::")[-1]
        a = acc[k][row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, cs in sorted(acc.items()):
    print(k, {c: round(v[0] / v[1]) for c, v in sorted(cs.items())})
PY
