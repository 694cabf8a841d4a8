"""Per-denoising-step kernel breakdown from a rocprofv3 kernel trace of tools/trace_fused_inference.py:
    python tools/infer_trace_summary.py gpurun_out/prof_infer7/infer_kernel_trace.csv"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_head_outputs" in r["Kernel_Name"] or "k_ddpm_step" in r["Kernel_Name"]]
N = 50
win = rows[idx[-N - 1] + 1: idx[-1] + 1]  # the last N denoising steps (graph replay)
t0, t1 = int(win[0]["Start_Timestamp"]), int(win[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
print(f"{(t1 - t0) / 1e3 / N:.1f} us per step, {len(win) / N:.1f} kernels per step, summed kernel time {busy / 1e3 / N:.1f} us per step")
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = re.sub(r"\(.*", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)[:90]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"{t / 1e3 / N:8.2f} us/step {c / N:6.1f} x {t / c / 1e3:7.2f} us  {n}")
