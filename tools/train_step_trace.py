"""What one policy training step costs ON THE GPU, from a rocprofv3 kernel trace of `bench.py --train-only`
(`rocprofv3 --kernel-trace --output-format csv -d DIR -o NAME -- python3 bench.py --train-only`).

Steps are delimited by the back-projection kernel (one launch per step and camera).  For the last steps of the trace: wall time
per step, the time at least one kernel was running (union of the dispatch intervals), the sum of the kernel durations, the number
of dispatches, the idle time between dispatches by gap size, and the kernels that carry the time.
Usage: python tools/train_step_trace.py DIR/NAME_kernel_trace.csv [steps] > profiles/rNN_train_step_trace.txt"""
import csv
import sys
from collections import defaultdict


def main(path, last=8):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [s for s, _, n in rows if "k_backproject" in n]
    if len(marks) < last + 1:
        raise SystemExit(f"only {len(marks)} steps in the trace")
    t0, t1 = marks[-last - 1], marks[-1]
    win = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
    wall = (t1 - t0) / last
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, _ in win:
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            gaps.append(s - cur_e)
            cur_s, cur_e = s, e
    busy += cur_e - cur_s
    tot = sum(e - s for s, e, _ in win)
    print(f"steps analysed: {last}; dispatches per step: {len(win) / last:.0f}")
    print(f"wall per step            {wall / 1e6:9.3f} ms")
    print(f"  >= 1 kernel running    {busy / last / 1e6:9.3f} ms")
    print(f"  nothing running        {(t1 - t0 - busy) / last / 1e6:9.3f} ms in {len(gaps) / last:.0f} gaps per step")
    print(f"sum of kernel durations  {tot / last / 1e6:9.3f} ms (overlapping branches count twice)")
    for lo, hi in ((0, 2000), (2000, 5000), (5000, 10000), (10000, 50000), (50000, 10 ** 12)):
        g = [x for x in gaps if lo <= x < hi]
        print(f"  gaps {lo / 1e3:5.0f}-{hi / 1e3 if hi < 10 ** 9 else float('inf'):5.0f} us: {len(g) / last:7.0f} per step, {sum(g) / last / 1e6:7.3f} ms")
    by = defaultdict(lambda: [0, 0])
    for s, e, n in win:
        by[n][0] += 1
        by[n][1] += e - s
    short = [(e - s) for s, e, _ in win if e - s < 10000]
    print(f"dispatches shorter than 10 us: {len(short) / last:.0f} per step, {sum(short) / last / 1e6:.3f} ms")
    print("ms/step  calls/step  avg_us  kernel")
    for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:45]:
        print(f"{t / last / 1e6:7.3f} {c / last:9.1f} {t / c / 1e3:8.1f}  {n[:150]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8)
