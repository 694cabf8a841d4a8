"""Per-kernel statistics (calls, total, average, min, max duration) from a rocprofv3 rocpd database
(`rocprofv3 --kernel-trace -d DIR -o NAME` writes NAME_results.db), as CSV on stdout.
Usage: python tools/rocpd_kernel_stats.py gpurun_out/prof/NAME_results.db > profiles/rNN_kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,ArchVGPR,AccumVGPR,SGPR,LDSBytes")
    for r in rows:
        name = r[0].replace('"', "'")
        print(f'"{name}",{r[1]},{r[2]},{r[3]:.1f},{100.0 * r[2] / total:.2f},{r[4]},{r[5]},{r[6]},{r[7]},{r[8]},{r[9]}')


if __name__ == "__main__":
    main(sys.argv[1])
