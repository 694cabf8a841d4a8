"""Register / LDS / scratch usage of every kernel of one csrc/*.hip file, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles without a GPU).  Usage: python tools/kernel_resources.py mmf_kernels_map.hip [name filter] > before.txt"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "nvblox_mindmap_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math".split()


def main(src, flt=""):
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                         cwd=CSRC, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = cur.replace("(anonymous namespace)::", "")  # (kernels in unnamed namespaces: keep the name, drop the tag)
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
            continue
        m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split(" ")[0]] = int(m.group(2))
    # `adm`: waves per SIMD the CU ADMITS -- min(8, floor(512 / ceil(vgpr / 8) * 8), floor(800 / (ceil(sgpr / 16) * 16 + 16))) (MI355X_MICROARCH.md,
    # "Residency"; LDS not counted: it depends on the workgroup size).  The compiler's `occ` ignores the scalar-register rule: a kernel at
    # 101-106 SGPRs shows 7 and runs at 6.
    print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s} {'adm':>4s}")
    for k, r in sorted(rows.items()):
        if flt in k:
            vg = max(-(-(r.get('VGPRs', 0) + r.get('AGPRs', 0)) // 8) * 8, 8)
            sg = -(-max(r.get('TotalSGPRs', 0), 1) // 16) * 16 + 16
            adm = min(8, 512 // vg, 800 // sg)
            print(f"{k[:70]:70s} {r.get('VGPRs', 0):5d} {r.get('AGPRs', 0):5d} {r.get('TotalSGPRs', 0):5d} {r.get('ScratchSize', 0):8d} {r.get('LDS', 0):7d} "
                  f"{r.get('Occupancy', 0):4d} {adm:4d}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
