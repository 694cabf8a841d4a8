"""Per-kernel averages of the SQ counters collected by tools/profile_sq.sh + the derived issue / latency fractions.
Usage: python tools/sq_summary.py out.json "<source label>" <dir>"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def _source_hash():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from nvblox_mindmap_amd._lib import source_hash

    return source_hash()


def short(name):
    name = re.sub(r"\(.*$", "", name).replace("void ", "").strip().replace(".kd", "")
    base = re.sub(r"<.*$", "", name).split("::")[-1]
    targs = re.search(r"<(.*)>$", name)
    return f"{base}<{targs.group(1)}>" if (targs and base.startswith("k_")) else base


def main(out, source, root):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                if not k.startswith("k_"):
                    continue
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {}
    for k, cs in sorted(acc.items()):
        v = {c: tot / n for c, (tot, n) in cs.items()}
        d = dict(v)
        d["dispatches"] = max(n for _, n in cs.values())
        waves, wc = v.get("SQ_WAVES"), v.get("SQ_WAVE_CYCLES")
        if waves:
            d["valu_per_wave"] = v.get("SQ_INSTS_VALU", 0.0) / waves
            d["salu_per_wave"] = v.get("SQ_INSTS_SALU", 0.0) / waves
            d["vmem_rd_per_wave"] = v.get("SQ_INSTS_VMEM_RD", 0.0) / waves
            d["lds_per_wave"] = v.get("SQ_INSTS_LDS", 0.0) / waves
        if wc:
            d["frac_parked"] = v.get("SQ_WAIT_ANY", 0.0) / wc
            d["frac_issuing"] = v.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
            d["frac_issue_stall"] = v.get("SQ_WAIT_INST_ANY", 0.0) / wc
            d["frac_issuing_valu"] = v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            d["frac_issuing_scalar"] = v.get("SQ_ACTIVE_INST_SCA", 0.0) / wc
            if v.get("SQ_BUSY_CYCLES"):
                d["wave_cycles_per_busy_cycle"] = wc / v["SQ_BUSY_CYCLES"]
        res[k] = d
    res["__source__"] = source
    res["__csrc_sha16__"] = _source_hash()  # the native sources these counters were collected on (bench.py: counters_stale)
    res["__units__"] = "SQ_*_CYCLES / WAIT / ACTIVE counters count quad-cycles summed over waves (MI355X_MICROARCH.md); fractions are of SQ_WAVE_CYCLES"
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, d in res.items():
        if isinstance(d, dict) and "frac_parked" in d:
            print(f"{k:60s} valu/wave {d.get('valu_per_wave', 0):7.0f} salu/wave {d.get('salu_per_wave', 0):6.0f}  parked {d['frac_parked']:.2f} "
                  f"issuing {d['frac_issuing']:.2f} issue-stall {d['frac_issue_stall']:.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3])
