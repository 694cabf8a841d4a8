"""Average PMC counter value per dispatch and kernel from rocprofv3 `--pmc X --output-format csv` runs.
Usage: python tools/pmc_summary.py out.json "<source label>" <dir_or_csv> [<dir_or_csv> ...]
The label (which run the numbers come from) is stored under "__source__"; bench.py quotes it as `roofline.traffic_source`.
Kernel names are shortened to the function name (no namespace, template arguments or parameters).  FETCH_SIZE / WRITE_SIZE
are reported in KB as the counters deliver them (gfx950: double FETCH_SIZE before comparing with byte counts, see
/opt/skills/guides/MI355X_MICROARCH.md)."""
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
    name = re.sub(r"\(.*$", "", name)          # parameters
    name = name.replace("void ", "").strip().replace(".kd", "")
    base = re.sub(r"<.*$", "", name).split("::")[-1]
    targs = re.search(r"<(.*)>$", name)
    # libmmfusion kernels keep their template arguments (k_feature_flat<false> reads the materialised image, <true> the low-res
    # map: different traffic); library kernels (at::native::...) are shortened to the function name
    return f"{base}<{targs.group(1)}>" if (targs and base.startswith("k_")) else base


def main(out, source, paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in paths:
        files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    a = acc[k][c]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    res = {}
    for k, cs in sorted(acc.items()):
        res[k] = {}
        for c, (tot, n) in cs.items():
            res[k][c + ("_KB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = tot / n
            res[k]["dispatches"] = n
    res["__source__"] = source
    res["__csrc_sha16__"] = _source_hash()  # the native sources these counters were collected on (bench.py: counters_stale)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    print(f"{len(res)} kernels -> {out}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
