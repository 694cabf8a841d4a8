#!/usr/bin/env python3
"""Which spec setting reproduces nvblox?  Given a file dumped by tools/dump_nvblox_golden.py from upstream nvblox_torch,
replay its stream on this repository's integrator once as specified and once per flipped spec item (the medium / low
confidence items of DESIGN.md section 3, listed in the file's meta as `spec_items`) and print the distance table.

    python tests/pin_report.py tests/golden/nvblox_small_patches.npz            # CPU oracle (no GPU needed)
    python tests/pin_report.py tests/golden/nvblox_bl_patches.npz --backend mmf  # HIP integrator on an MI355X

Test infrastructure (it drives the CPU oracle), hence under tests/."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import nvblox_golden_common as NG  # noqa: E402

COLUMNS = ["tsdf_blocks_missing", "tsdf_blocks_extra", "feature_blocks_missing", "feature_blocks_extra", "tsdf_max_abs_distance_diff",
           "tsdf_max_abs_weight_diff", "feature_max_abs_diff", "vertex_frac_beyond_1mm", "vertex_feature_frac_beyond_0.05"]


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("golden", nargs="?", help="a file dumped by tools/dump_nvblox_golden.py (omit with --self)")
    ap.add_argument("--backend", choices=["oracle", "mmf"], default="oracle")
    ap.add_argument("--self", dest="self_dump", action="store_true",
                    help="SENSITIVITY table: no upstream dump -- the stream is dumped from this repository's own integrator AS SPECIFIED "
                         "(the --backend), then replayed once per flipped item: how far each recollection risk moves the outputs, and "
                         "whether the north star / the reference's own e2e tolerances (mindmap/tests/utils/comparisons.py:95-109) would "
                         "still hold if upstream had the other variant")
    ap.add_argument("--config", default="bl")
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--hole-mode", default="patches")
    args = ap.parse_args(argv)
    make, device = (NG.oracle_backend, "cpu") if args.backend == "oracle" else (NG.mmf_backend, "cuda")
    if args.self_dump:
        kit = NG.load_kit()
        # every vertex (stride 1) and 24 sampled blocks per layer: the reference's tolerances are statements about all vertices
        gold = kit.replay(make(), args.config, args.hole_mode, args.frames, True, True, device=device, n_block_samples=24,
                          n_vertex_samples=1 << 40)
        args.golden = f"(self-dump of the {args.backend} as specified)"
    else:
        gold = np.load(args.golden, allow_pickle=False)
    meta = json.loads(str(gold["meta"]))
    rows = [("as specified", {})]
    by_hand = []
    for it in meta["spec_items"]:
        if it.get("param") is None:
            by_hand.append(f"{it['item']}  ({it.get('note', '')})")
            continue
        for v in it["flips"]:
            rows.append((f"{it['param']}={v:g}", {it["param"]: type(it["ours"])(v)}))
    print(f"{args.golden}: {meta['backend']}, config {meta['config']}, {meta['frames']} frames, holes {meta['hole_mode']}")
    print(f"{'setting':42s} " + " ".join(f"{c[:18]:>18s}" for c in COLUMNS) + "  north_star  ref_tol")
    results = []
    for name, over in rows:
        r = NG.compare(gold, NG.replay_like(gold, make(**over), device))
        results.append((name, r))
        print(f"{name:42s} " + " ".join(f"{r.get(c, float('nan')):18.6g}" for c in COLUMNS)
              + f"  {str(NG.passes_north_star(r)):>10s}  {str(NG.passes_reference_tolerances(r)):>7s}")
    print("\ncode-level spec items (no parameter to flip; check by hand if nothing above matches):")
    for b in by_hand:
        print("  -", b)
    return results


if __name__ == "__main__":
    main()
