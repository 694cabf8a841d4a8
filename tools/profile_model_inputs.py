"""The map -> model-input half of the hot path, alone: IsaacLabNvbloxMapper.get_nvblox_model_inputs
(mapping/isaaclab_nvblox_mapper.py:207-250 of the reference: update_feature_mesh / get_feature_mesh / AABB filter /
zero-row filter / sample_to_n_vertices) on a map built from the synthetic stream, and the facade's per-frame fusion call.

    python3 tools/profile_model_inputs.py --shape ref|bl [--iters 50] [--frames 12]

Run it under `rocprofv3 --kernel-trace --stats` (tools/profile_mesh.sh) for the per-kernel table; on its own it prints the
wall-clock per call as one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from nvblox_mindmap_amd.mapping.nvblox_mapper_constants import MAPPER_TO_ID  # noqa: E402


def build(shape: str, device, n_frames: int):
    return bench.build_facade(shape, device, n_frames)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ref", choices=["ref", "bl"])
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--frames", type=int, default=12)
    args = ap.parse_args()
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    cfg, C, frames, samples, ex, facade = build(args.shape, device, args.frames)

    def fuse(i):
        fr, smp = frames[i % len(frames)], samples[i % len(frames)]
        ex.next, ex.low = fr["features"], fr["lowres"]
        facade.decay()
        facade.update_reconstruction_from_sample(smp, "pov")

    for i in range(args.frames):
        fuse(i)
    torch.cuda.synchronize(device)
    import gc

    gc.collect()
    gc.freeze()  # a full collection of the interpreter's heap (tens of ms with torch + scipy loaded) is not the path's cost
    # facade fusion, wall clock per call (synchronised)
    t0 = time.perf_counter()
    per_iter = []
    for i in range(args.iters):
        t1 = time.perf_counter()
        fuse(i)
        per_iter.append((time.perf_counter() - t1) * 1e3)
    torch.cuda.synchronize(device)
    fusion_ms = (time.perf_counter() - t0) / args.iters * 1e3
    per_iter.sort()
    # map -> model inputs
    torch.manual_seed(0)
    out = facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.iters):
        out = facade.get_nvblox_model_inputs(MAPPER_TO_ID.STATIC, remove_zero_features=True)
    torch.cuda.synchronize(device)
    mi_ms = (time.perf_counter() - t0) / args.iters * 1e3
    V = facade.mapper.update_feature_mesh(MAPPER_TO_ID.STATIC)
    n_live = facade.mapper.tsdf_layer_view(MAPPER_TO_ID.STATIC).num_allocated_blocks()
    N = int(out["vertices"].shape[1])
    print(json.dumps({"shape": args.shape, "H": cfg.height, "W": cfg.width, "C": C, "mesh_vertices": V, "live_tsdf_blocks": n_live,
                      "sampled": N, "facade_fusion_ms": fusion_ms, "facade_fusion_host_ms_median_max": [per_iter[len(per_iter) // 2], per_iter[-1]], "map_to_model_input_ms": mi_ms,
                      "algorithmic_bytes": n_live * 512 * 8 + N * (12 + 4 * C) + N * 2 * C,
                      "iters": args.iters}))


if __name__ == "__main__":
    main()
