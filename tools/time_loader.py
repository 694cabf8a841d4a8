"""Loader throughput: samples/s of MindmapFrameDataset on the reference's real sample shape
(512x512 rgb + depth PNGs, one UNSAMPLED vertex-feature .zst with 768-channel f16 rows: 10-14 k vertices, ~20 MB), per worker
process and through a torch DataLoader, with the per-stage time inside a worker.  CPU only; the GPU side of the loader
(gpu_unpack) is timed by bench.py's `train.file_fed` leg.

    python tools/time_loader.py [--frames 32] [--workers 0 4 8 20] [--dir /tmp/mm_loader_demo] [--pin]
"""
import argparse
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.data_loading.dataset import MindmapFrameDataset, write_synthetic_demo  # noqa: E402
from nvblox_mindmap_amd.io import dataset_files as D  # noqa: E402
from nvblox_mindmap_amd.io import zstd  # noqa: E402


def stage_breakdown(ds, n=8):
    t = {"png_rgb": 0.0, "png_depth": 0.0, "file_read": 0.0, "zstd": 0.0, "unpickle": 0.0, "whole_getitem": 0.0}
    for i in range(n):
        it = ds.samples[i % len(ds)]
        a = time.perf_counter(); D.read_png(it["pov_rgb"]); b = time.perf_counter(); D.read_png(it["pov_depth"]); c = time.perf_counter()
        raw = open(it["vertex_features"], "rb").read(); d = time.perf_counter(); dec = zstd.decompress(raw); e = time.perf_counter()
        D._TensorUnpickler(io.BytesIO(dec)).load(); f = time.perf_counter()
        ds[i % len(ds)]; g = time.perf_counter()
        for k, v in zip(t, (b - a, c - b, d - c, e - d, f - e, g - f)):
            t[k] += v / n
    return {k: round(v * 1e3, 2) for k, v in t.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--workers", type=int, nargs="+", default=[0, 4, 8, 20])
    ap.add_argument("--dir", default="/tmp/mm_loader_demo")
    ap.add_argument("--feature-dim", type=int, default=768)
    ap.add_argument("--vertices", type=int, nargs=2, default=[10000, 14000])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--pin", action="store_true", help="pin_memory=True (needs a GPU)")
    a = ap.parse_args()
    demo = os.path.join(a.dir, "demo_00000")
    if not os.path.isdir(demo):
        t0 = time.perf_counter()
        write_synthetic_demo(demo, a.frames, image_size=(512, 512), feature_dim=a.feature_dim, ngrippers=2, vertex_count_range=a.vertices)
        print(f"wrote {a.frames} frames in {time.perf_counter() - t0:.1f} s")
    ds = MindmapFrameDataset(a.dir, seed=0)
    size = sum(os.path.getsize(p) for s in ds.samples for p in s.values()) / len(ds)
    frames = list(ds.samples)
    print(f"{len(ds)} samples, {size / 1e6:.2f} MB on disk per sample, host threads {os.cpu_count()}")
    print("per-stage ms (this process):", stage_breakdown(ds))
    for w in a.workers:
        # whole batches go to workers: several batches per worker and epoch, or most workers idle (frames are revisited)
        ds.samples = frames * max(1, -(-3 * max(w, 1) * a.batch // len(frames)))
        dl = torch.utils.data.DataLoader(ds, batch_size=a.batch, num_workers=w, shuffle=True, persistent_workers=w > 0, pin_memory=a.pin,
                                         prefetch_factor=2 if w > 0 else None)
        for i, _ in enumerate(dl):  # page cache + worker start
            if i >= max(w, 1):
                break
        t0 = time.perf_counter()
        n = 0
        for b in dl:
            n += b["rgb_u8"].shape[0]
        dt = time.perf_counter() - t0
        print(f"workers={w}: {n / dt:7.1f} samples/s")
        del dl


if __name__ == "__main__":
    main()
