"""Loader throughput: samples/s of MindmapFrameDataset on the reference's real sample shape
(512x512 rgb + depth PNGs, one vertex-feature .zst with 768-channel f16 rows), per worker process and through a
torch DataLoader.  CPU only; the GPU side of the loader (gpu_unpack) is timed by tests/bench on the GPU box.

    python tools/time_loader.py [--frames 64] [--workers 0 2 4 8] [--dir /tmp/mm_loader_demo]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nvblox_mindmap_amd.data_loading.dataset import MindmapFrameDataset, write_synthetic_demo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--workers", type=int, nargs="+", default=[0, 2, 4, 8])
    ap.add_argument("--dir", default="/tmp/mm_loader_demo")
    ap.add_argument("--feature-dim", type=int, default=768)
    a = ap.parse_args()
    demo = os.path.join(a.dir, "demo_00000")
    if not os.path.isdir(demo):
        t0 = time.perf_counter()
        write_synthetic_demo(demo, a.frames, image_size=(512, 512), feature_dim=a.feature_dim)
        print(f"wrote {a.frames} frames in {time.perf_counter() - t0:.1f} s")
    ds = MindmapFrameDataset(a.dir, seed=0)
    size = sum(os.path.getsize(p) for s in ds.samples for p in s.values()) / len(ds)
    print(f"{len(ds)} samples, {size / 1e6:.2f} MB on disk per sample")
    for w in a.workers:
        dl = torch.utils.data.DataLoader(ds, batch_size=8, num_workers=w, shuffle=False, persistent_workers=False)
        for _ in dl:  # page cache + worker start
            break
        t0 = time.perf_counter()
        n = 0
        for _ in range(2):
            for b in dl:
                n += b["rgb_u8"].shape[0]
        dt = time.perf_counter() - t0
        print(f"workers={w}: {n / dt:7.1f} samples/s")


if __name__ == "__main__":
    main()
