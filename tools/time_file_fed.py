"""Where does a file-fed training step spend its HOST time?  (bench.py's train.file_fed leg says how much slower than the
resident-batch step it is; this says why.)  Per step: seconds inside next(batch) (waiting for the loader + issuing the copies and
GPU-side transforms), seconds inside train_one_step (issuing the step's kernels), and the GPU time of the step (events).

    python tools/time_file_fed.py [--steps 12] [--workers 20] [--frames 32]
"""
import argparse
import os
import shutil
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--workers", type=int, default=None)
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--raw-cache", action="store_true", help="convert the demo's vertex features to the memory-mapped raw cache first")
    ap.add_argument("--host-threads", type=int, default=0, help="torch.set_num_threads of the training process (0: leave)")
    ap.add_argument("--prefetch", type=int, default=2)
    a = ap.parse_args()
    from torch.utils.data import DataLoader

    from nvblox_mindmap_amd.data_loading.dataset import DevicePrefetcher, MindmapFrameDataset, write_synthetic_demo
    from nvblox_mindmap_amd.diffuser_actor import DiffuserActorConfig
    from nvblox_mindmap_amd.training import build_model, build_optimizer, train_one_step

    dev = torch.device("cuda:0")
    cfg = DiffuserActorConfig()
    workers = a.workers if a.workers is not None else max(1, min(20, (os.cpu_count() or 1) - 2))
    root = tempfile.mkdtemp(prefix="mmf_file_fed_")
    try:
        write_synthetic_demo(os.path.join(root, "demo_00000"), a.frames, image_size=cfg.image_size, feature_dim=cfg.feature_dim,
                             num_history=cfg.num_history, prediction_horizon=cfg.prediction_horizon, ngrippers=cfg.ngrippers,
                             vertex_count_range=(10000, 14000))
        if a.raw_cache:
            from nvblox_mindmap_amd.io import vertex_cache

            vertex_cache.convert_dataset(root)
        if a.host_threads:
            torch.set_num_threads(a.host_threads)
        ds = MindmapFrameDataset(root, num_vertices=2048, use_raw_vertex_cache=a.raw_cache)
        ds.samples = ds.samples * max(1, -(-3 * workers * a.batch // len(ds.samples)))
        dl = DataLoader(ds, batch_size=a.batch, shuffle=True, num_workers=workers, drop_last=True, pin_memory=True,
                        persistent_workers=True, prefetch_factor=a.prefetch)
        torch.manual_seed(0)
        model = build_model(cfg, device=dev)
        opt = build_optimizer(model)

        def batches():
            while True:
                for b in DevicePrefetcher(dl, dev):
                    yield b

        def run(get, label):
            for _ in range(3):
                train_one_step(cfg, model, opt, get())
            torch.cuda.synchronize()
            t_next = t_step = 0.0
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
            t0 = time.perf_counter()
            ev[0].record()
            for i in range(a.steps):
                ta = time.perf_counter()
                b = get()
                tb = time.perf_counter()
                train_one_step(cfg, model, opt, b)
                tc = time.perf_counter()
                ev[i + 1].record()
                t_next += tb - ta
                t_step += tc - tb
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            gpu = [ev[i].elapsed_time(ev[i + 1]) for i in range(a.steps)]
            print(f"{label}: wall {wall / a.steps * 1e3:.1f} ms/step ({a.steps / wall:.2f} step/s) | host: next(batch) {t_next / a.steps * 1e3:.1f} ms, "
                  f"train_one_step issue {t_step / a.steps * 1e3:.1f} ms, all issued after {t_issue / a.steps * 1e3:.1f} ms/step | "
                  f"GPU between step markers: median {sorted(gpu)[len(gpu) // 2]:.1f} ms")

        it = batches()
        run(lambda: next(it), f"file-fed ({workers} workers)")
        resident = next(it)
        run(lambda: resident, "resident batch (same process, loader workers idle)")
        del it
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
