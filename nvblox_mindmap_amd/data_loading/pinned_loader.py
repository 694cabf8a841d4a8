"""Training batches assembled in place: a small thread pool writes every sample straight into its row of a preallocated,
pinned batch buffer ("must keep the loader off the critical path"; section 8(f) N4).

The reference feeds ``run_training`` from a torch ``DataLoader`` with 20 worker processes per GPU
(mindmap/data_loading/dataset.py:410-490, batching.py:213-262; cli/args.py: ``num_workers``): every sample is decoded into its
own tensors, ``default_collate`` copies them into a batch in shared memory, the pin-memory thread copies the batch again.  On
this machine the captured training step consumes 830 samples/s per GPU; that arrangement costs ~5 ms of CPU per sample --
8 GPUs x 830 x 5 ms = 33 cores against the 16 a training container gets.  Here:

  * ``slots`` batch buffers ``{key: [B, ...]}`` in pinned host memory, allocated once;
  * ``threads`` worker threads (not processes: no pickling, no shared-memory hand-over; the two reads of a sample are C calls
    that drop the interpreter lock -- ``mmf_host_read_file_at``, ``mmf_host_sample_vertex_file``, csrc/mmf_host_io.hip) fill
    rows: the image pixel blocks are ``pread`` from their raw copies (io/vertex_cache.py) into the row, the selected feature
    rows are copied from the mapped raw vertex file into the row.  ONE copy of every byte, page cache -> pinned buffer;
  * the small per-frame items (poses, intrinsics, gripper states: < 1 KB) are parsed once and kept;
  * a batch is handed out as views of its slot; the slot returns to the pool when the consumer releases it
    (``DevicePrefetcher``: behind the event that follows its host -> device copies).

The samples are the ones ``MindmapFrameDataset.__getitem__`` returns (same selection draws for a given seed, same values:
tests/test_cpu_pinned_loader.py); frames without fresh raw copies, or datasets with geometry augmentation, take that path and
are copied into the row (counted: ``stats()["slow_path_samples"]``).
"""
import ctypes as C
import os
import queue
import threading
import time
from typing import Dict, Iterator, List, Optional

import numpy as np
import torch

from .. import _lib
from ..io import vertex_cache as VC
from .dataset import MindmapFrameDataset
from .vertex_sampling import VertexSamplingMethod


class _Slow(Exception):
    """this sample cannot take the in-place path (no fresh raw copy, unsupported option): use ``__getitem__``."""


class HostBatch(dict):
    """``{key: tensor view of a slot}``; valid until released (``PinnedBatchLoader.release``) or, when never released explicitly,
    until the batch after the next one is requested."""
    slot: int = -1


class PinnedBatchLoader:
    def __init__(self, dataset: MindmapFrameDataset, batch_size: int, shuffle: bool = True, drop_last: bool = True, threads: int = 3,
                 slots: int = 4, seed: int = 0, pin_memory: Optional[bool] = None, rank: int = 0, world_size: int = 1):
        assert batch_size > 0 and threads > 0 and slots >= 2
        self.ds, self.B, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.n_threads, self.n_slots, self.seed, self.rank, self.world = int(threads), int(slots), int(seed), int(rank), int(world_size)
        self.epoch = 0
        pin = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        # shapes and dtypes of a sample (the reference path), drawn from a generator of our own: building a loader inside the training
        # process must not reseed / advance the generators the trainer draws from
        import random as _random

        first = dataset.get(0, generator=torch.Generator().manual_seed(int(seed)), pyrandom=_random.Random(int(seed)))
        self.keys = list(first.keys())
        self.buffers: List[Dict[str, torch.Tensor]] = []
        for _ in range(self.n_slots):
            self.buffers.append({k: torch.empty((self.B,) + tuple(v.shape), dtype=v.dtype, pin_memory=pin) for k, v in first.items()})
        self._np = [{k: (t.view(torch.uint8).numpy() if t.dtype == torch.bool else t.numpy()) for k, t in b.items()} for b in self.buffers]
        self.pinned = pin
        self._lib = _lib.lib()
        self._meta: Dict[int, object] = {}
        self._slow_lock = threading.Lock()  # the dataset's augmentor keeps the current sample's transform in ITS state: one sample at a time
        self._cv = threading.Condition()
        self._tasks: "queue.SimpleQueue" = queue.SimpleQueue()
        self._done = [0] * self.n_slots          # rows filled per slot
        self._outstanding = 0                    # tasks issued and not finished
        self._errors: List[BaseException] = []
        self._stats = {"samples": 0, "slow_path_samples": 0, "cpu_s": 0.0}
        self._threads = [threading.Thread(target=self._worker, args=(i,), daemon=True, name=f"mmf-loader-{i}") for i in range(self.n_threads)]
        for t in self._threads:
            t.start()
        self._free: List[int] = list(range(self.n_slots))
        self._pending_release: List[tuple] = []  # (slot, event) waiting for the device copies
        self._closed = False

    # ---- sample -> row ---------------------------------------------------------------------------------------------------------
    def _build_meta(self, idx: int):
        ds = self.ds
        if not ds.use_raw_vertex_cache or ds.augmentor is not None or ds.noiser is not None:
            raise _Slow()
        if ds.with_vertex_features and ds.method not in (VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, VertexSamplingMethod.RANDOM_WITH_REPLACEMENT):
            raise _Slow()
        it = ds.samples[idx]
        m = {"images": [], "small": {}}
        try:
            for cam in ds.cameras:
                for kind, item in (("rgb", 1), ("depth", 2)):
                    png = it[f"{cam}_{kind}"]
                    raw = png + ".raw"
                    if not os.path.exists(raw):
                        raise _Slow()
                    H, W, Cc, isz, at = VC.raw_image_header(raw, png)
                    if isz != item:
                        raise _Slow()
                    m["images"].append((raw.encode(), at, H * W * max(Cc, 1) * isz))
            if ds.with_vertex_features:
                raw = VC.raw_path_of(it["vertex_features"])
                if not os.path.exists(raw):
                    raise _Slow()
                m["vertex"] = (raw.encode(),) + tuple(VC.raw_header(raw, it["vertex_features"]))
        except VC.StaleRawCopy:
            raise _Slow()
        ncam = len(ds.cameras)
        m["small"]["camera_poses"] = np.stack([np.load(it[f"{c}_pose"]).astype(np.float32) for c in ds.cameras])
        m["small"]["intrinsics"] = np.stack([np.load(it[f"{c}_intrinsics"]).astype(np.float32) for c in ds.cameras])
        m["small"]["gripper_history"] = np.load(it["gripper_history"]).astype(np.float32)
        m["small"]["gt_gripper_pred"] = np.load(it["gt_gripper_pred"]).astype(np.float32)
        if "gt_head_yaw" in it:
            m["small"]["gt_head_yaw"] = np.load(it["gt_head_yaw"]).astype(np.float32)
        assert m["small"]["camera_poses"].shape[0] == ncam
        return m

    def _fill(self, idx: int, slot: int, row: int, tls) -> None:
        out = self._np[slot]
        m = self._meta.get(idx)
        if m is None:
            try:
                m = self._build_meta(idx)
            except _Slow:
                m = False
            self._meta[idx] = m
        if m is False:
            with self._slow_lock:
                # this thread's own generators (never torch.manual_seed / the process-wide ones: the trainer's noise comes from those)
                s = self.ds.get(idx, generator=tls["gen"], pyrandom=tls["pyrandom"])
            for k in self.keys:
                self.buffers[slot][k][row].copy_(s[k])
            tls["slow"] += 1
            return
        ds, L = self.ds, self._lib
        ncam = len(ds.cameras)
        for c in range(ncam):
            for j, key in enumerate(("rgb_u8", "depth_mm")):
                path, at, nbytes = m["images"][2 * c + j]
                dst = out[key][row, c]
                if dst.nbytes != nbytes:
                    raise ValueError(f"{path.decode()}: {nbytes} bytes of pixels, the batch buffer's row holds {dst.nbytes}")
                _lib.check(L.mmf_host_read_file_at(path, at, dst.ctypes.data, nbytes), "mmf_host_read_file_at")
        for k, v in m["small"].items():
            out[k][row] = v
        if ds.with_vertex_features:
            path, V, Cf, off_v, off_f = m["vertex"]
            want = ds.num_vertices
            if Cf != out["vertex_features"].shape[2]:
                raise ValueError(f"{path.decode()}: {Cf} feature channels, the batch buffer holds {out['vertex_features'].shape[2]}")
            gen = tls["gen"]
            if ds.seed is not None:  # the dataset's own rule: the draw of sample idx is seeded with seed + idx (dataset.py __getitem__)
                gen.manual_seed(ds.seed + idx)
            if V > want:
                if ds.method == VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT:
                    state = gen.get_state()
                    rows = tls["rows"]
                    if L.mmf_host_randperm_prefix(state.data_ptr(), state.numel(), V, want, rows.data_ptr()) != 0:
                        rows = torch.randperm(V, generator=gen)[:want].contiguous()
                    else:
                        gen.set_state(state)
                else:
                    rows = torch.randint(0, V, (want,), generator=gen)
                n_take = want
            else:
                rows = torch.arange(V, dtype=torch.int64)
                n_take = V
            vs = tls["verts16"]
            _lib.check(L.mmf_host_sample_vertex_file(path, off_v, off_f, V, Cf, rows.data_ptr(), n_take, vs.ctypes.data,
                                                     out["vertex_features"][row].ctypes.data), "mmf_host_sample_vertex_file")
            out["vertices"][row, :n_take] = vs[:n_take]  # float16 -> float32 (exact), as vertices[sel].to(float32)
            out["vertices_valid_mask"][row, :n_take] = 1
            if n_take < want:
                out["vertices"][row, n_take:] = 0
                out["vertex_features"][row, n_take:] = 0
                out["vertices_valid_mask"][row, n_take:] = 0

    def _worker(self, tid: int) -> None:
        torch.set_num_threads(1)
        gen = torch.Generator()
        gen.manual_seed((self.seed * 1000003 + 7919 * (tid + 1) + 104729 * self.rank) & 0x7FFFFFFF)
        import random as _random

        tls = {"gen": gen, "pyrandom": _random.Random(int(gen.initial_seed())), "rows": torch.empty(max(self.ds.num_vertices, 1), dtype=torch.int64),
               "verts16": np.empty((max(self.ds.num_vertices, 1), 3), dtype=np.float16), "slow": 0}
        while True:
            task = self._tasks.get()
            if task is None:
                return
            idx, slot, row = task
            c0 = time.thread_time()
            slow0 = tls["slow"]
            try:
                self._fill(idx, slot, row, tls)
            except BaseException as e:  # surfaced by the consumer's next()
                with self._cv:
                    self._errors.append(e)
            dt = time.thread_time() - c0
            with self._cv:
                self._done[slot] += 1
                self._outstanding -= 1
                self._stats["samples"] += 1
                self._stats["cpu_s"] += dt
                self._stats["slow_path_samples"] += tls["slow"] - slow0
                self._cv.notify_all()

    # ---- batches ---------------------------------------------------------------------------------------------------------------
    def samples_per_rank(self) -> int:
        """The SAME number on every rank: the index order is padded (wrapping around) to a multiple of the world size before it is
        strided, as torch's DistributedSampler -- and catalyst's DistributedSamplerWrapper over it, which the reference uses
        (mindmap/data_loading/dataset.py:566-583) -- does.  Ranks with different batch counts would leave one of them alone in the
        gradient all-reduce at the end of an epoch."""
        return -(-len(self.ds) // self.world)

    def __len__(self) -> int:
        n = self.samples_per_rank()
        return n // self.B if self.drop_last else -(-n // self.B)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = int(epoch)

    def _order(self) -> List[int]:
        n = len(self.ds)
        if self.shuffle:
            g = torch.Generator()
            g.manual_seed(self.seed + self.epoch)
            order = torch.randperm(n, generator=g).tolist()
        else:
            order = list(range(n))
        total = self.samples_per_rank() * self.world
        if total > n:
            order = order + order[: total - n]  # (pad by wrapping around, like DistributedSampler; n >= 1 here)
        return order[self.rank:total:self.world]  # rank-strided, as the reference's DistributedSamplerWrapper partitions

    def release(self, batch: HostBatch, event=None) -> None:
        """The consumer is done with the batch's host memory -- now, or (``event``: a recorded torch.cuda.Event) once the device
        work that reads it has completed."""
        if batch.slot < 0:
            return
        slot, batch.slot = batch.slot, -1
        if event is None:
            self._free.append(slot)
        else:
            self._pending_release.append((slot, event))

    def _reap(self, block: bool) -> None:
        keep = []
        for slot, ev in self._pending_release:
            if ev.query():
                self._free.append(slot)
            else:
                keep.append((slot, ev))
        self._pending_release = keep
        if block and not self._free and self._pending_release:
            slot, ev = self._pending_release.pop(0)
            ev.synchronize()
            self._free.append(slot)

    def _reclaim_all(self) -> None:
        """a new epoch starts (possibly after an abandoned one): wait for the rows in flight, take every slot back."""
        with self._cv:
            while self._outstanding > 0:
                self._cv.wait(timeout=1.0)
            self._errors.clear()
        for _, ev in self._pending_release:
            ev.synchronize()
        self._pending_release = []
        self._free = list(range(self.n_slots))

    def __iter__(self) -> Iterator[HostBatch]:
        if self._closed:
            raise RuntimeError("PinnedBatchLoader: closed")
        self._reclaim_all()
        order = self._order()
        self.epoch += 1
        nb = len(order) // self.B if self.drop_last else -(-len(order) // self.B)
        issued = 0
        inflight: List[tuple] = []   # (slot, rows) in batch order
        handed: List[HostBatch] = []
        while issued < nb or inflight:
            # implicit release: a batch that was never released explicitly lives until the one after the next is requested
            while len(handed) > 1:
                self.release(handed.pop(0))
            self._reap(block=False)
            while issued < nb and (self._free or not inflight):
                if not self._free:
                    if handed:
                        self.release(handed.pop(0))
                    self._reap(block=True)
                    if not self._free:
                        raise RuntimeError("PinnedBatchLoader: every slot is held by the consumer (release() batches, or raise `slots`)")
                slot = self._free.pop(0)
                idxs = order[issued * self.B:(issued + 1) * self.B]
                with self._cv:
                    self._done[slot] = 0
                    self._outstanding += len(idxs)
                for row, idx in enumerate(idxs):
                    self._tasks.put((idx, slot, row))
                inflight.append((slot, len(idxs)))
                issued += 1
            slot, rows = inflight.pop(0)
            with self._cv:
                while self._done[slot] < rows and not self._errors:
                    self._cv.wait(timeout=1.0)
                if self._errors:
                    raise self._errors.pop(0)
            hb = HostBatch({k: t[:rows] for k, t in self.buffers[slot].items()})
            hb.slot = slot
            handed.append(hb)
            yield hb
        for hb in handed:
            self.release(hb)

    def stats(self) -> Dict[str, float]:
        with self._cv:
            s = dict(self._stats)
        s["cpu_ms_per_sample"] = s["cpu_s"] / max(s["samples"], 1) * 1e3
        s["stale_raw_copies"] = VC.STALE_COUNT[0]
        s["threads"], s["slots"], s["pinned"] = self.n_threads, self.n_slots, self.pinned
        return s

    def reset_stats(self) -> None:
        with self._cv:
            self._stats = {"samples": 0, "slow_path_samples": 0, "cpu_s": 0.0}

    def close(self) -> None:
        if not self._closed:
            self._closed = True
            for _ in self._threads:
                self._tasks.put(None)
            for t in self._threads:
                t.join(timeout=5.0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def drain(dataset_path: str, seconds: float = 3.0, batch_size: int = 32, threads: int = 2, num_vertices: int = 2048, repeat: int = 16) -> Dict[str, float]:
    """Loader only (no GPU, no pinning): samples/s and CPU milliseconds per sample of one PinnedBatchLoader over ``seconds``."""
    ds = MindmapFrameDataset(dataset_path, num_vertices=num_vertices)
    ds.samples = ds.samples * max(1, repeat)
    ld = PinnedBatchLoader(ds, batch_size, threads=threads, slots=3, pin_memory=False)
    for i, _ in enumerate(ld):  # page cache, per-frame items
        if i >= 2:
            break
    ld.reset_stats()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for b in ld:
            n += b["rgb_u8"].shape[0]
            if time.perf_counter() - t0 >= seconds:
                break
    dt = time.perf_counter() - t0
    st = ld.stats()
    ld.close()
    return {"samples_per_s": n / dt, "cpu_ms_per_sample": st["cpu_ms_per_sample"], "slow_path_samples": st["slow_path_samples"], "threads": threads}


if __name__ == "__main__":
    import json
    import sys

    a = sys.argv[1:]
    if not a:
        raise SystemExit("usage: python -m nvblox_mindmap_amd.data_loading.pinned_loader <dataset path> [seconds] [threads]")
    print(json.dumps(drain(a[0], float(a[1]) if len(a) > 1 else 3.0, threads=int(a[2]) if len(a) > 2 else 2)))
