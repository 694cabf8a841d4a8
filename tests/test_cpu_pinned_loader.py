"""data_loading/pinned_loader.py: batches assembled in place (one copy per byte, page cache -> batch buffer) must be the batches the
reference-shaped path gives -- ``MindmapFrameDataset.__getitem__`` + ``default_collate`` (mindmap/data_loading/dataset.py:425-490)
-- sample for sample: same selection draws for a seeded dataset, same values, padding and validity masks; every sample of an epoch
exactly once; slots recycled; frames without raw copies take the slow path and still match; rank-strided partition."""
import os

import numpy as np
import pytest
import torch
from torch.utils.data import default_collate

from nvblox_mindmap_amd.data_loading.dataset import MindmapFrameDataset, write_synthetic_demo
from nvblox_mindmap_amd.data_loading.pinned_loader import PinnedBatchLoader
from nvblox_mindmap_amd.data_loading.vertex_sampling import VertexSamplingMethod
from nvblox_mindmap_amd.io import vertex_cache as VC


@pytest.fixture(scope="module")
def dataset_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pinned_ds"))
    write_synthetic_demo(os.path.join(d, "demo_00000"), 7, image_size=(40, 56), feature_dim=24, vertex_count_range=(700, 1500))
    write_synthetic_demo(os.path.join(d, "demo_00001"), 4, image_size=(40, 56), feature_dim=24, vertex_count_range=(50, 300), seed=5)  # fewer than asked: padded
    VC.convert_dataset(d)
    return d


def same_batch(a, b):
    assert set(a.keys()) == set(b.keys())
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("method", [VertexSamplingMethod.RANDOM_WITHOUT_REPLACEMENT, VertexSamplingMethod.RANDOM_WITH_REPLACEMENT])
def test_batches_equal_getitem_plus_collate(dataset_dir, method):
    ds = MindmapFrameDataset(dataset_dir, num_vertices=512, vertex_sampling_method=method, seed=11)
    ref = MindmapFrameDataset(dataset_dir, num_vertices=512, vertex_sampling_method=method, seed=11, use_raw_vertex_cache=False)
    ld = PinnedBatchLoader(ds, batch_size=4, shuffle=False, drop_last=False, threads=3, slots=2, pin_memory=False)
    assert len(ld) == 3
    seen = 0
    for bi, b in enumerate(ld):
        want = default_collate([ref[i] for i in range(bi * 4, min(bi * 4 + 4, len(ref)))])
        same_batch({k: v.clone() for k, v in b.items()}, want)
        seen += b["rgb_u8"].shape[0]
    assert seen == len(ds) == 11
    st = ld.stats()
    assert st["samples"] == 11 and st["slow_path_samples"] == 0 and st["cpu_ms_per_sample"] > 0
    ld.close()


def test_every_sample_once_per_epoch_and_slots_recycle(dataset_dir):
    ds = MindmapFrameDataset(dataset_dir, num_vertices=256, seed=None)
    ld = PinnedBatchLoader(ds, batch_size=2, shuffle=True, drop_last=True, threads=2, slots=2, seed=3, pin_memory=False)
    by_pose = {tuple(np.load(s["pov_pose"]).astype(np.float32).tolist()): i for i, s in enumerate(ds.samples)}
    orders = []
    for _ in range(2):
        got = []
        for b in ld:
            for p in b["camera_poses"][:, 0]:
                got.append(by_pose[tuple(p.tolist())])
        assert len(got) == 10 and len(set(got)) == 10  # 11 samples, drop_last
        orders.append(got)
    assert orders[0] != orders[1]  # a new permutation per epoch
    # unseeded selection: rows differ between epochs but are rows of the frame, valid mask all ones for the big frames
    b = next(iter(ld))
    assert b["vertices_valid_mask"].dtype == torch.bool and b["vertex_features"].dtype == torch.float16
    ld.close()


def test_frames_without_raw_copies_take_the_slow_path(tmp_path):
    d = str(tmp_path / "ds")
    write_synthetic_demo(os.path.join(d, "demo_00000"), 4, image_size=(32, 32), feature_dim=16, vertex_count_range=(300, 600), seed=2)
    ds = MindmapFrameDataset(d, num_vertices=128, seed=4)
    ld = PinnedBatchLoader(ds, batch_size=2, shuffle=False, threads=2, slots=2, pin_memory=False)
    got = [{k: v.clone() for k, v in b.items()} for b in ld]
    assert ld.stats()["slow_path_samples"] == 4
    ld.close()
    VC.convert_dataset(d)
    ds2 = MindmapFrameDataset(d, num_vertices=128, seed=4)
    ld2 = PinnedBatchLoader(ds2, batch_size=2, shuffle=False, threads=2, slots=2, pin_memory=False)
    for a, b in zip(got, ld2):
        same_batch(a, {k: v.clone() for k, v in b.items()})
    assert ld2.stats()["slow_path_samples"] == 0
    ld2.close()


def test_rank_strided_partition(dataset_dir):
    """Round-5 advisor finding: 11 samples over 2 ranks gave 6 and 5 batches -- the rank with the extra batch waits alone in the
    gradient all-reduce.  The order is padded (wrapping around) to a multiple of the world size first, like DistributedSampler: every
    rank gets the same count, the union covers every sample, only the padding repeats."""
    ds = MindmapFrameDataset(dataset_dir, num_vertices=64, seed=1)
    assert len(ds) == 11
    seen, lens = [], []
    for r in range(2):
        ld = PinnedBatchLoader(ds, batch_size=1, shuffle=True, drop_last=False, threads=1, slots=2, seed=9, pin_memory=False, rank=r, world_size=2)
        lens.append(len(ld))
        seen.append([tuple(b["camera_poses"][0, 0].tolist()) for b in ld])
        ld.close()
    assert lens == [6, 6] and len(seen[0]) == len(seen[1]) == 6
    assert len(set(seen[0]) | set(seen[1])) == 11 and len(set(seen[0]) & set(seen[1])) == 1  # one wrapped-around sample


@pytest.mark.parametrize("n_world_batch_drop", [(11, 2, 4, True), (11, 2, 4, False), (11, 3, 2, True), (11, 4, 3, False), (11, 8, 1, True)])
def test_every_rank_has_the_same_number_of_batches(dataset_dir, n_world_batch_drop):
    n, world, B, drop = n_world_batch_drop
    ds = MindmapFrameDataset(dataset_dir, num_vertices=64, seed=1)
    assert len(ds) == n
    lens, counts = [], []
    for r in range(world):
        ld = PinnedBatchLoader(ds, batch_size=B, shuffle=True, drop_last=drop, threads=1, slots=2, seed=2, pin_memory=False, rank=r, world_size=world)
        lens.append(len(ld))
        counts.append(sum(1 for _ in ld))
        ld.close()
    assert len(set(lens)) == 1 and counts == lens, (lens, counts)


def test_copied_dataset_keeps_its_raw_copies(tmp_path):
    """Round-4 advisor finding: a dataset copied without its modification times invalidated every raw copy silently.  Size + a
    content hash of the source decide now; a regenerated source of the same size is still refused."""
    import shutil

    d = str(tmp_path / "a")
    write_synthetic_demo(os.path.join(d, "demo_00000"), 2, image_size=(32, 32), feature_dim=16, vertex_count_range=(300, 600), seed=1)
    VC.convert_dataset(d)
    e = str(tmp_path / "b")
    shutil.copytree(d, e, copy_function=shutil.copyfile)  # contents only: fresh modification times
    for root, _, files in os.walk(e):
        for f in files:
            os.utime(os.path.join(root, f), (1.0e9, 1.0e9))
    before = VC.STALE_COUNT[0]
    ds = MindmapFrameDataset(e, num_vertices=128, seed=4)
    ld = PinnedBatchLoader(ds, batch_size=2, shuffle=False, threads=1, slots=2, pin_memory=False)
    list(ld)
    assert ld.stats()["slow_path_samples"] == 0 and VC.STALE_COUNT[0] == before
    ld.close()
    assert VC.convert_dataset(e) == 0  # nothing to rewrite
    # same size, other content: refused
    zst = ds.samples[0]["vertex_features"]
    blob = bytearray(open(zst, "rb").read())
    blob[len(blob) // 2] ^= 0xFF
    blob[10] ^= 0xFF
    open(zst, "wb").write(bytes(blob))
    with pytest.raises(VC.StaleRawCopy):
        VC.raw_header(VC.raw_path_of(zst), zst)
    assert VC.STALE_COUNT[0] == before + 1


def test_an_abandoned_epoch_gives_its_slots_back(dataset_dir):
    ds = MindmapFrameDataset(dataset_dir, num_vertices=64, seed=1)
    ld = PinnedBatchLoader(ds, batch_size=2, shuffle=False, threads=2, slots=2, pin_memory=False)
    for _ in range(3):
        for i, b in enumerate(ld):
            if i == 1:
                break  # two batches in flight / handed out when the consumer walks away
    assert sum(b["rgb_u8"].shape[0] for b in ld) == 10
    ld.close()


def test_host_io_entry_points_fail_loudly(tmp_path):
    """mmf_host_read_file_at / mmf_host_sample_vertex_file (csrc/mmf_host_io.hip): short files, missing files and out-of-range rows are
    errors with a message, not silent garbage in a batch buffer."""
    import ctypes as C

    from nvblox_mindmap_amd import _lib

    L = _lib.lib()
    buf = np.zeros(64, dtype=np.uint8)
    p = tmp_path / "f.bin"
    p.write_bytes(bytes(range(32)))
    assert L.mmf_host_read_file_at(str(p).encode(), 8, buf.ctypes.data, 16) == 0 and buf[:16].tolist() == list(range(8, 24))
    assert L.mmf_host_read_file_at(str(p).encode(), 8, buf.ctypes.data, 64) != 0 and "short read" in _lib.last_error()
    assert L.mmf_host_read_file_at(str(tmp_path / "missing").encode(), 0, buf.ctypes.data, 1) != 0 and "cannot open" in _lib.last_error()
    raw = str(tmp_path / "0000.nvblox_vertex_features.raw")
    verts, feats = torch.rand(10, 3), torch.randn(10, 16)
    VC.write_raw(raw, verts, feats)
    V, Cc, off_v, off_f = VC.raw_header(raw)
    rows = np.array([7, 0, 9], dtype=np.int64)
    v16, f16 = np.zeros((3, 3), np.float16), np.zeros((3, 16), np.float16)
    assert L.mmf_host_sample_vertex_file(raw.encode(), off_v, off_f, V, Cc, rows.ctypes.data, 3, v16.ctypes.data, f16.ctypes.data) == 0
    assert np.array_equal(v16, verts.half().numpy()[rows]) and np.array_equal(f16, feats.half().numpy()[rows])
    bad = np.array([10], dtype=np.int64)
    assert L.mmf_host_sample_vertex_file(raw.encode(), off_v, off_f, V, Cc, bad.ctypes.data, 1, v16.ctypes.data, f16.ctypes.data) != 0
    assert "out of range" in _lib.last_error()
    assert L.mmf_host_sample_vertex_file(raw.encode(), off_v, off_f, V + 1000, Cc, rows.ctypes.data, 3, v16.ctypes.data, f16.ctypes.data) != 0
    assert "smaller than its header" in _lib.last_error()


def test_slow_path_leaves_the_process_generators_alone(dataset_dir):
    """Round-5 advisor finding: the slow path (taken for every sample when an augmentor / noiser is set) ran ``__getitem__`` in a
    loader thread of the training process, where a seeded dataset's ``torch.manual_seed(seed + idx)`` reseeded the process-wide
    generators the trainer's diffusion noise comes from.  Loader threads now draw from generators of their own -- the same values for
    the same seed -- and neither building the loader nor an epoch through it moves torch's or Python's global streams."""
    import random

    from nvblox_mindmap_amd.data_loading.sample_transformer import GeometryNoiser

    ds = MindmapFrameDataset(dataset_dir, num_vertices=128, seed=21, geometry_noiser=GeometryNoiser(0.01, 1.0))
    torch.manual_seed(1234)
    random.seed(99)
    t_state, p_state = torch.get_rng_state().clone(), random.getstate()
    ld = PinnedBatchLoader(ds, batch_size=2, shuffle=False, drop_last=False, threads=2, slots=2, pin_memory=False)
    batches = [{k: v.clone() for k, v in b.items()} for b in ld]
    st = ld.stats()
    ld.close()
    assert st["slow_path_samples"] == len(ds)  # (a noiser is set: no sample can take the in-place path)
    assert torch.equal(torch.get_rng_state(), t_state) and random.getstate() == p_state
    # the selection draw of a seeded dataset is still the dataset's rule: sample idx seeded with seed + idx, the reference's draws
    ds_plain = MindmapFrameDataset(dataset_dir, num_vertices=128, seed=21)
    for i in range(len(ds)):
        want = ds_plain[i]
        got = {k: v[i % 2] for k, v in batches[i // 2].items()}
        assert torch.equal(got["vertex_features"], want["vertex_features"]) and torch.equal(got["vertices_valid_mask"], want["vertices_valid_mask"])
        assert not torch.equal(got["vertices"], want["vertices"]) or not bool(want["vertices_valid_mask"].any())  # (noised)
