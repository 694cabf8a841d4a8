"""Deferred feature rows (``mmf_set_deferred_feature_rows`` / ``Mapper.set_deferred_feature_rows``): a fused frame leaves its last
launch -- the row update of its survivor list -- to the next fused frame, which runs it beside its sphere trace; anything else
that takes the mapper runs it first.  Every observable state must be bit-identical to the undeferred sequence: at the end of a
stream, in the middle of it, and whatever call comes between two frames."""
import numpy as np
import os
import pytest
import torch

import fusion_common
from fusion_common import make_mapper, make_oracle, small_cfg
from nvblox_mindmap_amd import _lib
from nvblox_mindmap_amd import synthetic as S
from nvblox_mindmap_amd.nvblox_torch.mapper import integrate_frames_batch
from test_gpu_frame_batch import frame_args, same_maps, single
from test_gpu_fusion_parity import compare_features, compare_tsdf

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(fusion_common.NOT_DEFAULT_ROUTE, reason=fusion_common.ROUTE_SKIP_REASON)]


def pending(m, mapper_id=0):
    return _lib.lib().mmf_deferred_feature_rows_pending(m._h, mapper_id)


def stream_cfg(scale):
    base = small_cfg(scale)
    return S.StreamConfig(width=base.width, height=base.height, fx=base.fx, fy=base.fy, cx=base.cx, cy=base.cy, hole_mode="patches")


def pair(C, **kw):
    a, b = make_mapper(C, **kw), make_mapper(C, **kw)
    a.set_deferred_feature_rows(True)
    return a, b


def feed(cfg, mappers, index, C, k, decay=True):
    """the same frame into every mapper (each gets its own tensors: the deferred one keeps its feature image in use)"""
    out = []
    for m in mappers:
        e, f, dyn = frame_args(cfg, m, index, C, k)
        if decay:
            m.decay()
        out.append(single(e))
    for (dm, fm) in out[1:]:
        assert torch.equal(dm, out[0][0]) and torch.equal(fm, out[0][1])
    return f, dyn


@pytest.mark.parametrize("over", [{}, {"workspace_bounds_type": 0, "max_integration_distance_m": 2.5}], ids=["bounded", "hash_path"])
def test_stream_equals_the_undeferred_one_and_the_oracle(oracle_mod, over):
    """`hash_path` (round 5): an unbounded workspace -- the frame's launches are the scalable ones (k_front_compact_big, k_alloc_big,
    k_tsdf_pass<lazy>, k_sphere_alloc_big), which host the previous frame's gating and rows as the bounded launches do."""
    from oracle import image_ops as IO

    cfg, C = stream_cfg(2), 16
    d, e = pair(C, **over)
    orc = make_oracle(oracle_mod, C, **over)
    for step in range(8 if over else 6):
        f, dyn = feed(cfg, (d, e), (9 * step) % 200, C, step)
        assert pending(d) == 1 and pending(e) == 0
        odm, ofm = IO.frame_masks(~dyn, f["depth"], 0.3, 3, 4, 5, cfg.height, cfg.width)
        orc.decay()
        orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
        orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
        if step == 2:  # a reader in the middle of the stream sees the frame complete (and leaves nothing pending)
            same_maps(d, e)
            assert pending(d) == 0
    same_maps(d, e)
    assert pending(d) == 0
    compare_tsdf(orc, d)
    compare_features(orc, d)
    assert d.stats() == e.stats()


def test_lowres_feature_source_streams_too():
    """The facade's default feature source (the backbone's low-res map, sampled inside the row update) defers like the image."""
    cfg, C = stream_cfg(2), 16
    d, e = pair(C)
    gen = torch.Generator("cuda").manual_seed(11)
    for step in range(6):
        low = torch.rand(15, 20, 16, device="cuda", generator=gen)
        for m in (d, e):
            ea, _, _ = frame_args(cfg, m, 9 * step, C, step)
            m.decay()
            m.integrate_frame_lowres(ea["depth_frame"], ea["color_frame"], low.clone(), ea["input_mask"], ea["t_w_c"], ea["intrinsics"], 0.3,
                                     3, 4, 5, invert_input_mask=True)
        assert pending(d) == 1 and pending(e) == 0
        if step == 3:  # a full-resolution frame in between rides / hosts the same way
            feed(cfg, (d, e), 50, C, 7)
    same_maps(d, e)


@pytest.mark.parametrize("n", [2, 5, 8])
def test_batches_of_deferring_mappers(n):
    """``integrate_frames_batch`` with some mappers in deferred mode and some not: the deferring ones leave their tails to the next
    batch (roles of its launches 1 and 3), the others run launches 4 and 5 of the same batch; a single ``integrate_frame`` between
    two batches hosts / defers the same way."""
    cfg, C = stream_cfg(2), 16
    batched = [make_mapper(C) for _ in range(n)]
    alone = [make_mapper(C) for _ in range(n)]
    for q, m in enumerate(batched):
        m.set_deferred_feature_rows(q % 3 != 2)  # every third mapper does not defer
    for step in range(5):
        entries_b, entries_a = [], []
        for q in range(n):
            eb, _, _ = frame_args(cfg, batched[q], (17 * q + 9 * step) % 200, C, q + step)
            ea, _, _ = frame_args(cfg, alone[q], (17 * q + 9 * step) % 200, C, q + step)
            entries_b.append(eb)
            entries_a.append(ea)
        for m in batched + alone:
            m.decay()
        if step == 2:  # one mapper takes the single call in between
            single(entries_b[0])
            masks_b = [None] + integrate_frames_batch(entries_b[1:]) if n > 1 else [None]
        else:
            masks_b = integrate_frames_batch(entries_b)
        masks_a = [single(e) for e in entries_a]
        for mb, ma in zip(masks_b, masks_a):
            if mb is not None:
                assert torch.equal(mb[0], ma[0]) and torch.equal(mb[1], ma[1])
        for q, m in enumerate(batched):
            assert pending(m) == (1 if q % 3 != 2 else 0)
        if step == 3:
            same_maps(batched[0], alone[0])  # a reader completes one mapper's frame, the others stay pending
    for q in range(n):
        same_maps(batched[q], alone[q])


@pytest.mark.parametrize("batched", [False, True])
def test_image_size_changes_in_mid_stream(batched):
    """A larger image re-allocates internal images (synthetic depth, mask scratch) while a frame's appearance tail is pending."""
    C = 16
    d, e = pair(C)
    d2, e2 = pair(C)
    k = 0
    for scale in (4, 4, 2, 2, 4, 1, 2):
        cfg = stream_cfg(scale)
        if not batched:
            feed(cfg, (d, e), 11 * k, C, k)
        else:
            for m in (d, d2, e, e2):
                m.decay()
            integrate_frames_batch([frame_args(cfg, d, 11 * k, C, k)[0], frame_args(cfg, d2, 11 * k + 5, C, k + 1)[0]])
            single(frame_args(cfg, e, 11 * k, C, k)[0])
            single(frame_args(cfg, e2, 11 * k + 5, C, k + 1)[0])
        k += 1
    same_maps(d, e)
    if batched:
        same_maps(d2, e2)


def test_sequence_call_leaves_the_mapper_as_the_single_calls_do():
    cfg, C = stream_cfg(2), 16
    d, e = make_mapper(C), make_mapper(C)
    frames = [frame_args(cfg, d, 7 * k, C, k)[0] for k in range(6)]
    keys = ("depth_frame", "color_frame", "feature_frame", "input_mask", "t_w_c", "intrinsics", "min_depth_m",
            "input_mask_erosion_iterations", "valid_depth_mask_erosion_iterations", "border_percent", "invert_input_mask")
    masks = d.integrate_frame_sequence([{k: f[k] for k in keys} for f in frames])
    assert pending(d) == 0
    for f, (dm, fm) in zip(frames, masks):
        e.decay()
        dm2, fm2 = single(dict(f, mapper=e))
        assert torch.equal(dm, dm2) and torch.equal(fm, fm2)
    same_maps(d, e)
    single(dict(frames[0], mapper=d))  # the mode is off again
    assert pending(d) == 0


def test_full_size_stream():
    """BASELINE configs[2] at full size (640x480, C = 64): 8 frames with a decay() before each, deferred against undeferred."""
    cfg, C = S.StreamConfig(hole_mode="patches"), 64
    d, e = pair(C)
    for step in range(8):
        feed(cfg, (d, e), 5 * step, C, step)
    assert pending(d) == 1
    d.flush()
    assert pending(d) == 0
    same_maps(d, e)


def _standalone_feature(cfg, C, m):
    f = S.frame(cfg, 11, C)
    m.add_feature_frame(torch.from_numpy(np.ascontiguousarray(f["features"])).cuda(), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]))


def _lowres_frame(cfg, C, m):
    ea, f, dyn = frame_args(cfg, m, 17, C, 3)
    low = torch.rand(15, 20, 16, device="cuda", generator=torch.Generator("cuda").manual_seed(5))
    m.integrate_frame_lowres(ea["depth_frame"], ea["color_frame"], low, ea["input_mask"], ea["t_w_c"], ea["intrinsics"], 0.3, 3, 4, 5)


def _batch_call(cfg, C, m):
    ea, _, _ = frame_args(cfg, m, 23, C, 4)
    integrate_frames_batch([ea])


def _multi_call(cfg, C, m):
    ea, _, _ = frame_args(cfg, m, 29, C, 5)
    m.integrate_frame_multi(ea["depth_frame"], ea["color_frame"], ea["feature_frame"], ea["t_w_c"], ea["intrinsics"], 0.3, 5,
                            [dict(mapper_id=0, input_mask=ea["input_mask"], input_mask_erosion_iterations=3,
                                  valid_depth_mask_erosion_iterations=4)])


def _other_stream(cfg, C, m):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ea, _, _ = frame_args(cfg, m, 31, C, 6)
        single(ea)
    torch.cuda.current_stream().wait_stream(s)


def _same_pose_twice(cfg, C, m):  # the second frame finds its synthetic depth cached: no sphere-trace launch to ride in
    for _ in range(2):
        ea, _, _ = frame_args(cfg, m, 37, C, 7)
        single(ea)


BETWEEN = {"add_feature_frame": _standalone_feature, "lowres_frame": _lowres_frame, "frame_batch": _batch_call, "frame_multi": _multi_call,
           "other_stream": _other_stream, "same_pose_twice": _same_pose_twice, "decay": lambda cfg, C, m: m.decay(),
           "mesh": lambda cfg, C, m: m.update_feature_mesh(), "clear": lambda cfg, C, m: m.clear(),
           "color_frame": lambda cfg, C, m: m.add_color_frame(frame_args(cfg, m, 3, C, 1)[0]["color_frame"],
                                                              torch.from_numpy(S.frame(cfg, 3, C)["T_W_C"]), torch.from_numpy(S.frame(cfg, 3, C)["K"]))}


@pytest.mark.parametrize("what", sorted(BETWEEN))
def test_whatever_comes_between_two_frames(what):
    cfg, C = stream_cfg(2), 16
    d, e = pair(C)
    feed(cfg, (d, e), 0, C, 0)
    feed(cfg, (d, e), 4, C, 1)
    assert pending(d) == 1
    for m in (d, e):
        BETWEEN[what](cfg, C, m)
    feed(cfg, (d, e), 7, C, 2)
    feed(cfg, (d, e), 10, C, 3, decay=False)
    assert pending(d) == 1
    same_maps(d, e)


def test_reader_on_another_stream_sees_the_frame_complete():
    """A caller integrates on stream 1, records an event, makes stream 2 wait for it and reads the map on stream 2 -- valid without
    deferral.  With it, the frame's tail is enqueued on stream 1 only when the reader arrives (after the event): the reader's
    stream must then wait for that tail (round-3 advisor finding).  Stream 1 is kept busy behind the event, so an unordered
    read would run long before the tail."""
    cfg, C = stream_cfg(2), 16
    d, e = pair(C)
    feed(cfg, (d, e), 0, C, 0)
    feed(cfg, (d, e), 4, C, 1)
    assert pending(d) == 1
    s1, s2 = torch.cuda.current_stream(), torch.cuda.Stream()
    ev = torch.cuda.Event()
    ev.record(s1)
    big = torch.randn(8192, 8192, device="cuda")
    for _ in range(12):
        big = (big @ big).clamp_(-1.0, 1.0)  # ~100 ms of work on stream 1 behind the event
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        fd, wd, idd = d.feature_layer_view(0).get_all_blocks_split()
        fe, we, ide = e.feature_layer_view(0).get_all_blocks_split()
    assert pending(d) == 0
    torch.cuda.synchronize()
    assert torch.equal(idd, ide) and torch.equal(wd, we) and torch.equal(fd.view(torch.int16), fe.view(torch.int16))
    # mmf_flush on another stream orders that stream after the tail as well
    feed(cfg, (d, e), 9, C, 2)
    ev.record(s1)
    for _ in range(12):
        big = (big @ big).clamp_(-1.0, 1.0)
    s2.wait_event(ev)
    with torch.cuda.stream(s2):
        d.flush()
        wd2 = d.feature_layer_view(0).get_all_blocks_split()[1].clone()
    torch.cuda.synchronize()
    assert torch.equal(wd2, e.feature_layer_view(0).get_all_blocks_split()[1])


def test_switching_it_off_runs_what_is_pending():
    cfg, C = stream_cfg(2), 16
    d, e = pair(C)
    feed(cfg, (d, e), 100, C, 9)
    assert pending(d) == 1
    d.set_deferred_feature_rows(False)
    assert pending(d) == 0
    feed(cfg, (d, e), 104, C, 10)
    assert pending(d) == 0
    same_maps(d, e)


def test_checkpoint_of_a_deferred_stream(tmp_path):
    cfg, C = stream_cfg(2), 16
    d, e = pair(C)
    for step in range(3):
        feed(cfg, (d, e), 13 * step, C, step)
    d.save_map(str(tmp_path / "d.nvblx"))
    r = make_mapper(C)
    r.load_from_file(str(tmp_path / "d.nvblx"))
    same_maps(r, e)


def test_feature_image_modified_in_place_is_reported():
    cfg, C = stream_cfg(4), 16
    d = make_mapper(C)
    d.set_deferred_feature_rows(True)
    ea, _, _ = frame_args(cfg, d, 0, C, 0)
    single(ea)
    ea["feature_frame"].mul_(2.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        single(ea)
    # ... unless the update has run in the meantime
    d2 = make_mapper(C)
    d2.set_deferred_feature_rows(True)
    eb, _, _ = frame_args(cfg, d2, 0, C, 0)
    single(eb)
    d2.flush()
    eb["feature_frame"].mul_(2.0)
    single(eb)
