"""Long horizons: thousands of frames through the fused path against the CPU oracle, bit for bit.  What the short parity and
fuzz tests (a few dozen calls at most) never reach: the 8-bit grid tag wrapping several times, slow decay (0.98, the
reference's factor) deallocating blocks hundreds of frames after they were last seen and their slots being handed out again,
a map checkpoint taken and restored mid-stream, and -- on the hash path -- tombstones piling up to several table rebuilds.
Quarter-resolution images (160x120, 16 feature channels) keep the oracle's share to about a minute."""
import os

import numpy as np
import pytest
import torch

from fusion_common import make_mapper, make_oracle, small_cfg
from nvblox_mindmap_amd import synthetic as S
from test_gpu_fusion_parity import compare_features, compare_tsdf, dev

pytestmark = pytest.mark.gpu
C = 16


def stream_cfg():
    base = small_cfg(4)
    return S.StreamConfig(width=base.width, height=base.height, fx=base.fx, fy=base.fy, cx=base.cx, cy=base.cy, hole_mode="patches")


class FrameCache:
    def __init__(self, cfg):
        self.cfg, self.frames = cfg, {}

    def get(self, index):
        if index not in self.frames:
            f = S.frame(self.cfg, index, C)
            self.frames[index] = (f, {k: dev(f[k]) for k in ("depth", "rgb", "features")})
        return self.frames[index]


def compare_colors(orc, gpu):
    rgb, w, idx = gpu.color_layer_view(0).get_all_blocks_split()
    orgb, ow = orc.all_colors()
    assert np.array_equal(idx.cpu().numpy(), orc.block_indices(1))
    assert np.array_equal(w.cpu().numpy(), ow) and np.array_equal(rgb.cpu().numpy(), orgb)


def fused(gpu, orc, cache, index, k, min_d=0.3, k_in=3, k_depth=4, border=5):
    """decay + integrate_frame with a moving dynamic rectangle (the native call reads it inverted) on both sides."""
    from oracle import image_ops as IO

    f, d = cache.get(index)
    cfg = cache.cfg
    dyn = np.zeros(f["depth"].shape, dtype=bool)
    r0, c0 = (7 * k) % (cfg.height - 30), (11 * k) % (cfg.width - 40)
    dyn[r0:r0 + 24, c0:c0 + 32] = True
    odm, ofm = IO.frame_masks(~dyn, f["depth"], min_d, k_in, k_depth, border, cfg.height, cfg.width)
    orc.decay()
    orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"], odm.astype(np.uint8))
    orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"], odm.astype(np.uint8))
    orc.add_feature_frame(f["features"], f["T_W_C"], f["K"], ofm.astype(np.uint8))
    gpu.decay()
    gpu.integrate_frame(d["depth"], d["rgb"], d["features"], dev(dyn), torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"]), min_d,
                        k_in, k_depth, border, 0, invert_input_mask=True)


def pose_index(k):
    """orbit -> a long dwell on three neighbouring poses (everything only the rest of the orbit saw fades and is deallocated
    ~420 frames later: 5 * 0.98^n < 1e-3) -> orbit again (reallocation into reused slots)."""
    if k < 700:
        return (7 * k) % 200
    if k < 1250:
        return 40 + k % 3
    return (11 * k) % 200


@pytest.mark.parametrize("pipelined", [False, True])
def test_soak_bounded_2000_frames_with_checkpoint(oracle_mod, tmp_path, pipelined):
    """pipelined: consecutive frames software-pipelined (set_deferred_feature_rows: the appearance half of a frame as roles of the
    next frame's launches) -- the periodic comparisons and the checkpoint are the readers that complete the pending frame."""
    n_frames = int(os.environ.get("MMF_SOAK_FRAMES", "2000"))
    cache = FrameCache(stream_cfg())
    gpu, orc = make_mapper(C), make_oracle(oracle_mod, C)  # DRILL_IN_BOX box, 1 cm voxels, decay 0.98: the reference's mapper
    gpu.set_deferred_feature_rows(pipelined)
    live = []
    for k in range(n_frames):
        fused(gpu, orc, cache, pose_index(k), k)
        if k % 50 == 49:
            n = gpu.tsdf_layer_view(0).num_allocated_blocks()
            assert n == orc.num_blocks(0), (k, n, orc.num_blocks(0))
            live.append(n)
        if k % 250 == 249:
            mx, exact = compare_tsdf(orc, gpu)
            assert exact, k
            compare_features(orc, gpu)
        if k == n_frames // 2:  # checkpoint -> a NEW mapper carries on from the file
            path = str(tmp_path / "soak.nvblx")
            gpu.save_map(path, 0)
            fresh = make_mapper(C)
            fresh.load_from_file(path, 0)
            fresh.set_deferred_feature_rows(pipelined)
            gpu = fresh
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)
    compare_colors(orc, gpu)
    if n_frames >= 2000:
        assert min(live) < max(live) - 50, f"the dwell must deallocate blocks and the second orbit bring them back: {live}"
    ov, of = orc.feature_mesh()
    mesh = gpu.get_feature_mesh(0)
    assert np.array_equal(mesh.vertices().cpu().numpy(), ov)
    assert np.array_equal(mesh.vertex_features().cpu().numpy().view(np.uint16), of.view(np.uint16))


@pytest.mark.parametrize("pipelined", [False, True])
def test_soak_unbounded_500_frames_hash_churn(oracle_mod, tmp_path, pipelined):
    """pipelined (round 5): set_deferred_feature_rows on a large map -- the scalable launches host the previous frame's gating and rows.
    The hash path over 500 frames: 2 cm voxels, 3 m range, decay 0.9 (blocks die ~80 frames after they leave the view), a
    pool of 8 192 blocks (16 384 table entries: the tombstones force a rebuild every ~4 096 deallocations).  Fused frames with
    every fifth frame as the reference's stand-alone calls (eager decay, separate allocation launches)."""
    n_frames = int(os.environ.get("MMF_SOAK_FRAMES_UNBOUNDED", "500"))
    cache = FrameCache(stream_cfg())
    over = dict(workspace_bounds_type=0, voxel_size=0.02, max_integration_distance_m=3.0, tsdf_decay_factor=0.9, num_preallocated_blocks=8192)
    gpu, orc = make_mapper(C, **over), make_oracle(oracle_mod, C, **over)
    gpu.set_deferred_feature_rows(pipelined)
    rebuilds = 0
    for k in range(n_frames):
        index = (13 * k) % 200 if (k // 60) % 2 == 0 else 100 + k % 5
        if k % 5 == 4:
            f, d = cache.get(index)
            T, K = torch.from_numpy(f["T_W_C"]), torch.from_numpy(f["K"])
            orc.decay()
            gpu.decay()
            orc.add_depth_frame(f["depth"], f["T_W_C"], f["K"])
            gpu.add_depth_frame(d["depth"], T, K, None, 0)
            orc.add_color_frame(f["rgb"], f["T_W_C"], f["K"])
            gpu.add_color_frame(d["rgb"], T, K, mask_frame=None, mapper_id=0)
            orc.add_feature_frame(f["features"], f["T_W_C"], f["K"])
            gpu.add_feature_frame(d["features"], T, K, None, 0)
        else:
            fused(gpu, orc, cache, index, k)
        if k % 100 == 99:
            mx, exact = compare_tsdf(orc, gpu)
            assert exact, k
            compare_features(orc, gpu)
        if k == n_frames // 2:
            path = str(tmp_path / "soak_unbounded.nvblx")
            rebuilds += gpu.hash_state(0)["rebuilds"]
            gpu.save_map(path, 0)
            fresh = make_mapper(C, **over)
            fresh.load_from_file(path, 0)
            fresh.set_deferred_feature_rows(pipelined)
            gpu = fresh
    st = gpu.hash_state(0)
    assert st["table_entries"] == 16384 and st["live_blocks"] == orc.num_blocks(0)
    mx, exact = compare_tsdf(orc, gpu)
    assert exact
    compare_features(orc, gpu)
    compare_colors(orc, gpu)
    # the amortised rebuild keeps the tombstones under a quarter of the table, and its counter is exact (a scan of the table agrees):
    # keys that leave and re-enter the view take their tombstones back, so on an orbit a rebuild may never be needed
    assert gpu.count_tombstones(0) == st["tombstones"] and st["tombstones"] * 4 <= st["table_entries"], (rebuilds, st)
